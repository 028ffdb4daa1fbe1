"""Shared helpers for the test-suite: seeded synthetic inputs (tools/gen_synth.py) and pyramids."""
import functools

import numpy as np

import gen_synth
import oracle
from of_dis_amd.params import oppoint

_f32 = np.float32


@functools.lru_cache(maxsize=16)
def synth_case(width, height, seed=1234, channels=1, op_point=2, usetvref=None):
    """Returns (params, pyr_a[img,dx,dy], pyr_b[img,dx,dy], gt_flow, (ia, ib)) for a synthetic pair."""
    ia, ib, gt = gen_synth.make_pair(width, height, seed, channels)
    p = oppoint(op_point, width, height, noc=channels, usetvref=usetvref)
    O = oracle.c_oracle()
    pa = O.build_pyramid(p, ia)
    pb = O.build_pyramid(p, ib)
    return p, pa, pb, gt, (ia, ib)


def rand_planes(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(_f32)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=_f32)
    b = np.ascontiguousarray(b, dtype=_f32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    eq = (a == b) | (np.isnan(a) & np.isnan(b))
    if not eq.all():
        bad = np.argwhere(~eq)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ; first at {i}: {a[i]!r} vs {b[i]!r}; "
                             f"max abs diff {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))}")
