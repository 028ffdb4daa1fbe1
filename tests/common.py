"""Shared helpers for the test-suite: seeded synthetic inputs (tools/gen_synth.py) and pyramids."""
import functools

import numpy as np

import gen_synth
import oracle
from of_dis_amd.params import oppoint

_f32 = np.float32


def testhooks():
    """The TEST library (tests/csrc/ofdis_testhooks.hip, built by of_dis_amd.build next to the product): device kernels
    around the header-only helpers of ofdis_dev.h.  Not part of the shipped library."""
    import ctypes as C
    from of_dis_amd import build
    path = build.testhooks_path()
    if not __import__("os").path.exists(path):
        build.build()
    L = C.CDLL(path)
    VP = C.c_void_p
    L.ofdis_test_wave_sum.argtypes = [VP, VP, C.c_int, VP]
    L.ofdis_test_div_sqrt.argtypes = [VP, VP, VP, C.c_int, VP]
    L.ofdis_test_outlier_sq.restype = C.c_float
    L.ofdis_test_outlier_sq.argtypes = [C.c_float]
    return L


def wave_sum_test(gpu, x):
    x = np.ascontiguousarray(x, _f32)
    d, o = gpu.Dev(x), gpu.Dev(nbytes=x.nbytes)
    assert testhooks().ofdis_test_wave_sum(d.ptr, o.ptr, x.size, None) == 0
    gpu.check(gpu.lib().ofdis_sync(None))
    return o.get(x.shape)


def div_sqrt_test(gpu, a, b):
    """Rows: div_rn(a,b), a/b, sqrt_rn(|a|), sqrtf(|a|), the fused TV kernel's quotient a/b, its quotient b / sqrt(|a|),
    computed on the device (ofdis_dev.h)."""
    a, b = np.ascontiguousarray(a, _f32), np.ascontiguousarray(b, _f32)
    da, db, o = gpu.Dev(a), gpu.Dev(b), gpu.Dev(nbytes=6 * a.nbytes)
    assert testhooks().ofdis_test_div_sqrt(da.ptr, db.ptr, o.ptr, a.size, None) == 0
    gpu.check(gpu.lib().ofdis_sync(None))
    return o.get((6, a.size))


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
