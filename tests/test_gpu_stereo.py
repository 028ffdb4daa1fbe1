"""Stereo-depth mode (the reference's run_DE_* binaries, compile-time SELECTMODE=2; ofdis_params.selectmode = 2):
one horizontal displacement per patch / pixel, constrained to <= 0 for the left camera.  The checker is the reference
itself compiled in that mode (oracle/_ref/libofdis_ref_de_*.so); the C restatement covers optical flow only."""
import numpy as np
import pytest

import oracle
from common import assert_bits_equal, rand_planes, synth_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["per-stage kernels", "fused stereo kernel"])
def stereo_tv(gpu, request):
    """Every test of this file twice: with the refinement on de_system + de_sor per fixed-point iteration (what exact-contract
    contexts of fewer than 256 frames run by default) and with levels of at most 64 rows forced onto de_fused_kernel (all
    iterations in one launch; ofdis_tuning.fused_rgb_min = 1)."""
    old = gpu.set_tuning(fused_rgb_min=1 if request.param.startswith("fused") else (1 << 30))
    yield request.param
    gpu.restore_tuning(old)


def _case(w, h, seed, noc, opp, tv):
    p, pa, pb, _, _ = synth_case(w, h, seed, noc, opp, tv)
    p = p.copy(selectmode=2)
    # The synthetic pair moves by about (+6, -3) px, so matching the SECOND image against the first sees a negative
    # horizontal displacement, which is what the left camera's constraint (<= 0) admits.  The vertical component
    # only makes the 1-D search work harder -- parity, not accuracy, is what is tested.
    return p, pb, pa


def _ref(noc):
    kind = "de_int" if noc == 1 else "de_rgb"
    R = oracle.need_ref(kind, True)
    if R is None:
        pytest.skip("comparison against the compiled reference skipped: neither /root/reference nor oracle/_ref exists here")
    return R


@pytest.mark.parametrize("noc,opp", [(1, 2), (3, 3), (1, 3), (3, 2)])
def test_stereo_patchgrid_levels(gpu, noc, opp):
    p, pa, pb = _case(640, 480, 90, noc, opp, 0)
    R = _ref(noc)
    prev = None
    for l in range(p.sc_f, p.sc_l - 1, -1):
        rp, rflow = R.patchgrid_level(p, l, pa[0][l], pa[1][l], pa[2][l], pb[0][l], prev)
        gp, gflow = gpu.patchgrid_level(p, l, pa[0][l][None], pa[1][l][None], pa[2][l][None], pb[0][l][None],
                                        prev[None] if prev is not None else None)
        assert rflow.shape[-1] == 1
        assert (rp[:, 0] <= 0).all() and (rp[:, 1] == 0).all()     # left camera: disparity <= 0 (patch.cpp:190)
        assert_bits_equal(gp[0], rp, f"stereo patch displacements level {l}")
        assert_bits_equal(gflow[0], rflow, f"stereo dense displacement level {l}")
        prev = rflow


@pytest.mark.parametrize("noc,size,opp,solverit", [(1, (640, 480), 2, 0), (1, (1024, 436), 2, 0), (3, (320, 240), 3, 0),
                                                   (1, (333, 251), 3, 0), (1, (600, 300), 3, 5), (1, (1242, 375), 2, 5),
                                                   (1, (640, 480), 2, 1), (1, (520, 1100), 3, 2)])
def test_stereo_varref_levels(gpu, noc, size, opp, solverit):
    """Levels of at most 64 rows (several frames per wavefront), taller ones (one workgroup per frame, the wavefronts
    in lock step: operating point 3 ends at half resolution), and solver sweep counts on either side of what one pass
    pipelines (4, or 3 above 64 rows)."""
    p, pa, pb = _case(size[0], size[1], 91, noc, opp, 1)
    if solverit:
        p = p.copy(tv_solverit=solverit)
    R = _ref(noc)
    rng = np.random.default_rng(4)
    for l in range(p.sc_f, p.sc_l - 1, -1):
        w, h = p.level_size(l)
        flow = -np.abs(rand_planes(rng, h, w, 1, scale=1.5))
        flow[::7, ::5] = 0.3                                        # some positive values: clamped by the update
        ref = R.varref_level(p, l, pa[0][l], pb[0][l], flow)
        got = gpu.varref_level(p, l, pa[0][l][None], pb[0][l][None], flow[None])
        assert_bits_equal(got[0], ref, f"stereo varref level {l}")


@pytest.mark.parametrize("size,noc,opp,tv", [((1024, 436), 1, 2, 1), ((640, 480), 1, 2, 0), ((333, 251), 1, 1, 1),
                                             ((320, 240), 3, 3, 1), ((333, 251), 1, 3, 1), ((200, 160), 1, 4, 0), ((320, 240), 3, 2, 1)])
def test_stereo_flow_bit_exact(gpu, size, noc, opp, tv):
    p, pa, pb = _case(size[0], size[1], 92, noc, opp, tv)
    R = _ref(noc)
    ref = R.flow(p, pa[0], pa[1], pa[2], pb[0])
    assert ref.shape[-1] == 1 and (ref <= 0).all()
    got = gpu.flow(p, pa[0], pa[1], pa[2], pb[0])
    assert_bits_equal(got, ref, "stereo displacement vs reference sources (SELECTMODE=2)")
    S = oracle.need_ref("de_int", False) if noc == 1 else None
    if S is not None:
        seq = S.flow(p, pa[0], pa[1], pa[2], pb[0])
        assert np.abs(seq - got).mean() * (1 << p.sc_l) < 1e-3      # any summation order lands within the tolerance
    b = gpu.Batch(p, 3)
    for k in range(3):
        b.upload(k, pa[0], pa[1], pa[2], pb[0])
    b.run()
    out = b.download_all()
    full = b.upsample(size[0], size[1])
    b.close()
    assert out.shape[-1] == 1 and full.shape == (3, size[1], size[0], 1)
    for k in range(3):
        assert_bits_equal(out[k], ref, f"batch frame {k}")


@pytest.mark.parametrize("size,noc,opp,tv", [((640, 480), 1, 2, 1), ((333, 251), 1, 1, 0), ((320, 240), 3, 3, 1)])
def test_stereo_forward_backward(gpu, size, noc, opp, tv):
    """usefbcon in stereo mode: the backward grid is the right camera (displacement >= 0, patch.cpp:191-192,
    refine_variational.cpp:308-315), merged one-channel splats (patchgrid.cpp:366-371)."""
    p, pa, pb = _case(size[0], size[1], 93, noc, opp, tv)
    p = p.copy(usefbcon=1)
    R = _ref(noc)
    ref = R.flow(p, pa[0], pa[1], pa[2], pb[0], pyr_b_dx=pb[1], pyr_b_dy=pb[2])
    plain = R.flow(p.copy(usefbcon=0), pa[0], pa[1], pa[2], pb[0])
    assert not np.array_equal(ref, plain)
    got = gpu.flow(p, pa[0], pa[1], pa[2], pb[0], pyr_b_dx=pb[1], pyr_b_dy=pb[2])
    assert_bits_equal(got, ref, "stereo + usefbcon vs reference sources")
