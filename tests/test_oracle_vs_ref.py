"""CPU: pins the oracle.  The reference ships no tests or vectors, so the plain-C
restatement (oracle/ofdis_oracle.c) is checked bit-for-bit against the reference sources themselves,
compiled in place by oracle/Makefile (oracle/_ref) -- function by function and end to end, in both
reduction orders.  Needs /root/reference or a prebuilt oracle/_ref."""
import numpy as np
import pytest

import oracle
from common import assert_bits_equal, rand_planes, synth_case

_f32 = np.float32
# (skipped only on a machine with neither the reference sources nor a shipped build of them; a library missing where the
# build is expected fails in oracle.need_ref)
pytestmark = pytest.mark.skipif(not oracle.ref_expected(), reason="neither /root/reference nor oracle/_ref exists here")


@pytest.fixture(scope="module", params=[False, True], ids=["seq", "wave64"])
def pair(request):
    O = oracle.c_oracle()
    O.set_reduce_order(request.param)
    yield O, oracle.ref("int", request.param)
    O.set_reduce_order(False)


@pytest.fixture(scope="module")
def pair_rgb():
    O = oracle.c_oracle()
    O.set_reduce_order(False)
    return O, oracle.ref("rgb", False)


SHAPES = [(128, 56), (32, 14), (30, 17), (67, 33), (5, 4), (9, 5)]


@pytest.mark.parametrize("w,h", SHAPES)
def test_image_warp(pair, w, h):
    O, R = pair
    rng = np.random.default_rng(10)
    src = rand_planes(rng, 1, h, w, scale=50)
    wx, wy = rand_planes(rng, h, w, scale=3), rand_planes(rng, h, w, scale=3)
    wx[0, :3] = [0.0, w + 3.0, -w - 3.0]
    for a, b in zip(O.image_warp(src, wx, wy), R.image_warp(src, wx, wy)):
        assert_bits_equal(a.reshape(b.shape), b, "image_warp")


@pytest.mark.parametrize("w,h", SHAPES)
def test_derivatives_smoothness_data_laplacian(pair, w, h):
    O, R = pair
    rng = np.random.default_rng(11)
    im1, im2 = rand_planes(rng, 1, h, w, scale=60), rand_planes(rng, 1, h, w, scale=60)
    d = O.get_derivatives(im1, im2)
    assert_bits_equal(d, R.get_derivatives(im1, im2), "get_derivatives")
    uu, vv = rand_planes(rng, h, w, scale=2), rand_planes(rng, h, w, scale=2)
    qa = _f32(2.5)
    sh, sv = O.compute_smoothness(uu, vv, qa)
    rh, rv = R.compute_smoothness(uu, vv, qa)
    assert_bits_equal(sh, rh, "smooth_horiz")
    assert_bits_equal(sv, rv, "smooth_vert")
    mask = (rng.random((h, w)) > 0.2).astype(_f32)
    du, dv = rand_planes(rng, h, w, scale=0.3), rand_planes(rng, h, w, scale=0.3)
    hd, hg = _f32(5.0) * _f32(0.5) / _f32(3), _f32(10.0) * _f32(0.5) / _f32(3)
    s5 = O.compute_data(mask, du, dv, d, hd, hg)
    assert_bits_equal(s5, R.compute_data(mask, du, dv, d, hd, hg), "compute_data")
    assert_bits_equal(O.compute_data(mask, du, dv, d, 0.0, hg), R.compute_data(mask, du, dv, d, 0.0, hg),
                      "compute_data (delta=0 branch)")
    assert_bits_equal(O.sub_laplacian(s5[3], uu, sh, sv), R.sub_laplacian(s5[3], uu, sh, sv), "sub_laplacian")


@pytest.mark.parametrize("w,h,iters,slow", [(128, 56, 3, False), (32, 14, 1, False), (30, 17, 5, False),
                                             (7, 5, 2, False), (2, 2, 3, False), (1, 6, 2, False), (6, 1, 2, False),
                                             (20, 15, 3, True)])
def test_sor_coupled(pair, w, h, iters, slow):
    O, R = pair
    rng = np.random.default_rng(12)
    a11 = (rng.random((h, w)) * 5 + 0.5).astype(_f32)
    a22 = (rng.random((h, w)) * 5 + 0.5).astype(_f32)
    a12 = ((rng.random((h, w)) - 0.5) * 0.8).astype(_f32)
    b1, b2 = rand_planes(rng, h, w), rand_planes(rng, h, w)
    sh = (rng.random((h, w)) * 3 + 0.1).astype(_f32)
    sv = (rng.random((h, w)) * 3 + 0.1).astype(_f32)
    if (w * h) % 2 == 0:      # also exercise weights that are NOT zeroed on the last column / row
        sh[:, -1] = 0
        sv[-1, :] = 0
    du, dv = rand_planes(rng, h, w, scale=0.2), rand_planes(rng, h, w, scale=0.2)
    got = O.sor_coupled(du, dv, a11, a12, a22, b1, b2, sh, sv, iters, 1.6, slow=slow)
    ref = R.sor_coupled(du, dv, a11, a12, a22, b1, b2, sh, sv, iters, 1.6, slow=slow)
    names = ["du", "dv", "a11(inv)", "a12(inv)", "a22(inv)"]
    for n, a, b in zip(names[:2] if (slow or w < 2 or h < 2) else names, got, ref):
        assert_bits_equal(a, b, f"sor_coupled {n}")


def test_rgb_kernels(pair_rgb):
    O, R = pair_rgb
    rng = np.random.default_rng(13)
    w, h = 30, 17
    src = rand_planes(rng, 3, h, w, scale=50)
    wx, wy = rand_planes(rng, h, w, scale=3), rand_planes(rng, h, w, scale=3)
    for a, b in zip(O.image_warp(src, wx, wy), R.image_warp(src, wx, wy)):
        assert_bits_equal(a.reshape(b.shape), b, "rgb image_warp")
    im2 = rand_planes(rng, 3, h, w, scale=50)
    d = O.get_derivatives(src, im2)
    assert_bits_equal(d, R.get_derivatives(src, im2), "rgb get_derivatives")
    mask = (rng.random((h, w)) > 0.2).astype(_f32)
    du, dv = rand_planes(rng, h, w, scale=0.3), rand_planes(rng, h, w, scale=0.3)
    assert_bits_equal(O.compute_data(mask, du, dv, d, 0.8, 1.6), R.compute_data(mask, du, dv, d, 0.8, 1.6),
                      "rgb compute_data")


@pytest.mark.parametrize("size,opp,tv,cost", [((1024, 436), 2, 1, 0), ((640, 480), 2, 1, 0), ((1024, 436), 1, 0, 0),
                                               ((320, 240), 2, 1, 1), ((320, 240), 2, 1, 2), ((256, 128), 3, 1, 0)])
def test_levels_and_full_flow(pair, size, opp, tv, cost):
    O, R = pair
    p, pa, pb, _, _ = synth_case(size[0], size[1], 1234, 1, opp, tv)
    p = p.copy(costfct=cost)
    prev = None
    for l in range(p.sc_f, p.sc_l - 1, -1):
        op_, oflow = O.patchgrid_level(p, l, pa[0][l], pa[1][l], pa[2][l], pb[0][l], prev)
        rp_, rflow = R.patchgrid_level(p, l, pa[0][l], pa[1][l], pa[2][l], pb[0][l], prev)
        assert_bits_equal(op_, rp_, f"patch displacements level {l}")
        assert_bits_equal(oflow, rflow, f"dense flow level {l}")
        if tv:
            assert_bits_equal(O.varref_level(p, l, pa[0][l], pb[0][l], oflow),
                              R.varref_level(p, l, pa[0][l], pb[0][l], rflow), f"varref level {l}")
        prev = rflow
    assert_bits_equal(O.flow(p, pa[0], pa[1], pa[2], pb[0]), R.flow(p, pa[0], pa[1], pa[2], pb[0]), "OFClass")


def test_full_flow_rgb(pair_rgb):
    O, R = pair_rgb
    p, pa, pb, _, _ = synth_case(320, 240, 77, 3, 3, 1)
    p = p.copy(costfct=1, max_iter=8, min_iter=8)
    assert_bits_equal(O.flow(p, pa[0], pa[1], pa[2], pb[0]), R.flow(p, pa[0], pa[1], pa[2], pb[0]), "OFClass rgb L1")


def test_reduction_order_sensitivity_is_tiny():
    """The two valid 'Eigen' summation orders differ by ~1e-5 px; the bar is 1e-3."""
    p, pa, pb, _, _ = synth_case(1024, 436, 1234, 1, 2, 1)
    a = oracle.ref("int", False).flow(p, pa[0], pa[1], pa[2], pb[0])
    b = oracle.ref("int", True).flow(p, pa[0], pa[1], pa[2], pb[0])
    mean, mx, frac = oracle.epe_stats(a, b)
    s = 1 << p.sc_l
    assert mean * s < 1e-4 and mx * s < 1e-2, (mean * s, mx * s, frac)


def test_block_world_inputs_cpu():
    """The second input family (hard edges, occlusions, saturation): the C restatement against the reference sources
    compiled in place, both reduction orders, bit for bit."""
    import gen_synth
    from of_dis_amd.params import oppoint
    ia, ib = gen_synth.make_pair_blocks(320, 240, 21)
    p = oppoint(2, 320, 240)
    O = oracle.c_oracle()
    pa, pb = O.build_pyramid(p, ia), O.build_pyramid(p, ib)
    for wave64 in (False, True):
        assert oracle.need_ref("int", wave64) is not None
        O.set_reduce_order(wave64)
        a = O.flow(p, pa[0], pa[1], pa[2], pb[0])
        b = oracle.ref("int", wave64).flow(p, pa[0], pa[1], pa[2], pb[0])
        assert np.array_equal(a, b), ("reduce order", wave64)
    O.set_reduce_order(False)
