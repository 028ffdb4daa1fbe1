"""-m gpu: every HIP kernel against the oracle on the same seeded inputs, through the C ABI.

Bar: bit-exact.  The path is fp32 with separately rounded operations in the reference's order, and the
oracle (C restatement, pinned bit-for-bit to the reference compiled in place -- test_oracle_vs_ref.py)
is switched to the 64-lane butterfly reduction order the kernels use.
"""
import numpy as np
import pytest

from common import assert_bits_equal, div_sqrt_test, rand_planes, synth_case, wave_sum_test

pytestmark = pytest.mark.gpu
_f32 = np.float32

# ad-hoc campaigns: OFDIS_TEST_SEED_OFFSET=<n> shifts every seeded random draw below
_SEED_OFFSET = int(__import__("os").environ.get("OFDIS_TEST_SEED_OFFSET", "0"))


def test_wave_sum_order(gpu, orc):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((37, 64)) * 100).astype(_f32)
    got = wave_sum_test(gpu, x)
    for r in range(x.shape[0]):
        part = x[r].copy()
        o = 1
        while o < 64:
            part[0::2 * o] = part[0::2 * o] + part[o::2 * o]
            o *= 2
        assert np.all(got[r] == part[0]), (r, got[r][:4], part[0])


def test_trimmed_div_sqrt(gpu):
    """ofdis_dev.h div_rn / sqrt_rn == the compiler's IEEE expansion on their stated operand ranges, and == numpy's
    correctly rounded fp32 results (what the reference's x86 divss / sqrtss produce)."""
    rng = np.random.default_rng(5)
    n = 1 << 22
    # log-uniform magnitudes over 2^-40 .. 2^40, random signs, random mantissas
    def draw():
        m = rng.integers(0, 1 << 23, n, dtype=np.uint32)
        e = rng.integers(127 - 40, 127 + 40, n, dtype=np.uint32)
        sgn = rng.integers(0, 2, n, dtype=np.uint32)
        return ((sgn << 31) | (e << 23) | m).view(_f32)
    a, b = draw(), draw()
    # edge mantissas (all ones / all zeros / one off), zeros, inf and NaN numerators, zero and inf denominators
    edge = np.array([1.0, 1.9999999, 1.0000001, 1.5, 3.0, 0.33333334, 0.1, 0.01, 1e-6, 16777215.0], _f32)
    a[:100] = np.repeat(edge, 10)
    b[:100] = np.tile(edge, 10)
    a[100:108] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 0.0]
    b[100:108] = [3.0, 3.0, 2.0, 2.0, 2.0, np.inf, 0.0, 0.0]
    got = div_sqrt_test(gpu, a, b)
    with np.errstate(all="ignore"):
        q = (a / b).astype(_f32)
        r = np.sqrt(np.abs(a)).astype(_f32)
    for name, x, y in [("div vs compiler", got[0], got[1]), ("div vs numpy", got[0], q),
                       ("sqrt vs compiler", got[2], got[3]), ("sqrt vs numpy", got[2], r)]:
        same = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
        assert same.all(), (name, int((~same).sum()), a[~same][:4], b[~same][:4], x[~same][:4], y[~same][:4])
    # the fused TV kernel's quotient (shared refined reciprocal, no v_div_fixup): finite numerators and normal finite
    # denominators only; compared by value (a zero may carry the other sign)
    fin = np.isfinite(a) & np.isfinite(b) & (b != 0)
    assert np.array_equal(got[4][fin], q[fin]), int((got[4][fin] != q[fin]).sum())
    # ... and its quotient by a root
    pos = np.isfinite(a) & (np.abs(a) >= 2.0 ** -96) & np.isfinite(b)
    with np.errstate(all="ignore"):
        qs = (b / r).astype(_f32)
    assert np.array_equal(got[5][pos], qs[pos]), int((got[5][pos] != qs[pos]).sum())


@pytest.mark.parametrize("w,h,noc", [(128, 56, 1), (64, 28, 1), (32, 14, 1), (30, 17, 3), (67, 33, 1), (5, 4, 1)])
def test_image_warp(gpu, orc, w, h, noc):
    rng = np.random.default_rng(1)
    B = 3
    src = rand_planes(rng, B, noc, h, w, scale=50)
    wx = rand_planes(rng, B, h, w, scale=3)
    wy = rand_planes(rng, B, h, w, scale=3)
    wx[0, 0, :4] = [-0.0, 0.0, 1.0, -1.0]          # exact-integer and border cases
    wx[0, 1, :3] = [w + 5.0, -w - 5.0, 0.5]
    wy[0, 2, :3] = [h + 5.0, -h - 5.0, h - 1.0]
    dst, mask = gpu.image_warp(src, wx, wy)
    for b in range(B):
        rd, rm = orc.image_warp(src[b], wx[b], wy[b])
        assert_bits_equal(dst[b], rd.reshape(noc, h, w), f"warp dst frame {b}")
        assert_bits_equal(mask[b], rm, f"warp mask frame {b}")


@pytest.mark.parametrize("w,h,noc", [(128, 56, 1), (64, 28, 1), (240, 136, 3), (132, 40, 1)])
def test_image_warp_smooth_flow(gpu, orc, w, h, noc):
    """A smooth flow (what the densified flow of a level looks like): away from the borders the four pixels of a thread
    have consecutive tap columns in the same two source rows and the gray kernel takes its vector-load path (four loads
    instead of sixteen gathers); near the borders, at the wrap of the flow's integer part and for RGB the general path.
    Both in one image, and the integer-valued and out-of-image flows of the random test on top."""
    rng = np.random.default_rng(11)
    B = 3
    src = rand_planes(rng, B, noc, h, w, scale=50)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    wx = np.stack([(3.0 * np.sin(xx / 37.0 + b) + 1.5 * np.cos(yy / 23.0) + 0.37) for b in range(B)]).astype(_f32)
    wy = np.stack([(2.0 * np.cos(xx / 41.0 - b) - 1.0 * np.sin(yy / 29.0) - 0.21) for b in range(B)]).astype(_f32)
    wx[1, 5, 8:16] = [0.0, 1.0, -1.0, 2.0, 0.5, 0.5, 0.5, 0.5]     # a thread with integer flows, one with a constant flow
    wx[2, 3, :4] = -7.0                                           # left border: clamped taps, zero mask
    wy[2, h - 1, 4:12] = 0.75                                     # last row: the lower taps are clamped
    dst, mask = gpu.image_warp(src, wx, wy)
    for b in range(B):
        rd, rm = orc.image_warp(src[b], wx[b], wy[b])
        assert_bits_equal(dst[b], rd.reshape(noc, h, w), f"warp dst frame {b}")
        assert_bits_equal(mask[b], rm, f"warp mask frame {b}")
    assert mask.mean() > 0.9


@pytest.mark.parametrize("w,h,noc", [(128, 56, 1), (32, 14, 1), (30, 17, 3), (67, 33, 1), (5, 4, 1), (40, 5, 1)])
def test_get_derivatives(gpu, orc, w, h, noc):
    rng = np.random.default_rng(2)
    B = 2
    im1 = rand_planes(rng, B, noc, h, w, scale=60)
    im2 = rand_planes(rng, B, noc, h, w, scale=60)
    got = gpu.get_derivatives(im1, im2)
    for b in range(B):
        ref = orc.get_derivatives(im1[b], im2[b])
        assert_bits_equal(got[b], ref, f"derivatives frame {b}")


def _tv_inputs(rng, B, noc, h, w):
    mask = (rng.random((B, h, w)) > 0.1).astype(_f32)
    wx = rand_planes(rng, B, h, w, scale=2)
    wy = rand_planes(rng, B, h, w, scale=2)
    du = rand_planes(rng, B, h, w, scale=0.3)
    dv = rand_planes(rng, B, h, w, scale=0.3)
    derivs = rand_planes(rng, B, 8, noc, h, w, scale=20)
    return mask, wx, wy, du, dv, derivs


def _oracle_system(orc, p_alpha, p_gamma, p_delta, mask, wx, wy, du, dv, derivs):
    qa = _f32(0.25) * _f32(p_alpha)
    hg = _f32(p_gamma) * _f32(0.5) / _f32(3.0)
    hd = _f32(p_delta) * _f32(0.5) / _f32(3.0)
    sh, sv = orc.compute_smoothness(wx + du, wy + dv, qa)
    sys5 = orc.compute_data(mask, du, dv, derivs, hd, hg)
    b1 = orc.sub_laplacian(sys5[3], wx, sh, sv)
    b2 = orc.sub_laplacian(sys5[4], wy, sh, sv)
    return np.stack([sys5[0], sys5[1], sys5[2], b1, b2, sh, sv])


@pytest.mark.parametrize("w,h,noc", [(128, 56, 1), (32, 14, 1), (30, 17, 3), (67, 33, 1), (5, 4, 1), (2, 2, 1)])
def test_tv_system(gpu, orc, w, h, noc):
    rng = np.random.default_rng(3)
    B = 2
    mask, wx, wy, du, dv, derivs = _tv_inputs(rng, B, noc, h, w)
    got = gpu.tv_system(mask, wx, wy, du, dv, derivs, 10.0, 10.0, 5.0)
    for b in range(B):
        ref = _oracle_system(orc, 10.0, 10.0, 5.0, mask[b], wx[b], wy[b], du[b], dv[b], derivs[b])
        assert_bits_equal(got[b], ref, f"tv_system frame {b}")


def _spd_system(rng, B, h, w):
    a11 = (rng.random((B, h, w)) * 5 + 0.5).astype(_f32)
    a22 = (rng.random((B, h, w)) * 5 + 0.5).astype(_f32)
    a12 = ((rng.random((B, h, w)) - 0.5) * 0.8).astype(_f32)
    b1 = rand_planes(rng, B, h, w)
    b2 = rand_planes(rng, B, h, w)
    sh = (rng.random((B, h, w)) * 3 + 0.1).astype(_f32)
    sv = (rng.random((B, h, w)) * 3 + 0.1).astype(_f32)
    sh[:, :, -1] = 0            # what compute_smoothness guarantees (opticalflow_aux.c:154,163)
    sv[:, -1, :] = 0
    return np.stack([a11, a12, a22, b1, b2, sh, sv], 1)


@pytest.mark.parametrize("w,h,iters", [(128, 56, 3), (64, 28, 3), (32, 14, 3), (20, 15, 3), (40, 30, 1), (80, 60, 2),
                                        (128, 64, 4), (7, 5, 3), (2, 2, 3), (1, 5, 2), (30, 17, 5), (16, 70, 3),
                                        (33, 3, 3), (120, 68, 3), (37, 129, 2), (50, 300, 3), (24, 700, 1),
                                        (12, 1030, 2)])
def test_sor_coupled(gpu, orc, w, h, iters):
    rng = np.random.default_rng(4)
    B = 5
    sys = _spd_system(rng, B, h, w)
    du = rand_planes(rng, B, h, w, scale=0.2)
    dv = rand_planes(rng, B, h, w, scale=0.2)
    gu, gv = gpu.sor_coupled(du, dv, sys, iters, 1.6)
    for b in range(B):
        ru, rv, _, _, _ = orc.sor_coupled(du[b], dv[b], sys[b, 0], sys[b, 1], sys[b, 2], sys[b, 3], sys[b, 4],
                                          sys[b, 5], sys[b, 6], iters, 1.6)
        assert_bits_equal(gu[b], ru, f"sor du frame {b}")
        assert_bits_equal(gv[b], rv, f"sor dv frame {b}")


def test_sor_arbitrary_border_weights(gpu, orc):
    """sh/sv NOT zeroed on the last column/row: the reference still ignores the missing neighbours."""
    rng = np.random.default_rng(5)
    B, h, w = 2, 14, 32
    sys = _spd_system(rng, B, h, w)
    sys[:, 5, :, -1] = 0.7
    sys[:, 6, -1, :] = 0.9
    du = rand_planes(rng, B, h, w, scale=0.2)
    dv = rand_planes(rng, B, h, w, scale=0.2)
    gu, gv = gpu.sor_coupled(du, dv, sys, 3, 1.6)
    for b in range(B):
        ru, rv, _, _, _ = orc.sor_coupled(du[b], dv[b], *[sys[b, k] for k in range(7)], 3, 1.6)
        assert_bits_equal(gu[b], ru, "sor du")
        assert_bits_equal(gv[b], rv, "sor dv")


@pytest.mark.parametrize("size,opp,tv", [((1024, 436), 2, 1), ((640, 480), 2, 1), ((1024, 436), 1, 0)])
def test_patchgrid_levels(gpu, orc, size, opp, tv):
    p, pa, pb, _, _ = synth_case(size[0], size[1], 1234, 1, opp, tv)
    prev = None
    for l in range(p.sc_f, p.sc_l - 1, -1):
        rp, rflow = orc.patchgrid_level(p, l, pa[0][l], pa[1][l], pa[2][l], pb[0][l], prev)
        gp, gflow = gpu.patchgrid_level(p, l, pa[0][l][None], pa[1][l][None], pa[2][l][None], pb[0][l][None],
                                        None if prev is None else prev[None])
        assert_bits_equal(gp[0], rp, f"patch displacements level {l}")
        assert_bits_equal(gflow[0], rflow, f"dense flow level {l}")
        prev = rflow


@pytest.mark.parametrize("size", [(1024, 436), (640, 480)])
def test_varref_levels(gpu, orc, size, tv_variant):
    p, pa, pb, _, _ = synth_case(size[0], size[1], 1234, 1, 2, 1)
    rng = np.random.default_rng(6)
    for l in range(p.sc_f, p.sc_l - 1, -1):
        w, h = p.level_size(l)
        flow = rand_planes(rng, h, w, 2, scale=1.5)
        ref = orc.varref_level(p, l, pa[0][l], pb[0][l], flow)
        got = gpu.varref_level(p, l, pa[0][l][None], pb[0][l][None], flow[None])
        assert_bits_equal(got[0], ref, f"varref level {l}")


@pytest.mark.parametrize("alpha,gamma,delta,innerit,solverit,sor", [
    (10.0, 10.0, 0.0, 1, 3, 1.6),     # no brightness term (opticalflow_aux.c:352 branch not taken)
    (3.0, 0.0, 7.5, 2, 2, 1.9),       # no gradient term weight
    (25.0, 4.0, 1.0, 1, 1, 1.0),      # one sweep, plain Gauss-Seidel
    (1e-13, 10.0, 5.0, 1, 3, 1.6),    # weight below the fused kernel's range: the unfused path must take over
])
def test_varref_parameter_variants(gpu, orc, alpha, gamma, delta, innerit, solverit, sor, tv_variant):
    p, pa, pb, _, _ = synth_case(320, 240, 77, 1, 2, 1)
    p.tv_alpha, p.tv_gamma, p.tv_delta = alpha, gamma, delta
    p.tv_innerit, p.tv_solverit, p.tv_sor = innerit, solverit, sor
    rng = np.random.default_rng(8)
    for l in range(p.sc_f, p.sc_l - 1, -1):
        w, h = p.level_size(l)
        flow = rand_planes(rng, h, w, 2, scale=1.5)
        ref = orc.varref_level(p, l, pa[0][l], pb[0][l], flow)
        got = gpu.varref_level(p, l, pa[0][l][None], pb[0][l][None], flow[None])
        assert_bits_equal(got[0], ref, f"varref level {l}")


@pytest.mark.parametrize("w,h", [(38, 38), (64, 64), (56, 56), (24, 24), (16, 16), (40, 38), (36, 40), (17, 33), (64, 60)])
def test_varref_square_and_near_square_levels(gpu, orc, w, h, tv_variant):
    """The fused TV kernel's fill phase once produced an all-zero system (0/0) for the lane whose first fill pixel is a
    last-column pixel of the last image row -- exactly when w == h -- and 0 * NaN then poisoned that row."""
    import gen_synth
    from of_dis_amd.params import oppoint
    ia, ib, _ = gen_synth.make_pair(w, h, 5, 1)
    rng = np.random.default_rng(w * 100 + h)
    for innerit, solverit in ((1, 1), (2, 3)):
        p = oppoint(2, w, h).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4, tv_innerit=innerit, tv_solverit=solverit)
        p.width, p.height = w, h
        pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
        flow = rand_planes(rng, h, w, 2, scale=1.0)
        ref = orc.varref_level(p, 0, pa[0][0], pb[0][0], flow)
        got = gpu.varref_level(p, 0, pa[0][0][None], pb[0][0][None], flow[None])
        assert_bits_equal(got[0], ref, f"varref {w}x{h} innerit={innerit} solverit={solverit}")


@pytest.mark.parametrize("knobs", [{}, {"finish_fusion": 0}, {"fused_strip": 2}, {"prep_band_rows": 6}, {"fused_tall_group": 0}])
@pytest.mark.parametrize("w,h", [(120, 68), (128, 128), (16, 65), (100, 127), (64, 96), (17, 70), (128, 66), (96, 100),
                                 (156, 48), (192, 64), (129, 30), (256, 16), (200, 100), (193, 65), (255, 128), (160, 5),
                                 (135, 240), (64, 200), (100, 129), (128, 256), (20, 192), (150, 193),
                                 (120, 72), (40, 80), (90, 96), (33, 66), (60, 81)])
def test_varref_levels_of_65_to_128_rows(gpu, orc, w, h, knobs):
    """Levels wider than two wavefronts (the finest level of a 1242 x 375 KITTI pair at operating point 2 is 156 x 48: the
    row-marching warp + derivatives kernel with three / four wavefronts side by side) and
    levels taller than a wavefront has lanes (the finest level of a 1080p / 4K gray pair at operating point 2 is 120 x 68; a
    portrait 1080 x 1920 pair gives 135 x 240):
    the fused TV kernel with TWO to FOUR wavefronts per strip (ofdis_fused_tall.hip) -- rows 64 q .. 64 q + 63, every lane shift
    at a wavefront boundary bridged by an LDS mailbox -- and the row-marching warp + derivatives kernel in front of it must give the bits of
    the reference, for one / two / three sweeps, with and without the brightness term, for several frames (strips), with the
    separate finish kernel, and with the warp kernel cut into row bands."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(w * 1000 + h)
    # (65 ... 96 rows: several strips per workgroup with one shared tail wavefront -- 10 frames = a full group of 7 and a short one, or five strips of two)
    nfr = 10 if 64 < h <= 96 else 4
    pairs = [gen_synth.make_pair(w, h, 300 + k, 1) for k in range(2)]
    old = gpu.set_tuning(**knobs)
    try:
        for innerit, solverit, delta in ((1, 3, 5.0), (2, 1, 0.0), (3, 2, 5.0)):
            p = oppoint(2, w, h).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4, tv_innerit=innerit, tv_solverit=solverit, tv_delta=delta)
            p.width, p.height = w, h
            pyr = [(orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)) for ia, ib, _ in pairs]
            flows = [rand_planes(rng, h, w, 2, scale=(1.5, 0.3, 5.0, 1.0)[f % 4]) for f in range(nfr)]
            flows[2][h - 1, w - 1] = (2.5 * w, -2.5 * h)  # far outside the image: mask 0, clamped taps
            refs = [orc.varref_level(p, 0, pyr[f % 2][0][0][0], pyr[f % 2][1][0][0], flows[f]) for f in range(nfr)]
            got = gpu.varref_level(p, 0, np.stack([pyr[f % 2][0][0][0] for f in range(nfr)]),
                                   np.stack([pyr[f % 2][1][0][0] for f in range(nfr)]), np.stack(flows))
            for f in range(nfr):
                assert_bits_equal(got[f], refs[f], f"{w}x{h} innerit={innerit} solverit={solverit} delta={delta} {knobs} frame {f}")
    finally:
        gpu.restore_tuning(old)


@pytest.mark.parametrize("seed", range(48))
def test_random_varref_levels(gpu, orc, seed, tv_variant):
    """Random level geometry (mostly in the fused kernel's range: gray, 2 <= h <= 64, w >= 16; some outside it and some
    RGB), random TV parameters and a random incoming flow with out-of-image displacements."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(12000 + seed + _SEED_OFFSET)
    noc = 3 if seed % 6 == 5 else 1
    w = int(rng.integers(16, 140)) if seed % 5 else int(rng.integers(5, 16))
    h = int(rng.integers(4, 65)) if seed % 7 else int(rng.integers(65, 150))
    if seed % 9 == 0:
        h = w = int(rng.integers(16, 65))
    ia, ib, _ = gen_synth.make_pair(w, h, 12100 + seed, noc)
    p = oppoint(2, w, h, noc=noc).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4,
                                       tv_innerit=int(rng.integers(1, 4)), tv_solverit=int(rng.integers(1, 5)),
                                       tv_sor=float(rng.choice([1.0, 1.6, 1.95])), tv_alpha=float(rng.choice([1.0, 10.0, 40.0])),
                                       tv_gamma=float(rng.choice([0.0, 10.0, 20.0])), tv_delta=float(rng.choice([0.0, 5.0, 15.0])))
    p.width, p.height = w, h
    pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
    flow = rand_planes(rng, h, w, 2, scale=float(rng.choice([0.2, 1.5, 6.0])))
    flow[rng.integers(0, h), rng.integers(0, w)] = (3.0 * w, -3.0 * h)      # far outside: mask 0, clamped taps
    ref = orc.varref_level(p, 0, pa[0][0], pb[0][0], flow)
    got = gpu.varref_level(p, 0, pa[0][0][None], pb[0][0][None], flow[None])
    assert_bits_equal(got[0], ref, f"seed {seed}: {w}x{h} noc={noc} innerit={p.tv_innerit} solverit={p.tv_solverit}")


@pytest.mark.parametrize("w,h,nfr", [(300, 40, 2), (512, 224, 1), (600, 100, 3), (260, 64, 4), (257, 65, 2), (700, 17, 2), (1000, 256, 1)])
def test_wide_gray_levels_on_the_fused_kernels_by_records(gpu, orc, w, h, nfr):
    """Gray levels of more than 256 columns (the finest level of operating points 3 / 4): too wide for the row-marching warp +
    derivatives kernel, so -- in contexts of 16 (fused contract) / 512 (exact contract) frames and more, forced here -- the tiled warp kernel and the derivatives
    kernel in its record form feed the fused system + SOR kernels (one wavefront per frame up to 64 rows, two to four up to
    256): the bits of the per-stage kernels."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(34000 + w + h)
    p = oppoint(2, w, h, noc=1).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4, tv_innerit=int(rng.integers(1, 4)),
                                     tv_solverit=int(rng.integers(1, 4)), tv_delta=float(rng.choice([0.0, 5.0])))
    p.width, p.height = w, h
    ims_a, ims_b, flows, refs = [], [], [], []
    for k in range(nfr):
        ia, ib, _ = gen_synth.make_pair(w, h, 34100 + k + w, 1)
        pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
        flow = rand_planes(rng, h, w, 2, scale=1.5)
        ims_a.append(pa[0][0]); ims_b.append(pb[0][0]); flows.append(flow)
        refs.append(orc.varref_level(p, 0, pa[0][0], pb[0][0], flow))
    for knobs in ({"fused_rgb_min": 1}, {"fused_rgb_min": 1, "fused_tall_group": 0}, {}):
        old = gpu.set_tuning(**knobs)
        try:
            got = gpu.varref_level(p, 0, np.stack(ims_a), np.stack(ims_b), np.stack(flows))
        finally:
            gpu.restore_tuning(old)
        for k in range(nfr):
            assert_bits_equal(got[k], refs[k], f"{w}x{h} gray frame {k} {knobs} innerit={p.tv_innerit} solverit={p.tv_solverit}")


@pytest.mark.parametrize("knobs", [{}, {"fused_tall_group": 0}, {"fused_tall_group": 7}])
@pytest.mark.parametrize("w,h,nfr", [(120, 68, 3), (100, 127, 2), (64, 96, 4), (240, 136, 1), (60, 200, 2), (16, 65, 5), (130, 256, 1),
                                     (33, 80, 7)])
def test_tall_rgb_levels_on_the_fused_kernel(gpu, orc, w, h, nfr, knobs):
    """RGB levels of 65 ... 256 rows (the finest level of a 1920x1080 colour pair at operating point 2 is 120 x 68):
    tv_fused_tall_kernel with three derivative record arrays -- two to four wavefronts per strip, or heads + one shared tail
    wavefront -- gives the bits of n_inner x (tv_system + block SOR)."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(35000 + 3 * w + h)
    p = oppoint(2, w, h, noc=3).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4, tv_innerit=int(rng.integers(1, 4)),
                                     tv_solverit=int(rng.integers(1, 4)), tv_delta=float(rng.choice([0.0, 5.0])))
    p.width, p.height = w, h
    ims_a, ims_b, flows, refs = [], [], [], []
    for k in range(min(nfr, 3)):
        ia, ib, _ = gen_synth.make_pair(w, h, 35100 + k + w, 3)
        pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
        flow = rand_planes(rng, h, w, 2, scale=1.5)
        flow[rng.integers(0, h), rng.integers(0, w)] = (3.0 * w, -3.0 * h)
        ims_a.append(pa[0][0]); ims_b.append(pb[0][0]); flows.append(flow)
        refs.append(orc.varref_level(p, 0, pa[0][0], pb[0][0], flow))
    pick = [k % len(refs) for k in range(nfr)]
    old = gpu.set_tuning(fused_rgb_min=1, **knobs)
    try:
        got = gpu.varref_level(p, 0, np.stack([ims_a[k] for k in pick]), np.stack([ims_b[k] for k in pick]),
                               np.stack([flows[k] for k in pick]))
    finally:
        gpu.restore_tuning(old)
    for slot, k in enumerate(pick):
        assert_bits_equal(got[slot], refs[k], f"{w}x{h} rgb slot {slot} {knobs} innerit={p.tv_innerit} solverit={p.tv_solverit}")


@pytest.mark.parametrize("seed", range(36))
def test_rgb_levels_on_the_fused_system_and_solver(gpu, orc, seed):
    """RGB levels of at most 64 rows on the fused system + SOR kernel (ofdis_tuning.fused_rgb_min = 1 forces it for these
    one- to five-frame contexts; by default contexts of 16 / 512 frames and more -- fused / exact contract -- take it): the warp kernel, the derivatives kernel
    in its record form (three arrays of 8-float records in the diag layout, zeroed by the mask) and tv_fused_kernel with the
    RGB data term give the bits of n_inner x (tv_system + SOR) -- random geometry (16 <= w < 140, 4 <= h <= 64), TV
    parameters with and without the brightness term, 1-3 sweeps, a flow with out-of-image displacements."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(33000 + seed + _SEED_OFFSET)
    w = int(rng.integers(16, 140))
    h = int(rng.integers(4, 65)) if seed % 6 else 64
    nfr = int(rng.integers(1, 6))
    p = oppoint(2, w, h, noc=3).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4,
                                     tv_innerit=int(rng.integers(1, 4)), tv_solverit=int(rng.integers(1, 4)),
                                     tv_sor=float(rng.choice([1.0, 1.6, 1.95])), tv_alpha=float(rng.choice([1.0, 10.0, 40.0])),
                                     tv_gamma=float(rng.choice([0.0, 10.0, 20.0])), tv_delta=float(rng.choice([0.0, 5.0, 15.0])))
    p.width, p.height = w, h
    ims_a, ims_b, flows, refs = [], [], [], []
    for k in range(nfr):
        ia, ib, _ = gen_synth.make_pair(w, h, 33100 + 7 * seed + k, 3)
        pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
        flow = rand_planes(rng, h, w, 2, scale=float(rng.choice([0.2, 1.5, 6.0])))
        flow[rng.integers(0, h), rng.integers(0, w)] = (3.0 * w, -3.0 * h)
        ims_a.append(pa[0][0]); ims_b.append(pb[0][0]); flows.append(flow)
        refs.append(orc.varref_level(p, 0, pa[0][0], pb[0][0], flow))
    for knobs in ({"fused_tp_pipe": 0}, {"fused_tp_pipe": 2}):   # a wavefront per frame / per fixed-point iteration
        old = gpu.set_tuning(fused_rgb_min=1, **knobs)
        try:
            got = gpu.varref_level(p, 0, np.stack(ims_a), np.stack(ims_b), np.stack(flows))
        finally:
            gpu.restore_tuning(old)
        for k in range(nfr):
            assert_bits_equal(got[k], refs[k], f"seed {seed}: {w}x{h} rgb frame {k} {knobs} innerit={p.tv_innerit} solverit={p.tv_solverit} delta={p.tv_delta}")
    plain = gpu.varref_level(p, 0, np.stack(ims_a), np.stack(ims_b), np.stack(flows))   # (one frame: the per-stage kernels)
    for k in range(nfr):
        assert_bits_equal(plain[k], refs[k], f"seed {seed}: per-stage kernels, frame {k}")


@pytest.mark.parametrize("w,h,solverit", [(59, 59, 3), (102, 57, 2), (121, 76, 2), (62, 58, 2), (200, 130, 1), (16, 4, 3)])
def test_one_fixed_point_iteration_reads_no_scratch(gpu, orc, monkeypatch, w, h, solverit):
    """A level with ONE fixed-point iteration (tv_innerit = 1 at level 0) on the fused path never writes the du / dv array,
    and its sweep requests rows past its last column: those requests must not reach memory (round 6: they did, and a NaN
    pattern left there by an earlier allocation turned the last column into NaN -- found by a seed campaign, flaky by
    nature).  Contexts start from zeroed scratch; with OFDIS_POISON_SCRATCH=1 they start from NaN patterns instead, which
    is how this test (and, as a campaign, the whole GPU suite) shows that no result depends on either."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(31000 + w * h)
    ia, ib, _ = gen_synth.make_pair(w, h, 31100 + w, 1)
    p = oppoint(2, w, h, noc=1).copy(sc_f=0, sc_l=0, p_samp_s=4, imgpadding=4, tv_innerit=1, tv_solverit=solverit)
    p.width, p.height = w, h
    pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
    flow = rand_planes(rng, h, w, 2, scale=1.5)
    ref = orc.varref_level(p, 0, pa[0][0], pb[0][0], flow)
    monkeypatch.setenv("OFDIS_POISON_SCRATCH", "1")
    b = gpu.Batch(p, 2)
    assert np.isnan(b.download_all()).all(), "the poison hook is not live"
    b.close()
    for _ in range(2):
        got = gpu.varref_level(p, 0, pa[0][0][None], pb[0][0][None], flow[None])
        assert_bits_equal(got[0], ref, f"{w}x{h}, one fixed-point iteration, poisoned scratch")
    monkeypatch.delenv("OFDIS_POISON_SCRATCH")
    b = gpu.Batch(p, 2)
    assert not b.download_all().any(), "a context's scratch starts zeroed"
    b.close()
    assert_bits_equal(gpu.varref_level(p, 0, pa[0][0][None], pb[0][0][None], flow[None])[0], ref, "zeroed scratch")


@pytest.mark.parametrize("seed", range(32))
def test_random_patchgrid_levels(gpu, orc, seed):
    """Random patch size / overlap / iteration limits / cost function at one level with a random coarse flow that
    sends some patches out of bounds at the start and others over the outlier threshold."""
    import gen_synth
    from of_dis_amd.params import oppoint
    rng = np.random.default_rng(13000 + seed + _SEED_OFFSET)
    noc = 3 if seed % 4 == 3 else 1
    P = int(rng.choice([4, 8, 8, 8, 12, 6]))
    w, h = int(rng.integers(3 * P, 120)), int(rng.integers(3 * P, 90))
    w, h = w - w % 2, h - h % 2                                              # a coarser level of half the size exists
    ia, ib, _ = gen_synth.make_pair(w, h, 13100 + seed, noc)
    p = oppoint(2, w, h, noc=noc).copy(sc_f=0, sc_l=0, p_samp_s=P, imgpadding=P, patove=float(rng.choice([0.0, 0.4, 0.75])),
                                       max_iter=int(rng.integers(1, 14)), costfct=int(rng.integers(0, 3)),
                                       patnorm=int(rng.integers(0, 2)), res_thresh=float(rng.choice([0.0, 2.0])),
                                       dp_thresh=float(rng.choice([0.05, 0.3])), dr_thresh=float(rng.choice([0.95, 0.6])))
    p.min_iter = int(rng.integers(0, p.max_iter + 1))
    p.width, p.height = w, h
    pa, pb = orc.build_pyramid(p, ia), orc.build_pyramid(p, ib)
    prev = rand_planes(rng, h // 2, w // 2, 2, scale=float(rng.choice([0.3, 2.0, 8.0])))
    prev[0, 0] = (-2.0 * w, 0.0)
    prev[-1, -1] = (w, h)
    rp, rflow = orc.patchgrid_level(p, 0, pa[0][0], pa[1][0], pa[2][0], pb[0][0], prev)
    gp, gflow = gpu.patchgrid_level(p, 0, pa[0][0][None], pa[1][0][None], pa[2][0][None], pb[0][0][None], prev[None])
    assert_bits_equal(gp[0], rp, f"seed {seed}: patch displacements {w}x{h} P={P} noc={noc}")
    assert_bits_equal(gflow[0], rflow, f"seed {seed}: dense flow")


def test_streams_events_and_page_locked_copies(gpu):
    """include/ofdis.h version 3: asynchronous copies between page-locked host memory and the device on streams created through
    the ABI, ordered across streams by events -- the pieces of the sequence driver's upload | kernels | download pipeline."""
    L = gpu.lib()
    n = 1 << 22
    rng = np.random.default_rng(7)
    src = gpu.HostBuf((n,), np.float32)
    dst = gpu.HostBuf((n,), np.float32)
    src.array[:] = rng.standard_normal(n).astype(np.float32)
    dst.array[:] = 0
    d0, d1 = gpu.Dev(nbytes=4 * n), gpu.Dev(nbytes=4 * n)
    s_in, s_k, s_out = gpu.Stream(), gpu.Stream(), gpu.Stream()
    up, done = gpu.Event(), gpu.Event()
    for rep in range(3):
        gpu.check(L.ofdis_memcpy_h2d_async(d0.ptr, src.ptr, 4 * n, s_in.ptr))
        up.record(s_in)
        up.wait(s_k)                                                   # the "kernel" stream waits for the upload ...
        gpu.check(L.ofdis_memcpy_d2d(d1.ptr, d0.ptr, 4 * n, s_k.ptr))
        done.record(s_k)
        done.wait(s_out)                                               # ... the download stream for the kernel stream
        gpu.check(L.ofdis_memcpy_d2h_async(dst.ptr, d1.ptr, 4 * n, s_out.ptr))
        down = gpu.Event()
        down.record(s_out)
        down.sync()
        assert np.array_equal(dst.array, src.array), f"round {rep}"
        src.array[:] = src.array[::-1].copy()
        down.close()
    never = gpu.Event()                                                # a never-recorded event is complete
    never.wait(s_k)
    never.sync()
    s_k.sync()
    assert L.ofdis_event_record(None, s_k.ptr) != 0 and L.ofdis_memcpy_h2d_async(None, src.ptr, 4, s_in.ptr) != 0
    for o in (up, done, never, s_in, s_k, s_out):
        o.close()
    src.free()
    dst.free()
