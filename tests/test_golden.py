"""Golden vectors produced by the reference itself (tests/golden/make_golden.py, run where
/root/reference exists).  CPU part: the oracle restatement reproduces them bit-for-bit, so the oracle
stays pinned on boxes without the reference.  GPU part (-m gpu): the HIP path reproduces them."""
import glob
import os

import numpy as np
import pytest

import oracle
from common import assert_bits_equal
from of_dis_amd.params import OfdisParams

HERE = os.path.dirname(os.path.abspath(__file__))
_f32 = np.float32
FLOW_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "op*.npz")))


def load_case(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    p = OfdisParams()
    for n, v in zip(z["param_names"], z["params"]):
        setattr(p, str(n), type(getattr(p, str(n)))(v))
    return z, p


def np_pyramid(p, img_u8):
    """Independent numpy restatement of run_dense.cpp:130-178,298-311 (pad, 2x2 mean, Sobel/8, borders)."""
    img = img_u8.astype(np.float32)
    if img.ndim == 2:
        img = img[..., None]
    ho, wo = img.shape[:2]
    padw, padh = p.width - wo, p.height - ho
    img = np.pad(img, ((padh // 2, padh - padh // 2), (padw // 2, padw - padw // 2), (0, 0)), mode="edge")
    out = [[], [], []]
    for l in range(p.sc_f + 1):
        if l > 0:
            img = ((img[0::2, 0::2] + img[0::2, 1::2]) + (img[1::2, 0::2] + img[1::2, 1::2])) * _f32(0.25)
        r = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="reflect")
        gx = (r[:-2, 2:] - r[:-2, :-2]) + _f32(2) * (r[1:-1, 2:] - r[1:-1, :-2]) + (r[2:, 2:] - r[2:, :-2])
        gy = (r[2:, :-2] - r[:-2, :-2]) + _f32(2) * (r[2:, 1:-1] - r[:-2, 1:-1]) + (r[2:, 2:] - r[:-2, 2:])
        q = p.imgpadding
        out[0].append(np.pad(img, ((q, q), (q, q), (0, 0)), mode="edge"))
        out[1].append(np.pad(gx * _f32(0.125), ((q, q), (q, q), (0, 0))))
        out[2].append(np.pad(gy * _f32(0.125), ((q, q), (q, q), (0, 0))))
    return out


@pytest.mark.parametrize("name", FLOW_CASES)
def test_oracle_pyramid_matches_numpy(name):
    z, p = load_case(name)
    O = oracle.c_oracle()
    for img in (z["img_a"], z["img_b"]):
        got, ref = O.build_pyramid(p, img), np_pyramid(p, img)
        for k in range(3):
            for l in range(p.sc_f + 1):
                assert_bits_equal(got[k][l], ref[k][l], f"pyramid plane {k} level {l}")


@pytest.mark.parametrize("name", FLOW_CASES)
@pytest.mark.parametrize("order", ["seq", "w64"])
def test_oracle_reproduces_golden_flow(name, order):
    z, p = load_case(name)
    O = oracle.c_oracle()
    O.set_reduce_order(order == "w64")
    try:
        pa, pb = O.build_pyramid(p, z["img_a"]), O.build_pyramid(p, z["img_b"])
        out, levels = O.flow(p, pa[0], pa[1], pa[2], pb[0], want_levels=True)
        assert_bits_equal(out, z[f"flow_{order}"], "final flow")
        for l in range(p.sc_f, p.sc_l - 1, -1):
            assert_bits_equal(levels[l], z[f"lvl_l{l}_{order}"], f"level {l} flow")
    finally:
        O.set_reduce_order(False)


def test_oracle_reproduces_golden_kernels():
    z = np.load(os.path.join(HERE, "golden", "fdf_kernels.npz"))
    O = oracle.c_oracle()
    for t in ("int_", "rgb_"):
        src, im2, wx, wy, du, dv = (z[t + k] for k in ("src", "im2", "wx", "wy", "du", "dv"))
        noc, h, w = src.shape
        dst, mask = O.image_warp(src, wx, wy)
        assert_bits_equal(dst.reshape(noc, h, w), z[t + "warp_dst"], t + "warp")
        assert_bits_equal(mask, z[t + "warp_mask"], t + "mask")
        d = O.get_derivatives(src, im2)
        assert_bits_equal(d, z[t + "derivs"], t + "derivs")
        qa, hd, hg = _f32(2.5), _f32(5.0) * _f32(0.5) / _f32(3.0), _f32(10.0) * _f32(0.5) / _f32(3.0)
        sh, sv = O.compute_smoothness(wx + du, wy + dv, qa)
        assert_bits_equal(sh, z[t + "sh"], t + "sh")
        assert_bits_equal(sv, z[t + "sv"], t + "sv")
        s5 = O.compute_data(mask, du, dv, d, hd, hg)
        assert_bits_equal(s5, z[t + "data"], t + "data")
        b1, b2 = O.sub_laplacian(s5[3], wx, sh, sv), O.sub_laplacian(s5[4], wy, sh, sv)
        assert_bits_equal(b1, z[t + "b1"], t + "b1")
        assert_bits_equal(b2, z[t + "b2"], t + "b2")
        u, v, i11, i12, i22 = O.sor_coupled(du, dv, s5[0], s5[1], s5[2], b1, b2, sh, sv, 3, 1.6)
        assert_bits_equal(u, z[t + "sor_du"], t + "sor du")
        assert_bits_equal(v, z[t + "sor_dv"], t + "sor dv")
        assert_bits_equal(np.stack([i11, i12, i22]), z[t + "sor_inv"], t + "sor inverse")


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", FLOW_CASES)
def test_hip_reproduces_golden_flow(gpu, name):
    """u8 frames -> on-device pyramid -> hot path == the reference's output (wave64-order build)."""
    z, p = load_case(name)
    ia, ib = z["img_a"], z["img_b"]
    ho, wo = ia.shape[:2]
    b = gpu.Batch(p, 3)
    da, db = gpu.Dev(np.stack([ia] * 3)), gpu.Dev(np.stack([ib] * 3))
    b.build_pyramids_u8(da.ptr, db.ptr, wo, ho)
    b.run()
    out = b.download_all()
    for s in range(3):
        assert_bits_equal(out[s], z["flow_w64"], f"final flow (slot {s})")
    for l in range(p.sc_f, p.sc_l - 1, -1):
        assert_bits_equal(b.level_flow(l)[1], z[f"lvl_l{l}_w64"], f"level {l} flow")
    b.close()
    mean, mx, frac = oracle.epe_stats(out[0], z["flow_seq"])
    assert mean * (1 << p.sc_l) < 1e-3, (mean, mx, frac)   # north-star tolerance vs the sequential-sum build


MODE_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "mode_*.npz")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODE_CASES)
def test_hip_reproduces_golden_modes(gpu, name):
    """Forward-backward merging and stereo depth against vectors produced by the reference compiled in those modes
    (tests/golden/make_golden.py): needs neither /root/reference nor oracle/_ref."""
    z, p = load_case(name)
    ia, ib = z["img_a"], z["img_b"]
    ho, wo = ia.shape[:2]
    b = gpu.Batch(p, 2)
    da, db = gpu.Dev(np.stack([ia] * 2)), gpu.Dev(np.stack([ib] * 2))
    b.build_pyramids_u8(da.ptr, db.ptr, wo, ho)   # also exercises the second image's gradient pyramid (usefbcon)
    b.run()
    out = b.download_all()
    b.close()
    for s in range(2):
        assert_bits_equal(out[s], z["flow_w64"], f"{name} slot {s}")
    if "flow_seq" in z.files:
        d = np.sqrt(((out[0].astype(np.float64) - z["flow_seq"]) ** 2).sum(-1)).mean()
        assert d * (1 << p.sc_l) < 1e-3, d            # any summation order lands within the north-star tolerance


def test_golden_mode_vectors_are_present():
    assert len(MODE_CASES) == 4, MODE_CASES


@pytest.mark.gpu
def test_hip_reproduces_golden_kernels(gpu):
    z = np.load(os.path.join(HERE, "golden", "fdf_kernels.npz"))
    for t in ("int_", "rgb_"):
        src, im2, wx, wy, du, dv = (z[t + k] for k in ("src", "im2", "wx", "wy", "du", "dv"))
        dst, mask = gpu.image_warp(src[None], wx[None], wy[None])
        assert_bits_equal(dst[0], z[t + "warp_dst"], t + "warp")
        assert_bits_equal(mask[0], z[t + "warp_mask"], t + "mask")
        d = gpu.get_derivatives(src[None], im2[None])
        assert_bits_equal(d[0], z[t + "derivs"], t + "derivs")
        # tv_system takes alpha/gamma/delta: 4*2.5, 10, 5 reproduce the constants used for the vectors
        sys = gpu.tv_system(mask, wx[None], wy[None], du[None], dv[None], d, 10.0, 10.0, 5.0)[0]
        ref = np.concatenate([z[t + "data"][:3], z[t + "b1"][None], z[t + "b2"][None], z[t + "sh"][None], z[t + "sv"][None]])
        assert_bits_equal(sys, ref, t + "tv_system")
        u, v = gpu.sor_coupled(du[None], dv[None], sys[None], 3, 1.6)
        assert_bits_equal(u[0], z[t + "sor_du"], t + "sor du")
        assert_bits_equal(v[0], z[t + "sor_dv"], t + "sor dv")
