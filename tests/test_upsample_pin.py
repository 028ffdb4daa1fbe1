"""The step after the path (run_dense.cpp:406-414): flow x 2^lv_l, cv::resize(INTER_LINEAR) by 2^lv_l, crop.

OpenCV is not installed here, so the restatement (oracle_upsample_crop, and the device kernel behind
ofdis_batch_upsample) is pinned two ways that do not go through it:

 1. HAND VECTORS from OpenCV's documented resize rule (imgproc resize.cpp, INTER_LINEAR, float data): for a
    destination column dx the source coordinate is fx = (dx + 0.5) * (src_w / dst_w) - 0.5; sx = floor(fx); fx -= sx;
    sx < 0 -> (sx, fx) = (0, 0); sx >= src_w - 1 -> (sx, fx) = (src_w - 1, 0); the row is interpolated horizontally
    (S[sx] * (1 - fx) + S[sx + 1] * fx), then two such rows vertically with the same rule.  The expected values below
    are computed with exact rational arithmetic (fractions.Fraction) on integer-valued sources and power-of-two scales,
    where every product and sum is exactly representable in fp32 -- so any correct implementation must reproduce them
    BIT FOR BIT, whatever its rounding order.  The four borders (left, right, top, bottom), the corners, odd crops and
    the interior are covered.
 2. An independent library implementation of the same convention: torch.nn.functional.interpolate(mode="bilinear",
    align_corners=False), documented as OpenCV-compatible, on random data, to within a few ulp (its rounding order
    differs).
"""
from fractions import Fraction

import numpy as np
import pytest

from of_dis_amd.params import oppoint, padded_size

_f32 = np.float32


def _axis_rule(dst_n, src_n, scale):
    """(index, index+1 clamped, fraction) per destination coordinate, exactly (OpenCV resize.cpp, INTER_LINEAR)."""
    out = []
    for d in range(dst_n):
        f = (Fraction(2 * d + 1, 2)) / scale - Fraction(1, 2)
        s = f.numerator // f.denominator  # floor
        f -= s
        if s < 0:
            s, f = 0, Fraction(0)
        if s >= src_n - 1:
            s, f = src_n - 1, Fraction(0)
        out.append((s, min(s + 1, src_n - 1), f))
    return out


def _hand_upsample(flow, scale, left, top, wo, ho):
    """Exact expected full-resolution flow (Fractions), values scaled by `scale` first (run_dense.cpp:409)."""
    sh, sw, _ = flow.shape
    xs = _axis_rule(sw * scale, sw, scale)
    ys = _axis_rule(sh * scale, sh, scale)
    out = np.empty((ho, wo, 2), dtype=object)
    for y in range(ho):
        sy, sy1, fy = ys[y + top]
        for x in range(wo):
            sx, sx1, fx = xs[x + left]
            for c in range(2):
                v = lambda yy, xx: Fraction(int(flow[yy, xx, c])) * scale
                r0 = v(sy, sx) * (1 - fx) + v(sy, sx1) * fx
                r1 = v(sy1, sx) * (1 - fx) + v(sy1, sx1) * fx
                out[y, x, c] = r0 * (1 - fy) + r1 * fy
    return out


def _exact_f32(frac_array):
    a = np.array([[[float(v) for v in px] for px in row] for row in frac_array], dtype=np.float64)
    f = a.astype(_f32)
    assert np.array_equal(f.astype(np.float64), a), "hand vector not exactly representable in fp32: choose smaller integers"
    return f


# (width_org, height_org, finest level lv_l): level 3 = x8 (operating points 1, 2), level 2 = x4 (op 3), level 1 = x2
# (op 4); odd sizes are padded to a multiple of 2^level and cropped asymmetrically
PIN_CASES = [(64, 32, 3), (61, 27, 3), (40, 24, 2), (37, 21, 2), (19, 13, 1)]


def _case(w, h, lv, seed):
    p = oppoint(2, w, h)
    p.sc_f = p.sc_l = lv          # only the finest level and the padded size matter for this step
    p.width, p.height = padded_size(w, h, lv)
    assert p.sc_l == lv and (p.width, p.height) != (w, h) or (w % (1 << lv) == 0 and h % (1 << lv) == 0)
    sw, sh = p.width >> p.sc_l, p.height >> p.sc_l
    rng = np.random.default_rng(seed)
    flow = rng.integers(-9, 10, size=(sh, sw, 2)).astype(_f32)      # small integers: exact arithmetic throughout
    left, top = (p.width - w) // 2, (p.height - h) // 2
    return p, flow, left, top


@pytest.mark.parametrize("w,h,lv", PIN_CASES)
def test_oracle_upsample_matches_hand_vectors(orc, w, h, lv):
    p, flow, left, top = _case(w, h, lv, 31 + w)
    expect = _exact_f32(_hand_upsample(flow, 1 << p.sc_l, left, top, w, h))
    got = orc.upsample_crop(p, flow, w, h)
    assert got.shape == expect.shape
    assert np.array_equal(got.view(np.uint32), expect.view(np.uint32)), np.abs(got - expect).max()
    s = 1 << p.sc_l
    if left == 0 and top == 0:  # the documented border rule, spelled out: half a source pixel repeats the border value
        assert np.array_equal(got[0, : s // 2], np.repeat(flow[:1, 0] * s, s // 2, 0))


def test_oracle_upsample_matches_torch_convention(orc):
    """Random (non-integer) data against torch's align_corners=False bilinear, the OpenCV-compatible convention."""
    import torch
    import torch.nn.functional as F
    for (w, h, op) in [(1024, 436, 2), (333, 251, 1), (320, 240, 3)]:
        p = oppoint(op, w, h)
        sw, sh = p.width >> p.sc_l, p.height >> p.sc_l
        s = 1 << p.sc_l
        flow = (np.random.default_rng(w).standard_normal((sh, sw, 2)) * 3).astype(_f32)
        got = orc.upsample_crop(p, flow, w, h)
        t = torch.from_numpy(flow * _f32(s)).permute(2, 0, 1)[None]
        ref = F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        left, top = (p.width - w) // 2, (p.height - h) // 2
        ref = ref[top:top + h, left:left + w]
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        assert err.max() <= 4 * np.spacing(_f32(np.abs(ref).max())), (w, h, err.max())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,lv", PIN_CASES)
def test_device_upsample_matches_hand_vectors(gpu, w, h, lv):
    """ofdis_batch_upsample on flows planted in the context's result buffer: the same exact hand vectors."""
    p, flow, left, top = _case(w, h, lv, 31 + w)
    expect = _exact_f32(_hand_upsample(flow, 1 << p.sc_l, left, top, w, h))
    b = gpu.Batch(p, 2)
    L = gpu.lib()
    both = np.ascontiguousarray(np.stack([flow, -flow]))
    gpu.check(L.ofdis_memcpy_h2d(b.flow_ptr(), both.ctypes.data, both.nbytes))
    full = b.upsample(w, h)
    b.close()
    assert np.array_equal(full[0].view(np.uint32), expect.view(np.uint32))
    assert np.array_equal(full[1], -expect)  # by value: the negated plant holds -0.0, whose sign a sum may drop
