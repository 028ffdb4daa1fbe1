"""Seeded synthetic frame pairs for parity tests and the benchmark (SURVEY.md 8d recipe).

texture = sum_k a_k * sigma_k * gaussian_filter(N(0,1), sigma_k), sigma=(3,8,20), a=(1,1.5,2),
generated with a 64 px margin, normalised to mean 128 / std 45, rounded+clipped to uint8.
Ground-truth flow u = 6 + 4 sin(2pi .7 y/H + .3) + 2 cos(2pi 1.1 x/W), v = -3 + 3 cos(2pi .9 x/W + 1).
Second image = cubic map_coordinates of the (float) texture at (y - v, x - u).
Frame k of a sequence uses seed 1234 + k; RGB uses three textures (seeds 3s, 3s+1, 3s+2).
"""
import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates

MARGIN = 64


def _texture(rng, h, w):
    H, W = h + 2 * MARGIN, w + 2 * MARGIN
    tex = np.zeros((H, W), np.float64)
    for sigma, amp in ((3.0, 1.0), (8.0, 1.5), (20.0, 2.0)):
        tex += amp * sigma * gaussian_filter(rng.standard_normal((H, W)), sigma, mode="wrap")
    tex = (tex - tex.mean()) / tex.std() * 45.0 + 128.0
    return tex


def gt_flow(h, w, scale=1.0):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    u = 6 + 4 * np.sin(2 * np.pi * 0.7 * y / h + 0.3) + 2 * np.cos(2 * np.pi * 1.1 * x / w)
    v = -3 + 3 * np.cos(2 * np.pi * 0.9 * x / w + 1)
    return u * scale, v * scale


def make_pair(w, h, seed=1234, channels=1, flow_scale=1.0):
    """Returns (img_a, img_b) uint8 arrays of shape (h,w) or (h,w,3), and the GT flow (h,w,2) f32."""
    u, v = gt_flow(h, w, flow_scale)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a_ch, b_ch = [], []
    for c in range(channels):
        rng = np.random.default_rng(seed if channels == 1 else 3 * seed + c)
        tex = _texture(rng, h, w)
        a = tex[MARGIN:MARGIN + h, MARGIN:MARGIN + w]
        b = map_coordinates(tex, [yy - v + MARGIN, xx - u + MARGIN], order=3, mode="nearest")
        a_ch.append(np.clip(np.rint(a), 0, 255).astype(np.uint8))
        b_ch.append(np.clip(np.rint(b), 0, 255).astype(np.uint8))
    if channels == 1:
        ia, ib = a_ch[0], b_ch[0]
    else:
        ia, ib = np.stack(a_ch, -1), np.stack(b_ch, -1)
    return ia, ib, np.stack([u, v], -1).astype(np.float32)


def make_pair_blocks(w, h, seed=1234, channels=1, nrect=40, max_shift=9.0):
    """A second input family, as unlike the band-limited textures as possible: flat-shaded rectangles with hard edges on a
    noisy gradient background, each moving with its own (fractional, up to `max_shift` px) velocity and drawn back to front,
    so the pair has occlusions, disocclusions, motion discontinuities, saturated (0 / 255) regions and objects that leave the
    frame.  Returns (img_a, img_b) uint8 arrays of shape (h,w) or (h,w,3); there is no dense ground truth."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    rects = []
    for _ in range(nrect):
        rw, rh = rng.integers(6, max(8, w // 3)), rng.integers(6, max(8, h // 3))
        x0, y0 = rng.uniform(-rw / 2, w - rw / 2), rng.uniform(-rh / 2, h - rh / 2)
        vel = rng.uniform(-max_shift, max_shift, 2)
        col = rng.choice([0.0, 255.0, *rng.uniform(20, 235, 6)], size=channels)
        rects.append((x0, y0, rw, rh, vel, col, rng.uniform(0.0, 0.5)))

    def render(t):
        out = []
        for c in range(channels):
            img = 128.0 + 60.0 * np.sin(0.037 * (xx + 3.0 * t) + c) * np.cos(0.051 * (yy - 2.0 * t))
            for x0, y0, rw, rh, vel, col, tex in rects:
                cx, cy = x0 + t * vel[0], y0 + t * vel[1]
                # anti-aliased coverage of the (sub-pixel positioned) rectangle: hard edges, one blended pixel row / column
                cov = np.clip(xx + 0.5 - cx, 0, 1) * np.clip(cx + rw - (xx - 0.5), 0, 1) * \
                    np.clip(yy + 0.5 - cy, 0, 1) * np.clip(cy + rh - (yy - 0.5), 0, 1)
                shade = col[c] * (1.0 - tex) + tex * 255.0 * (((xx - cx).astype(int) ^ (yy - cy).astype(int)) & 4 > 0)
                img = img * (1 - cov) + shade * cov
            img = img + rng_noise[c]
            out.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        return out[0] if channels == 1 else np.stack(out, -1)
    rng_noise = [rng.normal(0, 2.0, (h, w)) for _ in range(channels)]
    ia = render(0.0)
    rng_noise = [rng.normal(0, 2.0, (h, w)) for _ in range(channels)]
    ib = render(1.0)
    return ia, ib


def write_pgm(path, img):
    img = np.ascontiguousarray(img)
    with open(path, "wb") as f:
        if img.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        else:
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=436)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--out", default="synth")
    a = ap.parse_args()
    ia, ib, _ = make_pair(a.width, a.height, a.seed, a.channels)
    ext = "pgm" if a.channels == 1 else "ppm"
    write_pgm(f"{a.out}_a.{ext}", ia)
    write_pgm(f"{a.out}_b.{ext}", ib)
