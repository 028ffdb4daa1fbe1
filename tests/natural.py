"""Natural images for the parity tests.

Every other parity input of this repository is synthetic (tools/gen_synth.py).  The build image happens to carry a few
photographs inside its Python packages -- scikit-image's data directory (among them a real Middlebury 2014 stereo pair,
`motorcycle_left.png` / `motorcycle_right.png`, with its ground-truth disparity) and scikit-learn's two sample JPEGs.  They
are NOT part of this repository and not of the reference: the tests read them where they lie if they are there (the GPU
box runs the same image) and skip otherwise; nothing is copied.  tests/golden/natural.json holds the SHA-256 of each input
as decoded here and of the reference build's output for it (made by tests/golden/make_natural.py), so that the CPU
restatement is pinned on photographs as well wherever the same files decode to the same bytes.
"""
import glob
import hashlib
import os

import numpy as np


def _dirs():
    out = []
    for pat in ("/opt/conda/lib/python3*/site-packages/skimage/data", "/usr/lib/python3*/site-packages/skimage/data",
                "/usr/local/lib/python3*/dist-packages/skimage/data", "/usr/local/lib/python3*/dist-packages/sklearn/datasets/images",
                "/opt/conda/lib/python3*/site-packages/sklearn/datasets/images"):
        out += sorted(glob.glob(pat))
    return out


def find(name):
    for d in _dirs():
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    return None


def load_rgb(name):
    """(h, w, 3) uint8, R G B, or None when the file or a decoder is missing."""
    p = find(name)
    if p is None:
        return None
    try:
        from PIL import Image
    except Exception:
        return None
    return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8))


def to_channels(rgb, channels):
    """What the run_OF_* binaries hand to the pipeline for a colour file: B G R for the RGB binaries, OpenCV's fixed-point
    BGR2GRAY for the gray ones (of_dis_amd/csrc/host/image_io.cpp; run_dense.cpp:204-222 via cv::imread)."""
    if channels == 3:
        return np.ascontiguousarray(rgb[..., ::-1])
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def similarity_warp(img, angle_deg, scale, tx, ty):
    """The second frame of a pair made from ONE photograph: img resampled (bilinear, float64, replicate border) under a
    rotation + scale about the centre + translation; returns (second image uint8, ground-truth flow (h, w, 2) f32 of the
    first frame's pixels)."""
    h, w = img.shape[:2]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    a = np.deg2rad(angle_deg)
    c, s = np.cos(a) * scale, np.sin(a) * scale
    # forward map of frame-1 pixels: p2 = R (p1 - centre) + centre + t
    fx = c * (xx - cx) - s * (yy - cy) + cx + tx
    fy = s * (xx - cx) + c * (yy - cy) + cy + ty
    flow = np.stack([fx - xx, fy - yy], -1).astype(np.float32)
    # frame 2 at pixel q shows frame 1 at R^-1 (q - centre - t) + centre
    det = c * c + s * s
    qx, qy = xx - cx - tx, yy - cy - ty
    sx = (c * qx + s * qy) / det + cx
    sy = (-s * qx + c * qy) / det + cy
    sx, sy = np.clip(sx, 0, w - 1), np.clip(sy, 0, h - 1)
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    ax, ay = sx - x0, sy - y0
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[..., None]
    out = ((1 - ay) * (1 - ax))[..., None] * src[y0, x0] + ((1 - ay) * ax)[..., None] * src[y0, x1] + \
        (ay * (1 - ax))[..., None] * src[y1, x0] + (ay * ax)[..., None] * src[y1, x1]
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return (out[..., 0] if img.ndim == 2 else out), flow


# name -> how to make the pair; every entry yields (first, second) as RGB uint8 arrays plus optional ground truth
def pair(name):
    """Returns (rgb_a, rgb_b, truth) or None.  truth: dict with 'flow' (h,w,2) or 'disparity' (h,w; inf = unknown)."""
    if name == "motorcycle":
        a, b = load_rgb("motorcycle_left.png"), load_rgb("motorcycle_right.png")
        if a is None or b is None:
            return None
        truth = {}
        p = find("motorcycle_disp.npz")
        if p is not None:
            truth["disparity"] = np.load(p)["arr_0"].astype(np.float32)
        return a, b, truth
    src = {"china": ("china.jpg", 1.2, 1.015, 4.3, -2.6), "astronaut": ("astronaut.png", -2.0, 0.99, -6.4, 3.2),
           "coffee": ("coffee.png", 0.8, 1.0, 9.5, 1.25), "chelsea": ("chelsea.png", -0.6, 1.02, -3.1, -4.7)}.get(name)
    if src is None:
        return None
    a = load_rgb(src[0])
    if a is None:
        return None
    b, flow = similarity_warp(a, *src[1:])
    return a, b, {"flow": flow}


PAIRS = ("motorcycle", "china", "astronaut", "coffee", "chelsea")


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
