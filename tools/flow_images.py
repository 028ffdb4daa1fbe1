"""Optical flow (or stereo disparity) of image pairs from Python, through the library's batch context -- what the run_OF_* /
run_DE_* binaries do (run_dense.cpp:185-431), for a list of pairs at once and without PyTorch: decode with PIL, upload the
8-bit frames, pad + pyramid + Sobel on the device, the hot path, x 2^lv_l upsample + crop on the device, write Middlebury .flo
(.pfm with --stereo).

    python tools/flow_images.py [--rgb] [--stereo] [--op 1..4] [--fused] img1a img1b out1.flo [img2a img2b out2.flo ...]

All pairs must have one size.  --fused selects the FMA / fast-reciprocal arithmetic contract (default: the exact one, bit for
bit what run_OF_INT / run_OF_RGB write)."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint, padded_size  # noqa: E402


def load(path, channels):
    """What the binaries hand to the pipeline: B G R for the RGB ones, OpenCV's fixed-point BGR2GRAY for the gray ones
    (of_dis_amd/csrc/host/image_io.cpp)."""
    from PIL import Image
    im = Image.open(path)
    if channels == 1 and im.mode in ("L", "1", "I;16"):
        return np.ascontiguousarray(np.asarray(im.convert("L"), dtype=np.uint8))
    rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    if channels == 3:
        return np.ascontiguousarray(rgb[..., ::-1])
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def write_flo(path, flow):
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PIEH" + struct.pack("<ii", w, h))
        f.write(np.ascontiguousarray(flow, np.float32).tobytes())


def write_pfm(path, disp):  # run_dense.cpp:60-81: rows bottom-up, values negated, little endian
    h, w = disp.shape[:2]
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n-1.000000\n" % (w, h))
        f.write(np.ascontiguousarray(-disp[::-1], np.float32).tobytes())


def main(argv):
    opts = {"--rgb": False, "--stereo": False, "--fused": False}
    op = 2
    args = []
    it = iter(argv)
    for a in it:
        if a in opts:
            opts[a] = True
        elif a == "--op":
            op = int(next(it))
        else:
            args.append(a)
    if not args or len(args) % 3:
        sys.exit(__doc__)
    noc = 3 if opts["--rgb"] else 1
    trip = [args[k:k + 3] for k in range(0, len(args), 3)]
    frames_a = [load(t[0], noc) for t in trip]
    frames_b = [load(t[1], noc) for t in trip]
    h, w = frames_a[0].shape[:2]
    if any(f.shape != frames_a[0].shape for f in frames_a + frames_b):
        sys.exit("all images must have one size")
    capi.set_tuning(contract=1 if opts["--fused"] else 0)
    p = oppoint(op, w, h, noc=noc).copy(selectmode=2 if opts["--stereo"] else 1)
    p.width, p.height = padded_size(w, h, p.sc_f)
    b = capi.Batch(p, len(trip))
    da, db = capi.Dev(np.stack(frames_a)), capi.Dev(np.stack(frames_b))
    b.build_pyramids_u8(da.ptr, db.ptr, w, h)
    b.run()
    full = b.upsample(w, h)          # [pairs][h][w][2] (one channel in stereo mode)
    b.close()
    da.free()
    db.free()
    for t, f in zip(trip, full):
        if opts["--stereo"]:
            write_pfm(t[2], f[..., 0])
        else:
            write_flo(t[2], f)
        mag = np.sqrt((f.astype(np.float64) ** 2).sum(-1))
        print(f"{t[2]}: {w}x{h}, mean |flow| {mag.mean():.3f} px, max {mag.max():.2f}")


if __name__ == "__main__":
    main(sys.argv[1:])
