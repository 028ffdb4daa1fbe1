"""Roofline of the image_warp kernel alone (measure on launches that move >> 10 MB):
python tools/warp_bench.py  ->  achieved GB/s of ofdis_image_warp (row-major, packed planes) for a few shapes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from of_dis_amd import capi  # noqa: E402

L = capi.lib()
dev = torch.device("cuda", 0)
for name, (B, noc, h, w) in {"op-2 level 3, 4096 pairs (gray)": (4096, 1, 56, 128), "op-2 level 3, 512 pairs": (512, 1, 56, 128),
                             "1080p RGB level 1, 64 pairs": (64, 3, 544, 960), "1080p RGB level 0, 16 pairs": (16, 3, 1088, 1920),
                             "1080p gray level 0, 64 pairs": (64, 1, 1088, 1920)}.items():
    g = torch.Generator(device=dev).manual_seed(1)
    src = torch.rand((B, noc, h, w), device=dev, generator=g) * 255
    # a smooth displacement field (what densification + refinement produce), a few pixels in magnitude
    yy, xx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32),
                            indexing="ij")
    ph = torch.rand((B, 1, 1), device=dev, generator=g) * 6.28
    wx = 3.0 * torch.sin(xx / 37.0 + ph) + 1.5 * torch.cos(yy / 23.0) + 0.37
    wy = 2.0 * torch.cos(xx / 41.0 - ph) - 1.0 * torch.sin(yy / 29.0) - 0.21
    if len(sys.argv) > 1 and sys.argv[1] == "random":
        wx = torch.randn((B, h, w), device=dev, generator=g) * 2
        wy = torch.randn((B, h, w), device=dev, generator=g) * 2
    dst = torch.empty_like(src)
    mask = torch.empty_like(wx)
    s = torch.cuda.Stream()
    args = (dst.data_ptr(), mask.data_ptr(), src.data_ptr(), wx.data_ptr(), wy.data_ptr(), w, h, noc, B, s.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(3):
        capi.check(L.ofdis_image_warp(*args))
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        capi.check(L.ofdis_image_warp(*args))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    byt = B * h * w * (8 + 4 * noc + 4 * noc + 4)
    print(f"{name}: {byt / 1e6:.0f} MB, {dt * 1e6:.1f} us, {byt / dt / 1e9:.0f} GB/s = {byt / dt / 8e12 * 100:.0f} % of 8 TB/s")
