"""Developer probe: HIP-event time of ofdis_batch_build_pyramids_u8 (the `e2e.build_pyramids` stage of bench.py) alone.

    [OFDIS_LIB=of_dis_amd/lib/ab_NAME/libofdis_hip.so] python tools/pyr_probe.py [pairs] [reps]
Prints the time per call, the algorithmic bytes (8-bit frames in once, level planes out) and the fraction of the HBM peak,
plus a checksum of the planes so that variants can be compared for identical output.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

W, H = 1024, 436
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
p = oppoint(2, W, H, noc=1, usetvref=True, verbosity=0)
g = torch.Generator(device=dev).manual_seed(7)
ia = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device=dev, generator=g)
ib = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device=dev, generator=g)
ts = torch.cuda.Stream(device=dev)
batch = capi.Batch(p, B)
torch.cuda.synchronize()
for _ in range(2):
    batch.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), W, H, ts.cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ts)
for _ in range(reps):
    batch.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), W, H, ts.cuda_stream)
e1.record(ts)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
byt = 2 * B * W * H
for l in range(p.sc_l, p.sc_f + 1):
    th, tw, _ = p.plane_shape(l)
    byt += B * 4 * th * tw * 4
batch.run(ts.cuda_stream)
torch.cuda.synchronize()
flow = batch.download_all()
print(f"build_pyramids {B} pairs: {ms:.4f} ms per call, {byt / 1e6:.0f} MB algorithmic, {byt / ms / 1e6:.0f} GB/s = "
      f"{byt / ms / 1e6 / 8000:.4f} of the HBM peak | flow checksum {float(abs(flow).astype('float64').sum()):.6f}")
batch.close()
