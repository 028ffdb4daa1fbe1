"""Developer probe: throughput when the per-GPU batch is split into S sub-batches on S HIP streams
(kernels of different levels / sub-batches can then overlap).  python tools/stream_probe.py B S [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

B, S = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
capi.check(capi.lib().ofdis_set_device(0))
p = oppoint(2, 1024, 436, verbosity=0)
per = B // S
ia, ib = bench.synth_frames_torch(per, 1024, 436, 1234, dev)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(S)]
batches = []
for s in streams:
    b = capi.Batch(p, per)
    b.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), 1024, 436, s.cuda_stream)
    batches.append(b)
torch.cuda.synchronize()
for _ in range(2):
    for b, s in zip(batches, streams):
        b.run(s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for b, s in zip(batches, streams):
        b.run(s.cuda_stream)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"B={B} streams={S} per={per}: {B * steps / el:.0f} frames/s, {el / steps * 1e3:.3f} ms/step")
