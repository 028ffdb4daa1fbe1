"""Flow mode beside the stereo-depth mode (ofdis_params.selectmode = 2, the reference's run_DE_* binaries) on the
same resident gray pairs: frames/s and the per-stage device times.  usage: stereo_probe.py [W H [N [contract]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from of_dis_amd import capi
from of_dis_amd.params import oppoint

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1242, 375)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
contract = sys.argv[4] if len(sys.argv) > 4 else "fused"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
capi.check(capi.lib().ofdis_set_device(0))
capi.set_tuning(contract=1 if contract == "fused" else 0)
for mode in (1, 2):
    p = oppoint(int(os.environ.get("OPP", "2")), W, H, noc=1, verbosity=0).copy(selectmode=mode)
    ia, ib = bench.synth_frames_range(0, min(n, 64), W, H, 1234, dev)
    reps = (n + ia.shape[0] - 1) // ia.shape[0]
    ia, ib = ia.repeat(reps, 1, 1)[:n].contiguous(), ib.repeat(reps, 1, 1)[:n].contiguous()
    s = torch.cuda.Stream(device=dev)
    b = capi.Batch(p, n)
    torch.cuda.synchronize()
    b.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), W, H, s.cuda_stream)
    dt = bench.timed_steps(torch, lambda: b.run(s.cuda_stream), 10, 3)
    b.timing(True)
    b.run(s.cuda_stream)
    torch.cuda.synchronize()
    rows = {name: round(b.kernel_time(k)[0], 3) for k, name in enumerate(capi.K_NAMES) if b.kernel_time(k)[1]}
    print(f"selectmode {mode} {W}x{H} {n} pairs {contract}: {dt * 1e3:.3f} ms/step  {n / dt:,.0f} frames/s  {rows}", flush=True)
    b.close()
