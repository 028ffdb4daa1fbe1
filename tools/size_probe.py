"""Developer probe: the batched path at another frame size / operating point / channel count / mode, with the per-stage table.
    [OPP=2] [NOC=1|3] [MODE=1|2 (2 = stereo depth)] [FBCON=1] [COST=0|1|2] python tools/size_probe.py W H [pairs=1024] [contract=fused]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

W, H = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
contract = sys.argv[4] if len(sys.argv) > 4 else "fused"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
capi.check(capi.lib().ofdis_set_device(0))
capi.set_tuning(contract=1 if contract == "fused" else 0)
noc = int(os.environ.get("NOC", "1"))
p = oppoint(int(os.environ.get("OPP", "2")), W, H, noc=noc, verbosity=0).copy(selectmode=int(os.environ.get("MODE", "1")),
                                                                          usefbcon=int(os.environ.get("FBCON", "0")),
                                                                          costfct=int(os.environ.get("COST", "0")))
ia, ib = bench.synth_frames_range(0, min(n, 64), W, H, 1234, dev, channels=noc)
reps = [(n + ia.shape[0] - 1) // ia.shape[0]] + [1] * (ia.dim() - 1)
ia, ib = ia.repeat(*reps)[:n].contiguous(), ib.repeat(*reps)[:n].contiguous()
s = torch.cuda.Stream(device=dev)
b = capi.Batch(p, n)
b.set_pipeline(2 if n >= 1024 else 1)
torch.cuda.synchronize()
b.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), W, H, s.cuda_stream)
dt = bench.timed_steps(torch, lambda: b.run(s.cuda_stream), 10, 3)
b.timing(True)
b.run(s.cuda_stream)
torch.cuda.synchronize()
rows = {}
for k, name in enumerate(capi.K_NAMES):
    ms, cnt = b.kernel_time(k)
    if cnt:
        rows[name] = {"ms": round(ms, 3), "launches": cnt}
print(json.dumps({"size": [W, H], "padded": [p.width, p.height], "levels": {l: p.level_size(l) for l in range(p.sc_l, p.sc_f + 1)},
                  "pairs": n, "contract": contract, "channels": noc, "opp": int(os.environ.get("OPP", "2")), "selectmode": p.selectmode, "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(n / dt, 1),
                  "kernels": rows}))
