"""Latency of the drop-in call ofdis_flow() (host pyramids in, host flow out), one pair at a time.
    [NOC=1|3] [OPP=2] [MODE=1|2] python tools/flow_latency.py [W H] [contract=exact]
    (default 1024 436, gray, operating point 2, optical flow; the library's default contract)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from common import synth_case
from of_dis_amd import capi
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 436)
contract = sys.argv[3] if len(sys.argv) > 3 else "exact"
noc, opp, mode = int(os.environ.get("NOC", "1")), int(os.environ.get("OPP", "2")), int(os.environ.get("MODE", "1"))
capi.set_tuning(contract=1 if contract == "fused" else 0)
p, pa, pb, _, _ = synth_case(W, H, 1234, noc, opp, 1)
p = p.copy(selectmode=mode)
if mode == 2:
    pa, pb = pb, pa   # negative horizontal motion: what the left camera's constraint admits
for _ in range(3):
    capi.flow(p, pa[0], pa[1], pa[2], pb[0])
n = 50 if opp <= 2 else 10
t0 = time.perf_counter()
for _ in range(n):
    capi.flow(p, pa[0], pa[1], pa[2], pb[0])
print(f"ofdis_flow {W}x{H} channels {noc} op-{opp} mode {mode}, {contract} contract, levels {[p.level_size(l) for l in range(p.sc_l, p.sc_f + 1)]}: "
      f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (incl. the ctypes marshalling of the binding)")
