"""Latency of the drop-in call ofdis_flow() (host pyramids in, host flow out), one pair at a time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from common import synth_case
from of_dis_amd import capi
p, pa, pb, _, _ = synth_case(1024, 436, 1234, 1, 2, 1)
for _ in range(3):
    capi.flow(p, pa[0], pa[1], pa[2], pb[0])
n = 50
t0 = time.perf_counter()
for _ in range(n):
    capi.flow(p, pa[0], pa[1], pa[2], pb[0])
print(f"ofdis_flow 1024x436 op-2: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (incl. ~0.3 ms of ctypes marshalling)")
