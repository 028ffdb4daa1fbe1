"""Developer helper: one ofdis_flow call per repetition with a debug build of the cross-CU fused TV kernel that prints, per
workgroup, when it started (100 MHz clock), how long its prologue took, its total time, and its stalls / polls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from of_dis_amd import capi
from common import synth_case
p, pa, pb, _, _ = synth_case(1024, 436, 1600, 1, 2, 1)
for rep in range(3):
    print("--- call", rep, flush=True)
    capi.flow(p, pa[0], pa[1], pa[2], pb[0])
