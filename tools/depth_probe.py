"""Developer probe: steady-state throughput of small shares with several passes in flight.

    python tools/depth_probe.py [pairs_per_pass=64] [contracts=exact,fused]

BASELINE configs[4] at 8 GPUs leaves 64 pairs per GPU: ONE pass over them is a ~0.25 ms dependency chain on a chip that is
~12 % occupied.  Frames are independent (run_dense.cpp:395 passes no initflow), so consecutive passes -- e.g. the 64-pair
shares of consecutive 512-pair batches -- can be in flight together: D contexts on D streams, pass k on slot k % D.
Prints, per depth D: ms per pass (steady state), frames/s, and whether every slot's result has the bits of the D = 1 run.
For comparison: single contexts of D x n pairs (what plain batching of the same frames gives).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import bench  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
contracts = (sys.argv[2] if len(sys.argv) > 2 else "exact,fused").split(",")
knobs = {}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    knobs[k] = int(v)
W, H = bench.WIDTH, bench.HEIGHT
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = capi.lib()
capi.check(L.ofdis_set_device(0))
p = oppoint(2, W, H, noc=1, usetvref=True, verbosity=0)
DMAX = 8
ia, ib = bench.synth_frames_range(0, DMAX * n, W, H, 1234, dev)
torch.cuda.synchronize()
out = {"pairs_per_pass": n, "knobs": knobs}


def flows(b, cnt):
    return bench.flows_tensor(capi, torch, b, p, cnt, dev).view(torch.int32)


for contract in contracts:
    old = capi.set_tuning(contract=1 if contract == "fused" else 0, **knobs)
    res = {}
    streams = [torch.cuda.Stream(device=dev) for _ in range(DMAX)]
    ctx = []
    for k in range(DMAX):
        b = capi.Batch(p, n)
        b.build_pyramids_u8(ia[k * n:].data_ptr(), ib[k * n:].data_ptr(), W, H, streams[k].cuda_stream)
        ctx.append(b)
    torch.cuda.synchronize()
    # reference bits: every slot alone
    for k in range(DMAX):
        ctx[k].run(streams[k].cuda_stream)
        torch.cuda.synchronize()
    ref = [flows(ctx[k], n).clone() for k in range(DMAX)]
    for graph in ((0, 1) if os.environ.get("DEPTH_GRAPH") else (0,)):
        for b in ctx:
            b.set_graph(graph)
        for D in [int(x) for x in os.environ.get("DEPTHS", "1,2,3,4,6,8").split(",")]:
            def loop(steps):
                for i in range(steps):
                    k = i % D
                    ctx[k].run(streams[k].cuda_stream)
            loop(4 * D)
            torch.cuda.synchronize()
            steps = 800
            dts = []
            for rep in range(3):
                t0 = time.perf_counter()
                loop(steps)
                torch.cuda.synchronize()
                dts.append((time.perf_counter() - t0) / steps)
            dt = sorted(dts)[1]
            same = all(bool(torch.equal(flows(ctx[k], n), ref[k])) for k in range(D))
            status = [ctx[k].status() for k in range(D)]
            res[f"D{D}" + ("_graph" if graph else "")] = {
                "ms_per_pass": round(dt * 1e3, 4), "frames_per_s": round(n / dt, 1), "all_ms": [round(x * 1e3, 4) for x in dts],
                "bit_identical_to_D1": same, "status": status}
    for b in ctx:
        b.close()
    # plain batching of the same frames: one context of m pairs, one stream
    for m in (() if os.environ.get("DEPTH_NOBATCH") else (n, 2 * n, 4 * n, 8 * n)):
        b = capi.Batch(p, m)
        s = streams[0].cuda_stream
        b.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), W, H, s)
        dt = bench.timed_steps(torch, lambda: b.run(s), 200, 10)
        # the first n frames against slot 0's bits
        same = bool(torch.equal(flows(b, n), ref[0]))
        res[f"batch{m}"] = {"ms_per_pass": round(dt * 1e3, 4), "frames_per_s": round(m / dt, 1), "first_slot_bits_equal": same}
        # the same context cut into 2 / 4 un-joined sub-batches (ofdis_batch_set_pipeline)
        if m >= 2 * n:
            for S in (2, 4):
                b.set_pipeline(S)
                dt = bench.timed_steps(torch, lambda: b.run(s), 200, 10)
                res[f"batch{m}_pipe{S}"] = {"ms_per_pass": round(dt * 1e3, 4), "frames_per_s": round(m / dt, 1)}
            b.set_pipeline(1)
        b.close()
    capi.restore_tuning(old)
    out[contract] = res
print(json.dumps(out, indent=1))
