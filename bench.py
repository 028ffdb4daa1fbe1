#!/usr/bin/env python3
"""Benchmark of the OF_DIS hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--tv on|off]

A "step" is one pass of the whole hot path (OFC::OFClass scope: per-level DIS search + densification
+ TV-L1 refinement, pyramids already resident in HBM) over one batch of B synthetic 1024x436 frame
pairs at operating point 2 (run_OF_INT).  One process per GPU; frames are independent, so ranks
share nothing (weak scaling: every rank processes its own batch; torch.distributed is used for the
start/stop barrier and the max-over-ranks time only).  Rank 0 prints ONE JSON line.

The JSON carries, besides the contract fields:
  roofline      the kernel class that takes the most time, algorithmic bytes / measured time
  kernels       the same figures for every kernel class (incl. the warp kernel of the north star)
  cpu_baseline  the reference CPU path (oracle/_ref, built from the reference sources) timed on one
                host core over a bounded sample of the same frames
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
WIDTH, HEIGHT = 1024, 436


def synth_frames_torch(n, w, h, seed, device):
    """n seeded band-limited 8-bit frame pairs, generated on the GPU (tools/gen_synth.py recipe:
    multi-scale Gaussian-filtered noise, smooth analytic flow, second frame by cubic back-warp)."""
    import math
    import torch
    import torch.nn.functional as F
    M = 64
    H, W = h + 2 * M, w + 2 * M
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    # Gaussian filtering as a product in the Fourier domain (periodic, like scipy's mode="wrap")
    fy = torch.fft.fftfreq(H, device=device).view(H, 1)
    fx = torch.fft.rfftfreq(W, device=device).view(1, W // 2 + 1)
    f2 = fy * fy + fx * fx
    transfer = sum(amp * sigma * torch.exp(-2.0 * math.pi ** 2 * sigma ** 2 * f2)
                   for sigma, amp in ((3.0, 1.0), (8.0, 1.5), (20.0, 2.0)))

    out_a, out_b = [], []
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    u = 6 + 4 * torch.sin(2 * math.pi * 0.7 * ys / h + 0.3) + 2 * torch.cos(2 * math.pi * 1.1 * xs / w)
    v = -3 + 3 * torch.cos(2 * math.pi * 0.9 * xs / w + 1)
    gx = (xs - u + M) / (W - 1) * 2 - 1
    gy = (ys - v + M) / (H - 1) * 2 - 1
    grid = torch.stack([gx, gy], -1)[None]
    chunk = 32
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        # (one noise field filtered by the sum of the three Gaussians: same spectrum family as the numpy
        # recipe's sum of three independently filtered fields; band-limited, |flow| <= 12 px)
        noise = torch.randn(m, 1, H, W, device=device, generator=g)
        tex = torch.fft.irfft2(torch.fft.rfft2(noise) * transfer, s=(H, W))
        tex = (tex - tex.mean((2, 3), keepdim=True)) / tex.std((2, 3), keepdim=True) * 45.0 + 128.0
        a = tex[:, :, M:M + h, M:M + w]
        b = F.grid_sample(tex, grid.expand(m, -1, -1, -1), mode="bicubic", padding_mode="border", align_corners=True)
        out_a.append(a.round().clamp(0, 255).to(torch.uint8)[:, 0].contiguous())
        out_b.append(b.round().clamp(0, 255).to(torch.uint8)[:, 0].contiguous())
    return torch.cat(out_a).contiguous(), torch.cat(out_b).contiguous()


def algorithmic_bytes(p, nframes, fused_tv=True):
    """ALGORITHMIC (compulsory) HBM bytes of ONE step per kernel class, summed over its launches
    (SURVEY.md 8d per-pixel figures; fused kernels: inputs read once + outputs written once)."""
    noc = p.noc
    out = {k: 0.0 for k in ("warp", "derivatives", "tv_system", "sor", "tv_finish", "patch_optimize", "densify", "tv_fused")}
    launches = dict.fromkeys(out, 0)
    for l in range(p.sc_l, p.sc_f + 1):
        w, h = p.level_size(l)
        npx = w * h * nframes
        nw, nh = p.grid(l)
        nop = nw * nh
        nv = noc * p.p_samp_s ** 2
        th, tw, _ = p.plane_shape(l)
        out["patch_optimize"] += nframes * (4 * th * tw * noc * 4 + (2 * (w // 2) * (h // 2) * 4 if l < p.sc_f else 0)
                                            + nop * 8 + nop * nv * 4)
        launches["patch_optimize"] += 1
        fused = p.usetvref and fused_tv and noc == 1 and h <= 64 and w >= 16 and p.tv_solverit <= 3
        out["densify"] += nframes * (nop * 8 + nop * nv * 4) + 8 * npx       # p, pweight in; wx, wy out
        launches["densify"] += 1
        if p.usetvref:
            n_inner = p.tv_innerit * (l + 1)
            out["warp"] += (8 + 4 * noc + 4 * noc + 4) * npx          # wx,wy + src once + dst + mask
            out["derivatives"] += 40 * noc * npx                      # I0,I1w in, 8 planes out
            if fused:   # system + SOR in one kernel: derivs, mask, wx, wy, du, dv in; du, dv out
                out["tv_fused"] += n_inner * (32 * noc + 20 + 8) * npx
                launches["tv_fused"] += 1
            else:
                out["tv_system"] += n_inner * (20 + 32 * noc + 28) * npx  # mask,wx,wy,du,dv + derivs in, 7 planes out
                out["sor"] += n_inner * 44 * npx                          # 7 planes + du,dv in, du,dv out (3 sweeps fused)
                launches["tv_system"] += n_inner
                launches["sor"] += n_inner
            out["tv_finish"] += 24 * npx
            launches["warp"] += 1
            launches["derivatives"] += 1
            launches["tv_finish"] += 1
    return out, launches


def cpu_baseline(p, batch, nsample, budget_s):
    """Reference CPU path (one thread) on `nsample` frames of this batch; pyramids copied back from HBM."""
    import ctypes as C
    import numpy as np
    import oracle
    from of_dis_amd import capi
    kind = "reference" if oracle.have_ref("int", False) else "port"
    R = oracle.ref("int", False) if kind == "reference" else oracle.c_oracle()
    if kind == "port":
        R.set_reduce_order(False)
    L = capi.lib()
    frames = []
    for f in range(nsample):
        planes = [[None] * (p.sc_f + 1) for _ in range(4)]
        for l in range(p.sc_l, p.sc_f + 1):
            n = batch.input_elems(l)
            for k in range(4):
                arr = np.empty(p.plane_shape(l), np.float32)
                capi.check(L.ofdis_memcpy_d2h(arr.ctypes.data, batch.input_ptr(l, k) + f * n * 4, n * 4))
                planes[k][l] = arr
        frames.append(planes)
    pq = p.copy(verbosity=0)
    # the metric's second half: end-point error of the HIP result against the reference's own output (its plain,
    # sequential-sum build -- NOT the defined-order build the bit-exact check uses), in full-resolution pixels: the
    # .flo is this flow times 2^sc_l before an interpolation that cannot increase a difference
    err = []
    for f, planes in enumerate(frames):  # also the warm-up
        ref = R.flow(pq, planes[0], planes[1], planes[2], planes[3])
        got = batch.download(f)
        err.append(np.sqrt(((got.astype(np.float64) - ref.astype(np.float64)) ** 2).sum(-1)) * (1 << p.sc_l))
    err = np.stack(err)
    epe = {"mean_px": float(err.mean()), "max_px": float(err.max()), "frac_above_1e-3": float((err > 1e-3).mean()),
           "frames": len(frames), "against": "reference CPU build with sequential sums" if kind == "reference"
           else "C restatement with sequential sums"}
    n_eval, t0 = 0, time.perf_counter()
    best = 1e9
    while True:
        for planes in frames:
            t1 = time.perf_counter()
            R.flow(pq, planes[0], planes[1], planes[2], planes[3])
            best = min(best, time.perf_counter() - t1)
            n_eval += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": round(n_eval / el, 2), "unit": "frames/s", "cores": 1, "kind": kind,
            "best_ms_per_frame": round(best * 1e3, 4), "epe_vs_reference": epe,
            "sample": f"{n_eval} OFClass-scope evaluations over {nsample} distinct frames of this batch, "
                      f"{el:.1f} s on one of {os.cpu_count()} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="frame pairs per GPU per step")
    ap.add_argument("--tv", choices=["on", "off"], default="on")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="ofdis_batch_set_pipeline: sub-batches on internal streams, consecutive steps overlap (1 = off)")
    ap.add_argument("--scope", choices=["ofclass", "e2e"], default="ofclass",
                    help="ofclass (the metric): pyramids resident in HBM -> level flow.  e2e (secondary, DESIGN.md 5): "
                         "8-bit frames resident in HBM -> pyramids -> flow -> full-resolution flow in HBM")
    args = ap.parse_args()

    import numpy as np
    import torch
    from of_dis_amd import capi
    from of_dis_amd.params import oppoint

    from of_dis_amd import shard
    rank, world, local_rank = shard.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    # developer switches for exercising the multi-rank control flow on a one-GPU box (tests only; the driver's
    # launch uses one GPU per rank over RCCL): all ranks on device 0, gloo instead of nccl
    backend = os.environ.get("OFDIS_BENCH_BACKEND", "nccl")
    if os.environ.get("OFDIS_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    L = capi.lib()
    capi.check(L.ofdis_set_device(local_rank))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the max-over-ranks tensor lives

    def barrier():
        shard.barrier(dist, torch.cuda.synchronize)

    tv = args.tv == "on"
    p = oppoint(2, WIDTH, HEIGHT, noc=1, usetvref=tv, verbosity=0)
    B = args.batch
    # weak scaling: rank r owns global frames [r*B, (r+1)*B); its generator is seeded by its first frame
    ia, ib = synth_frames_torch(B, WIDTH, HEIGHT, shard.frame_seed(1234, rank * B), dev)
    # a dedicated (non-default) stream: launches on HIP's legacy null stream carry implicit cross-stream
    # synchronisation and measure ~30 us slower per launch on this stack
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    batch = capi.Batch(p, B)
    batch.set_pipeline(args.pipeline)
    torch.cuda.synchronize()  # frames were generated on torch's default stream
    batch.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), WIDTH, HEIGHT, stream)
    torch.cuda.synchronize()

    e2e = args.scope == "e2e"
    full = torch.empty((B, HEIGHT, WIDTH, 2), dtype=torch.float32, device=dev) if e2e else None

    def step():
        if e2e:  # secondary scope: raw frames -> pyramids -> path -> full-resolution flow, everything in HBM
            batch.join(stream)  # a pipelined sub-batch of the previous step may still be reading the pyramids
            batch.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), WIDTH, HEIGHT, stream)
        batch.run(stream)
        if e2e:
            batch.upsample(WIDTH, HEIGHT, out_ptr=full.data_ptr(), stream=stream)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dist, red_dev)
    fps = shard.throughput([B] * world, args.steps, elapsed)

    result = None
    if rank == 0:
        # ---- per-kernel timing with HIP events on the launch stream (separate, untimed pass)
        nrep = 3
        batch.timing(True)
        for _ in range(nrep):
            batch.run(stream)
        torch.cuda.synchronize()
        abytes, _ = algorithmic_bytes(p, B, fused_tv=not os.environ.get("OFDIS_NO_FUSED"))
        kernels = {}
        for k, name in enumerate(capi.K_NAMES):
            ms, n = batch.kernel_time(k)
            if n == 0:
                continue
            per_step_ms = ms / nrep
            gbs = abytes[name] / (per_step_ms * 1e-3) / 1e9
            kernels[name] = {"launches_per_step": n // nrep, "ms_per_step": round(per_step_ms, 4),
                             "avg_launch_us": round(ms / n * 1e3, 2),
                             "algorithmic_MB_per_step": round(abytes[name] / 1e6, 2),
                             "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
        batch.timing(False)
        # HBM traffic per step from the PMC counters (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, calibrated in
        # profiles/r01_pmc_calibration.txt), collected by tools/pmc_traffic.py for this batch size
        traffic = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("batch") == B and tj.get("tv") == args.tv:
                traffic = tj["bytes_per_step"]
        except Exception:
            pass
        for name, k in kernels.items():
            if name in traffic:
                k["pmc_traffic_MB_per_step"] = round(traffic[name] / 1e6, 2)
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kernels[dom]["frac_of_hbm_peak"],
                    "traffic": traffic.get(dom),
                    "note": "achieved = algorithmic bytes of all launches of this kernel class in one step / their "
                            "summed HIP-event time (bytes/s); traffic = PMC HBM bytes of the same launches per step "
                            "(profiles/traffic.json). The two dominant kernels (tv_fused, patch_optimize) are VALU-issue "
                            "bound (PMC: profiles/*pmc_sq*), not HBM bound; see DESIGN.md section 4"}
        result = {
            "metric": "frames/sec at 1024\u00d7436 op-point-2 (INT)", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"run_OF_INT op-point-2, 1024x436 (padded 1024x448, levels 5-3), patch 8 overlap 0.4, "
                                   f"12 GN iterations, TV {'on (6/5/4 inner its, 3 SOR sweeps, alpha=gamma=10 delta=5)' if tv else 'off'}; "
                                   + ("end-to-end scope: 8-bit frames in HBM -> pyramids -> flow -> full-resolution flow in HBM (secondary)"
                                      if e2e else "OFClass scope, pyramids resident in HBM"),
                       "frames_per_gpu_per_step": B, "global_frames_per_step": B * world,
                       "parallelism": f"frame-sharded x{world}", "tv": args.tv,
                       "pipeline": f"{args.pipeline} sub-batches per GPU on internal HIP streams, consecutive steps "
                                   f"overlap inside the timed region" if args.pipeline > 1 else "off"},
            "roofline": roofline, "kernels": kernels,
        }
        if not args.no_parity:
            try:
                import oracle
                O = oracle.c_oracle()
                O.set_reduce_order(True)
                f = B - 1
                planes = [[None] * (p.sc_f + 1) for _ in range(4)]
                for l in range(p.sc_l, p.sc_f + 1):
                    n = batch.input_elems(l)
                    for k in range(4):
                        arr = np.empty(p.plane_shape(l), np.float32)
                        capi.check(L.ofdis_memcpy_d2h(arr.ctypes.data, batch.input_ptr(l, k) + f * n * 4, n * 4))
                        planes[k][l] = arr
                ref = O.flow(p, planes[0], planes[1], planes[2], planes[3])
                got = batch.download(f)
                result["parity_check"] = "bit-exact vs oracle (frame %d)" % f if np.array_equal(ref, got) else \
                    "MISMATCH vs oracle: mean EPE %.3g" % oracle.epe_stats(ref, got)[0]
            except Exception as e:  # the checker is optional for the measurement
                result["parity_check"] = f"not run ({type(e).__name__}: {e})"
        if world == 1 and tv and not e2e:
            # BASELINE.json also lists the same operating point with the refinement switched off (configs[1]); report it
            # next to the headline (configs[2], TV on -- what operating point 2 is in the reference, run_dense.cpp:259-265)
            try:
                p_off = oppoint(2, WIDTH, HEIGHT, noc=1, usetvref=False, verbosity=0)
                b_off = capi.Batch(p_off, B)
                b_off.set_pipeline(args.pipeline)
                b_off.build_pyramids_u8(ia.data_ptr(), ib.data_ptr(), WIDTH, HEIGHT, stream)
                for _ in range(args.warmup):
                    b_off.run(stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    b_off.run(stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                b_off.close()
                result["tv_off"] = {"workload": "same, TV off (BASELINE.json configs[1])", "value": round(B * args.steps / dt, 1),
                                    "unit": "frames/s", "ms_per_step": round(dt / args.steps * 1e3, 4)}
            except Exception as e:
                result["tv_off"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.cpu_seconds > 0:
            try:
                result["cpu_baseline"] = cpu_baseline(p, batch, min(16, B), args.cpu_seconds)
                result["speedup_vs_cpu_1core"] = round(fps / result["cpu_baseline"]["value"], 1)
            except Exception as e:
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 1, "kind": "unavailable",
                                          "sample": f"{type(e).__name__}: {e}"}
    barrier()
    batch.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
