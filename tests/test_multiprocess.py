"""CPU, world_size 2, gloo: the multi-GPU path's host logic -- disjoint/complete frame sharding, seeds that
follow the global frame index, barrier + max-over-ranks timing, report gathering -- with the oracle standing
in for the device (the checker is allowed in tests; the product never uses it).  A frame must give the
same bits whichever rank computes it."""
import os
import socket
import time

import numpy as np
import pytest

from of_dis_amd import shard


def test_frame_range_partitions():
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard.frame_range(total, r, world)
                assert 0 <= lo <= hi <= total
                seen += list(range(lo, hi))
            assert seen == list(range(total))
            sizes = [shard.frame_range(total, r, world)[1] - shard.frame_range(total, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.frame_range(10, 2, 2)


def test_bench_frames_depend_on_global_index_only():
    """bench.py's synthetic frames: a frame's content is a function of its global index, whatever range it is generated in
    (so a frame is the same problem on whichever rank it lands)."""
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    dev = torch.device("cpu")
    a0, b0 = bench.synth_frames_range(0, 70, 64, 48, 1234, dev)
    a1, b1 = bench.synth_frames_range(30, 70, 64, 48, 1234, dev)
    a2, b2 = bench.synth_frames_range(33, 34, 64, 48, 1234, dev)
    assert a0.shape == (70, 48, 64) and a0.dtype == torch.uint8
    assert torch.equal(a0[30:], a1) and torch.equal(b0[30:], b1)
    assert torch.equal(a0[33:34], a2) and torch.equal(b0[33:34], b2)
    assert not torch.equal(a0[0], a0[1]) and not torch.equal(a0[0], b0[0])
    c, _ = bench.synth_frames_range(0, 2, 64, 48, 99, dev, channels=3)
    assert c.shape == (2, 48, 64, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tools"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import gen_synth
    import oracle
    from of_dis_amd.params import oppoint
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    assert shard.env_rank() == (rank, world, rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.frame_range(total, rank, world)
    O = oracle.c_oracle()
    O.set_reduce_order(True)
    w, h = 256, 128
    p = oppoint(2, w, h)
    shard.barrier(dist)
    t0 = time.perf_counter()
    flows = {}
    for g in range(lo, hi):
        ia, ib, _ = gen_synth.make_pair(w, h, shard.frame_seed(1234, g))
        pa, pb = O.build_pyramid(p, ia), O.build_pyramid(p, ib)
        flows[g] = O.flow(p, pa[0], pa[1], pa[2], pb[0])
    if rank == 1:
        time.sleep(0.3)  # make the ranks' times differ: the reported time must be the slowest one
    shard.barrier(dist)
    mine = time.perf_counter() - t0
    tmax = shard.max_over_ranks(mine, dist)
    times = shard.gather_objects(mine, dist, world)
    counts = shard.gather_objects(hi - lo, dist, world)
    assert abs(tmax - max(times)) < 1e-9 and tmax >= mine
    assert sum(counts) == total
    fps = shard.throughput(counts, 1, tmax)
    assert abs(fps - total / max(times)) < 1e-6
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(k): v for k, v in flows.items()})
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    import gen_synth
    import oracle
    from of_dis_amd.params import oppoint
    total, world = 5, 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    got = {}
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for k in z.files:
            assert int(k) not in got, "frame computed twice"
            got[int(k)] = z[k]
    assert sorted(got) == list(range(total))
    # single-process result per frame, bit-identical
    O = oracle.c_oracle()
    O.set_reduce_order(True)
    p = oppoint(2, 256, 128)
    for g in range(total):
        ia, ib, _ = gen_synth.make_pair(256, 128, shard.frame_seed(1234, g))
        pa, pb = O.build_pyramid(p, ia), O.build_pyramid(p, ib)
        assert np.array_equal(got[g], O.flow(p, pa[0], pa[1], pa[2], pb[0])), g


def _bench_env(gpu):
    """Two ranks need two GPUs for RCCL (it refuses two ranks on one device); on a one-GPU box the developer switches of
    bench.py put both ranks on device 0 and rendezvous over gloo -- the same control flow otherwise."""
    if gpu.lib().ofdis_device_count() >= 2:
        return dict(os.environ), "rccl (torch.distributed nccl)"
    return dict(os.environ, OFDIS_BENCH_BACKEND="gloo", OFDIS_BENCH_SHARE_GPU="1"), "gloo"


def _run_bench(cmd, env):
    import json
    import subprocess
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def _parity_ok(d):
    """parity_check of a bench line: the exact contract's bit-exact check (a string; under the fused contract the line carries
    both contracts' checks in a dict)."""
    pc = d["parity_check"]
    return pc.startswith("bit-exact") if isinstance(pc, str) else pc["exact_contract"].startswith("bit-exact")


@pytest.mark.gpu
def test_bench_eight_ranks_control_flow(gpu):
    """BASELINE configs[4] as the driver will launch it on an 8-GPU node -- `bench.py --gpus 8`, 512 pairs in total, 64 per
    rank -- with all eight ranks on this box's one GPU over gloo (developer mode): spawn_ranks(8), frame_range(512, r, 8), the
    eight-way gathers, the max-over-ranks time and rank 0's re-computation of every other rank's frames all execute once."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OFDIS_BENCH_BACKEND="gloo", OFDIS_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    d = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--total-frames", "512", "--steps", "2",
                    "--warmup", "1", "--cpu-seconds", "0", "--no-extras"], env)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["global_frames_per_step"] == 512 and d["config"]["frames_per_gpu_per_step"] == 64
    rk = d["config"]["ranks"]
    assert rk["world_size"] == 8 and len(rk["pci_bus_ids"]) == 8 and rk["launch"] == "self-spawned by bench.py --gpus"
    assert rk["all_ranks_on_one_gpu_developer_mode"] and rk["distinct_gpus"] == 1
    assert d["multi_gpu_check"]["frames_compared"] == 448 and d["multi_gpu_check"]["bit_identical_to_1gpu"]
    assert _parity_ok(d) and d["contract"] in ("exact", "fused")


@pytest.mark.gpu
def test_bench_two_ranks_under_torchrun(gpu):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env, backend = _bench_env(gpu)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "64", "--cpu-seconds", "0", "--no-extras"]
    d = _run_bench(cmd, env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_frames_per_step"] == 128
    rk = d["config"]["ranks"]
    assert (rk["world_size"], rk["backend"], rk["launch"]) == (2, backend, "torch.distributed.run")
    # one PCI bus id per rank; distinct GPUs under RCCL (the one-GPU developer mode shares device 0)
    assert len(rk["pci_bus_ids"]) == 2 and rk["distinct_gpus"] == (2 if backend.startswith("rccl") else 1)
    assert _parity_ok(d)
    assert d["multi_gpu_check"]["bit_identical_to_1gpu"] and d["multi_gpu_check"]["frames_compared"] == 64  # the other rank's whole share
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks(gpu):
    """`python bench.py --gpus 2` as a PLAIN process (no torchrun): it starts one rank per GPU itself, the ranks
    rendezvous, and one JSON line with n_gpus = 2 comes back."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env, backend = _bench_env(gpu)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    d = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                    "--batch", "96", "--cpu-seconds", "0", "--no-extras"], env)
    assert d["n_gpus"] == 2 and d["config"]["global_frames_per_step"] == 192
    rk = d["config"]["ranks"]
    assert (rk["world_size"], rk["backend"], rk["launch"]) == (2, backend, "self-spawned by bench.py --gpus")
    assert d["multi_gpu_check"]["bit_identical_to_1gpu"]


@pytest.mark.gpu
def test_bench_batch512_block_with_two_ranks(gpu):
    """The secondary block `batch512` (BASELINE configs[4]: 512 pairs per step in total) in a two-rank run: 256 pairs per
    rank, collectives in step on both ranks, one JSON line."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env, _ = _bench_env(gpu)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    d = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                    "--batch", "256", "--cpu-seconds", "0", "--no-parity"], env)
    assert d["n_gpus"] == 2 and d["multi_gpu_check"]["bit_identical_to_1gpu"]
    b5 = d["batch512"]
    assert b5["scaling"] == "strong" and b5["value"] > 0 and "256 pairs on rank 0" in b5["workload"]
    assert len(b5["ms_per_step_per_rank"]) == 2 and d["strong_scaling"]["512"] == b5
    assert "4096" not in d["strong_scaling"]                       # 2048 pairs per rank do not fit a 256-pair share
    assert d["sustained"]["seconds"] >= 1.0 and d["sustained"]["value"] > 0
    assert "small_batch" not in d and "cpu_baseline" not in d      # one-GPU blocks only; --cpu-seconds 0


@pytest.mark.gpu
def test_bench_strong_scaling_partition(gpu):
    """BASELINE configs[4] in small: a FIXED batch (--total-frames) cut into contiguous per-rank shares; every frame of
    the other rank is re-computed on rank 0's GPU and must have the same bits."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env, _ = _bench_env(gpu)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    d = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-frames", "75", "--steps", "2",
                    "--warmup", "1", "--cpu-seconds", "0", "--no-extras"], env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_frames_per_step"] == 75 and d["config"]["frames_per_gpu_per_step"] == [38, 37]
    assert d["multi_gpu_check"] == {**d["multi_gpu_check"], "frames_compared": 37, "mismatches": 0, "bit_identical_to_1gpu": True}
    assert _parity_ok(d)


@pytest.mark.gpu
def test_bench_rccl_calls_with_one_rank(gpu):
    """Every RCCL call of the multi-rank path (communicator creation with a device id, barriers, the MAX all-reduce on a
    device tensor, the object gathers) on a one-rank communicator: what a one-GPU box can check of the RCCL side."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OFDIS_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "OFDIS_BENCH_BACKEND", "OFDIS_BENCH_SHARE_GPU"):
        env.pop(k, None)
    d = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                    "--batch", "64", "--cpu-seconds", "0", "--no-extras"], env)
    assert d["n_gpus"] == 1 and d["config"]["ranks"]["world_size"] == 1
    assert d["config"]["ranks"]["backend"] == "rccl (torch.distributed nccl)"
    assert _parity_ok(d)
