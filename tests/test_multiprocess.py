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


@pytest.mark.gpu
def test_bench_two_ranks_control_flow(gpu):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), on a
    one-GPU box: both ranks share device 0 and rendezvous over gloo (developer switches in bench.py) -- the barrier /
    max-over-ranks / rank-0 report path is the one the RCCL run takes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OFDIS_BENCH_BACKEND="gloo", OFDIS_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "64", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_frames_per_step"] == 128
    assert d["parity_check"].startswith("bit-exact")
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
