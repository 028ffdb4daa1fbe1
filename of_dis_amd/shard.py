"""Frame sharding across GPUs (one process per GPU).

Frame pairs are independent problems (the reference never carries state between pairs: `initflow` is
always null, run_dense.cpp:395), so the path shards with NO data-path collective: every rank owns a
contiguous block of frames and runs the whole hot path on it.  torch.distributed (RCCL on GPUs, gloo in the
CPU tests) is used only for the start/stop barrier, the max-over-ranks time and gathering small reports.
"""
import os


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def frame_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` frames owned by `rank`: sizes differ by at most one, earlier
    ranks take the remainder.  The union over ranks is exactly range(total)."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError((total, rank, world))
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frame_seed(base_seed, global_frame_index):
    """Seed of a synthetic frame: a function of its GLOBAL index only, so a frame is the same problem on
    whichever rank it lands (tools/gen_synth.py convention: frame k of a sequence uses seed 1234 + k)."""
    return base_seed + global_frame_index


def barrier(dist=None, device_sync=None):
    if device_sync:
        device_sync()
    if dist is not None:
        dist.barrier()
    if device_sync:
        device_sync()


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX all-reduce of a python float (the bench's elapsed time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj, dist=None, world=1):
    """All ranks' small python objects, in rank order (used for per-rank frame counts / reports)."""
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def throughput(frames_per_rank_list, steps, elapsed_max):
    """Whole-job frames/s: all frames processed by all ranks over the slowest rank's time."""
    return sum(frames_per_rank_list) * steps / elapsed_max
