"""Stream sharding across GPUs (one process per GPU).

Streams are independent (the reference keeps all state per Demodulator instance,
src/rtldavis/dsp.py:131-135), so the batch axis is cut into contiguous ranges, one per
rank, and nothing on the data path crosses GPUs: no RCCL collective is issued.  The only
exchange is the final gather of the (few bytes per packet) results to rank 0, done on the
host with ``torch.distributed.gather_object``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple


def shard_range(n_streams: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of streams owned by `rank`; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_records(local: Sequence[tuple], lo: int, group=None, dst: int = 0) -> Optional[List[tuple]]:
    """Gather (stream, call, packet) records from every rank to `dst`, renumbering local
    stream indices to global ones (local stream s -> lo + s) and concatenating in stream order."""
    import torch.distributed as dist

    glob = [(lo + s, c, p) for (s, c, p) in local]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return glob
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bucket = [None] * world if rank == dst else None
    dist.gather_object(glob, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    out: List[tuple] = []
    for part in bucket:  # rank order == stream order because shards are contiguous ranges
        out.extend(part)
    return out


def demodulate_sharded(n_streams: int, load_shard: Callable[[int, int], object],
                       demod_shard: Callable[[object], Sequence[tuple]], group=None) -> Optional[List[tuple]]:
    """Run `demod_shard(load_shard(lo, hi))` on this rank's stream range and gather to rank 0.

    `demod_shard` is normally ``lambda iq: BatchDemodulator(...).demodulate-and-records``;
    it is a parameter so the partition/gather logic can be exercised on CPU ranks."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_streams, world, rank)
    local = demod_shard(load_shard(lo, hi)) if hi > lo else []
    return gather_records(local, lo, group)


def gather_rank_lines(line: dict, group=None) -> List[dict]:
    """Every rank's own measurement line (bench.py: rank, device, value, ms_per_step, kernel_ms, roofline_frac, verified)
    on every rank, in rank order - BASELINE configs[4] asks for per-GPU MSamples/s next to the aggregate, and a straggler
    GPU is invisible in a MAX(elapsed) alone.  Host-side object gather; no data-path collective."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [line]
    bucket = [None] * dist.get_world_size(group)
    dist.all_gather_object(bucket, line, group=group)
    return sorted(bucket, key=lambda d: d["rank"])
