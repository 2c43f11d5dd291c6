"""N > 1 path on CPU: two gloo ranks shard a batch of streams, each demodulates its range
(with the C oracle standing in for the GPU - the partition/gather logic is what is under
test), rank 0 receives all packets in stream order.  No data-path collective exists."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from rtldavis_amd import synth
from rtldavis_amd.shard import demodulate_sharded, gather_rank_lines, shard_range

N_STREAMS = 5
SEEDS = list(range(N_STREAMS))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_records(raw):
    from oracle import c_oracle as CO
    res, _ = CO.demod_batch(raw, CO.make_cfg(), threads=1)
    return [(s, p.call, (p.index, bytes(p.data).hex())) for s, ps in enumerate(res) for p in ps]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(N_STREAMS, world, rank)
    out = demodulate_sharded(N_STREAMS, lambda a, b: synth.synth_streams(SEEDS[a:b], n_samples=6 * 8192),
                             _oracle_records)
    # bench.py's per_gpu lines: every rank's own figures on every rank, in rank order
    lines = gather_rank_lines({"rank": rank, "value": 100.0 + rank, "verified": True})
    assert [d["rank"] for d in lines] == list(range(world)) and [d["value"] for d in lines] == [100.0 + r for r in range(world)]
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_gather_in_stream_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _oracle_records(synth.synth_streams(SEEDS, n_samples=6 * 8192))
    assert got == want
    assert [g[0] for g in got] == sorted(g[0] for g in got)
