"""N>1 path on CPU: two gloo ranks shard a batch round-robin (chunk i -> rank i mod 2), each processes its
shard (with the CPU oracle standing in for the device — there is no GPU here), and the control-plane
aggregation (max time, summed bytes) plus per-chunk results are checked against the single-rank run."""
import hashlib
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

WORLD = 2
N_CHUNKS = 37


def _worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import oracle
    from cramjam_amd.shard import aggregate, shard_indices
    mine = list(shard_indices(N_CHUNKS, rank, WORLD))
    digests = {}
    for i in mine:
        raw = oracle.synth_v1(4096 + 17 * i, i)
        _, blk = oracle.lz4_compress_raw(raw)
        n, out = oracle.lz4_decompress_raw(blk, len(raw))
        assert out == raw
        digests[i] = hashlib.sha256(out).hexdigest()
    seconds = 1.0 + rank                      # rank 1 is "slower": the aggregate must report the max
    unc = sum(4096 + 17 * i for i in mine)
    dist.barrier()
    t, b = aggregate(dist, torch.device("cpu"), seconds, unc)
    gathered = [None] * WORLD
    dist.all_gather_object(gathered, digests)
    if rank == 0:
        q.put((t, b, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_round_robin_sharding():
    from cramjam_amd.shard import shard_count, shard_indices
    assert [list(shard_indices(7, r, 3)) for r in range(3)] == [[0, 3, 6], [1, 4], [2, 5]]
    assert sum(shard_count(N_CHUNKS, r, WORLD) for r in range(WORLD)) == N_CHUNKS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    t, b, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0                                                  # max over ranks
    assert b == float(sum(4096 + 17 * i for i in range(N_CHUNKS)))   # every chunk counted once
    merged = {}
    for d in gathered:
        assert not (set(d) & set(merged))                            # shards are disjoint
        merged.update(d)
    assert sorted(merged) == list(range(N_CHUNKS))
    import oracle
    for i in range(N_CHUNKS):                                        # identical to the 1-rank result
        assert merged[i] == hashlib.sha256(oracle.synth_v1(4096 + 17 * i, i)).hexdigest()
