"""CPU test of the N > 1 path: two gloo ranks shard chunk-proofs, reduce the report the way bench.py does."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aes_zero_knowledge_proof_circuit_amd import sharding  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import zko   # the checker stands in for the GPU prover here: byte-level ECB of this rank's message
    n_chunks, blocks = sharding.plan(10, 4)
    key, msg = sharding.rank_message(rank, blocks)
    ct = zko.aes_encrypt(msg, key)
    accepted = sum(1 for i in range(n_chunks) if len(ct[64 * i:64 * i + 64]) == 64)
    elapsed = 1.0 + rank          # rank 1 is the slow one
    e, a, t, neg = sharding.reduce_report(elapsed, accepted, n_chunks, 1)
    lo, hi = sharding.split_chunks(7, rank, world)
    out[rank] = (e, a, t, neg, blocks, key, msg[:4], (lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_report_reduction():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0[0] == r1[0] == 2.0                      # max over ranks
    assert (r0[1], r0[2], r0[3]) == (6, 6, 2)          # 3 chunk-proofs per rank, both negatives rejected
    assert r0[4] == 12                                 # 10 blocks rounded up to 3 chunks of 4
    assert r0[5] == r1[5] and r0[6] != r1[6]           # shared key, per-rank message
    assert r0[7] == (0, 4) and r1[7] == (4, 7)         # contiguous split of 7 chunks
    assert sharding.aggregate_value(2, 12, 3, 2.0) == 36.0


def test_plan_edges():
    assert sharding.plan(1, 1) == (1, 1)
    assert sharding.plan(64, 4) == (16, 64)
    assert sharding.plan(65, 4) == (17, 68)
    with pytest.raises(ValueError):
        sharding.plan(0, 4)
    assert [sharding.split_chunks(5, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
