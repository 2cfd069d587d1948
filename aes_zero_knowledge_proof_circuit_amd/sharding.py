"""Rank-level sharding of the chunk-proof workload (one process per GPU, torch.distributed over RCCL; gloo in CPU tests).

ECB blocks are independent (src/lib.rs:194) and so are their chunk-proofs: every rank proves its own share with a full key replica.
There is no data-path collective; the only communication is the reporting reduction at the end (max elapsed, summed counters).
"""
import numpy as np


def synthetic_bytes(nbytes, seed):
    return np.random.RandomState(seed & 0xFFFFFFFF).randint(0, 256, size=nbytes, dtype=np.uint8).tobytes()


def plan(blocks, chunk):
    """(n_chunks, padded_blocks): the last chunk must be full, so the message is rounded up to a multiple of the chunk size."""
    if blocks <= 0 or chunk <= 0:
        raise ValueError("blocks and chunk must be positive")
    n_chunks = (blocks + chunk - 1) // chunk
    return n_chunks, n_chunks * chunk


def rank_message(rank, blocks, base_seed=0x5EED):
    """weak scaling: every rank proves its own `blocks`-block message under one shared key"""
    return synthetic_bytes(16, base_seed), synthetic_bytes(16 * blocks, base_seed + 1 + rank)


def split_chunks(n_chunks, rank, world):
    """strong-scaling helper: contiguous share [lo, hi) of n_chunks for this rank (used by zkaes callers that shard ONE message)"""
    base, extra = divmod(n_chunks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_report(elapsed, accepted, total, negatives_ok, device=None):
    """max-over-ranks time and summed acceptance counters (no-op without an initialized process group)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed, accepted, total, negatives_ok
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([accepted, total, negatives_ok], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0]), int(c[1]), int(c[2])


def aggregate_value(world, blocks_per_rank, steps, elapsed_max):
    return world * blocks_per_rank * steps / elapsed_max
