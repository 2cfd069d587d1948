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


def rank_cpu_share(cpus, local_rank, local_world):
    """the `local_rank`-th of `local_world` contiguous shares of the sorted CPU list (sizes differ by at most one).  Linux numbers the cores of a socket contiguously
    and GPUs 0..3 / 4..7 of an 8-GPU MI355X node hang off sockets 0 / 1, so a contiguous share is also the NUMA-local one."""
    cpus = sorted(cpus)
    lo, hi = split_chunks(len(cpus), local_rank, local_world)
    return cpus[lo:hi]


def bind_rank_cpus(local_rank, local_world):
    """Pin this rank's process (prover threads, verifier pool, OpenMP) to its share of the CPUs it may run on; returns a description for the bench line, None where
    the platform has no sched_setaffinity or the share would be empty (fewer CPUs than ranks)."""
    import os
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    share = rank_cpu_share(os.sched_getaffinity(0), local_rank, local_world)
    if not share:
        return None
    os.sched_setaffinity(0, share)
    return {"local_rank": local_rank, "cpus": len(share), "first": share[0], "last": share[-1]}


def gather_affinities(affinity, device=None):
    """every rank's CPU share (bind_rank_cpus) on every rank, in rank order, for the bench line: [{"local_rank", "cpus", "first", "last"} | None, ...]"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [affinity]
    a = affinity or {}
    mine = torch.tensor([1 if affinity else 0, a.get("local_rank", -1), a.get("cpus", 0), a.get("first", -1), a.get("last", -1)], dtype=torch.int64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    out = []
    for p in parts:
        v = [int(x) for x in p.cpu()]
        out.append({"local_rank": v[1], "cpus": v[2], "first": v[3], "last": v[4]} if v[0] else None)
    return out


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


def gather_proofs(proofs, device=None):
    """The one exchange of a strong-scaling job ("one gather of proof bytes at the end", reference loop src/lib.rs:194):
    every rank contributes the serialized proofs of its contiguous chunk range; all ranks get the whole list back in chunk order.

    ONE all-gather of a [rows x stride] uint8 tensor per rank (rows = the largest share, stride = 4-byte length prefix + the longest proof;
    a Marlin proof here is 855 B), preceded by one all-reduce(MAX) that agrees on rows / stride.  ~0.9 KB per chunk-proof: latency-bound on xGMI.
    Without an initialized process group (single process) the list is returned unchanged; with one, the collective runs even for a 1-rank group
    (a 1-rank torchrun / the -m gpu test of one rank's configs[3] share still move their bytes through RCCL).
    """
    import torch
    import torch.distributed as dist
    proofs = [bytes(p) for p in proofs]
    if not (dist.is_available() and dist.is_initialized()):
        return proofs
    world = dist.get_world_size()
    shape = torch.tensor([len(proofs), max((len(p) for p in proofs), default=0)], dtype=torch.int64, device=device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)
    rows, stride = int(shape[0]), int(shape[1]) + 4
    buf = bytearray(max(rows, 1) * stride)
    for i, p in enumerate(proofs):
        buf[i * stride:i * stride + 4] = (len(p) + 1).to_bytes(4, "little")        # 0 marks an unused row
        buf[i * stride + 4:i * stride + 4 + len(p)] = p
    mine = torch.frombuffer(buf, dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = []
    for part in parts:
        raw = part.cpu().numpy().tobytes()
        for i in range(rows):
            n = int.from_bytes(raw[i * stride:i * stride + 4], "little")
            if n:
                out.append(raw[i * stride + 4:i * stride + 4 + n - 1])
    return out


def aggregate_value(world, blocks_per_rank, steps, elapsed_max):
    return world * blocks_per_rank * steps / elapsed_max


def point_range(n, rank, world):
    """contiguous share [lo, hi) of an n-point MSM for this rank"""
    return split_chunks(n, rank, world)


def msm_sharded_device(curve_id, bases_bytes, scalars_bytes, device):
    """The same point-range sharding with a DEVICE-RESIDENT exchange: every rank leaves the window sums of its slice (n_windows x 192 B XYZZ points,
    window plan of the whole MSM) in row `rank` of a [world, bytes] CUDA tensor, ONE torch.distributed.all_gather_into_tensor moves the rows
    HBM to HBM over RCCL / xGMI, and a device kernel adds the ranks' sums per window before the Horner pass (include/zkaes.h,
    zkaes_msm_window_sums_dev / zkaes_msm_fold_window_sums_dev).  Nothing of the data path is staged through host memory.
    """
    import torch
    import torch.distributed as dist
    from . import api
    n = len(scalars_bytes) // 32
    if len(bases_bytes) != 96 * n:
        raise ValueError("bases must be n x 96 bytes and scalars n x 32 bytes")
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    lo, hi = point_range(n, rank, world)
    _, _, nbytes = api.msm_sharded_plan(curve_id, n)
    buf = torch.zeros((world, nbytes), dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)                                   # libzkaes writes from its own stream: the zero-fill must have landed
    api.msm_window_sums_dev(curve_id, bases_bytes[96 * lo:96 * hi], scalars_bytes[32 * lo:32 * hi], n, buf[rank].data_ptr(), nbytes)   # returns after its stream has drained
    if dist.is_available() and dist.is_initialized():
        mine = buf[rank].clone()
        dist.all_gather_into_tensor(buf.view(-1), mine)              # RCCL: HBM -> HBM (also exercised with one rank)
        torch.cuda.synchronize(device)
    return api.msm_fold_window_sums_dev(curve_id, buf.data_ptr(), world, n)


def msm_sharded_srs_device(pk, scalars_bytes, device, offset=0):
    """ONE commitment-sized MSM over a proving key's own SRS, sharded by point range, on the path the prover itself runs (twisted Edwards window tables, one bucket set):
    every rank's share is ONE XYZZ point (192 B) left in row `rank` of a [world, 192] CUDA tensor, one all_gather_into_tensor over RCCL, one device fold
    (include/zkaes.h zkaes_pk_msm_partial_dev / zkaes_msm_fold_partials_dev).  scalars_bytes: the WHOLE scalar vector (n x 32 B Montgomery Fr), naming
    powers_of_g[offset .. offset + n)."""
    import torch
    import torch.distributed as dist
    from . import api
    n = len(scalars_bytes) // 32
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    lo, hi = point_range(n, rank, world)
    buf = torch.zeros((world, 192), dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)
    pk.msm_partial_dev(scalars_bytes[32 * lo:32 * hi], offset + lo, buf[rank].data_ptr())            # returns after its stream has drained
    if dist.is_available() and dist.is_initialized():
        mine = buf[rank].clone()
        dist.all_gather_into_tensor(buf.view(-1), mine)
        torch.cuda.synchronize(device)
    return api.msm_fold_partials_dev(377, buf.data_ptr(), world)


def msm_sharded(curve_id, bases_bytes, scalars_bytes, local_msm=None, device=None):
    """ONE multi-scalar multiplication sharded by point range over the ranks of the default process group (second row).

    Every rank runs the Pippenger MSM over its own slice of (base, scalar) pairs -- on its GPU through the C ABI -- and the per-rank partial
    sums (96 B affine + an infinity flag) are exchanged with ONE all-gather; each rank then folds the `world` partials with the host-side EC add
    (`zkaes_g1_sum`).  EC addition is not an RCCL reduction operator, hence all-gather + local add instead of an all-reduce; the payload is
    100 B per rank, so the exchange is latency- not bandwidth-bound on xGMI.  `local_msm` is a test seam (CPU tests supply the oracle's MSM
    because there is no GPU there); the product path leaves it None and fails loudly without a GPU.
    """
    import torch
    import torch.distributed as dist
    from . import api
    n = len(scalars_bytes) // 32
    if len(bases_bytes) != 96 * n:
        raise ValueError("bases must be n x 96 bytes and scalars n x 32 bytes")
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    lo, hi = point_range(n, rank, world)
    run = local_msm if local_msm is not None else (lambda b, s: api.msm(curve_id, b, s))
    if hi > lo:
        xy, inf = run(bases_bytes[96 * lo:96 * hi], scalars_bytes[32 * lo:32 * hi])
    else:
        xy, inf = bytes(96), True
    if world == 1:
        return bytes(xy), bool(inf)
    mine = torch.frombuffer(bytearray(bytes(xy) + bytes([1 if inf else 0, 0, 0, 0])), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    pts = []
    for p in parts:
        raw = bytes(p.cpu().numpy().tobytes())
        pts.append((raw[:96], raw[96] != 0))
    return api.g1_sum(curve_id, pts)
