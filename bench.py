#!/usr/bin/env python3
"""bench.py -- AES-ECB blocks proven per second (Marlin) on N x MI355X.

Default workload = BASELINE.json configs[2], the one the >=100x target is quoted on: ONE 4096-block (64 KiB) ECB message per GPU, proven as
682 chunk-proofs of 6 blocks + 1 of 4 with one proving key per chunk size (ECB blocks are independent, src/lib.rs:194; 6 blocks is the most
that fits |H| = 2^20, |K| = 2^22 and the reference's SRS literal).  The --steps timed steps SLICE that one message: step i proves the i-th
contiguous share of its chunk-proofs, witness generation -> serialized proof, SRS / index / circuit tables resident in HBM beforehand (the
region criterion times in the reference: benches/benchmark_encrypt.rs:45-47).  value = blocks of the whole message(s) / timed region.
The slices are issued through a 2-deep software pipeline (slice i+1 is submitted while slice i drains, as a streaming caller of the library would do), so
the chip does not idle between steps; the timed region is still exactly the --steps slices between two barriers.
Warm-up steps prove a separate short message.  Every timed proof is verified afterwards on the host (accept rate must be 100 %) together
with the reference's negative case (a wrong ciphertext must be rejected).

modes (all one process per GPU, torch.distributed over RCCL when launched under torchrun):
  headline (default)  every rank proves its own --blocks message                         -> "scaling": "weak", no data-path collective
  strong              ONE --blocks message (default 65536 = BASELINE configs[3]); rank r proves the contiguous chunk range
                      split_chunks(n, r, world); proofs are all-gathered and rank 0 verifies every one -> "scaling": "strong"
  batch               --proofs independent single-block proofs on one SRS (configs[4]), sharded the same way -> "scaling": "strong"

    python bench.py --gpus 1 --steps 4 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per prover context (ROCclr default: 4 for all streams of the process); must be in the environment before the first HIP call,
# which under torchrun is torch's, not libzkaes' (csrc/runtime.hip sets the same default when the library is loaded first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CIRCUIT_MODEL_NOTE = ("R1CS of the restated ark-r1cs-std 0.3.1 gadget semantics: 629,856 constraints / 3,002,900 non-zeros at 64 bytes; the reference's own SRS literal "
                      "(src/lib.rs:141) records 866,944 / 4,062,064 for that size and no variant of the source-less simpleworks shift/rotate calls reproduces it "
                      "(tools/circuit_variants.py, DESIGN.md section 2a) -- at the literal's density a 2^20 domain holds 4 blocks per chunk-proof, not 6")

from aes_zero_knowledge_proof_circuit_amd import sharding  # noqa: E402

synthetic = sharding.synthetic_bytes


def cpu_baseline(samples_small=3, samples_chunk=1, chunk_blocks=6, budget_s=150.0):
    """The CPU oracle (oracle/, a C restatement of the same algorithm; NOT arkworks) timed on this box's host cores, SRS + index prebuilt
    outside the timed part like the GPU side: `samples_small` one-block chunk-proofs and `samples_chunk` proofs at the bench's own chunk size."""
    from oracle import zko
    nthreads = int(zko.lib().zko_api_num_threads())
    out = {"unit": "blocks/s", "cores": nthreads, "kind": "port", "by_chunk": {}}
    t_begin = time.perf_counter()
    for blocks, samples in ((1, samples_small), (chunk_blocks, samples_chunk)):
        if samples <= 0 or (blocks != 1 and time.perf_counter() - t_begin > budget_s):
            continue
        t0 = time.perf_counter()
        cs, _ = zko.synth_aes(bytes(16 * blocks), bytes(16))
        ix = zko.Index(cs)                       # setup (SRS + index), outside the timed region like the GPU side
        t_index = time.perf_counter() - t0
        times = []
        for i in range(samples):
            msg, key = synthetic(16 * blocks, 0x5EED + 1 + i), synthetic(16, 0x5EED)
            t0 = time.perf_counter()
            cs, _ = zko.synth_aes(msg, key)
            ix.prove(cs)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s:
                break
        best = min(times)
        out["by_chunk"][str(blocks)] = {"blocks_per_s": round(blocks / best, 5), "best_s": round(best, 2), "samples_s": [round(t, 2) for t in times],
                                        "index_s": round(t_index, 1), "h": int(ix.info()["h"]), "k": int(ix.info()["k"])}
        del ix
    best_chunk = max(out["by_chunk"], key=lambda b: out["by_chunk"][b]["blocks_per_s"])
    out["value"] = out["by_chunk"][best_chunk]["blocks_per_s"]
    out["sample"] = "best of %s: %s" % (", ".join("%sx %s-block chunk-proof" % (len(v["samples_s"]), b) for b, v in out["by_chunk"].items()),
                                        "%s-block chunk, %.1f s per proof" % (best_chunk, out["by_chunk"][best_chunk]["best_s"]))
    out["threads_effective"] = "MSM: windows x point-slices tasks (all %d threads); NTT / polynomial loops: OpenMP static; synthesis + transcript: 1 thread" % nthreads
    return out


def even_slices(lo, hi, parts):
    """`parts` contiguous shares of [lo, hi), sizes differing by at most one"""
    return [tuple(lo + x for x in sharding.split_chunks(hi - lo, i, parts)) for i in range(parts)]


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["headline", "strong", "batch"], default="headline")
    ap.add_argument("--blocks", type=int, default=None, help="ECB blocks of the message (headline: per rank, default 4096; strong: whole job, default 65536)")
    ap.add_argument("--proofs", type=int, default=1024, help="batch mode: independent single-block proofs (whole job)")
    ap.add_argument("--chunk", type=int, default=6, help="blocks per chunk-proof (6 = the most that fits |H|=2^20, |K|=2^22 and the reference's SRS literal)")
    ap.add_argument("--contexts", type=int, default=16, help="chunk-proofs in flight per GPU (separate HIP streams)")
    ap.add_argument("--pipeline", type=int, default=2, help="timed slices in flight (1 = strictly one after the other: the chip drains at every step boundary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunk-samples", type=int, default=1, help="CPU-oracle samples at the bench's chunk size (0 = one-block samples only)")
    ap.add_argument("--serial-probe", type=int, default=2, help="chunk-proofs proven one at a time after the timed region for un-overlapped kernel durations (0 = off)")
    return ap


def run(args, api, dist_env=None):
    """the rank path.  `api` = aes_zero_knowledge_proof_circuit_amd.api (the CPU tests drive this exact function with a stub prover)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ            # under torchrun the RCCL group is created (and exercised) even for one rank
    # rehearsal hooks for a one-GPU box (not used by the driver): ZKAES_BENCH_BACKEND=gloo + ZKAES_BENCH_ONE_GPU=1 run N ranks on device 0
    backend = os.environ.get("ZKAES_BENCH_BACKEND", "nccl")
    if os.environ.get("ZKAES_BENCH_ONE_GPU"):
        local_rank = 0
    have_cuda = torch.cuda.is_available()
    if use_dist and not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_device = "cuda" if (use_dist and backend == "nccl") else None
    if api.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device: libzkaes has no CPU fallback")
    api.set_device(local_rank)

    mode = args.mode
    chunk = 1 if mode == "batch" else args.chunk
    total_blocks = args.proofs if mode == "batch" else (args.blocks or (65536 if mode == "strong" else 4096))
    n_full, rem = divmod(total_blocks, chunk)
    n_chunks = n_full + (1 if rem else 0)                    # chunk-proof i < n_full has `chunk` blocks, the last one (if rem) has `rem`
    # ---- this rank's share of the chunk-proof list
    if mode == "headline":
        lo, hi = 0, n_chunks                                 # weak scaling: every rank its own whole message
        key = synthetic(16, 0x5EED)
        msg = synthetic(16 * total_blocks, 0x5EED + 1 + rank)
        keys16 = None
    else:
        lo, hi = sharding.split_chunks(n_chunks, rank, world)   # strong scaling: contiguous chunk range of ONE job
        key = synthetic(16, 0x5EED)
        msg = synthetic(16 * total_blocks, 0x5EED + 1)
        keys16 = synthetic(16 * total_blocks, 0x5EED + 2) if mode == "batch" else None   # batch: one key per proof
    contexts = max(1, args.contexts)
    os.environ["ZKAES_CONTEXTS"] = str(contexts)
    t_setup = time.perf_counter()
    chunk_bytes = 16 * chunk
    need_full = min(hi, n_full) > lo or args.warmup > 0 or args.serial_probe > 0
    need_rem = bool(rem) and hi == n_chunks and hi > lo
    pk = vk = pk_rem = vk_rem = None
    if need_full:
        pk, vk = api.synthesize_keys(chunk_bytes)
    if need_rem:
        pk_rem, vk_rem = api.synthesize_keys(16 * rem)
    setup_s = time.perf_counter() - t_setup
    info = (pk or pk_rem).info()

    def barrier():
        if use_dist:
            dist.barrier()
        if have_cuda:
            torch.cuda.synchronize()

    def prove_range(a, b, source=None, source_keys=None):
        """chunk-proofs [a, b) of this job -> list of proof bytes (the remainder chunk, if inside the range, is proven alongside on its own key)"""
        m = source if source is not None else msg
        out_full, out_rem = [], []
        fa, fb = a, min(b, n_full)

        def run_rem():
            api.set_device(local_rank)                        # HIP's current device is per thread (libzkaes re-selects the key's device itself, too)
            out_rem.extend(pk_rem.encrypt_chunked(m[16 * chunk * n_full:16 * total_blocks], key))
        t = None
        if rem and b == n_chunks and b > a and source is None:
            t = threading.Thread(target=run_rem)
            t.start()
        if fb > fa:
            if mode == "batch":
                ks = source_keys if source_keys is not None else keys16
                out_full = pk.encrypt_batch([m[16 * i:16 * i + 16] for i in range(fa, fb)], [ks[16 * i:16 * i + 16] for i in range(fa, fb)])
            else:
                out_full = pk.encrypt_chunked(m[chunk_bytes * fa:chunk_bytes * fb], key)
        if t:
            t.join()
        return out_full + out_rem

    # ---- warm-up on a separate short message: `contexts` chunk-proofs per step (allocates every context's workspace outside the timed region)
    warm_n = min(contexts, max(1, n_full))
    warm_msg = synthetic(chunk_bytes * warm_n, 0x5EED + 7777 + rank)
    warm_keys = synthetic(16 * warm_n, 0x5EED + 7778)
    for _ in range(args.warmup):
        if mode == "batch":
            pk.encrypt_batch([warm_msg[16 * i:16 * i + 16] for i in range(warm_n)], [warm_keys[16 * i:16 * i + 16] for i in range(warm_n)])
        else:
            pk.encrypt_chunked(warm_msg, key)
    if need_rem and args.warmup > 0:
        pk_rem.encrypt_chunked(synthetic(16 * rem, 0x5EED + 7779), key)

    slices = even_slices(lo, hi, args.steps)
    api.msm_stats(reset=True)
    phase = dict(witness_ms=0.0, round1_ms=0.0, round2_ms=0.0, round3_ms=0.0, open_ms=0.0, total_ms=0.0)
    proofs = []
    barrier()
    t0 = time.perf_counter()
    def timed_slice(ab):
        api.set_device(local_rank)
        r = prove_range(*ab)
        t = pk.timings() if (pk is not None and ab[1] > ab[0]) else None      # wall times of ONE chunk-proof of the step (sampled; several are in flight concurrently)
        return r, t
    with ThreadPoolExecutor(max_workers=max(1, args.pipeline)) as ex:        # slice i+1 starts while slice i drains: both queue on the same prover contexts
        for r, t in ex.map(timed_slice, slices):
            proofs.extend(r)
            if t:
                for k, v in t.items():
                    phase[k] += v
    gathered = None
    if mode != "headline":
        gathered = sharding.gather_proofs(proofs, device=coll_device)    # the job's one exchange: ~855 B per chunk-proof to every rank
    barrier()
    elapsed = time.perf_counter() - t0
    stats = api.msm_stats()

    # ---- un-overlapped kernel durations: a few chunk-proofs one at a time (outside the timed region; cross-check for the rocprof one-context profile)
    serial = None
    if args.serial_probe > 0 and rank == 0 and pk is not None and mode != "batch":
        os.environ["ZKAES_CONTEXTS"] = "1"
        api.msm_stats(reset=True)
        ts = time.perf_counter()
        pk.encrypt_chunked(warm_msg[:chunk_bytes * min(args.serial_probe, warm_n)], key)
        t_serial = time.perf_counter() - ts
        s1 = api.msm_stats()
        os.environ["ZKAES_CONTEXTS"] = str(contexts)
        if s1["launches"]:
            serial = {"proofs": min(args.serial_probe, warm_n), "ms_per_proof": round(1e3 * t_serial / min(args.serial_probe, warm_n), 2),
                      "avg_launch_ms": round(s1["accumulate_ms"] / s1["launches"], 4), "launches": s1["launches"],
                      "achieved_GBs": round(128.0 * s1["points"] / 1e9 / (s1["accumulate_ms"] / 1e3), 2),
                      "int_multiplier_frac": round(3416.0 * s1["pairs"] / 1e12 / (s1["accumulate_ms"] / 1e3) / 28.1, 4)}

    # ---- acceptance: every timed proof must verify (host), the wrong-ciphertext negative must be rejected
    from oracle import zko   # checker only: byte-level AES for the expected ciphertext
    pool = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4))        # ctypes releases the GIL inside zkaes_verify_encryption

    def chunk_ct(i, ct_all):
        return ct_all[chunk_bytes * i:chunk_bytes * (i + 1)] if i < n_full else ct_all[chunk_bytes * n_full:]

    def check(job):
        i, p, ct_all = job
        k = vk if i < n_full else vk_rem
        return bool(api.verify_encryption(k, p, chunk_ct(i, ct_all)))

    accepted = total = 0
    rejected_wrong = 1
    if mode == "headline":
        ct = zko.aes_encrypt(msg, key)
        jobs = [(lo + j, p, ct) for j, p in enumerate(proofs)]
        accepted, total = sum(pool.map(check, jobs)), len(jobs)
    elif rank == 0:
        if mode == "batch":
            ct = b"".join(zko.aes_encrypt(msg[16 * i:16 * i + 16], keys16[16 * i:16 * i + 16]) for i in range(total_blocks))
        else:
            ct = zko.aes_encrypt(msg, key)
        if vk is None:
            _, vk = api.synthesize_keys(chunk_bytes)
        if rem and vk_rem is None:
            _, vk_rem = api.synthesize_keys(16 * rem)
        jobs = [(i, p, ct) for i, p in enumerate(gathered)]
        accepted, total = sum(pool.map(check, jobs)), len(jobs)
        if total != n_chunks:
            accepted = -1                                       # a proof went missing in the gather
    if (mode == "headline" or rank == 0) and (proofs or gathered):
        first_i = lo if mode == "headline" else 0
        first_p = proofs[0] if mode == "headline" else gathered[0]
        bad = bytearray(chunk_ct(first_i, ct)); bad[1] ^= 1; bad[-1] ^= 1
        rejected_wrong = int(not api.verify_encryption(vk if first_i < n_full else vk_rem, first_p, bytes(bad)))
    elapsed, acc_sum, tot_sum, neg_sum = sharding.reduce_report(elapsed, accepted, total, rejected_wrong, device=coll_device)

    out = None
    if rank == 0:
        job_blocks = total_blocks * (world if mode == "headline" else 1)
        value = job_blocks / elapsed
        # roofline of the dominant kernel (MSM bucket accumulation, kernels_msm.hip k_accumulate): algorithmic bytes per launch =
        # 128 B per point (96 B affine base + 32 B scalar) x points in the launch; duration from HIP events on the
        # kernel's own stream, accumulated over every launch of the timed region (rank 0's launches).
        acc_ms, pts, launches = stats["accumulate_ms"], stats["points"], stats["launches"]
        achieved = (128.0 * pts / 1e9) / (acc_ms / 1e3) if acc_ms > 0 else 0.0
        traffic = traffic_src = None
        for name in ("r02_pmc_k_accumulate_tables.json", "r02_pmc_k_accumulate.json", "r01_pmc_k_accumulate_v12.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                traffic = round(pmc["hbm_bytes_per_point_window"] * stats["pairs"] / max(launches, 1))
                traffic_src = "separate PMC run of the isolated kernel (profiles/%s: rocprofv3 --pmc bytes per (point, window) gather), scaled by this run's pairs per launch -- not counters of this run" % name
                break
            except Exception:
                pass
        try:
            copy_gbs = round(api.stream_copy_bench(1 << 30, 20), 1)
        except Exception:
            copy_gbs = None
        mads = 3416.0 * stats["pairs"] / 1e12
        workload = {
            "headline": "%d-block (%d B) ECB message per GPU as %d chunk-proofs of %d block(s)%s, sliced over the %d timed steps" % (total_blocks, 16 * total_blocks, n_full, chunk, (" + 1 of %d" % rem) if rem else "", args.steps),
            "strong": "ONE %d-block (%d B) ECB message as %d chunk-proofs of %d block(s)%s, chunk ranges sharded over %d rank(s), proofs all-gathered, rank 0 verifies all" % (total_blocks, 16 * total_blocks, n_full, chunk, (" + 1 of %d" % rem) if rem else "", world),
            "batch": "%d independent single-block proofs on one SRS (own key each), sharded over %d rank(s), proofs all-gathered, rank 0 verifies all" % (total_blocks, world),
        }[mode] + "; BLS12-377 Marlin, |H|=%d |K|=%d, universal SRS literals (866944,513,4062064)" % (info["h"], info["k"])
        out = {
            "metric": "AES-ECB blocks proven/sec (Marlin), proof verifies", "value": round(value, 4), "unit": "blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": "weak" if mode == "headline" else "strong", "vs_baseline": None,
            "dtype": "u32 limbs (253-bit Fr / 377-bit Fq modular integers)",
            "data": "synthetic (numpy MT19937 bytes, seed 0x5EED; key fixed, message per rank)" if mode == "headline" else "synthetic (numpy MT19937 bytes, seed 0x5EED; one job shared by all ranks)",
            "config": {"workload": workload, "mode": mode, "blocks_total": job_blocks, "chunk_blocks": chunk, "proofs_total": n_chunks * (world if mode == "headline" else 1),
                       "proofs_per_step_per_gpu": round((hi - lo) / args.steps, 2), "contexts_per_gpu": contexts,
                       "parallelism": "independent chunk-proofs per rank, no data-path collective" if mode == "headline" else "chunk range per rank + one all-gather of proof bytes",
                       "circuit_model": CIRCUIT_MODEL_NOTE},
            "proofs_verified": "%d/%d" % (acc_sum, tot_sum), "wrong_ciphertext_rejected": bool(neg_sum == world),
            "setup_s": round(setup_s, 2),
            "phase_ms_last_proof_avg": {k: round(v / args.steps, 2) for k, v in phase.items()},
            "roofline": {"bound": "hbm", "kernel": "k_accumulate (Pippenger bucket accumulation)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "peak_measured_stream_copy": copy_gbs,
                         "launches": launches, "avg_launch_ms": round(acc_ms / max(launches, 1), 4), "algorithmic_bytes_per_launch": round(128.0 * pts / max(launches, 1)),
                         "launch_overlap": round(acc_ms / (1e3 * elapsed), 3),
                         "one_context_probe": serial,
                         "int_multiplier": {"unit": "T v_mad_u64_u32/s", "achieved": round(mads / (acc_ms / 1e3), 2) if acc_ms > 0 else 0.0, "peak": 28.1,
                                            "frac": round(mads / (acc_ms / 1e3) / 28.1, 4) if acc_ms > 0 else 0.0, "frac_of_wall": round(mads / elapsed / 28.1, 4),
                                            "note": "the kernel's real roof: 3416 v_mad_u64_u32 per mixed add (6 products x 378 + 2 squarings x 287 + one two-product sum with a shared reduction, 574), one mixed add per "
                                                    "(point, window) pair; peak = rate of the same Fq product stream in isolation (tools/ubench/rates.hip: 74.4 G products/s x 378). frac divides by the SUM of in-situ "
                                                    "launch durations: launches of concurrent prover contexts overlap (launch_overlap = that sum / wall), so in-situ durations are stretched -- one_context_probe has the "
                                                    "un-overlapped figure; frac_of_wall divides by the whole timed region."},
                         "note": "integer-ALU bound (10 Fq limb products, 9 Montgomery reductions = 3416 v_mad_u64_u32 per mixed add); HBM fraction of O(1%) is the expected regime (BASELINE.md §3); traffic is ~16x the "
                                 "algorithmic bytes by construction: Pippenger gathers every 112-byte base once per window (15 windows), ~0.9 TB/s, not the limiter"},
        }
        if acc_sum != tot_sum or neg_sum != world or tot_sum != out["config"]["proofs_total"]:
            out["error"] = "verification failure"
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(samples_chunk=args.cpu_chunk_samples, chunk_blocks=chunk if chunk > 1 else 6)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = round(value / cb["value"], 2)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        if dist_env is None:
            dist.destroy_process_group()
    return out


def main(argv=None):
    args = build_parser().parse_args(argv)
    from aes_zero_knowledge_proof_circuit_amd import api
    run(args, api)


if __name__ == "__main__":
    main()
