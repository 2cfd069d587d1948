#!/usr/bin/env python3
"""A/B of libzkaes builds on one GPU box (box-to-box spread is ~4 %, so variants are only comparable inside one gpurun call).

    python tools/ab_lib.py [--bench "<bench.py args>"] [--msm LOG2N] [--rounds R] lib_a.so lib_b.so ...

For every library (interleaved, R rounds): the isolated table MSM (zkaes_msm_bench_synth, 2^LOG2N points, c = 20: whole pipeline and k_accumulate alone) and,
with --bench, one bench.py run.  "main" names the in-tree aes_zero_knowledge_proof_circuit_amd/libzkaes.so.  Variant libraries are built by hand from
csrc/build/*.o with one object swapped (see the commit that used it); nothing in the product reads a library path from the environment.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from aes_zero_knowledge_proof_circuit_amd import api
path = %(path)r
if path != "main":
    api.lib_path = lambda: path
out = {"lib": os.path.basename(path)}
if %(msm)d:
    t, a = api.msm_bench_synth(1 << %(msm)d, 20, 4)
    out["msm_ms"] = round(t, 3); out["accumulate_ms"] = round(a, 3)
bench_args = %(bench)r
if bench_args is not None:
    import io, contextlib
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(bench_args.split() + ([] if "--big-chunk" in bench_args else ["--big-chunk", "0"]))      # (an A/B compares the headline: no `big` leg)
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    out["blocks_per_s"] = d["value"]; out["verified"] = d["proofs_verified"]
    probe = d["roofline"].get("one_context_probe") or {}
    out["one_context_ms_per_proof"] = probe.get("ms_per_proof"); out["chip_ms_per_launch"] = probe.get("avg_launch_ms")
    for k in ("latency_ms", "alt"):
        if k in d:
            out[k] = d[k]
print("AB " + json.dumps(out), flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--bench", default=None)
    ap.add_argument("--msm", type=int, default=22)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    for r in range(a.rounds):
        for lib in a.libs:
            path = lib if lib == "main" else os.path.abspath(lib)
            code = CHILD % dict(root=ROOT, path=path, msm=a.msm, bench=a.bench)
            p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
            lines = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
            print(lines[-1] if lines else "AB " + json.dumps({"lib": lib, "error": (p.stderr or p.stdout)[-400:]}), flush=True)


if __name__ == "__main__":
    main()
