#!/usr/bin/env python3
"""Who competes with k_accumulate for VALU issue slots?  (run ON THE GPU BOX, from the repo root)

    python tools/pmc_by_kernel.py [tag=r04]          ->  gpurun_out/<tag>_valu_by_kernel.md        (21 six-block chunk-proofs: the headline shape)
    python tools/pmc_by_kernel.py <tag> batch        ->  gpurun_out/<tag>_valu_by_kernel_16B.md    (22 single-block proofs: BASELINE configs[4]'s shape)

In the saturated bench every kernel's cost is roughly its VALU work (k_accumulate already issues at the pipe's limit, so whatever else issues displaces it): one
rocprofv3 pass (`--pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv`, counters only) over a one-context run of 21 chunk-proofs, summed per kernel and
expressed per proof and relative to k_accumulate.  Setup kernels (SRS, tables, index) are listed apart.
"""
import csv
import os
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
out_dir = os.path.join("gpurun_out", "pmc_valu_" + tag)
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
batch = len(sys.argv) > 2 and sys.argv[2] == "batch"
ARGS = "--blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --calibrate-s 0 --big-chunk 0".split()
PROOFS = 21          # 20 timed 6-block chunk-proofs + 1 warm-up
if batch:
    ARGS = "--mode batch --proofs 21 --steps 1 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --calibrate-s 0 --big-chunk 0".split()
    PROOFS = 22      # 21 timed single-block proofs + 1 warm-up
cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "valu", "--", sys.executable, "bench.py"] + ARGS
subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
path = None
for root, _, files in os.walk(out_dir):
    for f in files:
        if f == "valu_counter_collection.csv":
            path = os.path.join(root, f)
SETUP = ("k_table_next", "k_convert_bases", "k_fixed_base", "k_power_scalars", "k_fill_powers", "k_index_", "k_twiddles29", "k_lagrange", "k_stream_copy")


def short(name):
    law = " [Edwards]" if ("EdwardsLaw" in name or "AccTE" in name) else (" [XYZZ]" if ("WeierLaw" in name or "Acc28" in name) else "")
    base = re.sub(r"<.*", "", name.replace("void ", "")).split("(")[0]
    if "rocprim" in name:
        m = re.search(r"detail::(\w+)", name.split("target_arch)", 1)[-1]) or re.search(r"(radix_sort_\w+|\w*scan\w*)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return base.replace("zk::gpu::", "") + law


agg = {}
for r in csv.DictReader(open(path)):
    k = short(r["Kernel_Name"])
    a = agg.setdefault(k, {"SQ_INSTS_VALU": 0.0, "SQ_WAVES": 0.0, "launches": set()})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["launches"].add(r.get("Dispatch_Id") or r.get("Correlation_Id") or len(a["launches"]))
acc = agg.get("k_accumulate [Edwards]", {}).get("SQ_INSTS_VALU", 0.0)
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])
lines = ["# wave-level VALU instructions per kernel, one-context run of %d %s (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace)" % (PROOFS, "single-block proofs" if batch else "six-block chunk-proofs"), "",
         "command: `python bench.py %s`" % " ".join(ARGS), "",
         "| kernel | launches | M wave-instructions per proof | relative to k_accumulate | instructions per wave |", "|---|---:|---:|---:|---:|"]
tot_other = 0.0
for k, a in rows:
    if any(s in k for s in SETUP):
        continue
    v = a["SQ_INSTS_VALU"]
    if k != "k_accumulate [Edwards]":
        tot_other += v
    lines.append("| %s | %d | %.2f | %.4f | %.0f |" % (k, len(a["launches"]), v / PROOFS / 1e6, v / acc if acc else 0.0, v / a["SQ_WAVES"] if a["SQ_WAVES"] else 0.0))
lines += ["", "everything but k_accumulate: %.2f M wave-instructions per proof = %.3f of k_accumulate's" % (tot_other / PROOFS / 1e6, tot_other / acc if acc else 0.0), "",
          "setup kernels (per key, not per proof):", ""]
for k, a in rows:
    if any(s in k for s in SETUP):
        lines.append("* %s: %.1f M wave-instructions in %d launches" % (k, a["SQ_INSTS_VALU"] / 1e6, len(a["launches"])))
dst = os.path.join("gpurun_out", "%s_valu_by_kernel%s.md" % (tag, "_16B" if batch else ""))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
