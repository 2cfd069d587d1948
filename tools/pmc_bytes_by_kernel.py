#!/usr/bin/env python3
"""Which kernels move the HBM bytes of a chunk-proof?  (run ON THE GPU BOX, from the repo root)

    python tools/pmc_bytes_by_kernel.py [tag=r05]   ->  gpurun_out/<tag>_bytes_by_kernel.md

Two separate rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with `--kernel-trace --output-format csv` only: MI355X_MICROARCH.md, HBM section) over the
one-context run of 21 six-block chunk-proofs, summed per kernel and expressed per proof.  Units as rocprofv3 reports them (KB); FETCH_SIZE on gfx950 tallies wide
coalesced reads at half their bytes (the guide's correction: double it for streaming kernels), WRITE_SIZE is uncalibrated in absolute terms but was found equal to the
bytes by construction for the coalesced writers here (k_part_hist: 64 B per scalar) -- so a kernel whose WRITE_SIZE is far above what it has to write is leaving
partly-filled lines (k_part_fine before round 5: 8.4x).
"""
import csv
import os
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out_dir = os.path.join("gpurun_out", "pmc_bytes_" + tag)
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
ARGS = "--blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --big-chunk 0".split()
PROOFS = 21
SETUP = ("k_table_next", "k_convert_bases", "k_fixed_base", "k_power_scalars", "k_fill_powers", "k_index_", "k_twiddles29", "k_lagrange", "k_stream_copy")


def short(name):
    law = " [Edwards]" if ("EdwardsLaw" in name or "AccTE" in name) else (" [XYZZ]" if ("WeierLaw" in name or "Acc28" in name) else "")
    base = re.sub(r"<.*", "", name.replace("void ", "")).split("(")[0]
    if "rocprim" in name:
        m = re.search(r"detail::(\w+)", name.split("target_arch)", 1)[-1]) or re.search(r"(radix_sort_\w+|\w*scan\w*)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return base.replace("zk::gpu::", "") + law


agg = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    name = "pmc_" + ctr.lower()
    cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", name, "--", sys.executable, "bench.py"] + ARGS
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
    path = None
    for root, _, files in os.walk(out_dir):
        for f in files:
            if f == name + "_counter_collection.csv":
                path = os.path.join(root, f)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr:
            continue
        a = agg.setdefault(short(r["Kernel_Name"]), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {"FETCH_SIZE": 0, "WRITE_SIZE": 0}})
        a[ctr] += float(r["Counter_Value"])
        a["n"][ctr] += 1
rows = sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))
lines = ["# HBM-side bytes per kernel, one-context run of %d six-block chunk-proofs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace)" % PROOFS, "",
         "command: `python bench.py %s`" % " ".join(ARGS), "",
         "FETCH_SIZE / WRITE_SIZE in MB as reported (raw; double FETCH_SIZE for wide coalesced reads on gfx950)", "",
         "| kernel | launches | FETCH_SIZE MB per proof | WRITE_SIZE MB per proof | FETCH MB per launch | WRITE MB per launch |", "|---|---:|---:|---:|---:|---:|"]
tf = tw = 0.0
for k, a in rows:
    if any(s in k for s in SETUP):
        continue
    n = max(a["n"]["FETCH_SIZE"], a["n"]["WRITE_SIZE"], 1)
    tf += a["FETCH_SIZE"]
    tw += a["WRITE_SIZE"]
    lines.append("| %s | %d | %.1f | %.1f | %.2f | %.2f |" % (k, n, a["FETCH_SIZE"] / PROOFS / 1e3, a["WRITE_SIZE"] / PROOFS / 1e3, a["FETCH_SIZE"] / n / 1e3, a["WRITE_SIZE"] / n / 1e3))
lines += ["", "all proof kernels: FETCH_SIZE %.1f MB, WRITE_SIZE %.1f MB per proof (raw)" % (tf / PROOFS / 1e3, tw / PROOFS / 1e3)]
dst = os.path.join("gpurun_out", "%s_bytes_by_kernel.md" % tag)
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:45]))
