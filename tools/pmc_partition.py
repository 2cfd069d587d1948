#!/usr/bin/env python3
"""HBM-side traffic of the table path's grouping kernels (run ON THE GPU BOX, from the repo root):

    python tools/pmc_partition.py [lg_n=22] [tag=r05]   ->  gpurun_out/<tag>_pmc_partition.json

Two separate rocprofv3 passes of `tools/ubench/msm_one.py <lg_n> 20` (FETCH_SIZE, WRITE_SIZE: `--pmc ... --kernel-trace --output-format csv` only, as
MI355X_MICROARCH.md's HBM section prescribes) + the kernel durations of a plain --kernel-trace pass, per k_part_* / k_order_* kernel, beside the bytes each
kernel has to move by construction.  FETCH_SIZE on gfx950 tallies wide coalesced reads at half their bytes (calibrated on k_part_hist: 32 B per scalar).
"""
import csv
import json
import os
import subprocess
import sys

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
n = 1 << lg
out_dir = os.path.join("gpurun_out", "pmc_part_" + tag)
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
NAMES = ("k_part_hist", "k_part_scatter", "k_part_fine", "k_order_hist", "k_order_scan", "k_order_scatter", "k_accumulate<")


def short(k):
    for s in NAMES:
        if s in k:
            return s.rstrip("<")
    return None


def run(name, extra):
    cmd = ["rocprofv3"] + extra + ["--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", name, "--", sys.executable, "tools/ubench/msm_one.py", str(lg), "20"]
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    found = {}
    for root, _, files in os.walk(out_dir):
        for f in files:
            if f.startswith(name + "_") and f.endswith(".csv"):
                found[f[len(name) + 1:-4]] = os.path.join(root, f)
    return found


res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = run("pmc_" + ctr.lower(), ["--pmc", ctr])
    rows = {}
    for r in csv.DictReader(open(files["counter_collection"])):
        s = short(r["Kernel_Name"])
        if s and r["Counter_Name"] == ctr:
            rows.setdefault(s, []).append(float(r["Counter_Value"]))
    for s, v in rows.items():
        res.setdefault(s, {})[ctr + "_raw_avg"] = sum(v) / len(v)
files = run("trace", [])
rows = {}
for r in csv.DictReader(open(files["kernel_trace"])):
    s = short(r["Kernel_Name"])
    if s:
        rows.setdefault(s, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for s, v in rows.items():
    res.setdefault(s, {})["avg_us"] = round(sum(v) / len(v), 1)
    res[s]["launches"] = len(v)
pairs = n * 13
must = {"k_part_hist": {"read": 32 * n, "write": 64 * n}, "k_part_scatter": {"read": 64 * n, "write": 6 * pairs}, "k_part_fine": {"read": 2 * (2 * pairs) + 4 * pairs, "write": 4 * pairs}}
for s, m in must.items():
    if s in res:
        res[s]["bytes_by_construction"] = m
out = {"n": n, "pairs": pairs, "units": "FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them (KB on this build if the values look 1000x small); durations in us", "kernels": res}
path = os.path.join("gpurun_out", tag + "_pmc_partition.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
