#!/usr/bin/env python3
"""PMC passes for the MSM bucket-accumulation kernel (run ON THE GPU BOX, from the repo root):

    python tools/pmc_accumulate.py [lg_n=22] [tag=r03] [window_bits=20]   ->  gpurun_out/<tag>_pmc_k_accumulate.json

window_bits: 20 = the prover's table path (13 balanced windows, one bucket set, twisted Edwards bases of 168 B), -1 = per-window buckets on Edwards bases
(15 windows of 17 bits: the lone-call path), 0 = per-window buckets on the Weierstrass model (112-byte bases: the generic zkaes_msm path).

Three separate rocprofv3 runs of `tools/ubench/msm_one.py <lg_n>` (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in a third), each with
`--pmc ... --kernel-trace --output-format csv` only (MI355X_MICROARCH.md, HBM / rocprofv3 sections).  The absolute FETCH_SIZE scale is calibrated on
kernels of the same run whose byte counts are known exactly: k_convert_bases / k_convert_bases_te (reads 96 B, writes 112 B / 168 B per point, 16 B-per-lane
array-of-structures access -- the pattern of k_accumulate's gathers) and k_digits / k_part_hist (reads 32 B per scalar, fully coalesced).
"""
import csv
import json
import os
import subprocess
import sys

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
wbits = int(sys.argv[3]) if len(sys.argv) > 3 else 20
edwards = wbits != 0
n = 1 << lg
out_dir = os.path.join("gpurun_out", "pmc_" + tag)
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "sq": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
          "valu": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"]}


def collect(name, counters):
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc_" + name, "--",
           sys.executable, "tools/ubench/msm_one.py", str(lg), str(wbits)]
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    path = None
    for root, _, files in os.walk(out_dir):
        for f in files:
            if f == "pmc_%s_counter_collection.csv" % name:
                path = os.path.join(root, f)
    rows = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        short = "k_accumulate" if "k_accumulate<" in k else ("k_convert_bases" if "k_convert_bases" in k else ("k_digits" if ("k_digits" in k or "k_part_hist" in k) else None))   # k_part_hist: the partition's pass over the scalars (reads 32 B each, coalesced; its writes are not in FETCH_SIZE)
        if short:
            rows.setdefault((short, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in rows.items()}, {k: len(v) for k, v in rows.items()}


def gather_calibration():
    """FETCH_SIZE of tools/ubench/gather_probe.hip (one 168-byte record per lane at random places of a 6 GB array, no reuse) against the bytes it
    touches at 64- and 128-byte granularity: how this counter tallies k_accumulate's kind of gather on gfx950"""
    exe = "/tmp/gather_probe"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "tools/ubench/gather_probe.hip", "-o", exe], check=True)
    cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc_gather", "--", exe]
    out = subprocess.run(cmd, check=True, env=env, capture_output=True, text=True, timeout=600).stdout
    line = [l for l in out.splitlines() if l.startswith("gather_probe")][0].split()
    info = {line[i]: int(line[i + 1]) for i in range(1, len(line), 2)}
    vals_ = []
    for root, _, files in os.walk(out_dir):
        for f in files:
            if f == "pmc_gather_counter_collection.csv":
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if "k_gather_probe" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                        vals_.append(float(r["Counter_Value"]))
    info["FETCH_SIZE_bytes"] = sum(vals_) / len(vals_) * 1024.0
    info["fetch_per_touched_64B_byte"] = info["FETCH_SIZE_bytes"] / info["touched_64B_bytes"]
    info["fetch_per_touched_128B_byte"] = info["FETCH_SIZE_bytes"] / info["touched_128B_bytes"]
    info["fetch_per_requested_byte"] = info["FETCH_SIZE_bytes"] / info["requested_bytes"]
    return info


vals, counts = {}, {}
for name, ctrs in PASSES.items():
    v, c = collect(name, ctrs)
    vals.update(v); counts.update(c)

KIB = 1024.0
nwin = 15 if wbits <= 0 else (253 + 1 + wbits - 1) // wbits
base_bytes = 192 if edwards else 112                   # one gathered base: Niels28 (te28.cuh: 168 B padded to a 64-byte aligned 192-byte record) or Affine28
n_conv = n * (nwin if wbits > 0 else 1)                # the table path converts all window copies in one launch
fetch = vals[("k_accumulate", "FETCH_SIZE")] * KIB
write = vals[("k_accumulate", "WRITE_SIZE")] * KIB
cal = {
    "k_convert_bases_known_read_bytes": 96 * n_conv, "k_convert_bases_FETCH_SIZE_bytes": vals[("k_convert_bases", "FETCH_SIZE")] * KIB,
    "k_convert_bases_known_write_bytes": base_bytes * n_conv, "k_convert_bases_WRITE_SIZE_bytes": vals[("k_convert_bases", "WRITE_SIZE")] * KIB,
    "k_digits_known_read_bytes": 32 * n, "k_digits_FETCH_SIZE_bytes": vals[("k_digits", "FETCH_SIZE")] * KIB,
}
cal["fetch_scale_aos_16B_per_lane"] = cal["k_convert_bases_FETCH_SIZE_bytes"] / cal["k_convert_bases_known_read_bytes"]
cal["fetch_scale_coalesced_stream"] = cal["k_digits_FETCH_SIZE_bytes"] / cal["k_digits_known_read_bytes"]
cal["write_scale"] = cal["k_convert_bases_WRITE_SIZE_bytes"] / cal["k_convert_bases_known_write_bytes"]
gcal = gather_calibration() if edwards else None
if gcal:
    # the gathers dominate the reads: scale FETCH_SIZE by what the counter reports per byte really touched (64-byte sectors) on the probe's identical pattern
    cal["gather_probe"] = gcal
    hbm = fetch / gcal["fetch_per_touched_64B_byte"] + write / cal["write_scale"]
else:
    hbm = fetch / cal["fetch_scale_aos_16B_per_lane"] + write / cal["write_scale"]
res = {
    "kernel": "k_accumulate (%s, %s), n = 2^%d points, %d signed-digit windows%s, averages over %d launches" % (
        tag, "twisted Edwards bases, 168 B" if edwards else "Weierstrass XYZZ, 112-byte bases", lg, nwin,
        " of 19-20 bits through window tables (one bucket set)" if wbits > 0 else " of 17 bits, per-window buckets", counts[("k_accumulate", "FETCH_SIZE")]),
    "command": "rocprofv3 --pmc <one counter group> --kernel-trace --output-format csv -- python tools/ubench/msm_one.py %d %d   (separate passes: %s)" % (lg, wbits, PASSES),
    "counters": {c: vals[("k_accumulate", c)] for grp in PASSES.values() for c in grp},
    # wave-cycle accounting (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES, all in
    # quad-cycles).  rocprofv3's derived VALUBusy falls back to a gfx94x formula on gfx950 and SQ_ACTIVE_INST_VALU reads identical to SQ_INSTS_VALU
    # here (an instruction count, not busy cycles), so no "VALU busy %" is derived from it: the issue-bound argument rests on the disassembly + rates.hip.
    "wave_cycle_shares": {
        "active_inst_any": vals[("k_accumulate", "SQ_ACTIVE_INST_ANY")] / vals[("k_accumulate", "SQ_WAVE_CYCLES")],
        "wait_inst_any_issue_stall": vals[("k_accumulate", "SQ_WAIT_INST_ANY")] / vals[("k_accumulate", "SQ_WAVE_CYCLES")],
        "wait_any_parked": vals[("k_accumulate", "SQ_WAIT_ANY")] / vals[("k_accumulate", "SQ_WAVE_CYCLES")],
    },
    "valu_instructions_per_point_window": vals[("k_accumulate", "SQ_INSTS_VALU")] * 64 / (n * nwin),
    "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them",
    "calibration": cal,
    "fetch_size_raw_bytes_per_point_window": fetch / (n * nwin), "write_size_raw_bytes_per_point_window": write / (n * nwin),
    "requested_bytes_per_point_window": base_bytes + 4,
    "algorithmic_bytes_per_launch": 128 * n,
    "hbm_bytes_per_launch": hbm,
    "hbm_bytes_per_point_window": hbm / (n * nwin),
    # the GPU box has no .git: the caller passes the commit of the code it sent (ZKAES_COMMIT=$(git rev-parse --short HEAD) in the gpurun command line); bench.py quotes it
    "measured_at_commit": os.environ.get("ZKAES_COMMIT", "unrecorded"),
}
# per-SIMD VALU issue share: every VALU instruction occupies its SIMD for one quad-cycle (tools/ubench/issue_cycles.hip), SQ_INSTS_VALU counts
# wave-instructions, GRBM_GUI_ACTIVE is summed over the 8 XCDs -> cycles of the kernel = GRBM_GUI_ACTIVE / 8, 1,024 SIMDs
res["valu_issue_share_of_simd_cycles"] = res["counters"]["SQ_INSTS_VALU"] * 4.0 / (1024.0 * res["counters"]["GRBM_GUI_ACTIVE"] / 8.0)
res["waves_per_simd_average"] = res["counters"]["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * res["counters"]["GRBM_GUI_ACTIVE"] / 8.0)
dst = os.path.join("gpurun_out", "%s_pmc_k_accumulate.json" % tag)
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
