#!/usr/bin/env python3
"""Where does a LONE 16-byte encrypt() spend its time?  (run ON THE GPU BOX, from the repo root)

    python tools/lone_timeline.py [nbytes=16] [tag=r04]   ->  gpurun_out/<tag>_lone_timeline_<nbytes>.md

One rocprofv3 --kernel-trace pass over a child that proves the same message a few times with pauses in between; the kernels of the LAST proof are laid on a
timeline: wall (first start to last end), busy (union of kernel intervals), the idle gaps with the kernels either side of them, and the time per kernel.
The gaps are host work on the critical path (transcript hashing between rounds, result copies, launches the device is waiting for).
"""
import csv
import os
import re
import subprocess
import sys

nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
CHILD = r'''
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("ZKAES_CONTEXTS", "1")
from aes_zero_knowledge_proof_circuit_amd import api, sharding
pk, vk = api.synthesize_keys(%d)
key, msg = sharding.rank_message(0, %d)
for _ in range(3): api.encrypt(msg, key, pk)
ts = []
for _ in range(4):
    time.sleep(0.2)
    t = time.perf_counter(); api.encrypt(msg, key, pk); ts.append(1e3 * (time.perf_counter() - t))
print("LONE_MS", " ".join("%%.2f" %% x for x in ts), flush=True)
''' % (nbytes, nbytes // 16)
out_dir = os.path.join("gpurun_out", "lone_%s_%d" % (tag, nbytes))
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
p = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "lone", "--", sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=1200)
host_ms = [l for l in p.stdout.splitlines() if l.startswith("LONE_MS")]
path = None
for root, _, files in os.walk(out_dir):
    for f in files:
        if f == "lone_kernel_trace.csv":
            path = os.path.join(root, f)
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last proof: everything after the last pause of more than 100 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(e for _, e, _ in rows[max(0, i - 64):i]) > 100e6:
        cut = i
proof = rows[cut:]


def short(name):
    base = re.sub(r"<.*", "", name.replace("void ", "")).split("(")[0].replace("zk::gpu::", "")
    if "rocprim" in name:
        m = re.search(r"detail::(\w+)", name.split("target_arch)", 1)[-1])
        return "rocprim::" + (m.group(1) if m else "kernel")
    return base


t0, t1 = proof[0][0], max(e for _, e, _ in proof)
busy, gaps, cur_end, last = 0, [], proof[0][0], None
for s, e, n in proof:
    if s > cur_end:
        gaps.append((s - cur_end, (cur_end - t0) / 1e6, short(last), short(n)))
        busy += 0
    busy += max(0, e - max(s, cur_end))
    if e > cur_end:
        cur_end, last = e, n
per = {}
for s, e, n in proof:
    k = short(n)
    a = per.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
lines = ["# timeline of one lone %d-byte encrypt() (last of four; rocprofv3 --kernel-trace)" % nbytes, "",
         "host-side wall of the four proofs under the tracer: %s ms" % (host_ms[0][8:] if host_ms else "?"), "",
         "kernels: %d   first start to last end: %.2f ms   device busy (union of kernel intervals): %.2f ms   idle inside: %.2f ms in %d gaps" %
         (len(proof), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)), "",
         "## idle gaps over 15 us", "", "| at ms | gap us | after | before |", "|---:|---:|---|---|"]
for g, at, a, b in sorted(gaps, key=lambda x: x[1]):
    if g > 15e3:
        lines.append("| %.2f | %.0f | %s | %s |" % (at, g / 1e3, a, b))
small = sum(g for g, _, _, _ in gaps if g <= 15e3)
lines += ["", "gaps of 15 us or less: %d, %.2f ms in total" % (sum(1 for g in gaps if g[0] <= 15e3), small / 1e6), "", "## kernel time (sum of durations; overlapping kernels count twice)", "",
          "| kernel | launches | total us | avg us |", "|---|---:|---:|---:|"]
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    lines.append("| %s | %d | %.0f | %.1f |" % (k, c, t / 1e3, t / c / 1e3))
lines += ["", "## in order", "", "| start ms | us | kernel |", "|---:|---:|---|"]
for s, e, n in proof:
    lines.append("| %.3f | %.1f | %s |" % ((s - t0) / 1e6, (e - s) / 1e3, short(n)))
dst = os.path.join("gpurun_out", "%s_lone_timeline_%d.md" % (tag, nbytes))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
