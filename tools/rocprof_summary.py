#!/usr/bin/env python3
"""Summarize a rocprofv3 --kernel-trace --stats rocpd database (…_results.db) into a small markdown table.

    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db profiles/r01_kernel_stats.md "command line that was profiled"
"""
import re
import sqlite3
import sys


def short(name):
    law = " [Edwards]" if ("EdwardsLaw" in name or "AccTE" in name) else (" [XYZZ]" if ("WeierLaw" in name or "Acc28" in name) else "")
    full = name
    name = re.sub(r"<.*", "", name.replace("void ", "")) + law
    if "rocprim" in name:
        # rocPRIM launches everything through a few kernel templates: keep the template's own name (onesweep_iteration_kernel, lookback_scan_kernel, ...) and,
        # for the generic ones, the algorithm named in their configuration
        base = name.split("::")[-1] or "kernel"
        if base not in ("kernel", "trampoline_kernel", ""):
            return "rocprim::" + base
        # trampoline_kernel<config, target_arch, algorithm<...>, ...>: the algorithm is the first detail:: name after the architecture argument
        tail = full.split("target_arch)", 1)[-1]
        m = re.search(r"detail::(\w+)", tail) or re.search(r"(radix_sort_\w+|merge_sort_\w+|\w*scan\w*|partition\w*|transform\w*|histogram\w*)", full)
        return "rocprim::" + (m.group(1) if m else base)
    if law:
        return name.split("(")[0].replace(law, "") + law
    return name.split("(")[0]


def main():
    db, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    agg = {}
    if len(sys.argv) > 4:           # optional: dump the full rocPRIM kernel names (they only differ in their template arguments)
        with open(sys.argv[4], "w") as f:
            for name, calls, total, avg, pct in rows:
                if "rocprim" in name:
                    f.write("%d calls %.1f ms: %s\n" % (calls, total / 1e3, name[:400]))
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls; a[1] += total; a[2] += pct
    lines = ["# rocprofv3 --kernel-trace --stats summary", "", "command: `%s`" % cmd, "", "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---:|---:|---:|---:|"]
    for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.2f | %.1f | %.2f |" % (k, calls, total / 1e3, total / calls, pct))   # rocpd top_kernels durations are in microseconds
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
