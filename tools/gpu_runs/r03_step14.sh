# round 3, step 14: where a lone 16-byte encrypt() spends its 34 ms: GPU kernel time (rocprofv3, lanes off = one stream) against wall time
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_step14
O=gpurun_out/r03_step14
for lanes in 0 1; do
rm -rf $O/prof
ZKAES_LANES=$lanes timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r03 -- python tools/ubench/lat_one.py 16 > $O/lat_$lanes.txt 2>&1
grep median $O/lat_$lanes.txt
db=$(find $O/prof -name "*_results.db" | head -1)
python - "$db" $lanes <<'PY' | tee $O/kernels_lanes_$lanes.txt
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, total_calls, total_duration, average from top_kernels").fetchall()
skip = ("k_table_next", "k_convert_bases", "k_fixed_base", "k_power_scalars", "k_fill_powers", "k_index", "k_twiddles", "k_lagrange")
rows = [r for r in rows if not any(s in r[0] for s in skip)]
tot = sum(r[2] for r in rows)
print("lanes=%s: GPU kernel time of 11 proofs (setup kernels excluded) %.1f ms => %.2f ms per proof" % (sys.argv[2], tot / 1e3, tot / 1e3 / 11))
for name, calls, total, avg in sorted(rows, key=lambda r: -r[2])[:22]:
    print("  %-90s calls %5d per-proof %7.2f ms avg %8.1f us" % (re.sub(r"\(.*", "", name)[:90], calls, total / 1e3 / 11, avg))
PY
done
rm -rf $O/prof
