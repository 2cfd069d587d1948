# round 3, step 8: per-kernel times of ONE 2^22-point table MSM, pre-split digits off / on (rocprofv3 --kernel-trace --stats), then the A/B of step 7 again
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_step8
O=gpurun_out/r03_step8
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or full_size or skewed" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
for ps in 0 1; do
rm -rf $O/prof
ZKAES_MSM_PRESPLIT=$ps timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r03 -- python tools/ubench/msm_one.py 22 20 > $O/msm_one_$ps.txt 2>&1
db=$(find $O/prof -name "*_results.db" | head -1)
python - "$db" <<'PY' | tee $O/kernels_presplit_$ps.txt
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, total_calls, total_duration, average from top_kernels").fetchall()
for name, calls, total, avg in sorted(rows, key=lambda r: -r[2])[3:16]:
    print("%-100s calls %4d avg %8.1f us" % (re.sub(r"\(.*", "", name)[:100], calls, avg))
PY
done
rm -rf $O/prof
for ps in 1 0 1 0; do
  ZKAES_MSM_PRESPLIT=$ps timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > $O/bench_$ps.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/bench_$ps.json').read().strip().splitlines()[-1]);print('presplit=$ps', d['value'], d['proofs_verified'])" | tee -a $O/ab.txt
done
