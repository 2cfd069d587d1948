# round 3, step 12: 16-bit sort keys for the pre-split table path (6-byte pairs through two radix passes; zero digits corrected from a list) against 32-bit keys
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_step12
O=gpurun_out/r03_step12
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py tests/test_distributed.py -m gpu -x -q -k "table or presplit or sparse or aes96 or full_size or skewed or key_flags or chunked_message or window_sum or sharded" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
for k in 1 0; do
rm -rf $O/prof
ZKAES_MSM_KEY16=$k timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r03 -- python tools/ubench/msm_one.py 22 20 > $O/msm_one_$k.txt 2>&1
db=$(find $O/prof -name "*_results.db" | head -1)
echo "== ZKAES_MSM_KEY16=$k  $(tail -1 $O/msm_one_$k.txt)" | tee -a $O/kernels.txt
python - "$db" <<'PY' | tee -a $O/kernels.txt
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, total_calls, total_duration, average from top_kernels").fetchall()
for name, calls, total, avg in sorted(rows, key=lambda r: -r[2])[3:15]:
    print("%-100s calls %4d avg %8.1f us" % (re.sub(r"\(.*", "", name)[:100], calls, avg))
PY
done
rm -rf $O/prof
for k in 1 0 1 0; do
  ZKAES_MSM_KEY16=$k timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > $O/bench_$k.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/bench_$k.json').read().strip().splitlines()[-1]);print('key16=$k', d['value'], d['proofs_verified'])" | tee -a $O/ab.txt
done
