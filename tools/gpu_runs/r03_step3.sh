# round 3, step 3: full -m gpu suite on the Edwards build, one-context rocprof summary, knock-in re-baseline
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_step3
O=gpurun_out/r03_step3
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -14 $O/gputest.log
rm -rf $O/prof_serial
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o r03 -- python bench.py --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
db=$(find $O/prof_serial -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_serial.md "rocprofv3 --kernel-trace --stats -- python bench.py --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline" | head -30
rm -rf $O/prof_serial
for k in 0 1 2 8 32; do
  ZKAES_KNOCKIN=$k timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > $O/bench_knockin_$k.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/bench_knockin_$k.json').read().strip().splitlines()[-1]);print('knockin=$k', d['value'], d['proofs_verified'])" | tee -a $O/knockin.txt
done
