cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_step24; mkdir -p $O
rm -rf $O/prof; timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o r03 -- python bench.py --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
db=$(find $O/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_serial.md "rocprofv3 --kernel-trace --stats -- python bench.py --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline" $O/rocprim_names.txt | head -40
rm -rf $O/prof
