# round 3: rocprofv3 summary of the driver's command on the final commit (rocPRIM algorithms named)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_final_prof; mkdir -p $O
rm -rf $O/prof; timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof -o r03 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof_driver_command.json 2> $O/bench.err
db=$(find $O/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_driver_command.md "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline" | head -24
rm -rf $O/prof
tail -c 300 $O/bench_under_rocprof_driver_command.json
