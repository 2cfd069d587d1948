set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02_gputest_1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_1.log
tail -30 gpurun_out/r02_gputest_1.log
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_prio1.json 2> gpurun_out/r02_bench_prio1.err; tail -c 1500 gpurun_out/r02_bench_prio1.json
ZKAES_STREAM_PRIORITY=0 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_prio0.json 2> gpurun_out/r02_bench_prio0.err; tail -c 600 gpurun_out/r02_bench_prio0.json
