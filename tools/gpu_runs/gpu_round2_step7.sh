set -x
mkdir -p gpurun_out
nproc > gpurun_out/r02_nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gputest_7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_7.log
tail -14 gpurun_out/r02_gputest_7.log
( time timeout 1200 python bench.py ) > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 2500 gpurun_out/r02_bench_default.json; tail -5 gpurun_out/r02_bench_default.err
