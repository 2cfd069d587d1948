set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02_gputest_final2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_final2.log
tail -9 gpurun_out/r02_gputest_final2.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r02_bench_final2.json 2> gpurun_out/r02_bench_final2.err; tail -c 300 gpurun_out/r02_bench_final2.json; tail -4 gpurun_out/r02_bench_final2.err
