set -x
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r02_gputest_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_final.log
tail -12 gpurun_out/r02_gputest_final.log
( time timeout 1200 python bench.py ) > gpurun_out/r02_bench_final_default.json 2> gpurun_out/r02_bench_final_default.err; tail -c 400 gpurun_out/r02_bench_final_default.json; tail -4 gpurun_out/r02_bench_final_default.err
prof() { name=$1; shift; rm -rf gpurun_out/prof_$name; timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o r02 -- python bench.py "$@" > gpurun_out/r02_bench_under_rocprof_$name.json 2> gpurun_out/r02_bench_under_rocprof_$name.err
  db=$(find gpurun_out/prof_$name -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/r02_kernel_stats_$name.md "rocprofv3 --kernel-trace --stats -- python bench.py $*" | head -12
  rm -rf gpurun_out/prof_$name
}
prof default --blocks 1024 --steps 4 --warmup 1 --no-cpu-baseline
prof serial --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --mode strong --blocks 4096 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_strong_1rank.json 2> gpurun_out/r02_bench_strong_1rank.err; tail -c 300 gpurun_out/r02_bench_strong_1rank.json
timeout 900 python bench.py --mode batch --proofs 1024 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_batch1024.json 2> gpurun_out/r02_bench_batch1024.err; tail -c 300 gpurun_out/r02_bench_batch1024.json
timeout 600 python bench.py --chunk 4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_chunk4.json 2> gpurun_out/r02_bench_chunk4.err; tail -c 300 gpurun_out/r02_bench_chunk4.json
timeout 600 python tools/latency.py > gpurun_out/r02_latency.json 2>&1; tail -3 gpurun_out/r02_latency.json
