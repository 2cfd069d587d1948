#!/bin/bash
# rocprofv3 kernel statistics of bench.py on the GPU box, summarized into gpurun_out/<tag>/ (copy what should be judged into profiles/).
#   gpurun -- 'bash tools/gpu_runs/prof.sh r05 serial'     one prover context: un-overlapped kernel durations, launches per proof (21 proofs)
#   gpurun -- 'bash tools/gpu_runs/prof.sh r05 driver'     the driver's own command (saturated: in-situ durations)
#   gpurun -- 'bash tools/gpu_runs/prof.sh r06 batch1'     one prover context over 43 single-block proofs (BASELINE configs[4]'s shape): where a small proof's time goes
# Counters (--pmc) are collected separately: tools/pmc_accumulate.py.  Round 1-3 ran ~35 one-off scripts from this directory; they are in the git history.
TAG=${1:-r05}; MODE=${2:-serial}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/${TAG}_prof_${MODE}; mkdir -p $O
if [ "$MODE" = batch1 ]; then ARGS="--mode batch --proofs 42 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --calibrate-s 0 --big-chunk 0"
elif [ "$MODE" = serial ]; then ARGS="--blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --calibrate-s 0 --big-chunk 0"
else ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --calibrate-s 0 --big-chunk 0"; fi
rm -rf $O/prof
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o $TAG -- python bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/bench.err
db=$(find $O/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_${MODE}.md "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" $O/rocprim_kernel_names.txt | head -30
rm -rf $O/prof
tail -c 400 $O/bench_under_rocprof.json
