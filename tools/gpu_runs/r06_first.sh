#!/bin/bash
# round 6, first GPU call: issue-cycle model of the hot loop, what the telemetry sources return on the box, the list of SQ counters, a baseline bench line of this box
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_first; mkdir -p $O
timeout 300 tools/ubench/issue_cycles.bin 3 > $O/issue_cycles.txt 2>&1
timeout 60 python tools/gpu_telemetry.py 2 > $O/telemetry_probe.txt 2>&1
timeout 120 rocprofv3-avail list > $O/counters_avail.txt 2>&1
timeout 60 rocm-smi --showclocks --showpower --showtemp --json > $O/rocm_smi.json 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_baseline.json 2> $O/bench_baseline.err
tail -c 600 $O/bench_baseline.json; cat $O/issue_cycles.txt; tail -3 $O/telemetry_probe.txt
