#!/bin/bash
# round 6, validation of the final commit: smoke(), the whole GPU suite, the driver's command with its wall time, one-context + driver rocprof summaries
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final2; mkdir -p $O
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -3 > $O/smoke.log; cat $O/smoke.log
timeout 2700 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "$F" | tail -16 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
S=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
E=$(date +%s); echo "driver command wall: $((E-S)) s" | tee $O/bench_driver_command.wall.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_final2/bench_driver_command.json'))
print(d['value'], d['proofs_verified'], d['ms_per_step'], d['latency_ms'], d['alt']['value'] if d.get('alt') else None)
print('big', {k: d['big'].get(k) for k in ('value', 'proofs_verified', 'ms_per_proof_in_flight', 'key_setup_s', 'error', 'skipped')})
print('roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'avg_launch_ms', 'kernel_share_of_step')}, d['roofline']['int_multiplier'].get('frac'))
print('telemetry', d['telemetry'].get('sclk_mhz'), d['telemetry'].get('socket_power_w'), d['telemetry'].get('limiter_residency'))
print('cpu', d['cpu_baseline']['value'], d.get('speedup_vs_cpu_baseline'))
PY
bash tools/gpu_runs/prof.sh r06f serial > $O/prof_serial.txt 2>&1; tail -5 $O/prof_serial.txt
bash tools/gpu_runs/prof.sh r06f driver > $O/prof_driver.txt 2>&1; tail -5 $O/prof_driver.txt
