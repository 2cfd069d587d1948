#!/bin/bash
# round 6, the driver's command + rocprof summaries on the round's last code (the GPU suite of the same commit: profiles/r06_pytest_gpu.log)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final3; mkdir -p $O
S=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
E=$(date +%s); echo "driver command wall: $((E-S)) s" | tee $O/bench_driver_command.wall.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_final3/bench_driver_command.json'))
print(d['value'], d['proofs_verified'], d['ms_per_step'], d['latency_ms'], d['alt']['value'] if d.get('alt') else None)
print('big', {k: d['big'].get(k) for k in ('value', 'proofs_verified', 'proof_bytes_equal_oracle_fixture', 'error', 'skipped')})
print('roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'avg_launch_ms', 'kernel_share_of_step')}, d['roofline']['int_multiplier'].get('frac'), d['roofline']['one_context_probe'])
print('telemetry', d['telemetry'].get('sclk_mhz'), d['telemetry'].get('power_w'), d['telemetry'].get('limiter_residency', {}).get('ppt'))
print('cpu', d['cpu_baseline']['value'], d.get('speedup_vs_cpu_baseline'), d['srs']['device_bytes_in_use_after_timed_region'] / 2**30)
PY
bash tools/gpu_runs/prof.sh r06g serial > $O/prof_serial.txt 2>&1; tail -3 $O/prof_serial.txt | cut -c1-300
bash tools/gpu_runs/prof.sh r06g driver > $O/prof_driver.txt 2>&1; tail -3 $O/prof_driver.txt | cut -c1-300
