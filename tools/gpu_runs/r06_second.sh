#!/bin/bash
# round 6, second GPU call: the driver's bench command with calibration + telemetry, then the GPU suite on the split host code
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_second; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_second/bench.json'))
print(d['value'], d['latency_ms'], d.get('telemetry'))
print(json.dumps(d['roofline']['int_multiplier'], indent=1)[:3000])
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
