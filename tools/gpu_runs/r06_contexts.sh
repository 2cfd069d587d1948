#!/bin/bash
# round 6: prover contexts per GPU on the final code, one box, two passes
for pass in 1 2; do for c in 8 10 12 14 16 20; do
  python bench.py --gpus 1 --steps 8 --warmup 3 --contexts $c --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --serial-probe 0 --calibrate-s 0 --big-chunk 0 2>/dev/null > /tmp/ctx.json
  python - "$c" <<'PY'
import json, sys
d = json.loads(open('/tmp/ctx.json').read().strip().splitlines()[-1])
print('contexts', sys.argv[1], d['value'], d['proofs_verified'], round(d['srs']['device_bytes_in_use_after_timed_region'] / 2**30, 1), 'GB')
PY
done; done
