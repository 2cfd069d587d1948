# round 3, step 13: single-call latency against the table threshold (MSMs below it run per-window buckets + a host-side Horner over 15 windows)
mkdir -p gpurun_out/r03_step13
O=gpurun_out/r03_step13
for m in 500000 200000 100000 30000; do
  ZKAES_MSM_TABLE_MIN=$m timeout 600 python tools/latency.py > $O/latency_$m.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/latency_$m.json').read().strip().splitlines()[-1]);print('table_min=$m', {k.split('_')[0]: v['median_ms'] for k,v in d.items()}, d['16_message_encryption']['phase_ms'])" | tee -a $O/latency.txt
done
