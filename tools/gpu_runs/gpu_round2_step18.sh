mkdir -p gpurun_out
out=gpurun_out/r02_final_knobs.txt; : > $out
for cfg in "16 2" "12 2" "20 2" "16 1" "16 3" "24 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$(( $1 > 16 ? 24 : 16 )) timeout 600 python bench.py --steps 8 --warmup 1 --contexts $1 --pipeline $2 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_knob_$1_$2.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_knob_$1_$2.json').read().strip().splitlines()[-1]);print('contexts=$1 pipeline=$2 steps=8', d['value'], d['proofs_verified'])" | tee -a $out
done
