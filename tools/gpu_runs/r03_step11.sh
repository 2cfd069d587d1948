# round 3, step 11: knob check on the final code -- prover contexts / hardware queues / pipeline depth (2048-block message, 4 steps)
mkdir -p gpurun_out/r03_step11
O=gpurun_out/r03_step11; : > $O/knobs.txt
for cfg in "16 2 16" "12 2 16" "20 2 24" "24 3 24" "32 3 32" "16 3 16" "8 2 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --contexts $1 --pipeline $2 --no-cpu-baseline --serial-probe 0 > $O/bench_$1_$2.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/bench_$1_$2.json').read().strip().splitlines()[-1]);print('contexts=$1 pipeline=$2 hw_queues=$3', d['value'], d['proofs_verified'])" | tee -a $O/knobs.txt
done
