O=gpurun_out/r03_step25; mkdir -p $O
run() { for r in 1 2 3; do
ZKAES_MSM_ORDER=$1 timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > $O/bench2_$1_$r.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench2_$1_$r.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['proofs_verified'])"
done; }
run sort; run counting; run sort; run counting
