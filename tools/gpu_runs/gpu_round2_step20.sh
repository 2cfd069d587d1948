mkdir -p gpurun_out
out=gpurun_out/r02_batch_contexts.txt; : > $out
python -c "import __graft_entry__ as g; g.smoke()" | tee -a $out
for cfg in "16 16" "24 24" "32 32" "24 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout 600 python bench.py --mode batch --proofs 1024 --steps 4 --warmup 1 --contexts $1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_batchctx_$1_$2.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_batchctx_$1_$2.json').read().strip().splitlines()[-1]);print('batch contexts=$1 queues=$2', d['value'], d['proofs_verified'])" | tee -a $out
done
