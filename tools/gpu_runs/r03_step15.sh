# round 3, step 15: k_reduce_l2 split into two roles (running sums / scalar product in separate workgroups): parity subset, single-call latency, bench
mkdir -p gpurun_out/r03_step15
O=gpurun_out/r03_step15
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py tests/test_distributed.py -m gpu -x -q -k "msm or table or aes96 or aes16 or ops_proofs or sharded or window_sum" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
timeout 600 python tools/latency.py > $O/latency.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/latency.json').read().strip().splitlines()[-1]);print('latency', {k.split('_')[0]: v['median_ms'] for k,v in d.items()})" | tee $O/summary.txt
ZKAES_LANES=0 timeout 300 python tools/ubench/lat_one.py 16 | tee -a $O/summary.txt
for r in 1 2; do
timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$r.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_$r.json').read().strip().splitlines()[-1]);print('bench', d['value'], d['proofs_verified'], d['roofline']['one_context_probe']['ms_per_proof'])" | tee -a $O/summary.txt
done
