# round 3, step 25: bucket visiting order by a counting sort on the 13-bit size key (3 launches) against the generic rocPRIM sort (ZKAES_MSM_ORDER=sort, ~16 launches per MSM)
mkdir -p gpurun_out/r03_step25
O=gpurun_out/r03_step25
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "msm or table or aes96 or full_size or skewed or ops_proofs or presplit or partition or sharded or fold" > $O/pytest.log 2>&1; tail -1 $O/pytest.log
probe() {
python - <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22):
    t, a = api.msm_bench_synth(n, 20, 6)
    print("n=2^%d table c=20: total %.3f ms  accumulate %.3f ms" % (n.bit_length() - 1, t, a), flush=True)
PY
timeout 300 python tools/latency.py > $O/latency_$1.json 2>/dev/null; python -c "
import json; l=json.load(open('$O/latency_$1.json')); print('$1 latency', {k: v['median_ms'] for k, v in l.items()})"
for r in 1 2; do
timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$1_$r.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_$1_$r.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['proofs_verified'], d['roofline']['one_context_probe']['avg_launch_ms'], d['roofline']['one_context_probe']['ms_per_proof'])"
done
}
echo "== counting (default)" | tee $O/ab.txt; probe counting 2>&1 | tee -a $O/ab.txt
export ZKAES_MSM_ORDER=sort
echo "== ZKAES_MSM_ORDER=sort" | tee -a $O/ab.txt; probe sort 2>&1 | tee -a $O/ab.txt
