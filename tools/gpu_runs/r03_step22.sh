# round 3, step 22: products of the hot loop with the reduction rows' "+ 2^28 - 1" as the start value of the column chains (ff28.cuh mul_biased, ZK_TE_BIASED):
# 3,603 instead of 3,691 VALU instructions per bucket addition.  Parity subset, probes, then the same with ZK_TE_BIASED=0.
mkdir -p gpurun_out/r03_step22
O=gpurun_out/r03_step22
probe() {
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or full_size or skewed or msm_matches or ops_proofs or presplit" > $O/pytest_$1.log 2>&1; tail -1 $O/pytest_$1.log
python - <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 22, 1 << 23):
    t, a = api.msm_bench_synth(n, 20, 4)
    print("n=2^%d table c=20: total %.3f ms  accumulate %.3f ms" % (n.bit_length() - 1, t, a), flush=True)
PY
for r in 1 2; do
timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$1_$r.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_$1_$r.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['proofs_verified'], d['roofline']['one_context_probe']['avg_launch_ms'], d['roofline']['one_context_probe']['ms_per_proof'])"
done
}
echo "== ZK_TE_BIASED=1" | tee $O/ab.txt; probe biased 2>&1 | tee -a $O/ab.txt
ZK_EXTRA_DEFINES="-DZK_TE_BIASED=0" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_TE_BIASED=0" | tee -a $O/ab.txt; probe plain 2>&1 | tee -a $O/ab.txt
