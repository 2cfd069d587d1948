# round 3, step 21: reduction rows with m = ~(t + 2^28 - 1) mod 2^28 (one v_bitop3_b32) and te_madd_signed with lazy Y1 -+ X1, 2 Z1, E, H (ZK_TE_LAZY=1):
# 3,820 / 3,691 VALU instructions per bucket addition against 3,926.  Parity subset on each build, then the probes.
mkdir -p gpurun_out/r03_step21
O=gpurun_out/r03_step21
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
echo "== ZK_TE_LAZY=0 (new reduction rows)" | tee $O/ab.txt; probe carry 2>&1 | tee -a $O/ab.txt
ZK_EXTRA_DEFINES="-DZK_TE_LAZY=1" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_TE_LAZY=1" | tee -a $O/ab.txt; probe lazy 2>&1 | tee -a $O/ab.txt
