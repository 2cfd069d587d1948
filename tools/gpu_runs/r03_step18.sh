# round 3, step 18: k_accumulate with the digit's sign as 56 selects (ZK_TE_SIGN_SELECT) and the next point gathered mid-addition into the current point's registers
# (ZK_ACC_PREFETCH=2): 3,926 instead of 4,013 VALU instructions per bucket addition, 173 instead of 199 VGPRs.  Parity subset on the new build, then A/B/C.
mkdir -p gpurun_out/r03_step18
O=gpurun_out/r03_step18
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or full_size or skewed or msm_matches or ops_proofs or presplit" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
probe() {
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
echo "== ZK_ACC_PREFETCH=2 ZK_TE_SIGN_SELECT=1 (new default)" | tee $O/ab.txt; probe new 2>&1 | tee -a $O/ab.txt
ZK_EXTRA_DEFINES="-DZK_ACC_PREFETCH=1 -DZK_TE_SIGN_SELECT=1" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_ACC_PREFETCH=1 ZK_TE_SIGN_SELECT=1" | tee -a $O/ab.txt; probe select 2>&1 | tee -a $O/ab.txt
ZK_EXTRA_DEFINES="-DZK_ACC_PREFETCH=1 -DZK_TE_SIGN_SELECT=0" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_ACC_PREFETCH=1 ZK_TE_SIGN_SELECT=0 (before)" | tee -a $O/ab.txt; probe old 2>&1 | tee -a $O/ab.txt
