# round 3, step 20: k_accumulate at 3 and 4 waves per SIMD (ZK_ACC_WAVES; the select / mid-addition-prefetch form needs 173 VGPRs: 168 costs 9 spilled dwords, 128 costs 61)
mkdir -p gpurun_out/r03_step20
O=gpurun_out/r03_step20
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
: > $O/ab.txt
for w in 3 4 2; do
ZK_EXTRA_DEFINES="-DZK_ACC_WAVES=$w" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_ACC_WAVES=$w" | tee -a $O/ab.txt; probe w$w 2>&1 | tee -a $O/ab.txt
done
