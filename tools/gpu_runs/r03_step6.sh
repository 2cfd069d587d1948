# round 3, step 6: A/B of 64-byte aligned, 192-byte SRS records (ZK_NIELS_PAD=1) against the packed 168-byte ones; library rebuilt on the box for the variant
mkdir -p gpurun_out/r03_step6
O=gpurun_out/r03_step6
probe() {
python - <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22, 1 << 23):
    t, a = api.msm_bench_synth(n, 20, 3)
    print("n=2^%d table c=20: total %.3f ms  accumulate %.3f ms" % (n.bit_length() - 1, t, a), flush=True)
PY
timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$1.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['proofs_verified'], d['roofline']['one_context_probe'])"
}
echo "== packed 168 B" | tee $O/ab.txt; probe packed 2>&1 | tee -a $O/ab.txt
cp aes_zero_knowledge_proof_circuit_amd/libzkaes.so /tmp/libzkaes_packed.so
ZK_EXTRA_DEFINES="-DZK_NIELS_PAD=1" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build_pad.log 2>&1; tail -1 $O/build_pad.log
echo "== padded 192 B, 64-byte aligned" | tee -a $O/ab.txt; probe padded 2>&1 | tee -a $O/ab.txt
