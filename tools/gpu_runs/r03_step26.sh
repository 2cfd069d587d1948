# round 3, step 26: lazy subtraction as v_sad_u32 (ZK_FF28_SAD) and negative digits' records gathered with their first two coordinates swapped (ZK_TE_PRESWAP):
# 3,550 -> 3,511 -> 3,498 VALU instructions per bucket addition.  Parity subset on the new default, then A/B/C on one box.
mkdir -p gpurun_out/r03_step26
O=gpurun_out/r03_step26
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or full_size or skewed or msm_matches or ops_proofs or presplit" > $O/pytest.log 2>&1; tail -1 $O/pytest.log
probe() {
python - <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 22, 1 << 23):
    t, a = api.msm_bench_synth(n, 20, 6)
    print("n=2^%d table c=20: total %.3f ms  accumulate %.3f ms" % (n.bit_length() - 1, t, a), flush=True)
PY
for r in 1 2; do
timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > $O/bench_$1_$r.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_$1_$r.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['proofs_verified'])"
done
}
echo "== ZK_FF28_SAD=1 ZK_TE_PRESWAP=1 (new)" | tee $O/ab.txt; probe new 2>&1 | tee -a $O/ab.txt
ZK_EXTRA_DEFINES="-DZK_TE_PRESWAP=0" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_FF28_SAD=1 ZK_TE_PRESWAP=0" | tee -a $O/ab.txt; probe sad 2>&1 | tee -a $O/ab.txt
ZK_EXTRA_DEFINES="-DZK_TE_PRESWAP=0 -DZK_FF28_SAD=0" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_FF28_SAD=0 ZK_TE_PRESWAP=0 (before)" | tee -a $O/ab.txt; probe old 2>&1 | tee -a $O/ab.txt
