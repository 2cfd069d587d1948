# round 3, step 23: D -+ C lazy as well (ZK_TE_LAZY=2, 3,550 VALU instructions per bucket addition) against ZK_TE_LAZY=1 (3,603); parity subset on the new default first;
# then the PMC passes of the isolated kernel (tools/pmc_accumulate.py, which now calibrates the coalesced-stream scale on k_split_hist)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_step23
O=gpurun_out/r03_step23
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
echo "== ZK_TE_LAZY=2" | tee $O/ab.txt; probe lazy2 2>&1 | tee -a $O/ab.txt
timeout 900 python tools/pmc_accumulate.py 22 r03_tables 20 > $O/pmc_tables.log 2>&1; tail -12 $O/pmc_tables.log; cp gpurun_out/r03_tables_pmc_k_accumulate.json $O/ 2>/dev/null
ZK_EXTRA_DEFINES="-DZK_TE_LAZY=1" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "== ZK_TE_LAZY=1" | tee -a $O/ab.txt; probe lazy1 2>&1 | tee -a $O/ab.txt
