# round 3, step 7: pre-split digits (stable 8-way split in the digit kernels + two-pass radix sort) against digits + three-pass radix sort
mkdir -p gpurun_out/r03_step7
O=gpurun_out/r03_step7
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or chunked_message or msm_matches or full_size or skewed" > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
probe() {
python - <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22):
    t, a = api.msm_bench_synth(n, 20, 3)
    print("n=2^%d table c=20: total %.3f ms  accumulate %.3f ms  rest %.3f ms" % (n.bit_length() - 1, t, a, t - a), flush=True)
PY
timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$1.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['proofs_verified'], d['roofline']['one_context_probe'])"
}
echo "== pre-split digits + 2-pass sort" | tee $O/ab.txt; probe presplit 2>&1 | tee -a $O/ab.txt
echo "== digits + 3-pass sort (ZKAES_MSM_PRESPLIT=0)" | tee -a $O/ab.txt; ZKAES_MSM_PRESPLIT=0 probe threepass 2>&1 | tee -a $O/ab.txt
echo "== pre-split again" | tee -a $O/ab.txt; probe presplit2 2>&1 | tee -a $O/ab.txt
