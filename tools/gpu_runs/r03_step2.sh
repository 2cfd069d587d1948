# round 3, step 2: twisted Edwards bucket law for the SRS paths -- parity subset, isolated MSM timing (Weierstrass vs Edwards), bench
mkdir -p gpurun_out/r03_step2
O=gpurun_out/r03_step2
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or ops_proofs or chunked_message or seeded or msm_matches or full_size or aes16 or witness or encrypt_16" > $O/pytest_subset.log 2>&1
tail -5 $O/pytest_subset.log
python - > $O/msm_isolated.txt 2>&1 <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22):
    for wb, name in ((0, "weierstrass per-window"), (-1, "edwards per-window"), (20, "edwards table c=20")):
        t, a = api.msm_bench_synth(n, wb, 3)
        print("n=2^%d %-24s total %.3f ms  accumulate %.3f ms" % (n.bit_length() - 1, name, t, a), flush=True)
PY
cat $O/msm_isolated.txt
timeout 900 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_step2/bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['proofs_verified'], 'setup_s', d['setup_s'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['one_context_probe'], d['roofline'].get('inconsistent'))
PY
tail -3 $O/bench_default.err
