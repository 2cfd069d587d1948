set -x
mkdir -p gpurun_out
for i in 1 2; do python tools/ubench/msm_one.py 22 0; python tools/ubench/msm_one.py 22 20; done > gpurun_out/r02_pretest_on.txt 2>&1
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_pretest_on.json 2>/dev/null
ZK_EXTRA_DEFINES="-DZK_CHEAP_PRETEST=0" python -m aes_zero_knowledge_proof_circuit_amd.build --force > /dev/null 2>&1
for i in 1 2; do python tools/ubench/msm_one.py 22 0; python tools/ubench/msm_one.py 22 20; done > gpurun_out/r02_pretest_off.txt 2>&1
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_pretest_off.json 2>/dev/null
python -m aes_zero_knowledge_proof_circuit_amd.build --force > /dev/null 2>&1
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_pretest_on2.json 2>/dev/null
cat gpurun_out/r02_pretest_on.txt gpurun_out/r02_pretest_off.txt
python - <<PY
import json
for n in ("pretest_on","pretest_off","pretest_on2"):
    d=json.loads(open("gpurun_out/r02_bench_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["proofs_verified"])
PY
