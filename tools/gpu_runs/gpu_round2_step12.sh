set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "not 4096 and not 1024 and not aes32 and not aes16_proof" > gpurun_out/r02_gputest_12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_12.log
tail -6 gpurun_out/r02_gputest_12.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_$name.json 2> gpurun_out/r02_bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["proofs_verified"], "acc avg", d["roofline"]["avg_launch_ms"], "overlap", d["roofline"]["launch_overlap"], d["roofline"]["one_context_probe"])
except Exception as e:
    print("$name ERR", e); print(open("gpurun_out/r02_bench_$name.err").read()[-1500:])
PY
}
run fold_dflt A=1
run fold_c19 ZKAES_MSM_TABLE_C=19
run fold_c21 ZKAES_MSM_TABLE_C=21
run fold_c22 ZKAES_MSM_TABLE_C=22
