set -x
mkdir -p gpurun_out
run() { name=$1; ctx=$2; shift; shift; env "$@" timeout 600 python bench.py --steps 4 --warmup 1 --contexts $ctx --serial-probe 0 --no-cpu-baseline > gpurun_out/r02_bench_$name.json 2> gpurun_out/r02_bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["proofs_verified"], "acc avg", d["roofline"]["avg_launch_ms"], "overlap", d["roofline"]["launch_overlap"], d["phase_ms_last_proof_avg"]["total_ms"])
except Exception as e:
    print("$name ERR", e); print(open("gpurun_out/r02_bench_$name.err").read()[-1500:])
PY
}
run dflt_c16 16 A=1
run min15_c16 16 ZKAES_MSM_TABLE_MIN=1500000
run dflt_c12 12 A=1
run dflt_c20_q24 20 GPU_MAX_HW_QUEUES=24
run c19_c16 16 ZKAES_MSM_TABLE_C=19
run c21_c16 16 ZKAES_MSM_TABLE_C=21
