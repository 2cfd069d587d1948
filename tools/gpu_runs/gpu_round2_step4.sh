set -x
mkdir -p gpurun_out
run() { name=$1; ctx=$2; shift; shift; env "$@" timeout 600 python bench.py --steps 4 --warmup 1 --contexts $ctx --serial-probe 0 --no-cpu-baseline > gpurun_out/r02_bench_$name.json 2> gpurun_out/r02_bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["proofs_verified"], "acc avg", d["roofline"]["avg_launch_ms"], "overlap", d["roofline"]["launch_overlap"], d["phase_ms_last_proof_avg"])
except Exception as e:
    print("$name ERR", e); print(open("gpurun_out/r02_bench_$name.err").read()[-1500:])
PY
}
run notab_c6 6 ZKAES_MSM_TABLES=0
run notab_c16 16 ZKAES_MSM_TABLES=0
run notab_c24 24 ZKAES_MSM_TABLES=0
run tab22_c16 16 A=1
run tab22_c24 24 A=1
