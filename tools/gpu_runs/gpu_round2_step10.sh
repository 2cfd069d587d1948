set -x
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" timeout 600 python bench.py --mode batch --proofs 512 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_$name.json 2> gpurun_out/r02_bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["proofs_verified"], "setup", d["setup_s"], "acc avg", d["roofline"]["avg_launch_ms"], "overlap", d["roofline"]["launch_overlap"])
except Exception as e:
    print("$name ERR", e); print(open("gpurun_out/r02_bench_$name.err").read()[-1500:])
PY
}
run batch_notab A=1
run batch_tab20 ZKAES_MSM_TABLES=1
run batch_tab20_min200 ZKAES_MSM_TABLES=1 ZKAES_MSM_TABLE_MIN=200000
run batch_tab18_min200 ZKAES_MSM_TABLES=1 ZKAES_MSM_TABLE_C=18 ZKAES_MSM_TABLE_MIN=200000
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r02_bench_driver_cmd.json 2> gpurun_out/r02_bench_driver_cmd.err; tail -c 600 gpurun_out/r02_bench_driver_cmd.json; tail -4 gpurun_out/r02_bench_driver_cmd.err
timeout 300 python bench.py --blocks 64 --steps 2 --warmup 1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_64block.json 2>/dev/null; tail -c 300 gpurun_out/r02_bench_64block.json
