set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "not 4096 and not 1024" > gpurun_out/r02_gputest_17.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_17.log
tail -4 gpurun_out/r02_gputest_17.log
python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python tools/latency.py > gpurun_out/r02_latency_lanes.json 2>&1; tail -c 1500 gpurun_out/r02_latency_lanes.json
ZKAES_LANES=0 timeout 600 python tools/latency.py > gpurun_out/r02_latency_nolanes.json 2>&1; tail -c 400 gpurun_out/r02_latency_nolanes.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_step17.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_step17.json').read().strip().splitlines()[-1]);print('default', d['value'], d['proofs_verified'], d['setup_s'], d['roofline']['one_context_probe'])"
