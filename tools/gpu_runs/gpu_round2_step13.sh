set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_13.log
tail -4 gpurun_out/r02_gputest_13.log
timeout 600 python tools/latency.py > gpurun_out/r02_latency.json 2>&1; tail -c 1200 gpurun_out/r02_latency.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_step13.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_step13.json').read().strip().splitlines()[-1]);print('default', d['value'], d['proofs_verified'], d['roofline']['one_context_probe'])"
