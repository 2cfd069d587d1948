set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "not 4096 and not 1024" > gpurun_out/r02_gputest_16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_16.log
tail -4 gpurun_out/r02_gputest_16.log
python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python tools/latency.py > gpurun_out/r02_latency.json 2>&1; tail -c 1500 gpurun_out/r02_latency.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_step16.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_step16.json').read().strip().splitlines()[-1]);print('default', d['value'], d['proofs_verified'], d['setup_s'], d['roofline']['one_context_probe'])"
