set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()"
timeout 900 python -m pytest tests -m gpu -x -q -k "not aes32 and not aes16_proof and not 1024 and not monolithic" > gpurun_out/r02_gputest_final3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest_final3.log
tail -3 gpurun_out/r02_gputest_final3.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_final3.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_final3.json').read().strip().splitlines()[-1]);print('default', d['value'], d['proofs_verified'], d['roofline']['traffic'], d['roofline']['one_context_probe'])"
