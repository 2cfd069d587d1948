# round 3, step 5: two-level bucket partition instead of digits + radix sort + bounds (table path); host waits drained before pageable copies
mkdir -p gpurun_out/r03_step5
O=gpurun_out/r03_step5
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or chunked_message or msm_matches or full_size or skewed" > $O/pytest_subset.log 2>&1
tail -5 $O/pytest_subset.log
python - > $O/msm_isolated.txt 2>&1 <<'PY'
import os
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22):
    t, a = api.msm_bench_synth(n, 20, 3)
    print("n=2^%d edwards table c=20 partition: total %.3f ms  accumulate %.3f ms  => everything but accumulate %.3f ms" % (n.bit_length() - 1, t, a, t - a), flush=True)
PY
cat $O/msm_isolated.txt
ZKAES_MSM_PARTITION=0 python - >> $O/msm_isolated.txt 2>&1 <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22):
    t, a = api.msm_bench_synth(n, 20, 3)
    print("n=2^%d edwards table c=20 radix sort: total %.3f ms  accumulate %.3f ms  => everything but accumulate %.3f ms" % (n.bit_length() - 1, t, a, t - a), flush=True)
PY
tail -2 $O/msm_isolated.txt
python - <<'PY'
import json, resource, subprocess, time, os
def run(tag, cmd, env=None):
    e = dict(os.environ); e.update(env or {})
    t0 = time.time(); r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    p = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=1500)
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN); wall = time.time() - t0
    open('gpurun_out/r03_step5/%s.json' % tag, 'w').write(p.stdout); open('gpurun_out/r03_step5/%s.err' % tag, 'w').write(p.stderr[-4000:])
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(tag, 'value', d['value'], d['proofs_verified'], 'wall %.1f s user %.1f s sys %.1f s -> %.2f busy cores' % (wall, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / wall), d['roofline'].get('one_context_probe'), flush=True)
    except Exception as ex:
        print(tag, 'FAILED', ex, p.stderr[-800:], flush=True)
run('bench_partition', ['python', 'bench.py', '--gpus', '1', '--steps', '8', '--warmup', '2', '--no-cpu-baseline'])
run('bench_radix', ['python', 'bench.py', '--gpus', '1', '--steps', '8', '--warmup', '2', '--no-cpu-baseline'], {'ZKAES_MSM_PARTITION': '0'})
run('bench_partition_spin', ['python', 'bench.py', '--gpus', '1', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--serial-probe', '0'], {'ZKAES_WAIT': 'spin'})
PY
