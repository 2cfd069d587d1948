# round 3, step 4: instruction-rate probes (FP64 FMA field products, global atomics), single-call latency with and without tables, PMC passes of the
# Edwards table-mode k_accumulate, the driver's own command with the CPU leg and host CPU time, `bench.py --gpus 2` launching its own ranks (one-GPU rehearsal)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_step4
O=gpurun_out/r03_step4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/rates.hip -o /tmp/rates && timeout 300 /tmp/rates > $O/rates.txt 2>&1; cat $O/rates.txt
timeout 600 python tools/latency.py > $O/latency_default.json 2> $O/latency_default.err; tail -c 600 $O/latency_default.json
ZKAES_MSM_TABLES=2 timeout 600 python tools/latency.py > $O/latency_tables.json 2> $O/latency_tables.err; tail -c 600 $O/latency_tables.json
timeout 900 python tools/pmc_accumulate.py 22 r03_tables 20 > $O/pmc_tables.log 2>&1; tail -30 $O/pmc_tables.log; cp gpurun_out/r03_tables_pmc_k_accumulate.json $O/ 2>/dev/null
python - <<'PY'
import json, resource, subprocess, time
def run(tag, cmd, env=None):
    import os
    e = dict(os.environ); e.update(env or {})
    t0 = time.time(); r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    p = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=1500)
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN); wall = time.time() - t0
    open('gpurun_out/r03_step4/%s.json' % tag, 'w').write(p.stdout); open('gpurun_out/r03_step4/%s.err' % tag, 'w').write(p.stderr[-4000:])
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(tag, 'value', d['value'], d['proofs_verified'], 'n_gpus', d['n_gpus'], 'wall %.1f s user %.1f s sys %.1f s -> %.2f busy cores' % (wall, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / wall), flush=True)
    except Exception as ex:
        print(tag, 'FAILED', ex, p.stderr[-500:], flush=True)
run('bench_nocpu_sleep', ['python', 'bench.py', '--gpus', '1', '--steps', '8', '--warmup', '2', '--no-cpu-baseline'])
run('bench_nocpu_spin', ['python', 'bench.py', '--gpus', '1', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--serial-probe', '0'], {'ZKAES_WAIT': 'spin'})
run('bench_gpus2_selflaunch', ['python', 'bench.py', '--gpus', '2', '--blocks', '1024', '--steps', '4', '--warmup', '1', '--contexts', '8', '--no-cpu-baseline', '--serial-probe', '0'], {'ZKAES_BENCH_ONE_GPU': '1', 'ZKAES_BENCH_BACKEND': 'gloo'})
run('bench_driver_command', ['python', 'bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5'])
PY
