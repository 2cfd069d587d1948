# round 3: the runs behind profiles/r03_* (final code): -m gpu suite + smoke, the driver's command with the CPU leg, rocprofv3 summaries of the driver's
# command and of the one-context run, PMC passes with the gather calibration, the other bench modes, single-call latency, CPU baseline with 3 samples
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r03_final
O=gpurun_out/r03_final
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python - <<'PY'
import json, resource, subprocess, time, os
O = 'gpurun_out/r03_final/'
def run(tag, cmd, env=None, timeout=2400):
    e = dict(os.environ); e.update(env or {})
    t0 = time.time(); r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    p = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=timeout)
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN); wall = time.time() - t0
    open(O + tag + '.json', 'w').write(p.stdout); open(O + tag + '.err', 'w').write(p.stderr[-4000:])
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        host = 'wall %.1f s, user %.1f s, sys %.1f s -> %.2f busy host cores' % (wall, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / wall)
        print(tag, d['value'], d['unit'], d['proofs_verified'], 'n_gpus', d['n_gpus'], d['scaling'], host, flush=True)
        open(O + tag + '.host.txt', 'w').write(host + '\n')
    except Exception as ex:
        print(tag, 'FAILED', ex, p.stderr[-600:], flush=True)
run('bench_driver_command', ['python', 'bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5'])
run('bench_strong_1rank_8192', ['python', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29533', 'bench.py', '--gpus', '1', '--mode', 'strong', '--steps', '20', '--warmup', '2', '--no-cpu-baseline'])
run('bench_chunk4', ['python', 'bench.py', '--chunk', '4', '--steps', '8', '--warmup', '2', '--no-cpu-baseline'])
run('bench_batch1024', ['python', 'bench.py', '--mode', 'batch', '--proofs', '1024', '--steps', '4', '--warmup', '1', '--no-cpu-baseline'])
run('bench_64_blocks', ['python', 'bench.py', '--blocks', '64', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'])
run('bench_gpus2_one_gpu_rehearsal', ['python', 'bench.py', '--gpus', '2', '--blocks', '1024', '--steps', '4', '--warmup', '1', '--contexts', '8', '--no-cpu-baseline', '--serial-probe', '0'], {'ZKAES_BENCH_ONE_GPU': '1', 'ZKAES_BENCH_BACKEND': 'gloo'})
run('bench_cpu_baseline_3_samples', ['python', 'bench.py', '--blocks', '512', '--steps', '2', '--warmup', '1', '--cpu-chunk-samples', '3'])
PY
prof() { name=$1; shift; rm -rf $O/prof_$name; timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r03 -- python bench.py "$@" > $O/bench_under_rocprof_$name.json 2> $O/bench_under_rocprof_$name.err
  db=$(find $O/prof_$name -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" $O/kernel_stats_$name.md "rocprofv3 --kernel-trace --stats -- python bench.py $*" | head -16
  rm -rf $O/prof_$name
}
prof driver_command --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
prof serial --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline
timeout 900 python tools/pmc_accumulate.py 22 r03_tables 20 > $O/pmc_tables.log 2>&1; tail -12 $O/pmc_tables.log; cp gpurun_out/r03_tables_pmc_k_accumulate.json $O/ 2>/dev/null
timeout 600 python tools/latency.py > $O/latency.json 2> $O/latency.err; tail -c 400 $O/latency.json
