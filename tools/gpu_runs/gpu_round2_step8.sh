set -x
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
prof() { name=$1; shift; rm -rf gpurun_out/prof_$name; timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o r02 -- python bench.py "$@" > gpurun_out/r02_bench_under_rocprof_$name.json 2> gpurun_out/r02_bench_under_rocprof_$name.err
  db=$(find gpurun_out/prof_$name -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/r02_kernel_stats_$name.md "rocprofv3 --kernel-trace --stats -- python bench.py $*" | head -30
  rm -rf gpurun_out/prof_$name
}
prof default --blocks 1024 --steps 4 --warmup 1 --no-cpu-baseline
prof serial --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline
timeout 900 python tools/pmc_accumulate.py 22 r02 > gpurun_out/r02_pmc.log 2>&1; tail -5 gpurun_out/r02_pmc.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/affine_batch.hip -o /tmp/affine_batch && timeout 300 /tmp/affine_batch > gpurun_out/r02_affine_batch.txt 2>&1; cat gpurun_out/r02_affine_batch.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --mode strong --blocks 4096 --steps 4 --warmup 1 > gpurun_out/r02_bench_strong_1rank.json 2> gpurun_out/r02_bench_strong_1rank.err; tail -c 700 gpurun_out/r02_bench_strong_1rank.json
timeout 900 python bench.py --mode batch --proofs 1024 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_batch1024.json 2> gpurun_out/r02_bench_batch1024.err; tail -c 500 gpurun_out/r02_bench_batch1024.json
timeout 600 python bench.py --chunk 4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_chunk4.json 2> gpurun_out/r02_bench_chunk4.err; tail -c 300 gpurun_out/r02_bench_chunk4.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_default_b.json 2> gpurun_out/r02_bench_default_b.err; python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_default_b.json').read().strip().splitlines()[-1]);print('default again', d['value'], d['roofline']['one_context_probe'])"
ZKO_TIMING=1 timeout 300 python -c "
import sys, time
sys.path.insert(0,'.')
from oracle import zko
from bench import synthetic
cs,_ = zko.synth_aes(bytes(16), bytes(16)); ix = zko.Index(cs)
t=time.perf_counter(); cs,_ = zko.synth_aes(synthetic(16,1), synthetic(16,2)); print('synth', time.perf_counter()-t); p = ix.prove(cs); print('total', time.perf_counter()-t)
" > gpurun_out/r02_oracle_timing_c1.txt 2>&1; cat gpurun_out/r02_oracle_timing_c1.txt
