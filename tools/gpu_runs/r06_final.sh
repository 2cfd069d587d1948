#!/bin/bash
# round 6, final validation on the final code: smoke(), the whole GPU suite, then BASELINE configs[3]'s 65,536-block (1 MiB) message proven on ONE GPU under a 1-rank torchrun
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/smoke.log; cat $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --mode strong --blocks 65536 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_strong_65536_1rank.json 2> $O/bench_strong_65536.err
python -c "
import json; d=json.load(open('gpurun_out/r06_final/bench_strong_65536_1rank.json')); print(d['value'], d['proofs_verified'], d['ms_per_step'], d['config']['workload'][:120], d['telemetry']['sclk_mhz'], d['telemetry']['limiter_residency']['ppt'])"
