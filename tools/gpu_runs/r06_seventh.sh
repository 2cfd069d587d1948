#!/bin/bash
# round 6, seventh GPU call: parity of the fused overflow tail, the calibration test, the RCCL banner redirect (stdout of a 1-rank torchrun must be ONE line), latency A/B against HEAD
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_seventh; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/pytest_kernels.log; cat $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "oracle or fixture or bytes" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/pytest_marlin.log; cat $O/pytest_marlin.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --mode strong --blocks 96 --steps 2 --warmup 1 --contexts 4 --no-cpu-baseline --alt-proofs 0 --latency-samples 0 --calibrate-s 0 > $O/strong_stdout.txt 2> $O/strong_stderr.txt; echo "stdout lines: $(wc -l < $O/strong_stdout.txt)  first chars: $(head -c 40 $O/strong_stdout.txt)"; grep -c "RCCL version" $O/strong_stderr.txt
timeout 2400 python tools/ab_lib.py --rounds 3 --msm 0 --bench "--gpus 1 --steps 4 --warmup 1 --blocks 384 --no-cpu-baseline --alt-proofs 0 --calibrate-s 0 --latency-samples 9" main tools/ab/libzkaes_head.so 2>&1 | grep "^AB" > $O/ab_fused_tail.txt; python - <<'PY'
import json
for l in open('gpurun_out/r06_seventh/ab_fused_tail.txt'):
    d=json.loads(l[3:]); print(d['lib'], d.get('blocks_per_s'), d['latency_ms']['16'], d['latency_ms']['32'], d['latency_ms']['64'], d['latency_ms']['min'])
PY
