#!/bin/bash
# round 6, fifth GPU call: whole GPU suite on the refactored prover (prove() split, advisor fixes, digits recomputed, lifetime + srs_hold tests), then A/B of the digit change
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_fifth; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 1500 python tools/ab_lib.py --rounds 2 --bench "--gpus 1 --steps 8 --warmup 2 --blocks 1536 --no-cpu-baseline --alt-proofs 0 --calibrate-s 0 --latency-samples 5" main tools/ab/libzkaes_prev.so 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/ab_digits.txt; cat $O/ab_digits.txt
