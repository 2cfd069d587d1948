#!/bin/bash
# round 6, sixth GPU call: parity of the canonical-scalar partition + three-way same-box A/B (digits stored / recomputed / canonical scalars stored)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_sixth; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/pytest_kernels.log; cat $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "oracle or fixture or bytes" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/pytest_marlin.log; cat $O/pytest_marlin.log
timeout 2400 python tools/ab_lib.py --rounds 3 --bench "--gpus 1 --steps 8 --warmup 2 --blocks 1536 --no-cpu-baseline --alt-proofs 0 --calibrate-s 0 --latency-samples 0" main tools/ab/libzkaes_prev.so tools/ab/libzkaes_recompute.so 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/ab_partition_three_way.txt; cat $O/ab_partition_three_way.txt
