#!/bin/bash
# round 6, third GPU call: parity of the compact reduction / class-sum / division kernels, then same-box A/B against the library at the previous commit
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_third; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_kernels.log; cat $O/pytest_kernels.log
timeout 1200 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "not 4096 and not batch_1024 and not 8190" 2>&1 | tail -8 > $O/pytest_marlin.log; cat $O/pytest_marlin.log
timeout 1500 python tools/ab_lib.py --rounds 2 --bench "--gpus 1 --steps 8 --warmup 2 --blocks 1536 --no-cpu-baseline --alt-proofs 0 --calibrate-s 0 --latency-samples 9" main tools/ab/libzkaes_base.so > $O/ab_reduce_terms.txt 2>&1; cat $O/ab_reduce_terms.txt
