#!/bin/bash
# round 6: the `big` leg by itself (28 and 13 blocks per chunk-proof over a larger universal SRS) + its GPU test
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_big; mkdir -p $O
timeout 900 python bench.py --big-only --big-chunk 28 --big-proofs 16 --big-contexts 4 > $O/big28.json 2> $O/big28.err; cat $O/big28.json | cut -c1-1500; tail -3 $O/big28.err | cut -c1-300
timeout 900 python bench.py --big-only --big-chunk 13 --big-proofs 32 --big-contexts 8 > $O/big13.json 2> $O/big13.err; cat $O/big13.json | cut -c1-1500
timeout 1200 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "twenty_eight" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $O/pytest_big.log; cat $O/pytest_big.log
