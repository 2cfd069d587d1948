#!/bin/bash
# round 6, fourth GPU call: lone-call timelines with the compact reduction kernels + the prove() split; proof-byte parity of the split; 8-rank rehearsal
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_fourth; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "not 4096 and not batch_1024 and not 8190" 2>&1 | tail -5 > $O/pytest_marlin.log; cat $O/pytest_marlin.log
timeout 600 python tools/lone_timeline.py 32 r06 > $O/lone32.txt 2>&1; head -30 $O/lone32.txt
timeout 600 python tools/lone_timeline.py 16 r06 > $O/lone16.txt 2>&1
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "eight_ranks" 2>&1 | tail -5 > $O/pytest_8ranks.log; cat $O/pytest_8ranks.log
