#!/bin/bash
# round 6: the 28-block leg after the vanishing-quotient schedule fix (short expansion step first)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_big3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "twenty_eight or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --big-only --big-chunk 13 --big-proofs 8 --big-contexts 4 > $O/big13.json 2> $O/big13.err; cut -c1-420 $O/big13.json
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o big -- python bench.py --big-only --big-chunk 28 --big-proofs 4 --big-contexts 1 > $O/big28_serial.json 2> $O/big28_serial.err
db=$(find $O/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_big28_serial.md "rocprofv3 --kernel-trace --stats -- python bench.py --big-only --big-chunk 28 --big-proofs 4 --big-contexts 1" $O/rocprim_names.txt | head -30
rm -rf $O/prof
cut -c1-400 $O/big28_serial.json
timeout 900 python bench.py --big-only --big-chunk 28 --big-proofs 48 --big-contexts 4 > $O/big28_48.json 2> $O/big28_48.err; cut -c1-420 $O/big28_48.json
