#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_big; mkdir -p $O
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o big -- python bench.py --big-only --big-chunk 28 --big-proofs 4 --big-contexts 1 > $O/big28_serial.json 2> $O/big28_serial.err
db=$(find $O/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_big28_serial.md "rocprofv3 --kernel-trace --stats -- python bench.py --big-only --big-chunk 28 --big-proofs 4 --big-contexts 1" $O/rocprim_names.txt | head -24
rm -rf $O/prof
cat $O/big28_serial.json | cut -c1-400
timeout 900 python bench.py --big-only --big-chunk 28 --big-proofs 48 --big-contexts 4 > $O/big28_48.json 2> $O/big28_48.err; cat $O/big28_48.json | cut -c1-420
timeout 900 python bench.py --big-only --big-chunk 28 --big-proofs 48 --big-contexts 6 > $O/big28_48_c6.json 2> $O/big28_48_c6.err; cat $O/big28_48_c6.json | cut -c1-420
