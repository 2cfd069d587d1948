set -x
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
prof() { name=$1; shift; rm -rf gpurun_out/prof_$name; env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o r02 -- python bench.py --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline > gpurun_out/r02_bench_prof_$name.json 2> gpurun_out/r02_bench_prof_$name.err
  db=$(find gpurun_out/prof_$name -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/r02_kernel_stats_$name.md "$* rocprofv3 --kernel-trace --stats -- python bench.py --blocks 120 --steps 2 --warmup 1 --contexts 1 --pipeline 1 --serial-probe 0 --no-cpu-baseline" | head -40
  rm -rf gpurun_out/prof_$name
}
prof serial_notab ZKAES_MSM_TABLES=0
prof serial_tab22 ZKAES_MSM_TABLES=1
prof serial_tab20 ZKAES_MSM_TABLE_C=20
