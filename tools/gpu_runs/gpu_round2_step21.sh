cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out
timeout 900 python tools/pmc_accumulate.py 22 r02_tables 20 > gpurun_out/r02_pmc_tables.log 2>&1; tail -25 gpurun_out/r02_pmc_tables.log
