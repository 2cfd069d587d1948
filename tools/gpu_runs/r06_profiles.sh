#!/bin/bash
# round 6: the profile set of the final code -- kernel statistics (one context / the driver's command / single-block proofs), PMC of the dominant kernel with the gather
# calibration (stamped with the commit passed in ZKAES_COMMIT), HBM bytes and VALU instructions per kernel for both proof sizes.  ~25 minutes of GPU box time.
#   gpurun --timeout 3000 -- "ZKAES_COMMIT=$(git rev-parse --short HEAD) bash tools/gpu_runs/r06_profiles.sh"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/gpu_runs/prof.sh r06 serial > gpurun_out/r06_prof_serial.log 2>&1
bash tools/gpu_runs/prof.sh r06 driver > gpurun_out/r06_prof_driver.log 2>&1
bash tools/gpu_runs/prof.sh r06 batch1 > gpurun_out/r06_prof_batch1.log 2>&1
timeout 900 python tools/pmc_accumulate.py 22 r06_tables 20 > gpurun_out/r06_pmc_accumulate.log 2>&1
timeout 900 python tools/pmc_bytes_by_kernel.py r06 > gpurun_out/r06_bytes.log 2>&1
timeout 900 python tools/pmc_by_kernel.py r06 > gpurun_out/r06_valu.log 2>&1
timeout 900 python tools/pmc_by_kernel.py r06 batch > gpurun_out/r06_valu_16B.log 2>&1
ls -la gpurun_out | tail -30
