#!/bin/bash
# column-wise product with the carry as the multiply-accumulate chain's addend (ff28.cuh ZK_FF28_COLUMNS) against the row-wise form, on the register-bucket probe
mkdir -p gpurun_out/r03_step17
for v in 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DZK_FF28_COLUMNS=$v -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/lds_buckets.hip -o /tmp/lds_buckets_$v \
    && { echo "== ZK_FF28_COLUMNS=$v"; timeout 60 /tmp/lds_buckets_$v; timeout 60 /tmp/lds_buckets_$v | head -2; } | tee -a gpurun_out/r03_step17/columns.txt
done
