#!/bin/bash
# north_star's LDS-bucket sketch against register-resident buckets (profiles/r03_lds_buckets.txt)
mkdir -p gpurun_out/r03_step16
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/lds_buckets.hip -o /tmp/lds_buckets \
  && timeout 60 /tmp/lds_buckets | tee gpurun_out/r03_step16/lds_buckets.txt
