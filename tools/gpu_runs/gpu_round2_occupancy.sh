# A/B of k_accumulate's occupancy / prefetch / radix knobs (csrc/kernels_msm.hip ZK_ACC_WAVES, ZK_ACC_PREFETCH; csrc/ec.cuh ZK_MSM_RADIX): rebuilds the library on the box per variant
mkdir -p gpurun_out
out=gpurun_out/r02_occupancy.txt; : > $out
for cfg in "28 2 1" "28 3 0" "28 2 0" "30 3 0" "30 2 0" "28 4 0"; do
  set -- $cfg
  ZK_MSM_RADIX=$1 ZK_EXTRA_DEFINES="-DZK_ACC_WAVES=$2 -DZK_ACC_PREFETCH=$3" python -m aes_zero_knowledge_proof_circuit_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do
    echo "radix=$1 waves=$2 prefetch=$3 classic $(python tools/ubench/msm_one.py 22 0) table20 $(python tools/ubench/msm_one.py 22 20)" | tee -a $out
  done
done
python -m aes_zero_knowledge_proof_circuit_amd.build --force > /dev/null 2>&1
