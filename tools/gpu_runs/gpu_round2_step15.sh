# full-bench A/B of k_accumulate's register footprint: 210 VGPRs (prefetch) vs 170 (no prefetch, 2 waves) vs 162 (3 waves): does leaving register space for the other contexts' short kernels pay?
mkdir -p gpurun_out
out=gpurun_out/r02_occupancy_bench.txt; : > $out
for cfg in "2 1" "2 0" "3 0" "2 1"; do
  set -- $cfg
  ZK_EXTRA_DEFINES="-DZK_ACC_WAVES=$1 -DZK_ACC_PREFETCH=$2" python -m aes_zero_knowledge_proof_circuit_amd.build --force > /dev/null 2>&1
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_occ_$1_$2.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r02_bench_occ_$1_$2.json').read().strip().splitlines()[-1]);print('waves=$1 prefetch=$2', d['value'], d['proofs_verified'], 'acc avg', d['roofline']['avg_launch_ms'], 'overlap', d['roofline']['launch_overlap'])" | tee -a $out
done
python -m aes_zero_knowledge_proof_circuit_amd.build --force > /dev/null 2>&1
