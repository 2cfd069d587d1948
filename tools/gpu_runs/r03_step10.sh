# round 3, step 10: k_accumulate<EdwardsLaw> occupancy / prefetch variants (library rebuilt on the box per variant), isolated 2^22 / 2^23-point table MSMs
mkdir -p gpurun_out/r03_step10
O=gpurun_out/r03_step10
probe() {
python - <<'PY'
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 22, 1 << 23):
    t, a = api.msm_bench_synth(n, 20, 4)
    print("  n=2^%d table c=20: total %.3f ms  accumulate %.3f ms" % (n.bit_length() - 1, t, a), flush=True)
PY
}
echo "== default (2 waves / SIMD, prefetch)" | tee $O/ab.txt; probe 2>&1 | tee -a $O/ab.txt
for v in "-DZK_ACC_PREFETCH=0" "-DZK_ACC_WAVES=3" "-DZK_ACC_WAVES=3 -DZK_ACC_PREFETCH=0" "-DZK_ACC_WAVES=1"; do
  ZK_EXTRA_DEFINES="$v" python -m aes_zero_knowledge_proof_circuit_amd.build --force > $O/build.log 2>&1
  echo "== $v" | tee -a $O/ab.txt; probe 2>&1 | tee -a $O/ab.txt
done
