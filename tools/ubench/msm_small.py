"""Isolated table-path MSMs (13 windows, one set of 2^19 buckets) from 2^17 to 2^23 points: whole pipeline and k_accumulate alone (run on the GPU box).
How much of a SMALL MSM's accumulation is fixed cost?  Compare with tools/ubench/gather_power.bin sweep (the bare loop at 3 ... 208 additions per lane)."""
import sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
for lg in (17, 18, 19, 20, 21, 22, 23):
    n = 1 << lg
    t, a = api.msm_bench_synth(n, 20, 6)
    print("n=2^%d  total %7.3f ms  accumulate %7.3f ms  -> %5.2f G pairs/s in k_accumulate, %5.1f additions per bucket" % (lg, t, a, n * 13 / a / 1e6, n * 13 / 524288.0), flush=True)
