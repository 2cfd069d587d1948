"""total / accumulate ms of the signed-digit Pippenger at the prover's MSM sizes (run on the GPU box)"""
import sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 3 << 20, 1 << 22, 3 << 22):
    t, a = api.msm_bench_synth(n, 0, 5)
    print("n=%9d  total %7.2f ms  accumulate %7.2f ms  -> %.2f G (point, window) pairs/s in k_accumulate" % (n, t, a, n * 15 / a / 1e6), flush=True)
