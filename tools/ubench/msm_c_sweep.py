"""window-size sweep of the signed-digit Pippenger at the prover's MSM sizes (run on the GPU box): total and accumulate ms per MSM"""
import os, sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 3 << 22):
    for c in (14, 15, 16, 17, 18, 19, 20):
        os.environ["ZKAES_MSM_C"] = str(c)
        t, a = api.msm_bench_synth(n, 0, 3)
        print("n=%9d c=%2d  total %7.2f ms  accumulate %7.2f ms  other %6.2f ms" % (n, c, t, a, t - a), flush=True)
