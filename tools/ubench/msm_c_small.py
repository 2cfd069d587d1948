import os, sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 17, 1 << 18, 1 << 19, 3 << 18, 1 << 20, 1 << 21):
    for c in (11, 12, 13, 15, 16, 17):
        os.environ["ZKAES_MSM_C"] = str(c)
        t, a = api.msm_bench_synth(n, 0, 5)
        print("n=%9d c=%2d  total %7.3f ms  accumulate %7.3f ms  other %6.3f ms" % (n, c, t, a, t - a), flush=True)
