import sys; sys.path.insert(0,'.')
from aes_zero_knowledge_proof_circuit_amd import api
for lg in (18, 20, 22):
    n = 1 << lg
    for c in (0, 16, 18, 20):
        if c and c > lg + 1: continue
        t, a = api.msm_bench_synth(n, c, 3)
        print("n=2^%d c=%2d  total %.2f ms  accumulate %.2f ms  -> %.1f Mpts/s" % (lg, c, t, a, n / t / 1e3), flush=True)
