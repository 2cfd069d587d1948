import sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
for n in (1 << 20, 1 << 22, 3 << 21):
    for c in (0, 17, 20):
        t, a = api.msm_bench_synth(n, c, 3)
        print("n=%9d %s  total %7.2f ms  accumulate %7.2f ms" % (n, "classic c=17" if c == 0 else "table c=%d" % c, t, a), flush=True)
