import sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
c = int(sys.argv[2]) if len(sys.argv) > 2 else 0
print(api.msm_bench_synth(n, c, 2))
