import sys; sys.path.insert(0, '.')
from aes_zero_knowledge_proof_circuit_amd import api
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
data = bytes(32 << lg)
for _ in range(3):
    api.ntt(377, data)
