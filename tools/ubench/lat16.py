import os, sys, time
sys.path.insert(0, '.')
os.environ["ZKAES_CONTEXTS"] = "1"
from aes_zero_knowledge_proof_circuit_amd import api, sharding
pk, vk = api.synthesize_keys(16)
key, msg = sharding.rank_message(0, 1)
for _ in range(6):
    api.encrypt(msg, key, pk)
