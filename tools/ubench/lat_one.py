import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("ZKAES_CONTEXTS", "1")
from aes_zero_knowledge_proof_circuit_amd import api, sharding
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pk, vk = api.synthesize_keys(nbytes)
key, msg = sharding.rank_message(0, nbytes // 16)
api.encrypt(msg, key, pk)
ts = []
for _ in range(10):
    t = time.perf_counter(); api.encrypt(msg, key, pk); ts.append(time.perf_counter() - t)
ts.sort(); print("median_ms %.2f" % (1e3 * ts[5]), pk.timings())
