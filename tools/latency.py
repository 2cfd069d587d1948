"""Single-proof latency of encrypt() at the reference's own criterion sizes (benches/benchmark_encrypt.rs: 16 / 32 / 64-byte messages), keys resident.

    python tools/latency.py   ->  one JSON line (run on the GPU box)
"""
import json, os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("ZKAES_CONTEXTS", "1")
from aes_zero_knowledge_proof_circuit_amd import api, sharding
from oracle import zko        # checker only: expected ciphertext

out = {}
for nbytes in (16, 32, 64, 96):
    pk, vk = api.synthesize_keys(nbytes)
    key, msg = sharding.rank_message(0, nbytes // 16)
    api.encrypt(msg, key, pk)
    ts = []
    for _ in range(10):                      # group.sample_size(10), benches/benchmark_encrypt.rs:43
        t = time.perf_counter(); proof = api.encrypt(msg, key, pk); ts.append(time.perf_counter() - t)
    assert api.verify_encryption(vk, proof, zko.aes_encrypt(msg, key))
    ts.sort()
    info = pk.info()
    out["%d_message_encryption" % nbytes] = {"median_ms": round(1e3 * ts[len(ts) // 2], 2), "min_ms": round(1e3 * ts[0], 2), "sorted_ms": [round(1e3 * t, 1) for t in ts], "h": info["h"], "k": info["k"], "phase_ms": {k: round(v, 2) for k, v in pk.timings().items()}}
print(json.dumps(out))
