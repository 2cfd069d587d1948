"""One MONOLITHIC proof of a B-block ECB message (BASELINE configs[1] shape "if memory allows"): own SRS literals, one prover context.

    ZKAES_CONTEXTS=1 python tools/monolith.py 64
"""
import json, os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("ZKAES_CONTEXTS", "1")
from aes_zero_knowledge_proof_circuit_amd import api, sharding
from oracle import zko        # checker only: expected ciphertext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NPAR = int(os.environ["ZKAES_CONTEXTS"])
info = api.circuit_info(api.CIRCUIT_AES, 16 * B)
srs = (int(info["constraints"]), int(info["instance"]), int(info["nnz_a"] + info["nnz_b"] + info["nnz_c"]))
key, msg = sharding.rank_message(0, B)
t0 = time.perf_counter()
pk, vk = api.synthesize_keys(16 * B, srs=srs)
t1 = time.perf_counter()
times = []
for _ in range(2):
    t = time.perf_counter()
    proof = api.encrypt(msg, key, pk)
    times.append(time.perf_counter() - t)
ct = zko.aes_encrypt(msg, key)
t2 = time.perf_counter()
ok = api.verify_encryption(vk, proof, ct)
bad = bytearray(ct); bad[5] ^= 1
rej = not api.verify_encryption(vk, proof, bytes(bad))
t3 = time.perf_counter()
batch = None
if NPAR > 1:      # NPAR independent B-block messages proven concurrently (one context each)
    msgs = [sharding.rank_message(i, B)[1] for i in range(NPAR)]
    pk.encrypt_batch(msgs, [key] * NPAR)
    t = time.perf_counter()
    proofs = pk.encrypt_batch(msgs, [key] * NPAR)
    dt = time.perf_counter() - t
    good = all(api.verify_encryption(vk, p, zko.aes_encrypt(m, key)) for m, p in zip(msgs, proofs))
    batch = {"proofs_in_flight": NPAR, "seconds": round(dt, 3), "blocks_per_s": round(NPAR * B / dt, 2), "all_verified": bool(good)}
print(json.dumps({"blocks": B, "batch": batch, "h": int(info["h"]), "k": int(info["k"]), "constraints": int(info["raw_constraints"]), "setup_s": round(t1 - t0, 2),
                  "prove_s": [round(x, 3) for x in times], "blocks_per_s": round(B / min(times), 2), "proof_bytes": len(proof), "verified": bool(ok),
                  "wrong_ciphertext_rejected": bool(rej), "verify_s": round((t3 - t2) / 2, 3), "phase_ms": pk.timings()}))
