"""One-off full-size parity check at the BENCH configuration (run on the GPU box; the CPU oracle needs ~10 minutes here):
the GPU proof of a 6-block (96-byte) message -- |H| = 2^20, |K| = 2^22, the reference's SRS literal -- against the CPU oracle's, byte for byte,
together with every prover polynomial.

    python tools/parity_full.py [blocks=6]  ->  one JSON line
"""
import hashlib, json, os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("ZKAES_CONTEXTS", "1")
from aes_zero_knowledge_proof_circuit_amd import api, sharding
from oracle import zko

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
key, msg = sharding.rank_message(0, B)
t0 = time.perf_counter()
pk, vk = api.synthesize_keys(16 * B)
proof = api.encrypt(msg, key, pk)
t1 = time.perf_counter()
cs, _ = zko.synth_aes(bytes(16 * B), bytes(16))
ix = zko.Index(cs)
t2 = time.perf_counter()
cs, _ = zko.synth_aes(msg, key)
ref = ix.prove(cs)
t3 = time.perf_counter()
polys = {name: bool(pk.debug_fetch(name) == ref.poly(name)) for name in zko.POLY_NAMES}
same = proof == ref.to_bytes()
info = pk.info()
print(json.dumps({"blocks": B, "h": info["h"], "k": info["k"], "proof_bytes_identical": bool(same), "polynomials_identical": polys,
                  "proof_sha256": hashlib.sha256(proof).hexdigest(), "verified": bool(api.verify_encryption(vk, proof, zko.aes_encrypt(msg, key))),
                  "gpu_setup_plus_proof_s": round(t1 - t0, 2), "oracle_index_s": round(t2 - t1, 1), "oracle_prove_s": round(t3 - t2, 1)}))
sys.exit(0 if same and all(polys.values()) else 1)
