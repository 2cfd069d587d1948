#!/usr/bin/env python3
"""Run ON THE GPU BOX: produce a verifying key + proof for the FIPS-197 block (tests/integration_tests.rs:313-337 inputs) so that the
CPU-only suite can exercise the host verifier on GPU-made data.  Output goes to gpurun_out/ (copy into tests/golden/).

    gpurun -- python tests/golden/make_gpu_fixtures.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aes_zero_knowledge_proof_circuit_amd import api  # noqa: E402

vec = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
pk, vk = api.synthesize_keys(16)
proof = api.encrypt(bytes(vec["plaintext"]), bytes(vec["key"]), pk)
assert api.verify_encryption(vk, proof, bytes(vec["ciphertext"]))
assert not api.verify_encryption(vk, proof, bytes(vec["wrong_ciphertext_16"]))
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "gpu_aes16_proof.bin"), "wb").write(proof)
open(os.path.join(out, "gpu_aes16_vk.bin"), "wb").write(vk.to_bytes())
open(os.path.join(out, "gpu_aes16_vk_ark.bin"), "wb").write(vk.to_ark_bytes())      # ark-serialize IndexVerifierKey layout (for a Rust verifier)
print("wrote", len(proof), "proof bytes and", len(vk.to_bytes()), "vk bytes")
