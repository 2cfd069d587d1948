#!/usr/bin/env python3
"""Run ON A GPU BOX: write the ark-serialize image of the arkworks IndexProverKey of a GPU-synthesized key (zkaes_pk_serialize_ark_to_file) for
integration/check_on_cargo_box.sh's `encrypt_with_gpu_key` step.

    python tools/make_pk_image.py OUT.bin [message_bytes=16]      # 1.25 GB for 16 bytes; carry it to the cargo box and export ZKAES_PK_IMAGE=OUT.bin

The image is the UNCOMPRESSED one (96-byte points): the harness reads it with IndexProverKey::deserialize_unchecked, which in ark-serialize 0.3 means the uncompressed
layout without any per-point check.  OUT.bin.vk is the compressed IndexVerifierKey (VerifyingKey::deserialize).
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aes_zero_knowledge_proof_circuit_amd import api  # noqa: E402

out = sys.argv[1]
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
pk, vk = api.synthesize_keys(nbytes)
size = pk.serialize_ark_to_file(out, uncompressed=True)
h = hashlib.sha256()
with open(out, "rb") as f:
    for blk in iter(lambda: f.read(1 << 24), b""):
        h.update(blk)
open(out + ".vk", "wb").write(vk.to_ark_bytes())
print("wrote %s: %d bytes, sha256 %s (+ %s.vk, the matching IndexVerifierKey)" % (out, size, h.hexdigest(), out))
