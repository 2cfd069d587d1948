"""MI355X-native drop-in for the proving hot path of lambdaclass/AES_zero_knowledge_proof_circuit.

Public surface (mirrors /root/reference/src/lib.rs): synthesize_keys, encrypt, verify_encryption.
Everything heavy lives in libzkaes.so (HIP kernels + C++ host, C ABI in include/zkaes.h); this package is the
ctypes binding a Python harness uses.  There is no CPU fallback: proving raises ZkAesError without a GPU.
"""
from .api import (ZkAesError, ProvingKey, VerifyingKey, synthesize_keys, encrypt, verify_encryption, lib, lib_path,  # noqa: F401
                  CIRCUIT_AES, CIRCUIT_OPS_XOR, CIRCUIT_OPS_ADD)
