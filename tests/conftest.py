import json
import os
import random
import sys

import pytest
try:
    import torch  # noqa: F401  -- BEFORE libzkaes.so: torch's wheel bundles its own ROCm runtime; whichever HIP runtime a process loads first is the one that sees the GPU,
    #                and the C-ABI library binds to an already-loaded one by SONAME while torch does not ("one process, two HIP runtimes")
except ImportError:      # only the ORDER matters, and only where torch exists: the CPU-only suites (test_ff28_host, test_marlin_cpu, ...) run without it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running (full-size CPU oracle proofs)")


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def zko():
    """The CPU oracle (test infrastructure)."""
    from oracle import zko as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def api():
    """ctypes binding of the product library; builds it if missing (hipcc cross-compiles without a GPU)."""
    from aes_zero_knowledge_proof_circuit_amd import api as m
    if not os.path.exists(m.lib_path()):
        from aes_zero_knowledge_proof_circuit_amd import build
        build.build()
    m.lib()
    return m


@pytest.fixture()
def rng():
    return random.Random(0x5EED)


def mt_bytes(n, seed=0x5EED):
    """synthetic key/message bytes (mt19937 stream, seed 0x5EED)"""
    import numpy as np
    return np.random.RandomState(seed & 0xffffffff).randint(0, 256, size=n, dtype=np.uint8).tobytes()
