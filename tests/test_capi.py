"""CPU tests of the C-ABI boundary: every symbol include/zkaes.h declares is exported, and the product refuses to work without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkaes.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkaes_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(api):
    L = api.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n


def test_no_cpu_fallback(api):
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.ZkAesError, match="no HIP device"):
        api.synthesize_keys(16)
    with pytest.raises(api.ZkAesError, match="no HIP device"):
        api.ntt(377, bytes(64))
    with pytest.raises(api.ZkAesError, match="no HIP device"):
        api.msm(377, bytes(96), bytes(32))
    with pytest.raises(api.ZkAesError, match="no HIP device"):
        api.int_rate_bench(0.1)                    # the calibration probe measures a GPU or fails: it never reports a host number


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "aes_zero_knowledge_proof_circuit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".cuh", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in text and "import oracle" not in text and "from oracle" not in text and "zko" not in text, os.path.join(dirpath, f)


def test_process_wide_settings_validate_without_a_device(api):
    """zkaes_set_default_contexts / zkaes_srs_hold only set process defaults (no device needed): the bound of 64 contexts per key is enforced at the boundary with a message,
    0 restores the built-in default, and holding / releasing the SRS cache with nothing in it is a no-op"""
    with pytest.raises(api.ZkAesError, match="at most 64"):
        api.set_default_contexts(65)
    api.set_default_contexts(3)
    api.set_default_contexts(0)
    api.srs_hold(True)
    api.srs_hold(False)
