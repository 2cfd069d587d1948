"""CPU tests of the R1CS side: the oracle's gate-level restatement (counts, satisfaction, structural pins) and the product's
circuit compiler (csrc/circuit.cpp) against it -- two independent implementations must emit identical matrices."""
import ctypes as C

import numpy as np
import pytest


def test_aes_circuit_counts_and_satisfaction(zko, vectors):
    key, pt = bytes(vectors["key"]), bytes(vectors["plaintext"])
    cs, ct = zko.synth_aes(pt, key)
    assert list(ct) == vectors["ciphertext"]
    c = cs.counts()
    # SURVEY.md §A.3 [MODEL] formula: constraints = 148,272 B + 36,768; instance = 128 B + 1
    assert (c["constraints"], c["instance"], c["witness"]) == (185_040, 129, 184_784)
    assert c["nnz_a"] + c["nnz_b"] + c["nnz_c"] == 882_002
    assert cs.is_satisfied() == 1
    ins, wit = cs.assignment()
    assert ins[0] == 1
    bits = [(b >> i) & 1 for b in vectors["ciphertext"] for i in range(8)]      # byte_to_field_array: LSB first (helpers/mod.rs:84-93)
    assert list(ins[1:]) == bits
    # message bits then key bits open the witness (src/lib.rs:70-88)
    assert list(wit[:128]) == [(b >> i) & 1 for b in pt for i in range(8)]
    assert list(wit[128:256]) == [(b >> i) & 1 for b in key for i in range(8)]


def test_64_byte_circuit_matches_the_srs_literal_pins(zko, vectors):
    cs, ct = zko.synth_aes(bytes(vectors["plaintext_64"]), bytes(vectors["key"]))
    assert list(ct) == vectors["ciphertext_64"]
    c = cs.counts()
    nc, ni, nnz = vectors["srs_literals"]
    assert c["instance"] == ni == 64 * 8 + 1                 # src/lib.rs:141: 513 instance variables
    assert c["constraints"] <= nc and c["nnz_a"] + c["nnz_b"] + c["nnz_c"] <= nnz
    assert (c["constraints"], c["witness"]) == (629_856, 628_832)
    assert cs.is_satisfied() == 1


def test_corrupted_witness_is_rejected(zko, vectors):
    """SURVEY.md section 5 fault injection: flip ONE witness bit of a satisfied system -> is_satisfied() names a violated constraint.
    Bits from every region of the variable order: a message bit, a key bit, key-schedule / round internals, the last witness."""
    key, pt = bytes(vectors["key"]), bytes(vectors["plaintext"])
    cs, _ = zko.synth_aes(pt, key)
    assert cs.is_satisfied() == 1
    _, wit = cs.assignment()
    n = len(wit)
    for idx in (0, 5, 128 + 77, 256 + 3, 5_000, 36_768 + 99, n // 2, n - 129, n - 1):
        cs.set_witness(idx, 1 - wit[idx])
        r = cs.is_satisfied()
        assert r < 0, "flipping witness %d went unnoticed" % idx          # -(row + 1) of the first violated constraint
        cs.set_witness(idx, wit[idx])
    assert cs.is_satisfied() == 1                                         # restored
    with pytest.raises(IndexError):
        cs.set_witness(n, 0)


def test_empty_and_ragged_messages(zko, api):
    cs, ct = zko.synth_aes(b"", bytes(16))                    # only the key schedule
    c = cs.counts()
    assert ct == b"" and c["instance"] == 1 and c["constraints"] == 36_768 and cs.is_satisfied() == 1
    with pytest.raises(ValueError, match="16 bytes"):
        zko.synth_aes(bytes(15), bytes(16))
    with pytest.raises(api.ZkAesError, match="Input must be 16 bytes length when adding round key"):
        api.circuit_info(api.CIRCUIT_AES, 17)


def test_ops_gates_kat(zko, vectors):
    # src/ops.rs:39-73: x, y = first two u32 of ChaCha20Rng::from_seed(seed) (gen_range over the full u32 range = next_u32)
    w = (C.c_uint32 * 2)()
    zko.lib().zko_api_chacha_words(bytes(vectors["ops_chacha20_seed"]), 20, C.c_size_t(2), w)
    x, y = int(w[0]), int(w[1])
    cs, z = zko.synth_ops("xor", x, y)
    assert z == x ^ y and cs.is_satisfied() == 1
    assert cs.counts()["constraints"] == 96                  # 64 booleanity + 32 xor
    if x + y < 2**32:
        cs, z = zko.synth_ops("add", x, y)
        assert z == x + y and cs.is_satisfied() == 1
    cs, z = zko.synth_ops("add", 0xFFFFFFFF, 1)
    assert z == 0 and cs.is_satisfied() == 1                 # result truncated to 32 bits, carry bit allocated
    assert cs.counts()["constraints"] == 64 + 33 + 1


@pytest.mark.parametrize("length", [0, 16, 32, 64, 96])
def test_product_compiler_equals_oracle_matrices(zko, api, length):
    info = api.circuit_info(api.CIRCUIT_AES, length)
    cs, _ = zko.synth_aes(bytes(length), bytes(16))
    raw = cs.counts()
    assert (info["raw_constraints"], info["raw_instance"], info["raw_witness"]) == (raw["constraints"], raw["instance"], raw["witness"])
    cs.pad_for_marlin()
    c = cs.counts()
    assert (info["constraints"], info["instance"], info["witness"]) == (c["constraints"], c["instance"], c["witness"])
    assert info["constraints"] == info["instance"] + info["witness"]          # square after padding
    for which in range(3):
        a = api.circuit_matrix(api.CIRCUIT_AES, length, which)
        b = cs.matrix(which)
        assert np.array_equal(a[0].astype(np.uint64), b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_product_ops_circuits_equal_oracle(zko, api):
    for kind, name in ((api.CIRCUIT_OPS_XOR, "xor"), (api.CIRCUIT_OPS_ADD, "add")):
        cs, _ = zko.synth_ops(name, 5, 7, field=377)
        cs.pad_for_marlin()
        for which in range(3):
            a = api.circuit_matrix(kind, 0, which)
            b = cs.matrix(which)
            assert np.array_equal(a[0].astype(np.uint64), b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
