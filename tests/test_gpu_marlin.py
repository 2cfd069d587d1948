"""GPU parity tests of the proving path through the C ABI: witness bits, intermediate polynomials and PROOF BYTES must equal the CPU
oracle's; every proof must be accepted by the verifier and wrong ciphertexts rejected (tests/integration_tests.rs:313-372)."""
import os

import numpy as np

import pytest

from conftest import mt_bytes

pytestmark = pytest.mark.gpu
SMALL_SRS = (200, 200, 600)


@pytest.fixture(scope="module")
def aes16(api):
    return api.synthesize_keys(16)


@pytest.mark.parametrize("name,kind", [("xor", 1), ("add", 2)])
def test_ops_proofs_byte_identical_to_oracle(zko, api, name, kind):
    pk, vk = api.synthesize_keys(0, circuit=kind, srs=SMALL_SRS)
    cs, _ = zko.synth_ops(name, 0, 0, field=377)
    ix = zko.Index(cs, srs=SMALL_SRS)
    for (x, y, seed) in [(0xDEADBEEF, 0x12345678, None), (0xFFFFFFFF, 1, bytes(range(32))), (0, 0, None)]:
        proof = pk.prove_ops(x, y, seed)
        cs, _ = zko.synth_ops(name, x, y, field=377)
        ref = ix.prove(cs, seed)
        for poly in zko.POLY_NAMES:
            assert pk.debug_fetch(poly) == ref.poly(poly), poly
        assert proof == ref.to_bytes()
        assert vk.verify(proof, b"") is True
    # index polynomials and commitments agree too
    for i, nm in enumerate(["row", "col", "a_val", "b_val", "c_val", "row_col"]):
        assert pk.debug_fetch(nm + "_evals") == ix.poly(i, 0), nm
        assert pk.debug_fetch(nm) == ix.poly(i, 1), nm


def test_aes_witness_bit_exact(zko, api, aes16, vectors):
    pk, _ = aes16
    for msg, key in [(bytes(vectors["plaintext"]), bytes(vectors["key"])), (bytes(16), bytes(16)), (mt_bytes(16, 1), mt_bytes(16, 2)), (b"\xff" * 16, b"\xff" * 16)]:
        cs, ct = zko.synth_aes(msg, key)
        cs.pad_for_marlin()
        ins, wit = cs.assignment()
        z = pk.witness(msg, key)
        assert z == ins + wit
    info = pk.info()
    assert (info["raw_constraints"], info["raw_instance"], info["raw_witness"]) == (185_040, 129, 184_784)
    assert (info["h"], info["k"], info["joint_nnz"]) == (1 << 18, 1 << 20, 728_810)


def test_encrypt_16_bytes_and_verify(api, aes16, vectors):
    # tests/integration_tests.rs:313-337
    pk, vk = aes16
    proof = api.encrypt(bytes(vectors["plaintext"]), bytes(vectors["key"]), pk.clone())
    assert api.verify_encryption(vk.clone(), proof, bytes(vectors["ciphertext"])) is True
    assert api.verify_encryption(vk, proof, bytes(vectors["wrong_ciphertext_16"])) is False
    assert api.proof_roundtrip(proof) == proof
    # deterministic under the reference's fixed prover seed; a fresh seed gives a different, still valid proof
    assert api.encrypt(bytes(vectors["plaintext"]), bytes(vectors["key"]), pk) == proof
    p2 = api.encrypt(bytes(vectors["plaintext"]), bytes(vectors["key"]), pk, zk_seed=bytes(range(32)))
    assert p2 != proof and api.verify_encryption(vk, p2, bytes(vectors["ciphertext"]))


def test_encrypt_rejects_bad_lengths(api, aes16):
    pk, _ = aes16
    with pytest.raises(api.ZkAesError, match="Input must be 16 bytes length when adding round key"):
        api.encrypt(bytes(15), bytes(16), pk)
    with pytest.raises(api.ZkAesError, match="InstanceDoesNotMatchIndex"):
        api.encrypt(bytes(32), bytes(16), pk)


def test_random_messages_verify_against_oracle_ciphertext(zko, api, aes16):
    pk, vk = aes16
    for i in range(3):
        msg, key = mt_bytes(16, 100 + i), mt_bytes(16, 200 + i)
        ct = zko.aes_encrypt(msg, key)
        proof = api.encrypt(msg, key, pk)
        assert api.verify_encryption(vk, proof, ct)
        bad = bytearray(ct)
        bad[i] ^= 0x80
        assert not api.verify_encryption(vk, proof, bytes(bad))


def test_encrypt_64_bytes_and_verify(api, vectors):
    # tests/integration_tests.rs:340-372 (same block four times)
    pk, vk = api.synthesize_keys(64)
    info = pk.info()
    assert (info["raw_constraints"], info["raw_instance"], info["h"], info["k"]) == (629_856, 513, 1 << 20, 1 << 22)
    proof = api.encrypt(bytes(vectors["plaintext_64"]), bytes(vectors["key"]), pk)
    assert api.verify_encryption(vk, proof, bytes(vectors["ciphertext_64"])) is True
    assert api.verify_encryption(vk, proof, bytes(vectors["wrong_ciphertext_64"])) is False


def test_monolithic_proof_beyond_the_reference_srs_literal(zko, api):
    """A 16-block message as ONE proof (|H| = 2^22, |K| = 2^24): does not fit the reference's SRS literal (src/lib.rs:141), so the key is
    synthesized for the circuit's own counts -- the same code path as tools/monolith.py's 64-block (BASELINE configs[1]) run."""
    blocks = 16
    ci = api.circuit_info(api.CIRCUIT_AES, 16 * blocks)
    assert ci["raw_constraints"] == 148_272 * blocks + 36_768                  # SURVEY.md A.3 formula
    with pytest.raises(api.ZkAesError):
        api.synthesize_keys(16 * blocks)                                         # the default literal is too small for 16 blocks
    pk, vk = api.synthesize_keys(16 * blocks, srs=(int(ci["constraints"]), int(ci["instance"]), int(ci["nnz_a"] + ci["nnz_b"] + ci["nnz_c"])))
    info = pk.info()
    assert (info["h"], info["k"]) == (1 << 22, 1 << 24)
    rs = np.random.RandomState(16)
    msg, key = rs.bytes(16 * blocks), rs.bytes(16)
    proof = api.encrypt(msg, key, pk)
    ct = zko.aes_encrypt(msg, key)
    assert len(proof) == 855
    assert api.verify_encryption(vk, proof, ct) is True
    bad = bytearray(ct); bad[-1] ^= 1
    assert api.verify_encryption(vk, proof, bytes(bad)) is False


@pytest.mark.slow
def test_aes16_proof_bytes_identical_to_oracle(zko, api, aes16, vectors):
    """Full-size parity: the GPU proof of the FIPS-197 block equals the CPU oracle's proof byte for byte (about a minute of CPU)."""
    pk, _ = aes16
    proof = api.encrypt(bytes(vectors["plaintext"]), bytes(vectors["key"]), pk)
    cs, _ = zko.synth_aes(bytes(16), bytes(16))
    ix = zko.Index(cs)
    cs, _ = zko.synth_aes(bytes(vectors["plaintext"]), bytes(vectors["key"]))
    ref = ix.prove(cs)
    for poly in zko.POLY_NAMES:
        assert pk.debug_fetch(poly) == ref.poly(poly), poly
    assert proof == ref.to_bytes()


@pytest.mark.slow
def test_aes32_proof_bytes_identical_to_oracle(zko, api):
    """The same byte-for-byte parity on a 2-block message (|H| = 2^19, |K| = 2^21: other NTT pass plans and MSM sizes than the 16-byte case),
    with a non-default prover seed so the mask polynomial and the hiding blinders differ as well (about two minutes of CPU)."""
    msg, key, seed = mt_bytes(32, 77), mt_bytes(16, 78), bytes(range(100, 132))
    pk, vk = api.synthesize_keys(32)
    info = pk.info()
    assert (info["h"], info["k"]) == (1 << 19, 1 << 21)
    proof = api.encrypt(msg, key, pk, zk_seed=seed)
    cs, _ = zko.synth_aes(bytes(32), bytes(16))
    ix = zko.Index(cs)
    cs, _ = zko.synth_aes(msg, key)
    ref = ix.prove(cs, seed)
    for poly in zko.POLY_NAMES:
        assert pk.debug_fetch(poly) == ref.poly(poly), poly
    assert proof == ref.to_bytes()
    assert api.verify_encryption(vk, proof, zko.aes_encrypt(msg, key))


def test_batch_of_independent_single_block_proofs(zko, api, aes16):
    """BASELINE config 5 shape (many small proofs on one SRS), reduced to 12 proofs: each (message, key) pair gets its own proof."""
    pk, vk = aes16
    msgs = [mt_bytes(16, 300 + i) for i in range(12)]
    keys = [mt_bytes(16, 400 + i) for i in range(12)]
    proofs = pk.encrypt_batch(msgs, keys)
    assert len(proofs) == 12
    for i, (m, k, p) in enumerate(zip(msgs, keys, proofs)):
        assert api.verify_encryption(vk, p, zko.aes_encrypt(m, k))
        assert not api.verify_encryption(vk, p, zko.aes_encrypt(m, keys[(i + 1) % 12]))
    # the concurrent contexts produce exactly what the single-proof entry point does
    assert proofs[3] == api.encrypt(msgs[3], keys[3], pk)


def test_chunked_message_equals_individual_proofs(api, aes16):
    pk, vk = aes16
    key, msg = mt_bytes(16, 500), mt_bytes(16 * 5, 501)
    proofs = pk.encrypt_chunked(msg, key)
    assert proofs == [api.encrypt(msg[16 * i:16 * i + 16], key, pk) for i in range(5)]


def test_empty_message_and_size_limits(api):
    """edge cases: a 0-byte message proves only the key schedule (no public inputs); 96 bytes is the largest plaintext that fits the
    reference's universal-SRS literal (src/lib.rs:141), 112 bytes must be refused like arkworks' IndexTooLarge."""
    pk, vk = api.synthesize_keys(0)
    info = pk.info()
    assert (info["raw_constraints"], info["raw_instance"]) == (36_768, 1)
    proof = api.encrypt(b"", bytes(range(16)), pk)
    assert api.verify_encryption(vk, proof, b"") is True
    assert api.verify_encryption(vk, proof, bytes(16)) is False        # instance length does not match the index
    with pytest.raises(api.ZkAesError, match="IndexTooLarge"):
        api.synthesize_keys(112)


def test_concurrent_encrypt_calls_on_one_key(zko, api, aes16):
    """callers may share a proving key across threads (the Rust wrapper is Send + Sync): concurrent zkaes_encrypt calls on one handle queue on the
    prover context instead of racing; every proof equals the one a lone call produces (proofs are deterministic)"""
    import threading
    pk, vk = aes16
    msgs = [mt_bytes(16, 900 + i) for i in range(6)]
    key = mt_bytes(16, 899)
    expected = [api.encrypt(m, key, pk) for m in msgs]
    got = [None] * len(msgs)

    def run(i):
        got[i] = api.encrypt(msgs[i], key, pk)
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(msgs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got == expected
    assert all(api.verify_encryption(vk, p, zko.aes_encrypt(m, key)) for m, p in zip(msgs, got))


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("ZKAES_LONG_TESTS"), reason="about ten minutes of CPU for the oracle: set ZKAES_LONG_TESTS=1 (tools/parity_full.py is the same check as a script)")
def test_aes96_proof_bytes_identical_to_oracle_at_the_bench_size(zko, api):
    """the bench configuration itself: 6-block chunk-proof, |H| = 2^20, |K| = 2^22, the reference's SRS literal"""
    msg, key = mt_bytes(96, 5), mt_bytes(16, 6)
    pk, vk = api.synthesize_keys(96)
    proof = api.encrypt(msg, key, pk)
    cs, _ = zko.synth_aes(bytes(96), bytes(16))
    ix = zko.Index(cs)
    cs, _ = zko.synth_aes(msg, key)
    ref = ix.prove(cs)
    for poly in zko.POLY_NAMES:
        assert pk.debug_fetch(poly) == ref.poly(poly), poly
    assert proof == ref.to_bytes()
