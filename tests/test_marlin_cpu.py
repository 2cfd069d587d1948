"""CPU tests at the Marlin boundary: the oracle prover (oracle/zko_marlin.c) against the product's host verifier
(csrc/marlin.cpp + csrc/pairing.hpp) -- independent code bases that only agree if both restate the same protocol."""
import os

import pytest

SMALL_SRS = (200, 200, 600)


def vk_for(api, zko, ix, n_public):
    info = ix.info()
    beta, gamma = ix.srs_scalars()
    return api.VerifyingKey.from_trapdoor(
        [info["num_variables"], info["num_constraints"], info["num_non_zero"], info["num_instance"], n_public, info["max_degree"], info["supported_degree"]],
        ix.comms(), zko.fr_pack([beta]), zko.fr_pack([gamma]))


@pytest.fixture(scope="module")
def xor_setup(zko, api):
    cs, _ = zko.synth_ops("xor", 0, 0, field=377)
    ix = zko.Index(cs, srs=SMALL_SRS)
    return ix, vk_for(api, zko, ix, 0)


def test_oracle_proof_accepted_by_product_verifier(zko, api, xor_setup):
    ix, vk = xor_setup
    info = ix.info()
    assert (info["h"], info["k"]) == (128, 256)
    cs, z = zko.synth_ops("xor", 0xDEADBEEF, 0x12345678, field=377)
    proof = ix.prove(cs).to_bytes()
    assert api.proof_roundtrip(proof) == proof                  # ark-serialize layout parses and re-serializes identically
    assert vk.verify(proof, b"") is True


def test_proofs_are_deterministic_and_seedable(zko, xor_setup):
    ix, _ = xor_setup
    p = []
    for seed in (None, None, bytes(range(32))):
        cs, _ = zko.synth_ops("xor", 7, 9, field=377)
        p.append(ix.prove(cs, seed).to_bytes())
    assert p[0] == p[1]                                        # generate_rand() is a fixed-seed StdRng
    assert p[0] != p[2]


def test_tampered_proofs_are_rejected(zko, api, xor_setup):
    ix, vk = xor_setup
    cs, _ = zko.synth_ops("xor", 1, 2, field=377)
    proof = ix.prove(cs).to_bytes()
    assert vk.verify(proof, b"")
    ev_off = len(proof) - (8 + 3 + 8 + 48 + 1 + 32 + 48 + 1 + 1) - 4 * 32     # first evaluation
    for off in (ev_off, ev_off + 32, ev_off + 64, ev_off + 96, len(proof) - 2 - 48 - 32):
        bad = bytearray(proof)
        bad[off] ^= 1
        assert vk.verify(bytes(bad), b"") is False
    bad = bytearray(proof)
    bad[20] ^= 1                                               # x coordinate of the first commitment: off-curve or a different point
    try:
        assert vk.verify(bytes(bad), b"") is False
    except api.ZkAesError:
        pass
    with pytest.raises(api.ZkAesError):
        vk.verify(proof[:-1], b"")
    with pytest.raises(api.ZkAesError):
        vk.verify(proof + b"\0", b"")


def test_add_circuit_with_wide_coefficients(zko, api):
    cs, _ = zko.synth_ops("add", 0, 0, field=377)
    ix = zko.Index(cs, srs=SMALL_SRS)
    vk = vk_for(api, zko, ix, 0)
    cs, z = zko.synth_ops("add", 0xFFFFFFFF, 0xFFFFFFFF, field=377)
    assert z == 0xFFFFFFFE
    assert vk.verify(ix.prove(cs).to_bytes(), b"")


def test_vk_transport_roundtrip(api, xor_setup):
    _, vk = xor_setup
    b = vk.to_bytes()
    assert api.VerifyingKey.from_bytes(b).to_bytes() == b
    with pytest.raises(api.ZkAesError):
        api.VerifyingKey.from_bytes(b[:-3])


GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "gpu_aes16_proof.bin")), reason="GPU-made fixture not generated yet")
def test_gpu_made_fixture_verifies_on_the_host(api, vectors):
    """A proof + verifying key produced on an MI355X by tests/golden/make_gpu_fixtures.py (committed as data)."""
    vk = api.VerifyingKey.from_bytes(open(os.path.join(GOLD, "gpu_aes16_vk.bin"), "rb").read())
    proof = open(os.path.join(GOLD, "gpu_aes16_proof.bin"), "rb").read()
    assert api.verify_encryption(vk, proof, bytes(vectors["ciphertext"])) is True
    assert api.verify_encryption(vk, proof, bytes(vectors["wrong_ciphertext_16"])) is False      # tests/integration_tests.rs:332-336
