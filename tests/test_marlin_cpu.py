"""CPU tests at the Marlin boundary: the oracle prover (oracle/zko_marlin.c) against the product's host verifier
(csrc/marlin.cpp + csrc/pairing.hpp) -- independent code bases that only agree if both restate the same protocol."""
import os

import pytest

SMALL_SRS = (200, 200, 600)


def vk_for(api, zko, ix, n_public):
    info = ix.info()
    beta, _, _ = ix.srs_trapdoor()
    return api.VerifyingKey.from_trapdoor(
        [info["num_variables"], info["num_constraints"], info["num_non_zero"], info["num_instance"], n_public, info["max_degree"], info["supported_degree"]],
        ix.comms(), zko.fr_pack([beta]))


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
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "gpu_aes16_proof.bin")), reason="GPU-made fixture not generated yet")
def test_gpu_made_fixture_verifies_on_the_host(api, vectors):
    """A proof + verifying key produced on an MI355X by tests/golden/make_gpu_fixtures.py (committed as data)."""
    vk = api.VerifyingKey.from_bytes(open(os.path.join(GOLD, "gpu_aes16_vk.bin"), "rb").read())
    proof = open(os.path.join(GOLD, "gpu_aes16_proof.bin"), "rb").read()
    assert api.verify_encryption(vk, proof, bytes(vectors["ciphertext"])) is True
    assert api.verify_encryption(vk, proof, bytes(vectors["wrong_ciphertext_16"])) is False      # tests/integration_tests.rs:332-336


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "gpu_aes16_vk_ark.bin")), reason="GPU-made fixture not generated yet")
def test_gpu_made_ark_layout_key_verifies_on_the_host(api, vectors):
    """the same key in the ark-serialize IndexVerifierKey layout (what integration/rust_verify_harness feeds to the reference verifier)"""
    raw = open(os.path.join(GOLD, "gpu_aes16_vk_ark.bin"), "rb").read()
    vk = api.VerifyingKey.from_ark_bytes(raw)
    assert vk.to_ark_bytes() == raw and len(raw) == 759
    proof = open(os.path.join(GOLD, "gpu_aes16_proof.bin"), "rb").read()
    assert api.verify_encryption(vk, proof, bytes(vectors["ciphertext"])) is True
    assert api.verify_encryption(vk, proof, bytes(vectors["wrong_ciphertext_16"])) is False


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "gpu_aes16_proof.bin")), reason="GPU-made fixture not generated yet")
def test_deserialize_proof_rejects_what_ark_serialize_rejects(api, zko, vectors):
    """ark-ec 0.3 GroupAffine::deserialize (behind the reference's deserialize_proof, src/lib.rs:52) rejects points outside the prime-order
    subgroup and SWFlags::from_u8 rejects the flag byte with infinity and sign both set; a lax reader makes proofs malleable (a cofactor-order
    component added to an opening witness survives the pairing check)."""
    proof = open(os.path.join(GOLD, "gpu_aes16_proof.bin"), "rb").read()
    vk = api.VerifyingKey.from_bytes(open(os.path.join(GOLD, "gpu_aes16_vk.bin"), "rb").read())
    assert api.proof_roundtrip(proof) == proof
    off = 8 + 8                                                   # Vec<Vec<Commitment>> length, first round length -> first commitment (48 B compressed G1)
    q = zko.Q377

    def with_point(xbytes):
        return proof[:off] + bytes(xbytes) + proof[off + 48:]
    # 1. on the curve, NOT in the subgroup: the smallest x with x^3 + 1 a square whose point has a cofactor component
    x = 1
    while True:
        x += 1
        rhs = (x * x * x + 1) % q
        if pow(rhs, (q - 1) // 2, q) != 1:
            continue
        bad = with_point(x.to_bytes(48, "little"))
        with pytest.raises(api.ZkAesError, match="subgroup"):
            api.proof_roundtrip(bad)
        with pytest.raises(api.ZkAesError, match="subgroup"):
            api.verify_encryption(vk, bad, bytes(vectors["ciphertext"]))
        break
    # 2. flag byte with infinity AND sign set
    b = bytearray(proof[off:off + 48]); b[47] |= 0xC0
    with pytest.raises(api.ZkAesError, match="flags"):
        api.proof_roundtrip(with_point(b))
    # 3. infinity flag with non-zero x bits: one encoding per point
    b = bytearray(proof[off:off + 48]); b[47] = (b[47] & 0x3f) | 0x40
    with pytest.raises(api.ZkAesError, match="infinity"):
        api.proof_roundtrip(with_point(b))
    # the canonical infinity encoding still parses (and the proof then simply fails to verify)
    inf = bytearray(48); inf[47] = 0x40
    assert api.proof_roundtrip(with_point(inf)) == with_point(inf)
    assert api.verify_encryption(vk, with_point(inf), bytes(vectors["ciphertext"])) is False
    # 4. an opening witness shifted by a cofactor-order point used to verify; it no longer parses.  (x = 0, y = 1) has order 3 on y^2 = x^3 + 1
    w_off = len(proof) - 1 - 1 - 48                               # ... w_gamma (48 B), Option tag, pc_proof.evals tag
    zero_x = bytearray(48)
    with pytest.raises(api.ZkAesError, match="subgroup"):
        api.proof_roundtrip(proof[:w_off] + bytes(zero_x) + proof[w_off + 48:])


def test_concurrent_first_use_of_point_decompression(api):
    """the square-root constants behind point decompression are initialised on first use; the first callers may race (verifier-only process,
    INTEGRATION.md: VkHandle is Sync).  Eight threads decompress at once in a FRESH process; all must agree with a later single-threaded call."""
    import subprocess
    import sys
    code = r"""
import sys, threading
sys.path.insert(0, %r)
from aes_zero_knowledge_proof_circuit_amd import api
import os
gold = os.path.join(%r, "tests", "golden")
raw = open(os.path.join(gold, "gpu_aes16_vk_ark.bin"), "rb").read()
out = [None] * 8
def run(i):
    out[i] = api.VerifyingKey.from_ark_bytes(raw).to_ark_bytes()
ts = [threading.Thread(target=run, args=(i,)) for i in range(8)]
[t.start() for t in ts]; [t.join() for t in ts]
assert all(o == raw for o in out), "racing first calls disagree"
assert api.VerifyingKey.from_ark_bytes(raw).to_ark_bytes() == raw
print("ok")
""" % (ROOT_DIR, ROOT_DIR)
    if not os.path.exists(os.path.join(GOLD, "gpu_aes16_vk_ark.bin")):
        pytest.skip("GPU-made fixture not generated yet")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


# ---------------- ark-serialize layout of the verifying key ----------------
def _g1_compressed(x, y, q):
    b = bytearray(x.to_bytes(48, "little"))
    if y > (q - y) % q:
        b[47] |= 0x80
    return bytes(b)


def _g2_compressed(P, q):
    (x0, x1), (y0, y1) = P
    ny = ((q - y0) % q, (q - y1) % q)
    b = bytearray(x0.to_bytes(48, "little") + x1.to_bytes(48, "little"))
    if (y1, y0) > (ny[1], ny[0]):                 # ark-ff QuadExtField order: c1 first, then c0
        b[95] |= 0x80
    return bytes(b)


def test_vk_ark_layout_against_the_python_curve_model(zko, api, xor_setup):
    """field by field: the C++ encoder against tools/curve_math.py (independent big-int model) for every group element of the key"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import curve_math as cm
    ix, vk = xor_setup
    info = ix.info()
    beta, g_or, gg_or = ix.srs_trapdoor()
    mb, g, gg, h = cm.ark_kzg10_setup_points()          # third, independent model of KZG10::setup's draws (tools/curve_math.py)
    assert mb == beta and zko.pt_unpack(g_or)[0] == g and zko.pt_unpack(gg_or)[0] == gg
    raw = vk.to_ark_bytes()
    q, F = cm.Q377, cm.Fq2(cm.Q377, -5)
    off = 0
    for name in ("num_variables", "num_constraints", "num_non_zero", "num_instance"):
        assert int.from_bytes(raw[off:off + 8], "little") == info[name]
        off += 8
    assert int.from_bytes(raw[off:off + 8], "little") == 6
    off += 8
    comms = zko.pt_unpack(ix.comms())
    for (x, y) in comms:
        assert raw[off:off + 48] == _g1_compressed(x, y, q) and raw[off + 48] == 0
        off += 49
    assert raw[off:off + 48] == _g1_compressed(g[0], g[1], q)
    assert raw[off + 48:off + 96] == _g1_compressed(gg[0], gg[1], q)
    off += 96
    assert raw[off:off + 96] == _g2_compressed(h, q)
    assert raw[off + 96:off + 192] == _g2_compressed(cm.ec2_mul(F, beta, h), q)
    off += 192
    assert raw[off] == 1 and int.from_bytes(raw[off + 1:off + 9], "little") == 2
    off += 9
    bounds = sorted((info["h"] - 2, info["k"] - 2))
    for b in bounds:
        assert int.from_bytes(raw[off:off + 8], "little") == b
        sp = cm.ec_mul(pow(beta, info["max_degree"] - b, cm.R377), g, q)
        assert raw[off + 8:off + 56] == _g1_compressed(sp[0], sp[1], q)
        off += 56
    assert int.from_bytes(raw[off:off + 8], "little") == info["max_degree"]
    assert int.from_bytes(raw[off + 8:off + 16], "little") == info["supported_degree"]
    assert off + 16 == len(raw) == 759


def test_vk_ark_roundtrip_verifies_and_rejects_garbage(zko, api, xor_setup):
    ix, vk = xor_setup
    raw = vk.to_ark_bytes()
    vk2 = api.VerifyingKey.from_ark_bytes(raw)           # G1 and G2 points are decompressed (Fq and Fq2 square roots)
    assert vk2.to_ark_bytes() == raw
    cs, _ = zko.synth_ops("xor", 0xCAFE, 0xF00D, field=377)
    proof = ix.prove(cs).to_bytes()
    assert vk2.verify(proof, b"") is True
    with pytest.raises(api.ZkAesError):
        api.VerifyingKey.from_ark_bytes(raw[:-1])
    with pytest.raises(api.ZkAesError):
        api.VerifyingKey.from_ark_bytes(raw + b"\0")
    bad = bytearray(raw)
    bad[32] = 7                                          # index_comms length
    with pytest.raises(api.ZkAesError):
        api.VerifyingKey.from_ark_bytes(bytes(bad))
    flipped = bytearray(raw)
    flipped[40 + 6 * 49 + 96 + 95] ^= 0x80               # sign flag of h: -h is a valid point, the pairing check must now fail
    assert api.VerifyingKey.from_ark_bytes(bytes(flipped)).verify(proof, b"") is False


def test_vk_ark_uncompressed_layout(zko, api, xor_setup):
    """serialize_uncompressed's image of the IndexVerifierKey -- what deserialize_unchecked reads (ark-serialize 0.3: it defaults to the UNCOMPRESSED layout): every G1 element
    is x || y (96 B), every G2 element x.c0 x.c1 y.c0 y.c1 (192 B), no flag bit set for finite points; scalars and lengths as in the compressed image"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import curve_math as cm
    ix, vk = xor_setup
    info = ix.info()
    beta, g_or, gg_or = ix.srs_trapdoor()
    _, g, gg, h = cm.ark_kzg10_setup_points()
    F = cm.Fq2(cm.Q377, -5)
    raw, comp = vk.to_ark_bytes_uncompressed(), vk.to_ark_bytes()
    le = lambda v: int(v).to_bytes(48, "little")
    g1 = lambda pt: le(pt[0]) + le(pt[1])
    g2 = lambda pt: le(pt[0][0]) + le(pt[0][1]) + le(pt[1][0]) + le(pt[1][1])
    assert raw[:40] == comp[:40]                              # index_info + the length of index_comms
    off = 40
    for pt in zko.pt_unpack(ix.comms()):
        assert raw[off:off + 96] == g1(pt) and raw[off + 96] == 0
        off += 97
    assert raw[off:off + 96] == g1(g) and raw[off + 96:off + 192] == g1(gg)
    off += 192
    assert raw[off:off + 192] == g2(h) and raw[off + 192:off + 384] == g2(cm.ec2_mul(F, beta, h))
    off += 384
    assert raw[off] == 1 and int.from_bytes(raw[off + 1:off + 9], "little") == 2
    off += 9
    for b in sorted((info["h"] - 2, info["k"] - 2)):
        assert int.from_bytes(raw[off:off + 8], "little") == b
        assert raw[off + 8:off + 104] == g1(cm.ec_mul(pow(beta, info["max_degree"] - b, cm.R377), g, cm.Q377))
        off += 104
    assert raw[off:] == comp[-16:] and off + 16 == len(raw) == 759 + 10 * 48 + 2 * 96


def _parse_pk_image(raw, p, expect_vk_prefix=0, g1_bytes=48):
    """walk the ark-serialize IndexProverKey image after its index_vk prefix; returns a dict of what it holds and checks that the file ends exactly"""
    import struct
    pos = [expect_vk_prefix]

    def u64():
        v = struct.unpack_from("<Q", raw, pos[0])[0]; pos[0] += 8; return v

    def u8():
        v = raw[pos[0]]; pos[0] += 1; return v

    def fr():
        v = int.from_bytes(raw[pos[0]:pos[0] + 32], "little"); pos[0] += 32; assert v < p; return v
    out = {}
    assert u64() == 6
    for _ in range(6):
        assert u64() == 0 and u8() == 0                       # empty blinding polynomial, no shifted_rand
    out["info"] = [u64() for _ in range(4)]
    out["nnz"] = []
    for _ in range(3):
        rows = u64(); nnz = 0
        for _ in range(rows):
            k = u64()
            for _ in range(k):
                fr(); col = u64(); assert col < out["info"][0]; nnz += 1
        assert rows == out["info"][1]
        out["nnz"].append(nnz)
    out["labels"] = []
    for _ in range(6):
        n = u64(); out["labels"].append(raw[pos[0]:pos[0] + n].decode()); pos[0] += n
        ln = u64(); pos[0] += 32 * ln; assert u8() == 0 and u8() == 0
    for _ in range(6):
        k = u64(); pos[0] += 32 * k
        assert u8() == 0 and u64() == k
        lg = struct.unpack_from("<I", raw, pos[0])[0]; pos[0] += 4; assert 1 << lg == k
        size_fe, size_inv, gen, gen_inv, cg_inv = fr(), fr(), fr(), fr(), fr()
        assert size_fe == k and size_fe * size_inv % p == 1 and gen * gen_inv % p == 1 and pow(gen, k, p) == 1 and pow(gen, k // 2, p) == p - 1
        out["k"] = k; out["coset_gen_inv"] = cg_inv
    npow = u64(); out["powers_at"] = pos[0]; pos[0] += g1_bytes * npow
    assert u8() == 1
    nshift = u64(); out["shifted_at"] = pos[0]; pos[0] += g1_bytes * nshift
    assert u64() == 3; pos[0] += g1_bytes * 3
    assert u8() == 1 and u64() == 2
    out["bounds"] = [u64(), u64()]
    out["max_degree"] = u64()
    out["powers"], out["shifted"] = npow, nshift
    assert pos[0] == len(raw), (pos[0], len(raw))
    return out


def test_oracle_pk_image_of_the_xor_circuit_is_well_formed(zko, tmp_path):
    """the checker's ark-serialize IndexProverKey writer (oracle/zko_marlin.c) on the src/ops.rs xor gate: every length field adds up, the Radix2 domain fields are
    consistent, labels and field order as ark-marlin's joint arithmetization [RECALL], the file ends where the committer key ends"""
    cs, _ = zko.synth_ops("xor", 0, 0, field=377)
    ix = zko.Index(cs, srs=(200, 200, 600))
    path = str(tmp_path / "pk_xor.bin")
    size = ix.pk_serialize_ark_to_file(path)
    raw = open(path, "rb").read()
    assert size == len(raw) > 0
    got = _parse_pk_image(raw, zko.R377)
    info = ix.info()
    assert got["info"] == [info["num_variables"], info["num_constraints"], info["num_non_zero"], info["num_instance"]]
    assert got["labels"] == ["row", "col", "a_val", "b_val", "c_val", "row_col"] and got["k"] == info["k"]
    assert got["powers"] == info["supported_degree"] + 1 and got["max_degree"] == info["max_degree"]
    assert got["bounds"] == sorted([info["h"] - 2, info["k"] - 2]) and got["shifted"] == max(got["bounds"]) + 1
    assert got["coset_gen_inv"] * 22 % zko.R377 == 1            # Fr377's multiplicative generator
    # the uncompressed image (what deserialize_unchecked reads): the same walk with 96-byte points, which are x || y of the committer key's powers
    size_u = ix.pk_serialize_ark_to_file(path, uncompressed=True)
    raw_u = open(path, "rb").read()
    gu = _parse_pk_image(raw_u, zko.R377, g1_bytes=96)
    assert size_u == len(raw_u) == size + 48 * (got["powers"] + got["shifted"] + 3)
    assert raw_u[:gu["powers_at"]] == raw[:got["powers_at"]]                      # everything before the first point is identical
    le = lambda v: int(v).to_bytes(48, "little")
    pts = zko.pt_unpack(ix.srs_powers(0, 4))
    for i, (x, y) in enumerate(pts):
        assert raw_u[gu["powers_at"] + 96 * i:gu["powers_at"] + 96 * (i + 1)] == le(x) + le(y)
    lowest = info["max_degree"] - max(got["bounds"])
    (x, y), = zko.pt_unpack(ix.srs_powers(lowest, 1))
    assert raw_u[gu["shifted_at"]:gu["shifted_at"] + 96] == le(x) + le(y)
