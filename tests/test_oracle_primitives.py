"""CPU tests pinning the oracle (oracle/) against the reference's golden vectors and against independent models
(hashlib, RFC vectors, Python big-int arithmetic in tools/curve_math.py)."""
import ctypes as C
import hashlib
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import curve_math as cm  # noqa: E402


def test_blake2s_matches_hashlib(zko):
    L = zko.lib()
    for msg in [b"", b"abc", b"x" * 63, b"y" * 64, b"z" * 65, bytes(range(256)) * 5]:
        out = C.create_string_buffer(32)
        L.zko_blake2s(out, msg, C.c_size_t(len(msg)))
        assert out.raw == hashlib.blake2s(msg).digest()


def test_chacha20_zero_key_keystream(zko):
    w = (C.c_uint32 * 16)()
    zko.lib().zko_api_chacha_words(bytes(32), 20, C.c_size_t(16), w)
    ks = b"".join(int(x).to_bytes(4, "little") for x in w)
    # well-known ChaCha20 keystream for the all-zero key / nonce / counter
    assert ks.hex().startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")


def test_chacha_block_counter_and_u64_pairs(zko):
    n = 200
    w = (C.c_uint32 * n)()
    zko.lib().zko_api_chacha_words(bytes(range(32)), 12, C.c_size_t(n), w)
    assert len(set(w)) > 190   # four-block refills keep producing fresh words past the 64-word buffer


def test_aes_fips197_every_round(zko, vectors):
    L = zko.lib()
    key = bytes(vectors["key"])
    pt = bytes(vectors["plaintext"])
    rk = (C.c_uint8 * (11 * 16))()
    L.zko_aes_derive_keys(rk, key)
    rks = [bytes(rk[16 * i:16 * i + 16]) for i in range(11)]
    assert list(rks[10]) == vectors["round_key_10"]
    buf = C.create_string_buffer(16)
    L.zko_aes_add_round_key(buf, pt, rks[0])
    state = buf.raw
    assert list(state) == vectors["expected_start_of_round"][0] == vectors["add_round_key_expected"]
    for r in range(1, 11):
        assert list(state) == vectors["expected_start_of_round"][r - 1]
        L.zko_aes_substitute_bytes(buf, state); sb = buf.raw
        assert list(sb) == vectors["expected_after_substituting_bytes"][r - 1]
        L.zko_aes_shift_rows(buf, sb); sr = buf.raw
        assert list(sr) == vectors["expected_after_shift_rows"][r - 1]
        if r != 10:
            L.zko_aes_mix_columns(buf, sr); mc = buf.raw
            assert list(mc) == vectors["expected_after_mix_columns"][r - 1]
        else:
            mc = sr
        L.zko_aes_add_round_key(buf, mc, rks[r]); state = buf.raw
    assert list(state) == vectors["ciphertext"]
    assert list(zko.aes_encrypt(pt, key)) == vectors["ciphertext"]
    assert list(zko.aes_encrypt(bytes(vectors["plaintext_64"]), key)) == vectors["ciphertext_64"]


def test_committed_oracle_fixtures_are_about_the_inputs_they_claim(zko, vectors):
    """tests/golden/oracle_aes64.json is the oracle's proof of the reference's OWN 64-byte test case (tests/integration_tests.rs:340-371): its message, key and ciphertext are the
    reference's numbers, the ciphertext is what the byte-level AES gives, and the proof is the 855 / 859-byte MarlinProof whose hash the file states (the -m gpu suite compares the GPU's
    bytes with it); the same consistency for the 6-block bench fixture."""
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = json.load(open(os.path.join(gold, "oracle_aes64.json")))
    assert bytes.fromhex(fx["message"]) == bytes(vectors["plaintext_64"]) and bytes.fromhex(fx["key"]) == bytes(vectors["key"])
    assert bytes.fromhex(fx["ciphertext"]) == bytes(vectors["ciphertext_64"]) == zko.aes_encrypt(bytes(vectors["plaintext_64"]), bytes(vectors["key"]))
    assert fx["blocks"] == 4 and fx["index"]["h"] == 1 << 20 and fx["index"]["k"] == 1 << 22
    big = json.load(open(os.path.join(gold, "oracle_aes448.json")))       # 28 blocks over a universal SRS of (2^22, 513, 2^24): NOT the reference's literal (bench.py's `big` leg)
    assert big["blocks"] == 28 and big["srs_literals"] == [1 << 22, 513, 1 << 24] and (big["index"]["h"], big["index"]["k"]) == (1 << 22, 1 << 24)
    assert big["index"]["num_constraints"] == 36768 + 148272 * 28
    for name in ("oracle_aes64.json", "oracle_aes96.json", "oracle_aes208.json", "oracle_aes448.json"):
        f = json.load(open(os.path.join(gold, name)))
        proof = bytes.fromhex(f["proof"])
        assert hashlib.sha256(proof).hexdigest() == f["proof_sha256"] and len(proof) in (855, 859)
        assert bytes.fromhex(f["ciphertext"]) == zko.aes_encrypt(bytes.fromhex(f["message"]), bytes.fromhex(f["key"]))
        assert set(f["poly_sha256"]) == set(zko.POLY_NAMES)


def test_sbox_equals_reference_lookup_table(zko, vectors):
    L = zko.lib()
    assert [L.zko_aes_substitute_byte(C.c_uint8(i)) for i in range(256)] == vectors["lookup_table"]


def test_gate_kats(zko, vectors):
    L = zko.lib()
    buf = C.create_string_buffer(16)
    L.zko_aes_mix_columns(buf, bytes(vectors["mix_columns_input"]))
    assert list(buf.raw) == vectors["mix_columns_expected"]
    L.zko_aes_substitute_bytes(buf, bytes(vectors["sub_bytes_input"]))
    assert list(buf.raw) == vectors["sub_bytes_expected"]
    data = bytes(range(16))
    L.zko_aes_shift_rows(buf, data)
    assert list(buf.raw) == [data[i] for i in vectors["shift_rows_index_map"]]


@pytest.mark.parametrize("cid", [377, 381])
def test_field_arithmetic_against_python_ints(zko, rng, cid):
    L = zko.lib()
    p = zko.FR[cid]
    for _ in range(50):
        a, b = rng.randrange(p), rng.randrange(p)
        out = C.create_string_buffer(32)
        L.zko_api_fr_mul(cid, zko.fr_pack([a], cid), zko.fr_pack([b], cid), out)
        assert zko.fr_unpack(out.raw, cid)[0] == a * b % p
        L.zko_api_fr_sub(cid, zko.fr_pack([a], cid), zko.fr_pack([b], cid), out)
        assert zko.fr_unpack(out.raw, cid)[0] == (a - b) % p
        L.zko_api_fr_inv(cid, zko.fr_pack([a], cid), out)
        assert zko.fr_unpack(out.raw, cid)[0] == pow(a, -1, p)
    q = zko.FQ[cid]
    for _ in range(20):
        a, b = rng.randrange(q), rng.randrange(q)
        out = C.create_string_buffer(48)
        L.zko_api_fq_mul(cid, zko.fq_to_mont(a, cid).to_bytes(48, "little"), zko.fq_to_mont(b, cid).to_bytes(48, "little"), out)
        assert zko.fq_from_mont(int.from_bytes(out.raw, "little"), cid) == a * b % q


@pytest.mark.parametrize("cid,gen,q", [(377, cm.G1_377, cm.Q377), (381, cm.G1_381, cm.Q381)])
def test_g1_scalar_mul_and_msm_against_python_ec(zko, rng, cid, gen, q):
    L = zko.lib()
    r = zko.FR[cid]
    g = C.create_string_buffer(96)
    L.zko_api_g1_generator(cid, g)
    assert zko.pt_unpack(g.raw, cid)[0] == gen
    ks = [rng.randrange(r) for _ in range(5)] + [0, 1, r - 1]
    pts = []
    for k in ks[:5]:
        out = C.create_string_buffer(96)
        inf = L.zko_api_g1_mul(cid, g.raw, zko.fr_pack([k], cid), out)
        assert not inf and zko.pt_unpack(out.raw, cid)[0] == cm.ec_mul(k, gen, q)
        pts.append(zko.pt_unpack(out.raw, cid)[0])
    # r * G = infinity
    out = C.create_string_buffer(96)
    assert L.zko_api_g1_mul(cid, g.raw, zko.fr_pack([0], cid), out) == 1
    # MSM (n = 40 exercises the c = ln(n)+2 window path, n = 5 the c = 3 path), incl. scalars 0 and 1
    for n in (5, 40):
        bases = [pts[i % 5] for i in range(n)]
        sc = [rng.randrange(r) for _ in range(n)]
        sc[0], sc[1] = 0, 1
        out = C.create_string_buffer(96)
        inf = L.zko_api_msm(cid, zko.pt_pack(bases, cid), zko.fr_pack(sc, cid), C.c_size_t(n), out)
        exp = None
        for b, s in zip(bases, sc):
            exp = cm.ec_add(exp, cm.ec_mul(s, b, q), q)
        assert not inf and zko.pt_unpack(out.raw, cid)[0] == exp


@pytest.mark.parametrize("cid", [377, 381])
def test_ntt_against_naive_dft(zko, rng, cid):
    L = zko.lib()
    p = zko.FR[cid]
    for n in (1, 2, 8, 64):
        a = [rng.randrange(p) for _ in range(n)]
        gbuf = C.create_string_buffer(32)
        L.zko_api_domain_gen(cid, C.c_size_t(n), gbuf)
        w = zko.fr_unpack(gbuf.raw, cid)[0]
        assert pow(w, n, p) == 1 and (n == 1 or pow(w, n // 2, p) == p - 1)
        assert w == pow(cm.FR[str(cid)]["root"], 1 << (cm.FR[str(cid)]["two_adicity"] - (n.bit_length() - 1)), p)
        buf = C.create_string_buffer(zko.fr_pack(a, cid), 32 * n)
        assert L.zko_api_ntt(cid, buf, C.c_size_t(n), 0) == 0
        got = zko.fr_unpack(buf.raw, cid)
        assert got == [sum(a[j] * pow(w, i * j, p) for j in range(n)) % p for i in range(n)]
        assert L.zko_api_ntt(cid, buf, C.c_size_t(n), 1) == 0
        assert zko.fr_unpack(buf.raw, cid) == a
        # coset transform = evaluation at g * w^i
        buf = C.create_string_buffer(zko.fr_pack(a, cid), 32 * n)
        L.zko_api_ntt(cid, buf, C.c_size_t(n), 2)
        g = cm.FR[str(cid)]["gen"]
        assert zko.fr_unpack(buf.raw, cid) == [sum(a[j] * pow(g * pow(w, i, p), j, p) for j in range(n)) % p for i in range(n)]


def test_fr_rand_is_montgomery_rejection_sampling(zko):
    # ark-ff: limbs from the rng, masked to 253 bits, rejected if >= p, used AS the Montgomery representation
    seed = bytes(range(32))
    n = 16
    words = (C.c_uint32 * 400)()
    zko.lib().zko_api_chacha_words(seed, 12, C.c_size_t(400), words)
    out = C.create_string_buffer(32 * n)
    zko.lib().zko_api_fr_rand_stream(377, seed, 12, C.c_size_t(n), out)
    got, i = [], 0
    while len(got) < n:
        limbs = [int(words[i + 2 * k]) | int(words[i + 2 * k + 1]) << 32 for k in range(4)]
        i += 8
        limbs[3] &= (1 << 61) - 1
        v = sum(l << (64 * k) for k, l in enumerate(limbs))
        if v < zko.R377:
            got.append(v)
    assert [int.from_bytes(out.raw[32 * k:32 * k + 32], "little") for k in range(n)] == got


def test_oracle_under_address_and_ub_sanitizers():
    """the CPU oracle built with -fsanitize=address,undefined walks every layer once at small sizes (byte-level AES KAT, one-block
    circuit satisfied / corrupted, NTT round trip on both fields, Pippenger MSM vs double-and-add, a complete Marlin index + proof of the ops xor gate):
    oracle/zko_selftest.c.  Any sanitizer report (or leak) fails the run."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-s", "-C", here, "selftest_asan"])
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(here, "selftest_asan")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
