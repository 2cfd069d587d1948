"""ctypes binding of include/zkaes.h, mirroring the reference's public functions (src/lib.rs:60,116,138)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CIRCUIT_AES, CIRCUIT_OPS_XOR, CIRCUIT_OPS_ADD = 0, 1, 2
KEY_NO_TABLES = 1                  # zkaes_synthesize_keys_ex2 flag: no fixed-base window tables (saves 10-42 GB per key)
PARITY = "parity"                  # zk_seed=PARITY: the reference's fixed ark_std::test_rng() stream for every proof (byte-parity tests only)


class ZkAesError(RuntimeError):
    """The Err(..) side of the reference's anyhow::Result."""


def lib_path():
    return os.path.join(HERE, "libzkaes.so")


_lib = None


def lib():
    """Load libzkaes.so (must have been built: python -m aes_zero_knowledge_proof_circuit_amd.build)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ZkAesError("libzkaes.so is not built (run: python -m aes_zero_knowledge_proof_circuit_amd.build)")
        L = C.CDLL(path)
        L.zkaes_last_error.restype = C.c_char_p
        for name in ("zkaes_bytes_free", "zkaes_pk_free", "zkaes_vk_free"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise ZkAesError(lib().zkaes_last_error().decode())


def _take(ptr, n):
    data = C.string_at(ptr, n.value)
    lib().zkaes_bytes_free(ptr)
    return data


def device_count():
    return lib().zkaes_device_count()


def set_device(ordinal):
    _check(lib().zkaes_set_device(int(ordinal)))


class VerifyingKey:
    def __init__(self, ptr):
        self._p = C.c_void_p(ptr)

    def clone(self):  # the reference passes keys by value; callers .clone() them (tests/integration_tests.rs:330)
        return VerifyingKey.from_bytes(self.to_bytes())

    def to_bytes(self):
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().zkaes_vk_serialize(self._p, C.byref(out), C.byref(n)))
        return _take(out, n)

    @staticmethod
    def from_bytes(b):
        p = C.c_void_p()
        _check(lib().zkaes_vk_deserialize(bytes(b), C.c_size_t(len(b)), C.byref(p)))
        return VerifyingKey(p.value)

    def to_ark_bytes(self):
        """ark-serialize 0.3 compressed IndexVerifierKey bytes (what the Rust side's CanonicalSerialize writes)"""
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().zkaes_vk_serialize_ark(self._p, C.byref(out), C.byref(n)))
        return _take(out, n)

    def to_ark_bytes_uncompressed(self):
        """serialize_uncompressed's image of the IndexVerifierKey (96-byte G1, 192-byte G2): what deserialize_unchecked reads"""
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().zkaes_vk_serialize_ark_uncompressed(self._p, C.byref(out), C.byref(n)))
        return _take(out, n)

    @staticmethod
    def from_ark_bytes(b):
        p = C.c_void_p()
        _check(lib().zkaes_vk_deserialize_ark(bytes(b), C.c_size_t(len(b)), C.byref(p)))
        return VerifyingKey(p.value)

    @staticmethod
    def from_trapdoor(info, index_comms, beta_mont):
        arr = (C.c_uint64 * 7)(*info)
        p = C.c_void_p()
        _check(lib().zkaes_vk_from_trapdoor(arr, bytes(index_comms), bytes(beta_mont), C.byref(p)))
        return VerifyingKey(p.value)

    def verify(self, proof, public_input_bits):
        acc = C.c_int()
        bits = bytes(public_input_bits)
        _check(lib().zkaes_verify(self._p, bytes(proof), C.c_size_t(len(proof)), bits, C.c_size_t(len(bits)), C.byref(acc)))
        return bool(acc.value)

    def __del__(self):
        try:
            if self._p:
                lib().zkaes_vk_free(self._p)
                self._p = None
        except Exception:
            pass


class ProvingKey:
    def __init__(self, ptr):
        self._p = C.c_void_p(ptr)

    def clone(self):  # device-resident and immutable: a clone is the same handle (benches/benchmark_encrypt.rs:46 clones per call)
        return self

    def info(self):
        out = (C.c_uint64 * 12)()
        _check(lib().zkaes_pk_info(self._p, out))
        keys = ["raw_constraints", "raw_instance", "raw_witness", "nnz_a", "nnz_b", "nnz_c", "constraints", "instance", "witness", "joint_nnz", "h", "k"]
        return dict(zip(keys, out))

    def timings(self):
        out = (C.c_double * 6)()
        _check(lib().zkaes_pk_timings(self._p, out))
        return dict(zip(["witness_ms", "round1_ms", "round2_ms", "round3_ms", "open_ms", "total_ms"], out))

    def debug_fetch(self, name):
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().zkaes_pk_debug_fetch(self._p, name.encode(), C.byref(out), C.byref(n)))
        return _take(out, n)

    def serialize_ark_to_file(self, path, uncompressed=False):
        """ark-serialize image of the arkworks IndexProverKey (index_vk, matrices, index polynomials + evaluations, committer key) streamed to `path`; returns its size.
        uncompressed=True: serialize_uncompressed's image (96-byte points) -- what IndexProverKey::deserialize_unchecked reads"""
        n = C.c_uint64()
        _check(lib().zkaes_pk_serialize_ark_to_file_ex(self._p, os.fsencode(path), 1 if uncompressed else 0, C.byref(n)))
        return int(n.value)

    def set_contexts(self, n):
        """proofs in flight per multi-proof call on this key (1..64; 0 = the process default)"""
        _check(lib().zkaes_pk_set_contexts(self._p, C.c_size_t(int(n))))

    def contexts(self):
        n = C.c_size_t()
        _check(lib().zkaes_pk_get_contexts(self._p, C.byref(n)))
        return int(n.value)

    def srs_info(self):
        """the universal SRS behind the key (one per process, device and SRS literals, shared by every key over it)"""
        out, secs = (C.c_uint64 * 6)(), (C.c_double * 2)()
        _check(lib().zkaes_pk_srs_info(self._p, out, secs))
        d = dict(zip(["max_degree", "points_per_copy", "copies", "bytes", "keys_sharing", "lagrange_bytes"], [int(v) for v in out]))
        d["srs_build_s"], d["setup_s"] = float(secs[0]), float(secs[1])
        return d

    def op_lists(self, message, secret_key, throughput_path=True):
        """the transforms and MSMs the library ACTUALLY launches for one proof on this key (zkaes_pk_op_lists): dict with "ntt" [[points, transforms per launch], ...] and
        "msm" [[points, kind], ...] + the circuit sizes.  Process-global recorder: no other proof may be in flight."""
        import json
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().zkaes_pk_op_lists(self._p, bytes(message), C.c_size_t(len(message)), bytes(secret_key), 1 if throughput_path else 0, C.byref(out), C.byref(n)))
        return json.loads(_take(out, n).decode())

    def tables_built(self):
        """(built, bytes): does the key hold the fixed-base window tables of its SRS?"""
        b, n = C.c_int(), C.c_uint64()
        _check(lib().zkaes_pk_tables_built(self._p, C.byref(b), C.byref(n)))
        return bool(b.value), int(n.value)

    def msm_partial_dev(self, scalars_bytes, offset, dev_ptr, dev_bytes=192):
        """this rank's share of ONE MSM over the key's SRS powers [offset, offset + n) on the prover's own path (Edwards tables); the partial sum (one XYZZ point,
        192 B) stays in device memory at dev_ptr"""
        n = len(scalars_bytes) // 32
        _check(lib().zkaes_pk_msm_partial_dev(self._p, bytes(scalars_bytes), C.c_size_t(n), C.c_size_t(offset), C.c_void_p(dev_ptr), C.c_size_t(dev_bytes)))

    def witness(self, message, secret_key):
        n = C.c_size_t()
        _check(lib().zkaes_aes_witness(self._p, bytes(message), C.c_size_t(len(message)), bytes(secret_key), None, C.c_size_t(0), C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(lib().zkaes_aes_witness(self._p, bytes(message), C.c_size_t(len(message)), bytes(secret_key), buf, n, C.byref(n)))
        return buf.raw

    def prove_ops(self, x, y, zk_seed=None):
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().zkaes_prove_ops(self._p, C.c_uint32(x), C.c_uint32(y), zk_seed, C.byref(out), C.byref(n)))
        return _take(out, n)

    @staticmethod
    def _seed_arg(zk_seed):
        """None -> fresh OS seed per call (the unseeded C entry points); PARITY -> NULL seed = fixed test_rng stream; else 32 bytes"""
        if zk_seed is None or zk_seed == PARITY:
            return None
        if len(zk_seed) != 32:
            raise ZkAesError("zk_seed must be 32 bytes")
        return bytes(zk_seed)

    def encrypt_chunked(self, message, secret_key, zk_seed=None, first_proof_index=0):
        """chunk-proofs of a long ECB message.  zk_seed: None = a fresh OS seed per call, 32 bytes = caller's seed (proof i uses index first_proof_index + i),
        PARITY = the reference's fixed prover randomness for every proof"""
        if len(secret_key) != 16:
            raise ZkAesError("secret_key must be 16 bytes")
        seed = self._seed_arg(zk_seed)
        chunk = (self.info()["raw_instance"] - 1) // 8          # 8 public-input bits per ciphertext byte
        n_chunks = len(message) // chunk
        lens = (C.c_size_t * max(n_chunks, 1))()
        out, n = C.c_void_p(), C.c_size_t()
        if zk_seed is None:
            _check(lib().zkaes_encrypt_chunked(bytes(message), C.c_size_t(len(message)), bytes(secret_key), self._p, C.byref(out), C.byref(n), lens, C.c_size_t(n_chunks)))
        else:
            _check(lib().zkaes_encrypt_chunked_seeded_at(bytes(message), C.c_size_t(len(message)), bytes(secret_key), self._p, seed, C.c_uint64(first_proof_index), C.byref(out), C.byref(n), lens,
                                                         C.c_size_t(n_chunks)))
        blob = _take(out, n)
        proofs, off = [], 0
        for i in range(n_chunks):
            proofs.append(blob[off:off + lens[i]])
            off += lens[i]
        return proofs

    def encrypt_batch(self, messages, secret_keys, zk_seed=None, first_proof_index=0):
        """n independent proofs: messages = list of equal-length byte strings (the key's plaintext length), secret_keys = list of 16-byte keys; zk_seed as encrypt_chunked"""
        n = len(messages)
        chunk = (self.info()["raw_instance"] - 1) // 8
        if len(secret_keys) != n:
            raise ZkAesError("one secret key per message")
        if any(len(m) != chunk for m in messages):
            raise ZkAesError("every message must be %d bytes (the key's plaintext length)" % chunk)
        if any(len(k) != 16 for k in secret_keys):
            raise ZkAesError("secret_key must be 16 bytes")
        seed = self._seed_arg(zk_seed)
        lens = (C.c_size_t * max(n, 1))()
        out, total = C.c_void_p(), C.c_size_t()
        mb, kb = b"".join(bytes(m) for m in messages), b"".join(bytes(k) for k in secret_keys)
        if zk_seed is None:
            _check(lib().zkaes_encrypt_batch(C.c_size_t(n), mb, kb, self._p, C.byref(out), C.byref(total), lens))
        else:
            _check(lib().zkaes_encrypt_batch_seeded_at(C.c_size_t(n), mb, C.c_size_t(len(mb)), kb, C.c_size_t(len(kb)), self._p, seed, C.c_uint64(first_proof_index), C.byref(out), C.byref(total), lens))
        blob = _take(out, total)
        proofs, off = [], 0
        for i in range(n):
            proofs.append(blob[off:off + lens[i]])
            off += lens[i]
        return proofs

    def free(self):
        """release the key now (device memory: index, prover contexts, and the universal SRS if this was its last key) instead of at garbage collection"""
        if self._p:
            lib().zkaes_pk_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synthesize_keys(plaintext_length, circuit=CIRCUIT_AES, srs=(866_944, 513, 4_062_064), flags=0):
    """zk_aes::synthesize_keys (src/lib.rs:138-174) -> (ProvingKey, VerifyingKey).  flags: KEY_NO_TABLES"""
    pk, vk = C.c_void_p(), C.c_void_p()
    _check(lib().zkaes_synthesize_keys_ex2(int(circuit), C.c_size_t(plaintext_length), C.c_size_t(srs[0]), C.c_size_t(srs[1]), C.c_size_t(srs[2]), C.c_uint(flags), C.byref(pk), C.byref(vk)))
    return ProvingKey(pk.value), VerifyingKey(vk.value)


def encrypt(message, secret_key, proving_key, zk_seed=None):
    """zk_aes::encrypt (src/lib.rs:60-114) -> serialized MarlinProof bytes."""
    if len(secret_key) != 16:
        raise ZkAesError("secret_key must be 16 bytes")
    out, n = C.c_void_p(), C.c_size_t()
    _check(lib().zkaes_encrypt_seeded(bytes(message), C.c_size_t(len(message)), bytes(secret_key), proving_key._p, zk_seed, C.byref(out), C.byref(n)))
    return _take(out, n)


def verify_encryption(verifying_key, proof, ciphertext):
    """zk_aes::verify_encryption (src/lib.rs:116-136) -> bool."""
    acc = C.c_int()
    _check(lib().zkaes_verify_encryption(verifying_key._p, bytes(proof), C.c_size_t(len(proof)), bytes(ciphertext), C.c_size_t(len(ciphertext)), C.byref(acc)))
    return bool(acc.value)


def proof_roundtrip(proof):
    out, n = C.c_void_p(), C.c_size_t()
    _check(lib().zkaes_proof_roundtrip(bytes(proof), C.c_size_t(len(proof)), C.byref(out), C.byref(n)))
    return _take(out, n)


def circuit_info(circuit, plaintext_length):
    out = (C.c_uint64 * 12)()
    _check(lib().zkaes_circuit_info(int(circuit), C.c_size_t(plaintext_length), out))
    keys = ["raw_constraints", "raw_instance", "raw_witness", "nnz_a", "nnz_b", "nnz_c", "constraints", "instance", "witness", "joint_nnz", "h", "k"]
    return dict(zip(keys, out))


def circuit_matrix(circuit, plaintext_length, which):
    rows, nnz = C.c_uint64(), C.c_uint64()
    _check(lib().zkaes_circuit_matrix(int(circuit), C.c_size_t(plaintext_length), which, C.byref(rows), C.byref(nnz), None, None, None))
    rowptr = np.zeros(rows.value + 1, dtype=np.uint32)
    col = np.zeros(max(nnz.value, 1), dtype=np.uint32)
    coeff = np.zeros(max(nnz.value, 1), dtype=np.int64)
    _check(lib().zkaes_circuit_matrix(int(circuit), C.c_size_t(plaintext_length), which, None, None, rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                      coeff.ctypes.data_as(C.c_void_p)))
    return rowptr, col[:nnz.value], coeff[:nnz.value]


def ntt(field_id, data_mont_bytes, inverse=False):
    n = len(data_mont_bytes) // 32
    buf = C.create_string_buffer(bytes(data_mont_bytes), len(data_mont_bytes))
    _check(lib().zkaes_ntt(int(field_id), buf, C.c_size_t(n), 1 if inverse else 0))
    return buf.raw


def ntt_coset(field_id, data_mont_bytes, coset_c, lg_big, inverse=False):
    n = len(data_mont_bytes) // 32
    buf = C.create_string_buffer(bytes(data_mont_bytes), len(data_mont_bytes))
    _check(lib().zkaes_ntt_coset(int(field_id), buf, C.c_size_t(n), 1 if inverse else 0, int(coset_c), int(lg_big)))
    return buf.raw


def ntt_batch(field_id, vectors, cosets=None, lg_big=0, inverse=False):
    """several transforms of one size in shared launches: vectors = list of equal-length Montgomery byte strings, cosets[i] = 0 (plain) or the coset index of vector i"""
    count, n = len(vectors), len(vectors[0]) // 32
    if any(len(v) != 32 * n for v in vectors):
        raise ZkAesError("ntt_batch: vectors of one length")
    buf = C.create_string_buffer(b"".join(bytes(v) for v in vectors), 32 * n * count)
    cs = (C.c_int * count)(*[int(c) for c in cosets]) if cosets is not None else None
    _check(lib().zkaes_ntt_batch(int(field_id), buf, C.c_size_t(n), count, 1 if inverse else 0, cs, int(lg_big)))
    return [buf.raw[32 * n * i:32 * n * (i + 1)] for i in range(count)]


def msm(curve_id, bases_bytes, scalars_bytes):
    n = len(scalars_bytes) // 32
    out = C.create_string_buffer(96)
    inf = C.c_int()
    _check(lib().zkaes_msm(int(curve_id), bytes(bases_bytes), bytes(scalars_bytes), C.c_size_t(n), out, C.byref(inf)))
    return out.raw, bool(inf.value)


def g1_sum(curve_id, points):
    """Host-side sum of [(xy_bytes96, is_inf), ...] -> (xy_bytes96, is_inf)."""
    n = len(points)
    buf = b"".join(bytes(p[0]) for p in points)
    inf = (C.c_int * max(n, 1))(*[1 if p[1] else 0 for p in points])
    out = C.create_string_buffer(96)
    oinf = C.c_int()
    _check(lib().zkaes_g1_sum(int(curve_id), buf, inf, C.c_size_t(n), out, C.byref(oinf)))
    return out.raw, bool(oinf.value)


def msm_sharded_plan(curve_id, n_total):
    """(window_bits, n_windows, bytes_per_rank) of a point-range-sharded MSM over n_total points"""
    c, w, b = C.c_int(), C.c_int(), C.c_size_t()
    _check(lib().zkaes_msm_sharded_plan(int(curve_id), C.c_size_t(n_total), C.byref(c), C.byref(w), C.byref(b)))
    return c.value, w.value, b.value


def msm_window_sums_dev(curve_id, bases_bytes, scalars_bytes, n_total, dev_ptr, dev_bytes):
    """Pippenger over this rank's slice; the window sums stay in device memory at dev_ptr (an int device address, e.g. tensor.data_ptr())"""
    n = len(scalars_bytes) // 32
    _check(lib().zkaes_msm_window_sums_dev(int(curve_id), bytes(bases_bytes), bytes(scalars_bytes), C.c_size_t(n), C.c_size_t(n_total), C.c_void_p(dev_ptr), C.c_size_t(dev_bytes)))


def msm_fold_window_sums_dev(curve_id, dev_ptr, world, n_total):
    out = C.create_string_buffer(96)
    inf = C.c_int()
    _check(lib().zkaes_msm_fold_window_sums_dev(int(curve_id), C.c_void_p(dev_ptr), int(world), C.c_size_t(n_total), out, C.byref(inf)))
    return out.raw, bool(inf.value)


def msm_fold_partials_dev(curve_id, dev_ptr, world):
    out = C.create_string_buffer(96)
    inf = C.c_int()
    _check(lib().zkaes_msm_fold_partials_dev(int(curve_id), C.c_void_p(dev_ptr), int(world), out, C.byref(inf)))
    return out.raw, bool(inf.value)


def msm_table(curve_id, bases_bytes, scalars_bytes, window_bits, srs=False):
    """precomputed-window MSM.  srs=True (377 only): the prover's SRS path on the twisted Edwards model -- bases MUST lie in the prime-order subgroup"""
    n = len(scalars_bytes) // 32
    out = C.create_string_buffer(96)
    inf = C.c_int()
    if srs:
        if int(curve_id) != 377:
            raise ValueError("the SRS (twisted Edwards) table path exists for BLS12-377 only")
        _check(lib().zkaes_msm_table_srs(bytes(bases_bytes), bytes(scalars_bytes), C.c_size_t(n), int(window_bits), out, C.byref(inf)))
    else:
        _check(lib().zkaes_msm_table(int(curve_id), bytes(bases_bytes), bytes(scalars_bytes), C.c_size_t(n), int(window_bits), out, C.byref(inf)))
    return out.raw, bool(inf.value)


def msm_bench(curve_id, bases_bytes, scalars_bytes, reps=3):
    n = len(scalars_bytes) // 32
    t, a = C.c_double(), C.c_double()
    _check(lib().zkaes_msm_bench(int(curve_id), bytes(bases_bytes), bytes(scalars_bytes), C.c_size_t(n), int(reps), C.byref(t), C.byref(a)))
    return t.value, a.value


def msm_bench_synth(n, window_bits=0, reps=3, want_point=False):
    t, a = C.c_double(), C.c_double()
    out = C.create_string_buffer(96)
    _check(lib().zkaes_msm_bench_synth(C.c_size_t(n), int(window_bits), int(reps), C.byref(t), C.byref(a), out))
    return (t.value, a.value, out.raw) if want_point else (t.value, a.value)


def set_default_contexts(n):
    """zkaes_set_default_contexts: the process default of prover contexts per key (0 = back to ZKAES_CONTEXTS / 12); also what key synthesis reserves beside the window tables"""
    _check(lib().zkaes_set_default_contexts(C.c_size_t(n)))


def srs_hold(hold=True):
    """zkaes_srs_hold: keep every universal / Lagrange SRS resident after its last key is freed (hold=False releases them again)"""
    _check(lib().zkaes_srs_hold(1 if hold else 0))


def int_rate_bench(seconds=0.5):
    """zkaes_int_rate_bench: per-box calibration of the integer roof (include/zkaes.h)"""
    out = (C.c_double * 8)()
    _check(lib().zkaes_int_rate_bench(C.c_double(seconds), out))
    return {"fq_products_per_s": out[0], "fq_stream_sclk_mhz": out[1], "mad_per_s": 378.0 * out[0], "hot_loop_l2_additions_per_s": out[2], "hot_loop_l2_sclk_mhz": out[3],
            "hot_loop_cycles_per_addition_per_wave": out[4], "rounds": int(out[5])}


def stream_copy_bench(nbytes=1 << 30, reps=20):
    """Measured HBM stream-copy rate (read + write GB/s) of a plain 16 B/lane copy kernel -- printed beside the nominal peak."""
    g = C.c_double()
    _check(lib().zkaes_stream_copy_bench(C.c_size_t(nbytes), int(reps), C.byref(g)))
    return g.value


def mem_info():
    """(free, total) bytes of the current device"""
    f, t = C.c_uint64(), C.c_uint64()
    _check(lib().zkaes_mem_info(C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def msm_stats(reset=False):
    out = (C.c_double * 5)()
    _check(lib().zkaes_msm_stats(out, 1 if reset else 0))
    return dict(accumulate_ms=out[0], total_ms=out[1], points=int(out[2]), launches=int(out[3]), pairs=int(out[4]))
