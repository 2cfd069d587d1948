"""oracle/zko.py -- TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/libzko.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libzko.so")

R377 = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
Q377 = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
R381 = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
Q381 = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
FR = {377: R377, 381: R381}
FQ = {377: Q377, 381: Q381}


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s", "libzko.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        L.zko_api_synth_aes.restype = C.c_void_p
        L.zko_api_synth_ops.restype = C.c_void_p
        L.zko_api_index.restype = C.c_void_p
        L.zko_api_prove.restype = C.c_void_p
        L.zko_api_proof_bytes.restype = C.c_size_t
        L.zko_api_proof_poly_len.restype = C.c_size_t
        L.zko_api_now.restype = C.c_double
        L.zko_aes_substitute_byte.restype = C.c_uint8
    return _lib


# ---- Montgomery <-> int helpers (R = 2^256 for Fr, 2^384 for Fq) ----
def fr_to_mont(x, cid=377):
    return (x % FR[cid]) * (1 << 256) % FR[cid]


def fr_from_mont(m, cid=377):
    return m * pow(1 << 256, -1, FR[cid]) % FR[cid]


def fq_to_mont(x, cid=377):
    return (x % FQ[cid]) * (1 << 384) % FQ[cid]


def fq_from_mont(m, cid=377):
    return m * pow(1 << 384, -1, FQ[cid]) % FQ[cid]


def fr_pack(vals, cid=377):
    """list of canonical ints -> bytes of Montgomery limbs"""
    return b"".join(fr_to_mont(v, cid).to_bytes(32, "little") for v in vals)


def fr_unpack(buf, cid=377):
    return [fr_from_mont(int.from_bytes(buf[i:i + 32], "little"), cid) for i in range(0, len(buf), 32)]


def pt_pack(pts, cid=377):
    return b"".join(fq_to_mont(x, cid).to_bytes(48, "little") + fq_to_mont(y, cid).to_bytes(48, "little") for (x, y) in pts)


def pt_unpack(buf, cid=377):
    out = []
    for i in range(0, len(buf), 96):
        out.append((fq_from_mont(int.from_bytes(buf[i:i + 48], "little"), cid), fq_from_mont(int.from_bytes(buf[i + 48:i + 96], "little"), cid)))
    return out


class CS:
    """A synthesized constraint system held by the oracle."""

    def __init__(self, ptr):
        self.ptr = C.c_void_p(ptr)
        self.owned = True

    def counts(self):
        out = (C.c_uint64 * 6)()
        lib().zko_api_cs_counts(self.ptr, out)
        return dict(constraints=out[0], instance=out[1], witness=out[2], nnz_a=out[3], nnz_b=out[4], nnz_c=out[5])

    def assignment(self):
        c = self.counts()
        ins = C.create_string_buffer(c["instance"])
        wit = C.create_string_buffer(c["witness"])
        lib().zko_api_cs_assignment(self.ptr, ins, wit)
        return ins.raw, wit.raw

    def matrix(self, which):
        import numpy as np
        c = self.counts()
        nnz = [c["nnz_a"], c["nnz_b"], c["nnz_c"]][which]
        rowptr = np.zeros(c["constraints"] + 1, dtype=np.uint64)
        col = np.zeros(max(nnz, 1), dtype=np.uint32)
        coeff = np.zeros(max(nnz, 1), dtype=np.int64)
        lib().zko_api_cs_matrix(self.ptr, which, rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), coeff.ctypes.data_as(C.c_void_p))
        return rowptr, col[:nnz], coeff[:nnz]

    def set_witness(self, idx, val):
        """fault injection: overwrite witness variable idx (tests flip a bit and expect is_satisfied() to fail)"""
        if lib().zko_api_cs_set_witness(self.ptr, C.c_size_t(idx), int(val)) != 0:
            raise IndexError(idx)

    def is_satisfied(self):
        return lib().zko_cs_is_satisfied(self.ptr)

    def pad_for_marlin(self):
        lib().zko_cs_pad_for_marlin(self.ptr)

    def free(self):
        if self.owned and self.ptr:
            lib().zko_cs_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synth_aes(msg, key, field=377):
    ct = C.create_string_buffer(max(len(msg), 1))
    p = lib().zko_api_synth_aes(field, bytes(msg), C.c_size_t(len(msg)), bytes(key), ct)
    if not p:
        raise ValueError("Input must be 16 bytes length when adding round key")
    return CS(p), ct.raw[:len(msg)]


def synth_ops(which, x, y, field=381):
    out = C.c_uint32()
    p = lib().zko_api_synth_ops(field, {"xor": 0, "add": 1}[which], C.c_uint32(x), C.c_uint32(y), C.byref(out))
    return CS(p), out.value


def aes_encrypt(msg, key):
    out = C.create_string_buffer(max(len(msg), 1))
    lib().zko_aes_encrypt_ecb(out, bytes(msg), C.c_size_t(len(msg)), bytes(key))
    return out.raw[:len(msg)]


class Index:
    def __init__(self, cs, srs=(866_944, 513, 4_062_064)):
        """universal_setup(srs literals) + index; takes ownership of `cs` (src/lib.rs:141,173)."""
        self.ptr = C.c_void_p(lib().zko_api_index(cs.ptr, C.c_size_t(srs[0]), C.c_size_t(srs[1]), C.c_size_t(srs[2])))
        cs.owned = False
        cs.ptr = None
        self.field = 377

    def info(self):
        out = (C.c_uint64 * 8)()
        lib().zko_api_index_info(self.ptr, out)
        keys = ["num_variables", "num_constraints", "num_non_zero", "num_instance", "h", "k", "max_degree", "supported_degree"]
        return dict(zip(keys, out))

    def poly(self, which, form):
        k = self.info()["k"]
        buf = C.create_string_buffer(32 * k)
        lib().zko_api_index_poly(self.ptr, which, form, buf)
        return buf.raw

    def comms(self):
        buf = C.create_string_buffer(96 * 6)
        lib().zko_api_index_comms(self.ptr, buf)
        return buf.raw

    def srs_trapdoor(self):
        """(beta, g, gamma_g): the KZG10::setup trapdoor scalar and its two random G1 base points (96-byte affine, Montgomery coordinates)"""
        b = C.create_string_buffer(32)
        g = C.create_string_buffer(96)
        gg = C.create_string_buffer(96)
        lib().zko_api_srs_info(self.ptr, b, g, gg)
        return fr_unpack(b.raw)[0], g.raw, gg.raw

    def srs_powers(self, start, count):
        buf = C.create_string_buffer(96 * count)
        rc = lib().zko_api_srs_powers(self.ptr, C.c_size_t(start), C.c_size_t(count), buf)
        if rc:
            raise IndexError("power not held by the committer key")
        return buf.raw

    def pk_serialize_ark_to_file(self, path, uncompressed=False):
        """ark-serialize image of the IndexProverKey minus its index_vk prefix -> file; returns the size (uncompressed: 96-byte points, what deserialize_unchecked reads)"""
        lib().zko_api_pk_serialize_mode.restype = C.c_uint64
        return int(lib().zko_api_pk_serialize_mode(self.ptr, os.fsencode(path), 1 if uncompressed else 0))

    def prove(self, cs, zk_seed=None):
        p = lib().zko_api_prove(self.ptr, cs.ptr, zk_seed)
        if not p:
            raise RuntimeError("InstanceDoesNotMatchIndex")
        return Proof(p)

    def __del__(self):
        try:
            if self.ptr:
                lib().zko_index_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


POLY_NAMES = ["w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"]


class Proof:
    def __init__(self, ptr):
        self.ptr = C.c_void_p(ptr)

    def to_bytes(self, curve=377):
        n = lib().zko_api_proof_bytes(self.ptr, curve, None, C.c_size_t(0))
        buf = C.create_string_buffer(n)
        lib().zko_api_proof_bytes(self.ptr, curve, buf, C.c_size_t(n))
        return buf.raw

    def poly(self, name):
        i = POLY_NAMES.index(name)
        n = lib().zko_api_proof_poly_len(self.ptr, i)
        buf = C.create_string_buffer(32 * max(n, 1))
        lib().zko_api_proof_poly(self.ptr, i, buf)
        return buf.raw[:32 * n]

    def scalars(self):
        buf = C.create_string_buffer(32 * 11)
        lib().zko_api_proof_scalars(self.ptr, buf)
        v = fr_unpack(buf.raw)
        return dict(zip(["alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma", "opening_challenge", "g_1", "g_2", "t", "z_b"], v))

    def __del__(self):
        try:
            if self.ptr:
                lib().zko_proof_free(self.ptr)
                self.ptr = None
        except Exception:
            pass
