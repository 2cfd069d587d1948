//! Bindings + safe wrappers for libzkaes (include/zkaes.h): the MI355X implementation of zk_aes::{synthesize_keys, encrypt, verify_encryption}.
//!
//! In `zk-aes` the three functions of src/lib.rs (:60, :116, :138) become one-liners over `zkaes_sys::{synthesize_keys, encrypt,
//! verify_encryption}`; proofs cross the boundary as the ark-serialize bytes of `ark_marlin::Proof`, verifying keys (optionally) as the
//! ark-serialize bytes of `IndexVerifierKey` (`VerifyingKey::to_ark_bytes`), the proving key stays a device-resident handle.
use anyhow::{anyhow, Result};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_uint};
use std::sync::Arc;

#[repr(C)]
pub struct zkaes_pk { _private: [u8; 0] }
#[repr(C)]
pub struct zkaes_vk { _private: [u8; 0] }

extern "C" {
    fn zkaes_last_error() -> *const c_char;
    fn zkaes_bytes_free(p: *mut u8);
    fn zkaes_pk_free(pk: *mut zkaes_pk);
    fn zkaes_vk_free(vk: *mut zkaes_vk);
    fn zkaes_synthesize_keys(plaintext_length: usize, pk: *mut *mut zkaes_pk, vk: *mut *mut zkaes_vk) -> c_int;
    fn zkaes_synthesize_keys_ex2(circuit_kind: c_int, plaintext_length: usize, srs_num_constraints: usize, srs_num_variables: usize, srs_num_non_zero: usize, flags: c_uint,
                                 pk: *mut *mut zkaes_pk, vk: *mut *mut zkaes_vk) -> c_int;
    fn zkaes_encrypt(message: *const u8, message_len: usize, secret_key: *const u8, pk: *const zkaes_pk, proof: *mut *mut u8, proof_len: *mut usize) -> c_int;
    fn zkaes_verify_encryption(vk: *const zkaes_vk, proof: *const u8, proof_len: usize, ciphertext: *const u8, ciphertext_len: usize, accepted: *mut c_int) -> c_int;
    fn zkaes_encrypt_chunked(message: *const u8, message_len: usize, secret_key: *const u8, pk: *const zkaes_pk, proofs: *mut *mut u8, proofs_len: *mut usize,
                             proof_lens: *mut usize, n_chunks: usize) -> c_int;
    fn zkaes_encrypt_chunked_seeded_at(message: *const u8, message_len: usize, secret_key: *const u8, pk: *const zkaes_pk, zk_seed32: *const u8, first_proof_index: u64,
                                       proofs: *mut *mut u8, proofs_len: *mut usize, proof_lens: *mut usize, n_chunks: usize) -> c_int;
    fn zkaes_pk_serialize_ark_to_file_ex(pk: *const zkaes_pk, path: *const c_char, uncompressed: c_int, bytes_written: *mut u64) -> c_int;
    fn zkaes_pk_set_contexts(pk: *mut zkaes_pk, n: usize) -> c_int;
    fn zkaes_pk_get_contexts(pk: *const zkaes_pk, n: *mut usize) -> c_int;
    fn zkaes_srs_hold(hold: c_int) -> c_int;
    fn zkaes_set_default_contexts(n: usize) -> c_int;
    fn zkaes_pk_srs_info(pk: *const zkaes_pk, out: *mut u64, secs: *mut f64) -> c_int;
    fn zkaes_pk_tables_built(pk: *const zkaes_pk, built: *mut c_int, table_bytes: *mut u64) -> c_int;
    fn zkaes_vk_serialize_ark(vk: *const zkaes_vk, out: *mut *mut u8, out_len: *mut usize) -> c_int;
    fn zkaes_vk_deserialize_ark(bytes: *const u8, len: usize, vk: *mut *mut zkaes_vk) -> c_int;
}

fn last_error() -> anyhow::Error {
    // thread-local message of the failing call (mirrors the anyhow::Error the reference returns)
    let msg = unsafe { CStr::from_ptr(zkaes_last_error()) }.to_string_lossy().into_owned();
    anyhow!(msg)
}
fn take_bytes(p: *mut u8, n: usize) -> Vec<u8> {
    let v = unsafe { std::slice::from_raw_parts(p, n) }.to_vec();
    unsafe { zkaes_bytes_free(p) };
    v
}

struct PkHandle(*mut zkaes_pk);
struct VkHandle(*mut zkaes_vk);
// the handles are immutable after synthesis and libzkaes serialises access to its prover contexts internally
unsafe impl Send for PkHandle {}
unsafe impl Sync for PkHandle {}
unsafe impl Send for VkHandle {}
unsafe impl Sync for VkHandle {}
impl Drop for PkHandle { fn drop(&mut self) { unsafe { zkaes_pk_free(self.0) } } }
impl Drop for VkHandle { fn drop(&mut self) { unsafe { zkaes_vk_free(self.0) } } }

/// Keys are passed by value in the reference API and callers `.clone()` them per call (tests/integration_tests.rs:330): Clone = Arc clone.
#[derive(Clone)]
pub struct ProvingKey(Arc<PkHandle>);
#[derive(Clone)]
pub struct VerifyingKey(Arc<VkHandle>);

impl ProvingKey {
    /// Streams the ark-serialize image of the arkworks `IndexProverKey` this key corresponds to, to run the reference's CPU `encrypt()` (src/lib.rs:60) on a GPU-made key.
    /// `uncompressed = true` (1.25 GB for a 16-byte key, 96-byte points): read it back with `ProvingKey::deserialize_unchecked(BufReader::new(File::open(path)?))` --
    /// in ark-serialize 0.3 `deserialize_unchecked` reads the UNCOMPRESSED layout and does no per-point work.  `uncompressed = false` (0.65 GB, 48-byte points) is what
    /// `ProvingKey::serialize` writes: read it with `ProvingKey::deserialize` (a square root and a subgroup check per SRS point -- tens of minutes for a 64-byte key).
    pub fn serialize_ark_to_file(&self, path: &std::path::Path, uncompressed: bool) -> Result<u64> {
        let c = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|_| anyhow::anyhow!("path contains a NUL byte"))?;
        let mut n = 0u64;
        if unsafe { zkaes_pk_serialize_ark_to_file_ex((self.0).0, c.as_ptr(), uncompressed as c_int, &mut n) } != 0 { return Err(last_error()); }
        Ok(n)
    }
    /// Proofs in flight per `encrypt_chunked` call on this key (1..=64; 0 = the process default, 12).  Replaces the ZKAES_CONTEXTS environment round-trip.
    pub fn set_contexts(&self, n: usize) -> Result<()> {
        if unsafe { zkaes_pk_set_contexts((self.0).0, n) } != 0 { return Err(last_error()); }
        Ok(())
    }
    pub fn contexts(&self) -> Result<usize> {
        let mut n = 0usize;
        if unsafe { zkaes_pk_get_contexts((self.0).0, &mut n) } != 0 { return Err(last_error()); }
        Ok(n)
    }
    /// The universal SRS behind the key -- ONE per process, device and SRS literals, as `generate_universal_srs` at src/lib.rs:139-141 is one for every circuit size:
    /// (max_degree, points per copy, copies, device bytes, keys sharing it now, Lagrange-basis bytes)
    pub fn srs_info(&self) -> Result<[u64; 6]> {
        let mut out = [0u64; 6];
        if unsafe { zkaes_pk_srs_info((self.0).0, out.as_mut_ptr(), std::ptr::null_mut()) } != 0 { return Err(last_error()); }
        Ok(out)
    }
    /// (built, bytes): whether the key holds the fixed-base window tables of its SRS (skipped under KEY_NO_TABLES or when device memory is short)
    pub fn tables_built(&self) -> Result<(bool, u64)> {
        let (mut b, mut n) = (0 as c_int, 0u64);
        if unsafe { zkaes_pk_tables_built((self.0).0, &mut b, &mut n) } != 0 { return Err(last_error()); }
        Ok((b != 0, n))
    }
}

impl VerifyingKey {
    /// ark-serialize bytes of `ark_marlin::IndexVerifierKey` -- feed to `simpleworks::marlin::VerifyingKey::deserialize`
    pub fn to_ark_bytes(&self) -> Result<Vec<u8>> {
        let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
        if unsafe { zkaes_vk_serialize_ark((self.0).0, &mut p, &mut n) } != 0 { return Err(last_error()); }
        Ok(take_bytes(p, n))
    }
    pub fn from_ark_bytes(bytes: &[u8]) -> Result<Self> {
        let mut vk = std::ptr::null_mut();
        if unsafe { zkaes_vk_deserialize_ark(bytes.as_ptr(), bytes.len(), &mut vk) } != 0 { return Err(last_error()); }
        Ok(VerifyingKey(Arc::new(VkHandle(vk))))
    }
}

/// zk_aes::synthesize_keys (src/lib.rs:138)
pub fn synthesize_keys(plaintext_length: usize) -> Result<(ProvingKey, VerifyingKey)> {
    let (mut pk, mut vk) = (std::ptr::null_mut(), std::ptr::null_mut());
    if unsafe { zkaes_synthesize_keys(plaintext_length, &mut pk, &mut vk) } != 0 { return Err(last_error()); }
    Ok((ProvingKey(Arc::new(PkHandle(pk))), VerifyingKey(Arc::new(VkHandle(vk)))))
}

/// Key synthesis options (include/zkaes.h ZKAES_KEY_*)
pub const KEY_NO_TABLES: u32 = 1;   // this key does not use (or build) the window tables of the universal SRS (31.4 GB, shared by all keys); multi-proof calls run ~9 % slower

/// `synthesize_keys` with options: `flags` = KEY_NO_TABLES to keep the key small (the tables are also skipped automatically when the device is short of memory)
pub fn synthesize_keys_with(plaintext_length: usize, flags: u32) -> Result<(ProvingKey, VerifyingKey)> {
    let (mut pk, mut vk) = (std::ptr::null_mut(), std::ptr::null_mut());
    // circuit kind 0 = the AES circuit; the universal-SRS literals of src/lib.rs:141
    if unsafe { zkaes_synthesize_keys_ex2(0, plaintext_length, 866_944, 513, 4_062_064, flags as c_uint, &mut pk, &mut vk) } != 0 { return Err(last_error()); }
    Ok((ProvingKey(Arc::new(PkHandle(pk))), VerifyingKey(Arc::new(VkHandle(vk)))))
}

/// The process default of prover contexts per key (0 = back to ZKAES_CONTEXTS / 12); lower it before synthesizing keys for much larger chunk sizes (see include/zkaes.h).
pub fn set_default_contexts(n: usize) -> Result<()> {
    if unsafe { zkaes_set_default_contexts(n) } != 0 { return Err(last_error()); }
    Ok(())
}

/// Keep the universal SRS (31.4 GB of window tables for the reference's literals) resident after the last key over it is dropped; `false` releases it again.
/// For callers that create and drop keys in turn (one key per request size): without it every first key over an idle SRS rebuilds the tables (~1.5 s).
pub fn srs_hold(hold: bool) -> Result<()> {
    if unsafe { zkaes_srs_hold(hold as c_int) } != 0 { return Err(last_error()); }
    Ok(())
}

/// zk_aes::encrypt (src/lib.rs:60): returns the ark-serialize bytes of the MarlinProof (`deserialize_proof(bytes)` gives the arkworks type)
pub fn encrypt(message: &[u8], secret_key: &[u8; 16], proving_key: &ProvingKey) -> Result<Vec<u8>> {
    let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
    if unsafe { zkaes_encrypt(message.as_ptr(), message.len(), secret_key.as_ptr(), (proving_key.0).0, &mut p, &mut n) } != 0 { return Err(last_error()); }
    Ok(take_bytes(p, n))
}

/// zk_aes::verify_encryption (src/lib.rs:116): Ok(false) for a wrong ciphertext, Err only for malformed input
pub fn verify_encryption(verifying_key: &VerifyingKey, proof: &[u8], ciphertext: &[u8]) -> Result<bool> {
    let mut accepted: c_int = 0;
    if unsafe { zkaes_verify_encryption((verifying_key.0).0, proof.as_ptr(), proof.len(), ciphertext.as_ptr(), ciphertext.len(), &mut accepted) } != 0 { return Err(last_error()); }
    Ok(accepted != 0)
}

/// Prover randomness of a multi-proof call
pub enum ZkSeed<'a> {
    /// a fresh 32-byte seed from the operating system per call (the default of the C entry point): zero-knowledge across proofs
    Fresh,
    /// caller's seed; proof i of the call draws from StdRng(Blake2s(seed || (first_proof_index + i))) -- a job split over several calls / ranks under one
    /// seed passes the job-global index of each call's first proof
    Seeded { seed: &'a [u8; 32], first_proof_index: u64 },
    /// the reference's fixed `test_rng` stream in EVERY proof (src/lib.rs:65): byte-parity with the CPU path, NOT zero-knowledge across proofs -- tests only
    ReferenceParity,
}

/// Long ECB messages (not in the reference API: its SRS literal caps one proof at 96 bytes): ceil(len / chunk) independent chunk-proofs on the key of
/// `chunk` bytes, many in flight on the GPU (ECB blocks are independent, src/lib.rs:194).
pub fn encrypt_chunked(message: &[u8], secret_key: &[u8; 16], proving_key: &ProvingKey, chunk_len: usize, zk_seed: ZkSeed) -> Result<Vec<Vec<u8>>> {
    if chunk_len == 0 || message.len() % chunk_len != 0 { return Err(anyhow!("message length must be a multiple of the key's plaintext length")); }
    let n = message.len() / chunk_len;
    let (mut p, mut total) = (std::ptr::null_mut(), 0usize);
    let mut lens = vec![0usize; n.max(1)];
    let rc = match zk_seed {
        ZkSeed::Fresh => unsafe { zkaes_encrypt_chunked(message.as_ptr(), message.len(), secret_key.as_ptr(), (proving_key.0).0, &mut p, &mut total, lens.as_mut_ptr(), n) },
        ZkSeed::Seeded { seed, first_proof_index } => unsafe {
            zkaes_encrypt_chunked_seeded_at(message.as_ptr(), message.len(), secret_key.as_ptr(), (proving_key.0).0, seed.as_ptr(), first_proof_index, &mut p, &mut total, lens.as_mut_ptr(), n)
        },
        ZkSeed::ReferenceParity => unsafe {
            zkaes_encrypt_chunked_seeded_at(message.as_ptr(), message.len(), secret_key.as_ptr(), (proving_key.0).0, std::ptr::null(), 0, &mut p, &mut total, lens.as_mut_ptr(), n)
        },
    };
    if rc != 0 { return Err(last_error()); }
    let blob = take_bytes(p, total);
    let mut out = Vec::with_capacity(n);
    let mut off = 0;
    for l in lens.iter().take(n) { out.push(blob[off..off + l].to_vec()); off += l; }
    Ok(out)
}
