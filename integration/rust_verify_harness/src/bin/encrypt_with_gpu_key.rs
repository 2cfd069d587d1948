//! usage: encrypt_with_gpu_key <pk_ark.bin> <vk_ark.bin>
//!
//! pk_ark.bin  zkaes_pk_serialize_ark_to_file_ex(.., uncompressed = 1) output for a 16-byte key: the serialize_uncompressed image of ark_marlin::IndexProverKey
//!             (96-byte G1 points, ~1.25 GB; made on an MI355X with `python tools/make_pk_image.py pk_ark.bin`, not committed).  It MUST be the uncompressed image:
//!             ark-serialize 0.3's `deserialize_unchecked` defaults to `deserialize_uncompressed`, and GroupAffine reads x || y (96 bytes) without any check there.
//!             (The compressed image -- zkaes_pk_serialize_ark_to_file -- goes with `ProvingKey::deserialize`: a square root + subgroup check per SRS power.)
//! vk_ark.bin  the matching verifying key (tests/golden/gpu_aes16_vk_ark.bin)
//!
//! Runs the REFERENCE's own CPU prover, zk_aes::encrypt (reference src/lib.rs:60-114), on a proving key that libzkaes synthesized on the GPU, and checks the proof
//! with zk_aes::verify_encryption: the GPU-made IndexProverKey image is then a drop-in for what synthesize_keys (src/lib.rs:138-174) returns.
//! Source only: never compiled in the build image (no Rust toolchain there).
use anyhow::{anyhow, Result};
use ark_serialize::CanonicalDeserialize;
use simpleworks::marlin::{ProvingKey, VerifyingKey};
use std::{env, fs, fs::File, io::BufReader};

fn main() -> Result<()> {
    let args: Vec<String> = env::args().collect();
    if args.len() < 3 {
        return Err(anyhow!("usage: {} <pk_ark.bin> <vk_ark.bin>", args[0]));
    }
    let proving_key = ProvingKey::deserialize_unchecked(BufReader::new(File::open(&args[1])?)).map_err(|e| anyhow!("proving key: {e:?}"))?;
    let verifying_key = VerifyingKey::deserialize(&fs::read(&args[2])?[..]).map_err(|e| anyhow!("verifying key: {e:?}"))?;
    // FIPS-197 Appendix B (reference tests/integration_tests.rs:313-337)
    let message: [u8; 16] = [0x32, 0x43, 0xf6, 0xa8, 0x88, 0x5a, 0x30, 0x8d, 0x31, 0x31, 0x98, 0xa2, 0xe0, 0x37, 0x07, 0x34];
    let secret_key: [u8; 16] = [0x2b, 0x7e, 0x15, 0x16, 0x28, 0xae, 0xd2, 0xa6, 0xab, 0xf7, 0x15, 0x88, 0x09, 0xcf, 0x4f, 0x3c];
    let ciphertext: [u8; 16] = [0x39, 0x25, 0x84, 0x1d, 0x02, 0xdc, 0x09, 0xfb, 0xdc, 0x11, 0x85, 0x97, 0x19, 0x6a, 0x0b, 0x32];
    let proof = zk_aes::encrypt(&message, &secret_key, proving_key)?;
    let accepted = zk_aes::verify_encryption(verifying_key, &proof, &ciphertext)?;
    println!("reference encrypt() on the GPU-made proving key -> verify_encryption {accepted}");
    if !accepted {
        return Err(anyhow!("proof made with the GPU-synthesized key was rejected"));
    }
    Ok(())
}
