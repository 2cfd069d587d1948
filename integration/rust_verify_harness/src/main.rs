//! usage: zkaes-verify-harness <vk_ark.bin> <proof.bin> <ciphertext.bin> [expect-reject]
//!
//! vk_ark.bin      zkaes_vk_serialize_ark() output  (ark-serialize IndexVerifierKey, 759 bytes for the AES keys)
//! proof.bin       zkaes_encrypt() output           (ark-serialize ark_marlin::Proof, 855 bytes)
//! ciphertext.bin  the AES-128-ECB ciphertext the proof is about
//!
//! Committed sample inputs: tests/golden/gpu_aes16_vk_ark.bin, tests/golden/gpu_aes16_proof.bin and the FIPS-197 ciphertext
//! 3925841d02dc09fbdc118597196a0b32 (tests/golden/reference_vectors.json).
use anyhow::{anyhow, Result};
use ark_serialize::CanonicalDeserialize;
use simpleworks::marlin::VerifyingKey;
use std::{env, fs};

fn main() -> Result<()> {
    let args: Vec<String> = env::args().collect();
    if args.len() < 4 {
        return Err(anyhow!("usage: {} <vk_ark.bin> <proof.bin> <ciphertext.bin> [expect-reject]", args[0]));
    }
    let vk_bytes = fs::read(&args[1])?;
    let proof_bytes = fs::read(&args[2])?;
    let ciphertext = fs::read(&args[3])?;
    let expect_reject = args.get(4).map(|s| s == "expect-reject").unwrap_or(false);

    let verifying_key = VerifyingKey::deserialize(&vk_bytes[..]).map_err(|e| anyhow!("verifying key: {e:?}"))?;
    let proof = zk_aes::deserialize_proof(proof_bytes)?; // re-export of simpleworks::marlin::serialization::deserialize_proof
    let accepted = zk_aes::verify_encryption(verifying_key, &proof, &ciphertext)?;
    println!("verify_encryption -> {accepted}");
    if accepted == expect_reject {
        return Err(anyhow!("unexpected verdict"));
    }
    Ok(())
}
