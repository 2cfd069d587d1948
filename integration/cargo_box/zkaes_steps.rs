//! integration/cargo_box/zkaes_steps.rs -- copied by integration/check_on_cargo_box.sh into a clone of the reference as `examples/zkaes_steps.rs`.
//! Runs the UNMODIFIED reference (`synthesize_keys` + `encrypt`, src/lib.rs:138,60) on an N-byte message with the debug logger on, so that
//! `debug_constraint_system_status` (src/helpers/mod.rs:66-82) prints constraints / instance / witness / non-zeros after every step:
//!
//!     RUST_LOG=debug cargo run --release --locked --example zkaes_steps -- 64 2> steps_64.log
//!
//! Source only: there is no Rust toolchain in the image this repository is built in (never compiled there).
use anyhow::Result;

fn main() -> Result<()> {
    env_logger::init();
    let len: usize = std::env::args().nth(1).and_then(|s| s.parse().ok()).unwrap_or(64);
    let message: Vec<u8> = (0..len).map(|i| (i as u8).wrapping_mul(37).wrapping_add(11)).collect();
    let secret_key = [0x2bu8, 0x7e, 0x15, 0x16, 0x28, 0xae, 0xd2, 0xa6, 0xab, 0xf7, 0x15, 0x88, 0x09, 0xcf, 0x4f, 0x3c];
    let (proving_key, _verifying_key) = zk_aes::synthesize_keys(len)?;
    let _proof = zk_aes::encrypt(&message, &secret_key, proving_key)?;
    eprintln!("zkaes_steps: proved {len} bytes");
    Ok(())
}
