#!/usr/bin/env bash
# integration/check_on_cargo_box.sh -- everything this repository could not check in its build image (no Rust toolchain, no network), in ONE run on
# any box with cargo + network + python3.  No GPU needed: it checks the restated layers against the REAL reference, using artefacts an MI355X made.
#
#   integration/check_on_cargo_box.sh [WORKDIR]          (default WORKDIR = ./cargo_box_out)
#
# 1. R1CS size, step by step (un-caps SURVEY.md 8c / DESIGN.md 2a): clones lambdaclass/AES_zero_knowledge_proof_circuit, adds ONE example file
#    (integration/cargo_box/zkaes_steps.rs; the crate's sources stay untouched), runs it under RUST_LOG=debug for 16 and 64 bytes and diffs the
#    constraints / instance / witness / non-zero counters `debug_constraint_system_status` prints after every step (reference src/lib.rs:77-287,
#    src/helpers/mod.rs:66-82) against integration/expected_step_counts_{16,64}.json (= tools/circuit_step_counts.py = oracle/zko_r1cs.c =
#    csrc/circuit.cpp).  The first diverging step names the gadget behind the 629,856-vs-866,944 gap; "CONFIRMED" pins the gadget layer.
# 2. Proof + key formats and the whole Marlin restatement: feeds the committed GPU-made verifying key + proof (tests/golden/gpu_aes16_*.bin) to the
#    unmodified zk_aes::verify_encryption through integration/rust_verify_harness -- must accept the FIPS-197 ciphertext and reject a flipped one.
#    With ZKAES_PK_IMAGE=<file made by tools/make_pk_image.py on an MI355X> also the third artefact of SURVEY.md 8 f1: the reference's OWN CPU prover, zk_aes::encrypt
#    (src/lib.rs:60-114), runs on the GPU-synthesized IndexProverKey image and its proof must verify (rust_verify_harness/src/bin/encrypt_with_gpu_key.rs).
# 3. The real CPU numbers for BASELINE.md: `cargo criterion --bench benchmark` (reference Makefile:9-10; falls back to `cargo bench`).
# Every step writes its log under WORKDIR and the script ends with a summary; a failing step does not stop the later ones.
# ZKAES_CARGO_BOX_DRY_RUN=1 skips everything that needs cargo / git and only runs the log differ on WORKDIR/steps_{16,64}.log (used by
# tests/test_cargo_box_script.py with synthetic logs).
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(dirname "$HERE")"
WORK="${1:-$PWD/cargo_box_out}"
REF_URL="${ZKAES_REFERENCE_URL:-https://github.com/lambdaclass/AES_zero_knowledge_proof_circuit}"
DRY="${ZKAES_CARGO_BOX_DRY_RUN:-0}"
PY="${PYTHON:-python3}"
mkdir -p "$WORK"
declare -A RESULT

step_counts() {          # $1 = message bytes
    local n="$1" log="$WORK/steps_$1.log"
    if [ "$DRY" != "1" ]; then
        ( cd "$WORK/reference" && RUST_LOG=debug cargo run --release --locked --example zkaes_steps -- "$n" ) > "$WORK/steps_$n.stdout" 2> "$log"
        if [ $? -ne 0 ]; then RESULT["steps_$n"]="cargo run failed (see $log)"; return; fi
    fi
    if [ ! -s "$log" ]; then RESULT["steps_$n"]="no log at $log"; return; fi
    "$PY" "$REPO/tools/circuit_step_counts.py" --diff "$log" --bytes "$n" --expected "$HERE/expected_step_counts_$n.json" > "$WORK/steps_$n.diff"
    local rc=$?
    RESULT["steps_$n"]="$(tail -1 "$WORK/steps_$n.diff") (rc $rc, details: $WORK/steps_$n.diff)"
}

if [ "$DRY" != "1" ]; then
    command -v cargo > /dev/null || { echo "cargo not found: this script needs a Rust toolchain"; exit 2; }
    if [ ! -d "$WORK/reference/.git" ]; then git clone --depth 1 "$REF_URL" "$WORK/reference" || { echo "cannot clone $REF_URL"; exit 2; }; fi
    mkdir -p "$WORK/reference/examples"
    cp "$HERE/cargo_box/zkaes_steps.rs" "$WORK/reference/examples/zkaes_steps.rs"
fi
step_counts 16
step_counts 64

if [ "$DRY" != "1" ]; then
    printf '\x39\x25\x84\x1d\x02\xdc\x09\xfb\xdc\x11\x85\x97\x19\x6a\x0b\x32' > "$WORK/ct_ok.bin"       # FIPS-197 App. B ciphertext (tests/integration_tests.rs:52-64)
    printf '\x39\x25\x84\x1d\x02\xdc\x09\xfb\xdc\x11\x85\x97\x19\x6a\x0b\x33' > "$WORK/ct_bad.bin"
    H="$HERE/rust_verify_harness/Cargo.toml"
    G="$REPO/tests/golden"
    cargo run --release --manifest-path "$H" --bin zkaes-verify-harness -- "$G/gpu_aes16_vk_ark.bin" "$G/gpu_aes16_proof.bin" "$WORK/ct_ok.bin" > "$WORK/verify_accept.log" 2>&1
    RESULT[verify_accept]="rc $? (expected 0: the unmodified verifier accepts the GPU-made proof; $WORK/verify_accept.log)"
    cargo run --release --manifest-path "$H" --bin zkaes-verify-harness -- "$G/gpu_aes16_vk_ark.bin" "$G/gpu_aes16_proof.bin" "$WORK/ct_bad.bin" expect-reject > "$WORK/verify_reject.log" 2>&1
    RESULT[verify_reject]="rc $? (expected 0: a wrong ciphertext is Ok(false); $WORK/verify_reject.log)"
    if [ -n "${ZKAES_PK_IMAGE:-}" ]; then
        cargo run --release --manifest-path "$H" --bin encrypt_with_gpu_key -- "$ZKAES_PK_IMAGE" "${ZKAES_PK_IMAGE}.vk" > "$WORK/encrypt_with_gpu_key.log" 2>&1
        RESULT[encrypt_with_gpu_key]="rc $? (expected 0: the reference's CPU encrypt() proves with the GPU-made proving key; $WORK/encrypt_with_gpu_key.log)"
    fi
    ( cd "$WORK/reference" && { cargo criterion --bench benchmark || cargo bench --bench benchmark; } ) > "$WORK/criterion.log" 2>&1
    RESULT[criterion]="rc $? (Encryption/{16,32,64}_message_encryption times for BASELINE.md section 1: $WORK/criterion.log; cores: $(nproc))"
fi

echo "==== check_on_cargo_box summary ===="
for k in steps_16 steps_64 verify_accept verify_reject encrypt_with_gpu_key criterion; do
    [ -n "${RESULT[$k]+x}" ] && echo "$k: ${RESULT[$k]}"
done
fail=0
for k in steps_16 steps_64; do case "${RESULT[$k]:-missing}" in *CONFIRMED*) ;; *) fail=1 ;; esac; done
# which figure of the bench line the real R1CS density selects (bench.py: `value` = 6 blocks per chunk-proof under the restated gadget layer -- 629,856 rows at 64 bytes;
# `alt` = 4 blocks per chunk-proof, what fits |H| = 2^20 at the density of the reference's SRS literal -- 866,944 rows)
case "${RESULT[steps_64]:-missing}" in
    *CONFIRMED*) echo "bench figure: the 64-byte circuit has the restated density (6 blocks fit a 2^20 domain) -> the headline \`value\` of bench.py applies" ;;
    *) echo "bench figure: the 64-byte circuit does NOT have the restated density -> read \`alt\` (4 blocks per chunk-proof) in the bench line, not \`value\`, until csrc/circuit.cpp follows the diverging gadget" ;;
esac
if [ "$DRY" != "1" ]; then for k in verify_accept verify_reject; do case "${RESULT[$k]:-missing}" in "rc 0"*) ;; *) fail=1 ;; esac; done; fi
exit $fail
