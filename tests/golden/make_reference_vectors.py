#!/usr/bin/env python3
"""Extract the known-answer DATA (inputs / expected outputs) held by the reference's own tests into
tests/golden/reference_vectors.json.  Run in the build container (needs /root/reference):

    python tests/golden/make_reference_vectors.py

Sources (all FIPS-197 App. A/B values):
  tests/integration_tests.rs:52-64   plaintext, key, ciphertext
  tests/integration_tests.rs:67-276  per-round states (start, after SubBytes, after ShiftRows, after MixColumns)
  tests/integration_tests.rs:332-335, 341-370  wrong ciphertexts (negative cases), 64-byte case
  src/aes_circuit.rs:433-694         the 256 lookup-table constants
  src/aes_circuit.rs:722-725,739-750,811-814,839-845  gate KATs
  src/ops.rs:41-47                   ChaCha20 seed of the toy-op tests
Only numbers are copied, no source text.
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def hexbytes(text):
    return [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})\b", text)]


def block_after(src, marker, start=0):
    """bytes of the bracketed literal that follows `marker`"""
    i = src.index(marker, start)
    j = src.index("[", src.index("=", i))
    depth, k = 0, j
    while True:
        if src[k] == "[":
            depth += 1
        elif src[k] == "]":
            depth -= 1
            if depth == 0:
                break
        k += 1
    return hexbytes(src[j:k + 1]), k


def main():
    it = open(os.path.join(REF, "tests/integration_tests.rs")).read()
    ac = open(os.path.join(REF, "src/aes_circuit.rs")).read()
    ops = open(os.path.join(REF, "src/ops.rs")).read()
    v = {}
    pt, p = block_after(it, "let plaintext: [u8; 16]")
    key, p = block_after(it, "let key: [u8; 16]", p)
    ct, p = block_after(it, "let expected_output", p)
    v["plaintext"], v["key"], v["ciphertext"] = pt, key, ct
    for name in ("expected_start_of_round", "expected_after_substituting_bytes", "expected_after_shift_rows", "expected_after_mix_columns"):
        data, p = block_after(it, "let " + name, p)
        assert len(data) % 16 == 0
        v[name] = [data[i:i + 16] for i in range(0, len(data), 16)]
    t16 = it.index("fn test_encrypt_a_16_bytes_plaintext")
    w16, _ = block_after(it, "let wrong_ciphertext", t16)
    v["wrong_ciphertext_16"] = w16
    t64 = it.index("fn test_one_round_aes_encryption_of_a_64_bytes_plaintext")
    pt64, q = block_after(it, "let plaintext: [u8; 64]", t64)
    ct64, q = block_after(it, "let expected_ciphertext", q)
    w64, q = block_after(it, "let wrong_ciphertext", q)
    v["plaintext_64"], v["ciphertext_64"], v["wrong_ciphertext_64"] = pt64, ct64, w64
    # lookup table constants: UInt8::new_constant(cs, 0x..)
    lt = ac[ac.index("pub fn lookup_table"):ac.index("#[cfg(test)]")]
    table = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})_u8|0x([0-9a-fA-F]{2})\b", lt) for x in x if x]
    assert len(table) == 256, len(table)
    v["lookup_table"] = table
    tests = ac[ac.index("#[cfg(test)]"):]
    ark, q = block_after(tests, "let expected_primitive_result")
    v["add_round_key_expected"] = ark
    mix_in, q = block_after(tests, "let value_to_mix = UInt8Gadget::new_witness_vec")
    v["mix_columns_input"] = mix_in[:16]
    mix_out, q = block_after(tests, "let expected_primitive_mixed_value")
    v["mix_columns_expected"] = mix_out
    sub_in, q2 = block_after(tests, "let value_to_substitute = UInt8Gadget::new_witness_vec")
    v["sub_bytes_input"] = sub_in[:16]
    sub_out, q2 = block_after(tests, "let expected_primitive_substituted_value")
    v["sub_bytes_expected"] = sub_out
    ke = tests[tests.index("fn key_expansion_circuit"):]
    rk10 = hexbytes(ke[ke.index("result.get(10)"):])[:16]
    v["round_key_10"] = rk10
    seed = [int(x) for x in re.findall(r"\d+", ops[ops.index("let seed = ["):ops.index("];", ops.index("let seed = ["))])]
    assert len(seed) == 32
    v["ops_chacha20_seed"] = seed
    v["shift_rows_index_map"] = [0, 5, 10, 15, 4, 9, 14, 3, 8, 13, 2, 7, 12, 1, 6, 11]   # src/aes_circuit.rs:772-789
    v["srs_literals"] = [866944, 513, 4062064]                                          # src/lib.rs:141
    json.dump(v, open(OUT, "w"), indent=0, separators=(",", ":"))
    print("wrote", OUT, {k: (len(x) if isinstance(x, list) else x) for k, x in v.items()})


if __name__ == "__main__":
    main()
