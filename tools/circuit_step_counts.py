#!/usr/bin/env python3
"""tools/circuit_step_counts.py -- the R1CS counters after EVERY step the reference logs, for a step-by-step diff against a real `cargo run`.

The reference prints (constraints, instance variables, witness variables, A+B+C non-zeros) through `debug_constraint_system_status`
(/root/reference/src/helpers/mod.rs:66-82) after every step of `encrypt` (/root/reference/src/lib.rs:77,89,110) and of
`encrypt_and_generate_constraints` (src/lib.rs:183,189,197-270,287) when run with RUST_LOG=debug.  This repository restates the gadget layer from the
published crates and gets 629,856 / 3,002,900 at 64 bytes where the reference's own SRS literal (src/lib.rs:141) says 866,944 / 4,062,064; no variant of
the source-less simpleworks calls reproduces the literal (tools/circuit_variants.py).  One run on a box with cargo settles it:
this tool emits OUR counts per step (same symbolic executor as tools/circuit_variants.py, base variant = oracle/zko_r1cs.c = csrc/circuit.cpp) and
diffs them against the log of that run, so the first diverging step names the gadget.

    python tools/circuit_step_counts.py --emit 16 > integration/expected_step_counts_16.json      # one block (what src/main.rs proves)
    python tools/circuit_step_counts.py --emit 64 > integration/expected_step_counts_64.json      # the size of the SRS literal
    RUST_LOG=debug cargo run --release 2> run.log ; python tools/circuit_step_counts.py --diff run.log --bytes 16
(integration/check_on_cargo_box.sh does all of it.)
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import circuit_variants as cv  # noqa: E402


def steps(nbytes, V=None):
    """[(message, constraints, instance, witness, nnz)] in the order `encrypt` logs them (src/lib.rs:60-114 + 176-293)"""
    if nbytes % 16:
        raise ValueError("message length must be a multiple of 16")
    V = V or cv.Variant()
    nblocks = nbytes // 16
    cs = cv.CS()
    out = []

    def snap(msg):
        out.append((msg, cs.ncons, cs.ninst, cs.nwit, sum(cs.nnz)))
    msg = [cv.u8_alloc(cs) for _ in range(16 * nblocks)]
    snap("After allocating the message")                                            # src/lib.rs:77
    key = [cv.u8_alloc(cs) for _ in range(16)]
    snap("After allocating the secret key")                                         # src/lib.rs:89
    snap("After generating the lookup table")                                       # src/lib.rs:183 (256 constants: no variables)
    rk = cv.derive_keys(cs, V, key)
    snap("After deriving the round keys")                                           # src/lib.rs:189
    ct = []
    for blk in range(nblocks):
        st = [cv.u8_xor(cs, a, b) for a, b in zip(msg[16 * blk:16 * blk + 16], key)]
        snap("After adding round key in round 0")                                   # src/lib.rs:197
        for rnd in range(1, 10):
            st = [cv.substitute_byte(cs, V, b) for b in st]
            snap("After substituting bytes in round %d" % rnd)                      # src/lib.rs:208
            st = cv.shift_rows(cs, V, st)
            snap("After shifting rows in round %d" % rnd)                           # src/lib.rs:216
            st = cv.mix_columns(cs, V, st)
            snap("After mixing columns in round %d" % rnd)                          # src/lib.rs:224
            st = [cv.u8_xor(cs, a, b) for a, b in zip(st, rk[rnd])]
            snap("After adding round key in round %d" % rnd)                        # src/lib.rs:235
        st = [cv.substitute_byte(cs, V, b) for b in st]
        snap("After substituting bytes in round 10")                                # src/lib.rs:248
        st = cv.shift_rows(cs, V, st)
        snap("After shifting rows in round 10")                                     # src/lib.rs:256
        st = [cv.u8_xor(cs, a, b) for a, b in zip(st, rk[10])]
        snap("After adding round key in round 10")                                  # src/lib.rs:267
        ct += st
    for by in ct:
        pub = cv.u8_alloc(cs, inp=True)
        cv.enforce_equal_u8(cs, pub, by)
    snap("After enforcing that the obtained ciphertext is equal to the given one")  # src/lib.rs:287
    snap("Before generating the proof")                                             # src/lib.rs:110
    return out


KEYS = ("constraints", "instance", "witness", "nnz")
PATTERNS = (("message", re.compile(r"CONSTRAINT SYSTEM STATUS: (.*?)\s*$")), ("constraints", re.compile(r"Number of constraints: (\d+)")),
            ("instance", re.compile(r"Number of variables: (\d+)")), ("witness", re.compile(r"Number of witnesses: (\d+)")), ("nnz", re.compile(r"Number of non-zero: (\d+)")))


def parse_log(text):
    """entries of a RUST_LOG=debug run, in order: [{"message", "constraints", "instance", "witness", "nnz"}]"""
    entries, cur = [], None
    for line in text.splitlines():
        for name, rx in PATTERNS:
            m = rx.search(line)
            if not m:
                continue
            if name == "message":
                cur = {"message": m.group(1)}
                entries.append(cur)
            elif cur is not None and name not in cur:
                cur[name] = int(m.group(1))
            break
    return [e for e in entries if all(k in e for k in KEYS)]


def split_passes(entries):
    """`synthesize_keys` runs the circuit once on zeros (src/lib.rs:144-171: no 'After allocating' lines), `encrypt` runs it again: cut at every
    'After generating the lookup table' and attach a directly preceding 'After allocating ...' pair to the pass it opens"""
    passes = []
    for i, e in enumerate(entries):
        if e["message"] == "After generating the lookup table":
            start = i
            while start > 0 and entries[start - 1]["message"].startswith("After allocating") and (not passes or start - 1 > passes[-1][1]):
                start -= 1
            passes.append([start, i])
    out = []
    for n, (start, _) in enumerate(passes):
        end = passes[n + 1][0] if n + 1 < len(passes) else len(entries)
        out.append(entries[start:end])
    return out


def diff(expected, log_entries):
    """compare every pass of the log with the expected sequence; returns (report lines, number of differing steps)"""
    exp = [dict(zip(("message",) + KEYS, e)) for e in expected]
    lines, bad = [], 0
    passes = split_passes(log_entries)
    if not passes:
        return ["no 'CONSTRAINT SYSTEM STATUS' entries found: was the run started with RUST_LOG=debug?"], 1
    for pi, p in enumerate(passes):
        has_alloc = p[0]["message"].startswith("After allocating")
        want = exp if has_alloc else [e for e in exp if not e["message"].startswith("After allocating") and e["message"] != "Before generating the proof"]
        lines.append("pass %d (%s): %d logged steps, %d expected" % (pi, "encrypt" if has_alloc else "synthesize_keys", len(p), len(want)))
        prev_g = prev_w = dict.fromkeys(KEYS, 0)
        first = True
        for k in range(min(len(p), len(want))):
            g, w = p[k], want[k]
            if g["message"] != w["message"]:
                lines.append("  step %d: message mismatch: logged %r, expected %r -- the step sequences differ, stopping this pass" % (k, g["message"], w["message"]))
                bad += 1
                break
            if any(g[x] != w[x] for x in KEYS):
                bad += 1
                dg = {x: g[x] - prev_g[x] for x in KEYS}
                dw = {x: w[x] - prev_w[x] for x in KEYS}
                tag = "FIRST DIVERGENCE" if first else "differs"
                first = False
                lines.append("  step %d %-70s %s: logged %s, expected %s; this step added %s (reference) vs %s (restatement)" % (
                    k, repr(g["message"]), tag, [g[x] for x in KEYS], [w[x] for x in KEYS], [dg[x] for x in KEYS], [dw[x] for x in KEYS]))
            prev_g, prev_w = g, w
        if len(p) != len(want):
            bad += 1
            lines.append("  step count differs (%d logged, %d expected)" % (len(p), len(want)))
    lines.append("RESULT: %s" % ("every logged step equals the restated circuit: gadget-layer counts CONFIRMED" if bad == 0 else "%d step(s) differ -- see FIRST DIVERGENCE for the gadget to fix" % bad))
    return lines, bad


def main(argv):
    if len(argv) >= 2 and argv[0] == "--emit":
        nbytes = int(argv[1])
        st = steps(nbytes)
        json.dump({"bytes": nbytes, "fields": ["message"] + list(KEYS), "source": "tools/circuit_step_counts.py (base variant of tools/circuit_variants.py = oracle/zko_r1cs.c = csrc/circuit.cpp)",
                   "steps": [list(s) for s in st]}, sys.stdout, indent=0)
        sys.stdout.write("\n")
        return 0
    if len(argv) >= 2 and argv[0] == "--diff":
        nbytes = int(argv[argv.index("--bytes") + 1]) if "--bytes" in argv else 16
        if "--expected" in argv:
            expected = [tuple(s) for s in json.load(open(argv[argv.index("--expected") + 1]))["steps"]]
        else:
            expected = steps(nbytes)
        lines, bad = diff(expected, parse_log(open(argv[1], errors="replace").read()))
        print("\n".join(lines))
        return 1 if bad else 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
