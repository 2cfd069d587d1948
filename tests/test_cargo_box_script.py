"""Dry-run tests of integration/check_on_cargo_box.sh and tools/circuit_step_counts.py: the one run on a box with cargo must be decisive, so the parts
that do not need cargo (log parsing, pass splitting, the step-by-step diff, the summary / exit code) are exercised here against synthetic
`RUST_LOG=debug` logs in env_logger's format -- one that equals the restated circuit, and one where a single gadget (ShiftRows allocating fresh
witness bytes, a suspect named in DESIGN.md section 2a) makes the counters diverge."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import circuit_step_counts as sc  # noqa: E402
import circuit_variants as cv  # noqa: E402


def fake_log(steps, with_keygen_pass=True):
    """what env_logger prints for src/main.rs-like runs: synthesize_keys' pass (no 'After allocating' lines, src/lib.rs:144-171), then encrypt's"""
    def block(s):
        m, c, i, w, z = s
        pre = "[2023-03-01T12:00:00Z DEBUG zk_aes::helpers] "
        return "\n".join([pre + "CONSTRAINT SYSTEM STATUS: " + m, pre + "Number of constraints: %d" % c, pre + "Number of variables: %d" % i,
                          pre + "Number of witnesses: %d" % w, pre + "Number of non-zero: %d" % z])
    out = ["[2023-03-01T12:00:00Z INFO  something] unrelated line"]
    if with_keygen_pass:
        out += [block(s) for s in steps if not s[0].startswith("After allocating") and s[0] != "Before generating the proof"]
    out += [block(s) for s in steps]
    return "\n".join(out) + "\n"


def test_step_counts_reproduce_the_known_totals_and_the_committed_json():
    for nbytes, total in ((16, (185_040, 129, 184_784, 882_002)), (64, (629_856, 513, 628_832, 3_002_900))):
        st = sc.steps(nbytes)
        assert st[-1][1:] == total and st[-2][1:] == total
        assert [s[0] for s in st[:4]] == ["After allocating the message", "After allocating the secret key", "After generating the lookup table", "After deriving the round keys"]
        assert st[0][1:] == (8 * nbytes, 1, 8 * nbytes, 3 * 8 * nbytes)                  # booleanity rows only: (1 - b) * b = 0 has 2 + 1 + 0 non-zeros
        assert len(st) == 4 + (nbytes // 16) * (1 + 9 * 4 + 3) + 2
        committed = json.load(open(os.path.join(ROOT, "integration", "expected_step_counts_%d.json" % nbytes)))
        assert [tuple(s) for s in committed["steps"]] == st


def test_diff_confirms_an_identical_log_and_localises_a_diverging_gadget():
    st = sc.steps(16)
    lines, bad = sc.diff(st, sc.parse_log(fake_log(st)))
    assert bad == 0 and "CONFIRMED" in lines[-1] and lines[0].startswith("pass 0 (synthesize_keys)") and any(l.startswith("pass 1 (encrypt)") for l in lines)
    # a reference whose rotate_left allocates fresh witness bytes: first divergence = the first ShiftRows... of the KEY SCHEDULE (rotate_word), i.e. 'After deriving the round keys'
    other = sc.steps(16, cv.Variant(rot="wit"))
    lines, bad = sc.diff(st, sc.parse_log(fake_log(other)))
    assert bad > 0
    first = [l for l in lines if l.startswith("  step") and "FIRST DIVERGENCE" in l]
    assert len(first) == 2 and all("'After deriving the round keys'" in l for l in first)          # one per pass
    # a shift that allocates: diverges at the first MixColumns, not before
    other = sc.steps(16, cv.Variant(shift="wit"))
    lines, bad = sc.diff(st, sc.parse_log(fake_log(other, with_keygen_pass=False)))
    first = [l for l in lines if l.startswith("  step") and "FIRST DIVERGENCE" in l]
    assert bad > 0 and len(first) == 1 and "'After mixing columns in round 1'" in first[0]
    assert sc.diff(st, sc.parse_log("nothing here"))[1] == 1


def test_check_on_cargo_box_dry_run(tmp_path):
    script = os.path.join(ROOT, "integration", "check_on_cargo_box.sh")
    for nbytes in (16, 64):
        (tmp_path / ("steps_%d.log" % nbytes)).write_text(fake_log(sc.steps(nbytes)))
    env = dict(os.environ, ZKAES_CARGO_BOX_DRY_RUN="1", PYTHON=sys.executable)
    r = subprocess.run(["bash", script, str(tmp_path)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("CONFIRMED") == 2 and "headline `value` of bench.py applies" in r.stdout
    # the reference's literal-sized circuit (a fake with more rows) must make the script fail and say where
    (tmp_path / "steps_64.log").write_text(fake_log(sc.steps(64, cv.Variant(shift="wit_eq", rot="wit_eq"))))
    r = subprocess.run(["bash", script, str(tmp_path)], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "steps_64" in r.stdout and "differ" in r.stdout and "read `alt`" in r.stdout
    assert "FIRST DIVERGENCE" in (tmp_path / "steps_64.diff").read_text()
