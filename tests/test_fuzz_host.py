"""Sanitizer + fuzz coverage of the product's HOST side that takes untrusted bytes (SURVEY.md section 5 "race detection / sanitizers").

csrc/circuit.cpp + csrc/marlin_codec.cpp + csrc/capi_host.cpp -- the verifier (src/lib.rs:116-136), the proof / verifying-key (de)serialisers behind the re-exported
deserialize_proof (src/lib.rs:52), the transcript and the BLS12-377 pairing -- are built a second time, without HIP, with -fsanitize=address,undefined, and driven by
tests/fuzz_host.cpp over mutations of the committed GPU-made fixtures:
  * >= 10^5 deterministic mutated inputs (8 seeds in parallel, ~1.5 min on 8 cores), weighted towards the parsers;
  * a short coverage-guided libFuzzer leg on the same target.
Any sanitizer report, any accepted forgery (a proof, key encoding or statement other than the golden one that verifies) and any non-canonical accepted encoding fails.
No GPU and no oracle involved.
"""
import math
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aes_zero_knowledge_proof_circuit_amd", "csrc")
OUT = os.path.join(CSRC, "build")
GOLD = os.path.join(ROOT, "tests", "golden")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SOURCES = [os.path.join(ROOT, "tests", "fuzz_host.cpp")] + [os.path.join(CSRC, f) for f in ("circuit.cpp", "marlin_codec.cpp", "capi_host.cpp")]
# -asan-globals=0: this toolchain's ASan trips over its own registration of merged string literals ("odr-violation: global '.str'") before main() runs; heap, stack
# and use-after-free checking -- what parsers of untrusted bytes need -- are unaffected
COMMON = ["-x", "c++", "-O2", "-g", "-std=c++17", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-mllvm", "-asan-globals=0", "-I", CSRC]
CASES = 100_000

pytestmark = pytest.mark.skipif(not (os.path.exists(CLANG) and os.path.exists(os.path.join(GOLD, "gpu_aes16_proof.bin"))), reason="needs the ROCm clang and the GPU-made fixtures")


def _build(name, extra):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, name)
    deps = SOURCES + [os.path.join(CSRC, h) for h in ("marlin.hpp", "marlin_host.hpp", "capi_common.hpp", "pairing.hpp", "transcript.hpp", "ff.cuh", "ec.cuh", "circuit.hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call([CLANG] + COMMON + extra + SOURCES + ["-o", exe])
    return exe


@pytest.fixture(scope="module")
def binaries():
    with ThreadPoolExecutor(max_workers=2) as ex:
        a = ex.submit(_build, "mutate_host", ["-DZKAES_FUZZ_STANDALONE", "-fsanitize=address,undefined"])
        b = ex.submit(_build, "fuzz_host", ["-fsanitize=fuzzer,address,undefined"])
        return a.result(), b.result()


def _env():
    return dict(os.environ, ZKAES_FUZZ_GOLDEN=GOLD, ASAN_OPTIONS="abort_on_error=1:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")


def test_hundred_thousand_mutated_inputs_under_asan_ubsan(binaries):
    exe, _ = binaries
    procs = max(1, min(8, os.cpu_count() or 1))
    per = math.ceil(CASES / procs)
    ps = [subprocess.Popen([exe, str(seed), str(per)], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for seed in range(1, procs + 1)]
    total = 0
    for p in ps:
        out, err = p.communicate(timeout=1500)
        assert p.returncode == 0, "sanitizer report or finding:\n" + (err or out)[-4000:]
        line = [ln for ln in out.splitlines() if ln.startswith("mutate_host:")][-1]
        total += int(line.split(" cases ok")[0].split()[-1])
    assert total >= CASES


def test_libfuzzer_coverage_guided_leg(binaries, tmp_path):
    _, exe = binaries
    corpus = tmp_path / "corpus"
    corpus.mkdir()
    proof = open(os.path.join(GOLD, "gpu_aes16_proof.bin"), "rb").read()
    ark = open(os.path.join(GOLD, "gpu_aes16_vk_ark.bin"), "rb").read()
    private = open(os.path.join(GOLD, "gpu_aes16_vk.bin"), "rb").read()
    ct = bytes([0x39, 0x25, 0x84, 0x1d, 0x02, 0xdc, 0x09, 0xfb, 0xdc, 0x11, 0x85, 0x97, 0x19, 0x6a, 0x0b, 0x32])
    for i, (sel, body) in enumerate([(0, proof), (240, ark), (250, private), (252, proof), (253, ark), (254, ct)]):     # selector bytes: tests/fuzz_host.cpp
        (corpus / ("seed%d" % i)).write_bytes(bytes([sel]) + body)
    r = subprocess.run([exe, "-max_total_time=20", "-seed=1", "-max_len=1024", "-print_final_stats=1", str(corpus)], env=_env(), cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stderr or r.stdout)[-4000:]
    assert not [f for f in os.listdir(tmp_path) if f.startswith(("crash-", "leak-", "timeout-", "oom-"))]
    runs = [int(ln.split()[-1]) for ln in r.stderr.splitlines() if ln.startswith("stat::number_of_executed_units")]
    assert runs and runs[0] > 50
