#!/usr/bin/env python3
"""Instruction counts of the shipped gfx950 code object, per kernel (no GPU needed).

    python tools/disasm_count.py [object] [kernel-substring ...]

Default object: csrc/build/kernels_msm.hip.o; default kernels: k_accumulate.  Prints VALU / v_mad_u64_u32 / MFMA / LDS / scratch counts
and the VGPR / scratch figures from the kernel descriptor notes -- the numbers DESIGN.md section 3 quotes for k_accumulate<EdwardsLaw>.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_object(obj, workdir):
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], cwd=workdir, capture_output=True, text=True)
    co = None
    for f in os.listdir(os.path.dirname(obj)):
        if f.startswith(os.path.basename(obj)) and "gfx950" in f:
            co = os.path.join(os.path.dirname(obj), f)
    if co is None:
        raise SystemExit("no gfx950 bundle in %s\n%s" % (obj, out.stderr))
    return co


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def main():
    args = sys.argv[1:]
    obj = os.path.join(ROOT, "aes_zero_knowledge_proof_circuit_amd", "csrc", "build", "kernels_msm.hip.o")
    if args and os.path.exists(args[0]):
        obj = os.path.abspath(args.pop(0))
    wanted = args or ["k_accumulate<"]
    with tempfile.TemporaryDirectory() as td:
        co = code_object(obj, td)
        try:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        finally:
            for f in os.listdir(os.path.dirname(obj)):
                if f.startswith(os.path.basename(obj) + "."):
                    os.remove(os.path.join(os.path.dirname(obj), f))
    funcs, cur = {}, None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur and "\t" in line:
            ins = line.strip().split()
            if ins:
                funcs[cur].append(ins[0])
    meta = {}
    for blk in re.split(r"\n\s+- ", notes):
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
            meta[name.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"), spill=g("vgpr_spill_count"))
    names = list(funcs)
    dem = dict(zip(names, demangle(names)))
    for n in names:
        d = dem[n]
        if not any(w in d for w in wanted) or not funcs[n]:
            continue
        ins = funcs[n]
        valu = [i for i in ins if i.startswith("v_") and not i.startswith("v_mfma")]
        row = dict(valu=len(valu), mad_u64_u32=ins.count("v_mad_u64_u32"), lshl_add_u64=ins.count("v_lshl_add_u64"), bitop3=sum(i.startswith("v_bitop3") for i in ins),
                   sad_u32=ins.count("v_sad_u32"), cndmask=sum(i.startswith("v_cndmask") for i in ins), mfma=sum(i.startswith("v_mfma") for i in ins),
                   lds=sum(i.startswith("ds_") for i in ins), scratch_ops=sum(i.startswith("scratch_") for i in ins), global_loads=sum(i.startswith("global_load") for i in ins),
                   s_nop=ins.count("s_nop"), total=len(ins))
        row.update(meta.get(n, {}))
        print(d[:150])
        print("   " + "  ".join("%s=%s" % kv for kv in row.items()))


if __name__ == "__main__":
    main()
