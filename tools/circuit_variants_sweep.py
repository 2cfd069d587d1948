#!/usr/bin/env python3
"""tools/circuit_variants_sweep.py -- the wider sweep behind DESIGN.md section 2a (output: profiles/r02_circuit_variants_sweep.txt).

On top of tools/circuit_variants.py's shift / rotate variants it varies (i) Boolean::conditionally_select: A = as restated, B = general case with a
booleanity check on the result, C = constant folding only when both branches are constant, Cb = C + booleanity; (ii) MixColumns also in round 10;
(iii) the key schedule re-derived per block.  One line per variant: select shift rotate mc10 key_per_block constraints nnz (delta to the literal).

    python tools/circuit_variants_sweep.py > profiles/r02_circuit_variants_sweep.txt      # ~7 minutes
"""
import itertools
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import circuit_variants as cv
T = cv.TARGET
orig_select = cv.b_select
def make_select(mode):
    def sel(cs, cond, t, f):
        if mode == 'A': return orig_select(cs, cond, t, f)
        if cond == cv.TRUE: return t
        if cond == cv.FALSE: return f
        if cond[0] == 'not': return sel(cs, cv.b_not(cond), f, t)
        if mode == 'C':   # fold only when both branches constant
            if t[0]=='c' and f[0]=='c':
                return orig_select(cs, cond, t, f)
            return ('is', cv.sel_general(cs, cond, t, f))
        if mode == 'Cb':   # as C, with booleanity on the result
            if t[0]=='c' and f[0]=='c':
                return orig_select(cs, cond, t, f)
            r = cv.sel_general(cs, cond, t, f)
            a={}; cv.lc_add(a,1,cv.ONE); cv.lc_add(a,-1,r); cs.enforce(a,{r:1},{})
            return ('is', r)
        if mode == 'B':   # arms exist; general case with booleanity
            if f == cv.FALSE or t == cv.FALSE or t == cv.TRUE or f == cv.TRUE:
                return orig_select(cs, cond, t, f)
            r = cv.sel_general(cs, cond, t, f)
            a={}; cv.lc_add(a,1,cv.ONE); cv.lc_add(a,-1,r); cs.enforce(a,{r:1},{})
            return ('is', r)
    return sel
def synth(V, nblocks, mc10, kpb):
    cs = cv.CS()
    msg = [cv.u8_alloc(cs) for _ in range(16*nblocks)]
    key = [cv.u8_alloc(cs) for _ in range(16)]
    rk = None
    if not kpb: rk = cv.derive_keys(cs, V, key)
    ct = []
    for blk in range(nblocks):
        if kpb: rk = cv.derive_keys(cs, V, key)
        st = [cv.u8_xor(cs,a,b) for a,b in zip(msg[16*blk:16*blk+16], key)]
        for rnd in range(1,10):
            st = [cv.substitute_byte(cs,V,b) for b in st]
            st = cv.shift_rows(cs,V,st)
            st = cv.mix_columns(cs,V,st)
            st = [cv.u8_xor(cs,a,b) for a,b in zip(st, rk[rnd])]
        st = [cv.substitute_byte(cs,V,b) for b in st]
        st = cv.shift_rows(cs,V,st)
        if mc10: cv.mix_columns(cs,V,st)
        st = [cv.u8_xor(cs,a,b) for a,b in zip(st, rk[10])]
        ct += st
    for by in ct:
        pub = cv.u8_alloc(cs, inp=True)
        cv.enforce_equal_u8(cs, pub, by)
    return cs
res = []
for sel in ['A','B','C','Cb']:
    cv.b_select = make_select(sel)
    for s, r in itertools.product(['free','wit','fill_wit','wit_eq'], ['free','wit']):
        for mc10 in (0,1):
            for kpb in (0,1):
                V = cv.Variant(shift=s, rot=r)
                c1 = synth(V,1,mc10,kpb); c2 = synth(V,2,mc10,kpb)
                nc = c1.ncons + 3*(c2.ncons-c1.ncons); nz = sum(c1.nnz)+3*(sum(c2.nnz)-sum(c1.nnz))
                print(sel, s, r, mc10, kpb, nc, nz, nc-T[0], nz-T[2], "MATCH" if (nc,nz)==(T[0],T[2]) else "", flush=True)
