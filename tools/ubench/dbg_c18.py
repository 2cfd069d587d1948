import sys, time, ctypes as C; sys.path.insert(0,'.')
import numpy as np
from aes_zero_knowledge_proof_circuit_amd import api
from oracle import zko
def rand_fr_mont(n, p, seed):
    rs = np.random.RandomState(seed); out = bytearray()
    for _ in range(n): out += (int.from_bytes(rs.bytes(32), "little") % p).to_bytes(32, "little")
    return bytes(out)
n = 1<<12
sc0 = rand_fr_mont(n, zko.R377, 1)
out = C.create_string_buffer(96*n); zko.lib().zko_api_fixed_base(377, sc0, C.c_size_t(n), out); bases = out.raw
scalars = rand_fr_mont(n, zko.R377, 2)
ref = C.create_string_buffer(96); zko.lib().zko_api_msm(377, bases, scalars, C.c_size_t(n), ref)
for c in (16,17,18,19,20):
    t=time.time(); got, inf = api.msm_table(377, bases, scalars, c); dt=time.time()-t
    print(c, got==ref.raw, "%.2fs"%dt, flush=True)
