import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import duckhts_amd, orc, cases
d = cases.ALL_CASES['basic_small_blocks']()
z = orc.bgzf_inflate_all(d)
ctx = duckhts_amd.Context(0); ctx.open(d); nb = ctx.bgzf_index()
out, bst = ctx.bgzf_inflate(0, nb, len(z["data"]))
print(nb, ''.join('x' if b else '.' for b in bst))
L = ctx.L
for s in [0,1,2,31,32,33,34,35,63,64]:
    m = (C.c_uint32*4)()
    L.dhts_debug_meta(C.c_void_p(ctx.h), C.c_int64(s), m)
    print(s, 'ntok', m[0], 'nlit', m[1], 'outlen', m[2], 'status', C.c_int32(m[3]).value, 'expect ulen', z['ulen'][s], 'clen', z['clen'][s])
