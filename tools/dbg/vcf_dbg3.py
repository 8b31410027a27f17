"""debug aid: the encoded record of one line of a VCF text case"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import duckhts_amd
import vcf_text_cases as V
name, row = sys.argv[1], int(sys.argv[2])
data = dict(V.all_cases())[name]
ctx = duckhts_amd.Context(0)
ctx.open(data); ctx.bgzf_index()
sc = duckhts_amd.BcfScan(ctx)
b = sc.next_batch(0)
L = duckhts_amd.lib()
L.dhts_debug_vcf_records.restype = C.c_int64
L.dhts_debug_vcf_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64]
buf = np.zeros(1 << 22, np.uint8); ro = np.zeros(int(b.n_rows) + 1, np.uint32)
n = L.dhts_debug_vcf_records(ctx.h, buf.ctypes.data, buf.nbytes, ro.ctypes.data, int(b.n_rows))
for r in (row, row + 1):
    o = int(ro[r]); ls, li = int(buf[o:o+4].view(np.uint32)[0]), int(buf[o+4:o+8].view(np.uint32)[0])
    print("row", r, "off", o, "l_shared", ls, "l_indiv", li, "next", int(ro[r + 1]) if r + 1 < len(ro) else None)
    print(" core", buf[o+8:o+32].tobytes().hex())
    print(" indiv", buf[o+8+ls:o+8+ls+li].tobytes().hex())
# the same records as a BCF file through the binary path
import gzip, io, struct
import bamwriter, orc
raw = gzip.GzipFile(fileobj=io.BytesIO(data)).read() if data[:2] == b"\x1f\x8b" else data
hdr = b"".join(l + b"\n" for l in raw.split(b"\n") if l.startswith(b"#")) + b"\0"
nr = int(b.n_rows)
end = int(ro[nr - 1]); end += 8 + int(buf[end:end+4].view(np.uint32)[0]) + int(buf[end+4:end+8].view(np.uint32)[0])
bcf = bamwriter.bgzf_file(b"BCF\x02\x02" + struct.pack("<I", len(hdr)) + hdr + buf[:end].tobytes())
e2 = orc.bcf_read(bcf); g2 = duckhts_amd.read_bcf(bcf)
print("as BCF: oracle rows", e2["n_rows"], "product rows", g2["n_rows"], "diff", orc.bcf_cols_diff(e2, g2))
e1 = orc.bcf_read(data)
print("oracle text vs oracle on the device's records:", orc.bcf_cols_diff(e1, e2))
