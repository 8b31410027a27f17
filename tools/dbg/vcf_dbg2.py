"""debug aid: every differing cell of one VCF text case, three runs"""
import os, sys, gzip, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import duckhts_amd, orc
import vcf_text_cases as V
name = sys.argv[1]
data = dict(V.all_cases())[name]
exp = orc.bcf_read(data)
raw = gzip.GzipFile(fileobj=io.BytesIO(data)).read() if data[:2] == b"\x1f\x8b" else data
lines = [l for l in raw.split(b"\n") if l and not l.startswith(b"#")]
for rep in range(3):
    got = duckhts_amd.read_bcf(data)
    tot = 0
    for ca, cb in zip(exp["cols"], got["cols"]):
        for k in ("valid", "fixed", "llen"):
            if k in ca and len(ca[k]) == len(cb[k]) and not np.array_equal(ca[k], cb[k]):
                bad = np.nonzero(ca[k] != cb[k])[0]
                tot += len(bad)
                print("rep", rep, ca["name"], k, "rows", bad[:12], "exp", ca[k][bad[:4]], "got", cb[k][bad[:4]])
    print("rep", rep, "differing cells", tot, flush=True)
r = int(sys.argv[2]) if len(sys.argv) > 2 else 257
print(lines[r])
