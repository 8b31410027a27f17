"""debug aid: one VCF text case through the C ABI against the oracle"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import duckhts_amd
import orc
import vcf_text_cases as V
name = sys.argv[1] if len(sys.argv) > 1 else "numbers_plain"
data = dict(V.all_cases())[name]
exp = orc.bcf_read(data)
got = duckhts_amd.read_bcf(data)
print(name, "rows", got["n_rows"], exp["n_rows"], "diff", orc.bcf_cols_diff(exp, got), flush=True)
import numpy as np
for ca, cb in zip(exp["cols"], got["cols"]):
    for k in ("valid", "fixed", "llen", "cfixed", "sbytes"):
        if k in ca and not np.array_equal(ca[k], cb[k]):
            n = min(len(ca[k]), len(cb[k])); bad = np.nonzero(ca[k][:n] != cb[k][:n])[0]
            r = int(bad[0]) if len(bad) else n
            print("col", ca["name"], k, "first diff at", r, "exp", ca[k][max(0, r - 1):r + 3], "got", cb[k][max(0, r - 1):r + 3])
            if k in ("valid", "fixed", "llen"):
                import bamwriter, zlib
                raw = data
                if raw[:2] == b"\x1f\x8b":
                    import gzip, io
                    raw = gzip.GzipFile(fileobj=io.BytesIO(raw)).read()
                lines = [l for l in raw.split(b"\n") if l and not l.startswith(b"#")]
                print("line:", lines[r][:300])
            break
    else:
        continue
    break
