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
