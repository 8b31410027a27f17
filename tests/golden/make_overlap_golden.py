"""Golden vector for the interval overlap join: 400 seeded intervals against the reads of range.bam, expected pairs produced by the
REFERENCE's own cgranges (oracle/_ref/libcgranges.so, compiled from /root/reference/third_party/cgranges by oracle/Makefile).
Run from the repo root (needs /root/reference): python tests/golden/make_overlap_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
import region_oracle as ro  # noqa: E402

t = orc.bam_read(open(os.path.join(ROOT, "tests", "golden", "range.bam"), "rb").read())
names = [bytes(x).decode() for x in t["ref_names"]]
rng = np.random.default_rng(3)
n = 400
tid = rng.integers(0, len(names), n).astype(np.int32)
beg = rng.integers(0, int(max(t["POS"])) + 200, n).astype(np.int64)
end = beg + rng.choice([0, 1, 5, 50, 500, 5000], n)
q = []
for i in range(t["n_rows"]):
    ti, st = int(t["tid"][i]), int(t["POS"][i]) - 1
    q.append((names[ti] if ti >= 0 else "*", st, ro.endpos(st, int(t["FLAG"][i]), t["CIGAR"][i])))
ref = ro.cgranges_overlap(os.path.join(ROOT, "oracle", "_ref", "libcgranges.so"), names, tid, beg, end, q)
json.dump({"tid": tid.tolist(), "beg": beg.tolist(), "end": end.tolist(), "queries": q, "overlaps": [x.tolist() for x in ref]},
          open(os.path.join(ROOT, "tests", "golden", "overlap_range_bam.json"), "w"))
print("rows", len(ref), "pairs", sum(len(x) for x in ref))
