"""read_bcf through the table function on the 1 M-record synthetic BCF (the BCF leg of tools/bench_surface.py alone)"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_surface as S
from duckhts_amd import synth
d = tempfile.mkdtemp(dir="/tmp"); bcf = os.path.join(d, "s.bcf")
synth.bcf_segment(1_000_000, seed=43)[0].tofile(bcf)
size = os.path.getsize(bcf)
S.run("read_bcf", bcf, proj=[0], repeat=1)
for name, proj in (("count(*) (CHROM)", [0]), ("core 7 columns", list(range(7))), ("all 111 columns", None)):
    for thr in (1, 8):
        for cache, stream in (("0", "1"), ("0", "0"), ("1", "1")):
            rows, dt, runs = S.run("read_bcf", bcf, proj=proj, threads=thr, env={"DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": cache, "DHTS_STREAM": stream})
            warm = sorted(runs[1:])[len(runs[1:]) // 2]
            print(json.dumps({"operator": "read_bcf through the table function", "projection": name, "DHTS_THREADS": thr, "file_MB": round(size / 1e6, 1),
                              "mode": "resident" if cache == "1" else ("file read every query, staging overlapped" if stream == "1" else "file read every query, staged first (DHTS_STREAM=0)"),
                              "warm_query_s": round(warm, 4), "records_per_s": round(rows / warm, 1)}), flush=True)
os.remove(bcf)
