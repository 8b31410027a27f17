"""op_trace.py [segments] -- the read_bam operator (mini host, all 13 columns, 8 fill threads) on a generated file of `segments` x 4 M records
with DHTS_TRACE=1: stage timings of a query that reads the file (DHTS_FILE_CACHE=0) and of one that finds it resident."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from duckhts_amd import synth  # noqa: E402
import duckhts_amd  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path = os.path.join(bench.scratch_dir(reps * 420_000_000), "dhts_op_trace.bam")
bench.generate_file(path, synth, 42, 4_000_000, reps, min(os.cpu_count() or 1, 32))
host = os.path.join(ROOT, "tests", "minihost", "minihost")
try:
    for env_extra, tag in (({"DHTS_FILE_CACHE": "0"}, "file read every query"), ({}, "file resident after the first query")):
        env = dict(os.environ, DHTS_THREADS=os.environ.get("OP_THREADS", "16"), DHTS_TRACE="1", **env_extra)
        r = subprocess.run([host, duckhts_amd.LIB_PATH, "read_bam", path, "-t", os.environ.get("OP_THREADS", "16"), "-r", "3"], capture_output=True, text=True, env=env)
        print(f"==== {tag} ({os.path.getsize(path) / 1e9:.2f} GB)")
        print(r.stdout[-600:]); print(r.stderr[-2500:])
finally:
    os.remove(path)
