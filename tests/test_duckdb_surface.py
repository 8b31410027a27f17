"""The outer drop-in boundary: duckhts_init_c_api + read_bam bind/init/local_init/scan, driven by the mini DuckDB
host (tests/minihost).  Expectations restate /root/reference/test/sql/duckhts.test (read_bam section) and the
error strings of src/bam_reader.c; DataChunks are compared with the oracle chunk-for-chunk (2048-row chunks)."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import cases
import duckhts_amd
import orc
from conftest import GOLDEN, ROOT

HOST = os.path.join(ROOT, "tests", "minihost", "minihost")
TYPE_W = {1: 1, 7: 2, 4: 4, 10: 4, 5: 8, 11: 8}
VARCHAR = 17


def run_host(path, named=(), proj=None, fn="read_bam"):
    out = tempfile.NamedTemporaryFile(suffix=".chunks", delete=False).name
    cmd = [HOST, duckhts_amd.LIB_PATH, fn, path]
    for k, v in named:
        cmd += ["-n", f"{k}={v}"]
    if proj is not None:
        cmd += ["-p", ",".join(map(str, proj))]
    cmd += ["-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return r.returncode, r.stdout.strip(), out


def parse_chunks(fn):
    d = open(fn, "rb").read()
    p = 0
    (nc,) = struct.unpack_from("<I", d, p); p += 4
    schema = []
    for _ in range(nc):
        (t,) = struct.unpack_from("<I", d, p); p += 4
        schema.append((d[p:p + 64].split(b"\0")[0].decode(), t)); p += 64
    (nproj,) = struct.unpack_from("<I", d, p); p += 4
    chunks = []
    while p < len(d):
        (n,) = struct.unpack_from("<Q", d, p); p += 8
        cols = []
        for _ in range(nproj):
            (t,) = struct.unpack_from("<I", d, p); p += 4
            words = (n + 63) // 64
            val = np.frombuffer(d, np.uint64, words, p); p += 8 * words
            if t == VARCHAR:
                vals = []
                for _r in range(n):
                    (ln,) = struct.unpack_from("<I", d, p); p += 4
                    if ln == 0xFFFFFFFF:
                        vals.append(None)
                    else:
                        vals.append(d[p:p + ln]); p += ln
            else:
                w = TYPE_W[t]
                vals = np.frombuffer(d, {1: np.uint8, 2: np.uint16, 4: np.int32, 8: np.int64}[w], n, p).copy(); p += w * n
            cols.append((t, val, vals))
        chunks.append((n, cols))
    return schema, chunks


SCHEMA = [("QNAME", 17), ("FLAG", 7), ("RNAME", 17), ("POS", 5), ("MAPQ", 4), ("CIGAR", 17), ("RNEXT", 17), ("PNEXT", 5),
          ("TLEN", 5), ("SEQ", 17), ("QUAL", 17), ("READ_GROUP_ID", 17), ("SAMPLE_ID", 17)]


def test_entrypoint_and_errors_without_gpu_dependency():
    """bind-time error strings are the reference's (src/bam_reader.c:416,446); they do not need a device"""
    rc, out, _ = run_host("")
    assert rc == 3 and out == "ERROR bind: read_bam requires a file path"
    rc, out, _ = run_host("/no/such/file.bam")
    assert rc == 3 and out == "ERROR bind: Failed to open SAM/BAM/CRAM file: /no/such/file.bam"
    rc, out, _ = run_host("x.bam", named=[("bogus", "1")])
    assert rc == 3 and "unknown named parameter" in out          # only the reference's five named parameters are registered
    for k in ("region", "index_path", "reference", "standard_tags", "auxiliary_tags"):
        rc, out, _ = run_host("/no/such/file.bam", named=[(k, "x")])
        assert "unknown named parameter" not in out


def expect_chunks(exp, proj):
    names = [s[0] for s in SCHEMA]
    n = exp["n_rows"]
    for c0 in range(0, n, 2048):
        c1 = min(n, c0 + 2048)
        yield c1 - c0, [(names[j], exp[names[j]][c0:c1]) for j in proj]


def compare(path_bytes, proj=None, tmp_path=None):
    fn = os.path.join(str(tmp_path), "in.bam")
    open(fn, "wb").write(path_bytes)
    exp = orc.bam_read(path_bytes)
    rc, out, dump = run_host(fn, proj=proj)
    assert rc == 0, out
    schema, chunks = parse_chunks(dump)
    assert schema == SCHEMA
    proj = list(range(13)) if proj is None else proj
    want = list(expect_chunks(exp, proj))
    assert [c[0] for c in chunks] == [w[0] for w in want]              # full 2048-row chunks except the last
    for (n, cols), (_, wcols) in zip(chunks, want):
        for (t, val, vals), (name, w) in zip(cols, wcols):
            if t == VARCHAR:
                assert list(vals) == list(w), name
                bits = [(int(val[i >> 6]) >> (i & 63)) & 1 for i in range(n)]
                assert bits == [0 if x is None else 1 for x in w], name
            else:
                assert np.array_equal(vals.astype(np.int64), np.asarray(w).astype(np.int64)), name
    assert f"rows={exp['n_rows']} " in out and "max_threads=1" in out
    return exp


@pytest.mark.gpu
def test_read_bam_range_all_columns(tmp_path):
    exp = compare(open(os.path.join(GOLDEN, "range.bam"), "rb").read(), None, tmp_path)
    assert exp["n_rows"] == 112                                         # duckhts.test:129-131


@pytest.mark.gpu
@pytest.mark.parametrize("proj", [[0], [1, 3, 4], [12, 11, 0], [10, 9, 5, 2, 6], [3]])
def test_read_bam_projection_pushdown(tmp_path, proj):
    compare(cases.case_basic(n=3000, payload=20000), proj, tmp_path)    # > one chunk, NULLs in RG/SAMPLE


@pytest.mark.gpu
def test_read_bam_multichunk_and_error_stop(tmp_path):
    from duckhts_amd import synth
    arr, _ = synth.bam_segment(10000, seed=5, threads=2)
    compare(arr.tobytes(), None, tmp_path)
    compare(cases.case_error_midfile("cigar_qlen"), [0, 1], tmp_path)   # silent stop after 77 rows
    compare(cases.case_quirks(), None, tmp_path)
    compare(cases.case_header_only(), None, tmp_path)


@pytest.mark.gpu
def test_read_bam_bind_errors_on_gpu(tmp_path):
    fn = os.path.join(str(tmp_path), "notbam.bam")
    open(fn, "wb").write(b"this is not a BGZF file at all, just text\n" * 10)
    rc, out, _ = run_host(fn)
    assert rc == 3 and out == "ERROR bind: Failed to read SAM/BAM/CRAM header"          # bam_reader.c:461
    good = os.path.join(str(tmp_path), "noindex.bam")
    open(good, "wb").write(cases.case_basic(n=50))
    rc, out, _ = run_host(good, named=[("region", "chr1:1-100")])
    assert rc == 3 and out == "ERROR bind: Region query requires an index (.bai/.csi/.crai)"   # bam_reader.c:647-648
