"""The drop-in boundary as a reference maintainer would use it.

(i)  register_read_bam_function / register_read_bcf_function have external linkage and are driven WITHOUT
     duckhts_init_c_api, exactly as /root/reference/src/duckhts.c:13-16,54-55 does: the host fills the `duckdb_ext_api`
     global (DUCKDB_EXTENSION_API_INIT, duckdb_capi/duckdb_extension.h:1153-1158) and calls the two functions.
(ii) tools/append_extension_footer.py writes the 534-byte metadata footer of
     /root/reference/r/Rduckhts/tools/append_extension_metadata.R:17-72, checked field by field.
"""
import ctypes
import os
import subprocess
import sys

import pytest

import duckhts_amd
from conftest import GOLDEN, ROOT
import re

HOST = os.path.join(ROOT, "tests", "minihost", "minihost")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def host(*args):
    r = subprocess.run([HOST, *args], capture_output=True, text=True)
    return r.returncode, r.stdout.strip()


def test_registration_symbols_are_exported_and_declared():
    L = ctypes.CDLL(duckhts_amd.LIB_PATH)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "duckhts_extension.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(dhts_[a-z0-9_]+|duckhts_[a-z0-9_]+|register_[a-z0-9_]+)\s*\(", text)))
    assert {"duckhts_init_c_api", "register_read_bam_function", "register_read_bcf_function", "dhts_set_duckdb_api"} <= set(names)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/duckhts_extension.h but not exported"
    # the API-table global is a data symbol of 357 pointers, all NULL until a host fills it
    tab = (ctypes.c_void_p * 357).in_dll(L, "duckdb_ext_api")
    assert all(v is None for v in tab)


def catalog(mode):
    rc, out = host(*mode, duckhts_amd.LIB_PATH, "--catalog", "-")
    assert rc == 0, out
    return out.splitlines()


def test_direct_registration_without_the_entrypoint():
    """src/duckhts.c:54-55 order (read_bcf, then read_bam), names, named parameters (VARCHAR=17, BOOLEAN=1), pushdown"""
    want = ["TF read_bcf pushdown=1 bind=1 init=1 local_init=1 func=1 named=region:17,index_path:17,tidy_format:1",
            "TF read_bam pushdown=1 bind=1 init=1 local_init=1 func=1 named=region:17,index_path:17,reference:17,standard_tags:1,auxiliary_tags:1"]
    want += ["TF bgzip pushdown=0 bind=1 init=1 local_init=0 func=1 named=output_path:17,threads:4,level:4,keep:1,overwrite:1",            # src/bgzip.c:336-357
             "TF bgunzip pushdown=0 bind=1 init=1 local_init=0 func=1 named=output_path:17,threads:4,keep:1,overwrite:1",                 # :359-380
             "TF bam_index pushdown=0 bind=1 init=1 local_init=0 func=1 named=index_path:17,min_shift:4,threads:4",                        # src/hts_index_builder.c:326-344
             "TF bcf_index pushdown=0 bind=1 init=1 local_init=0 func=1 named=index_path:17,min_shift:4,threads:4",                        # :346-364
             "TF tabix_index pushdown=0 bind=1 init=1 local_init=0 func=1 named=preset:17,index_path:17,min_shift:4,threads:4,seq_col:4,start_col:4,end_col:4,comment_char:17,skip_lines:4"]   # :366-390
    assert catalog(["--direct"]) == want
    assert catalog([]) == want                              # the entrypoint registers the same seven, in src/duckhts.c's order


def test_direct_registration_binds_like_the_entrypoint():
    # bind-time behaviour that needs no device: the reference's error strings (src/bam_reader.c:416,446; src/bcf_reader.c:461,494)
    assert host("--direct", duckhts_amd.LIB_PATH, "read_bam", "") == (3, "ERROR bind: read_bam requires a file path")
    assert host("--direct", duckhts_amd.LIB_PATH, "read_bam", "/no/such.bam") == (3, "ERROR bind: Failed to open SAM/BAM/CRAM file: /no/such.bam")
    assert host("--direct", duckhts_amd.LIB_PATH, "read_bcf", "") == (3, "ERROR bind: read_bcf requires a file path")
    assert host("--direct", duckhts_amd.LIB_PATH, "read_bcf", "/no/such.bcf") == (3, "ERROR bind: Failed to open BCF/VCF file: /no/such.bcf")
    # src/bgzip.c:101-105,139-150; src/hts_index_builder.c:121-125,232-239
    assert host(duckhts_amd.LIB_PATH, "bgzip", "") == (3, "ERROR bind: bgzip requires a file path")
    assert host(duckhts_amd.LIB_PATH, "bgunzip", "") == (3, "ERROR bind: bgunzip requires a file path")
    assert host(duckhts_amd.LIB_PATH, "bgzip", __file__, "-n", "output_path=" + __file__) == (3, f"ERROR bind: bgzip: output '{__file__}' already exists (use overwrite := TRUE to replace)")
    assert host(duckhts_amd.LIB_PATH, "bam_index", "") == (3, "ERROR bind: bam_index requires a file path")
    assert host(duckhts_amd.LIB_PATH, "bcf_index", "") == (3, "ERROR bind: bcf_index requires a file path")
    assert host(duckhts_amd.LIB_PATH, "tabix_index", "") == (3, "ERROR bind: tabix_index requires a file path")
    assert host(duckhts_amd.LIB_PATH, "tabix_index", "x.gz", "-n", "preset=bam") == (3, "ERROR bind: tabix_index: preset must be one of vcf, bed, gff, sam")


@pytest.mark.gpu
def test_direct_registration_scans(tmp_path):
    """the directly registered functions are the same callbacks: duckhts.test:129-131 (112 rows) and :74-76 (15 rows)"""
    rc, out = host("--direct", duckhts_amd.LIB_PATH, "read_bam", os.path.join(GOLDEN, "range.bam"))
    assert rc == 0 and out.startswith("OK rows=112 "), out
    rc, out = host("--direct", duckhts_amd.LIB_PATH, "read_bcf", os.path.join(GOLDEN, "vcf_file.bcf"))
    assert rc == 0 and out.startswith("OK rows=15 "), out


def test_extension_footer_layout(tmp_path):
    import append_extension_footer as aef
    lib = tmp_path / "lib.so"
    body = os.urandom(1000)
    lib.write_bytes(body)
    out = tmp_path / "duckhts.duckdb_extension"
    assert aef.main(["--library-file", str(lib), "--out-file", str(out), "--extension-version", "v9.9.9",
                     "--duckdb-version", "v1.2.0", "--duckdb-platform", "linux_amd64"]) == 0
    d = out.read_bytes()
    assert len(d) == 1000 + 534 and d[:1000] == body
    f = d[1000:]
    # append_extension_metadata.R:17-22: as.raw(c(0,147,4,16)) "duckdb_signature" as.raw(c(128,4))
    assert f[:22] == bytes([0, 147, 4, 16]) + b"duckdb_signature" + bytes([128, 4])
    fields = [f[22 + 32 * i: 22 + 32 * (i + 1)] for i in range(8)]
    # :57-64: three empty fields, ABI type, extension version, DuckDB version, platform, "4"
    want = [b"", b"", b"", b"C_STRUCT", b"v9.9.9", b"v1.2.0", b"linux_amd64", b"4"]
    for got, w in zip(fields, want):
        assert got == w + b"\0" * (32 - len(w))
    assert f[22 + 256:] == b"\0" * 256 and len(f) == 22 + 256 + 256            # :65
    # a field longer than 32 bytes is cut, not overflowed (:24-30)
    assert aef.padded("x" * 40) == b"x" * 32


def test_extension_footer_on_the_built_library(tmp_path):
    import append_extension_footer as aef
    out = tmp_path / "duckhts.duckdb_extension"
    n = aef.append(duckhts_amd.LIB_PATH, str(out), extension_version="test")
    assert n == os.path.getsize(duckhts_amd.LIB_PATH) + 534
    # still a loadable shared object (dlopen ignores trailing bytes) exporting the entrypoint DuckDB looks up: <name>_init_c_api
    L = ctypes.CDLL(str(out))
    assert hasattr(L, "duckhts_init_c_api")


def test_bgzf_wrap_is_valid_bgzf_without_a_gpu():
    """dhts_bgzf_wrap (host only): stored-block BGZF with CRC-32 / ISIZE per block and htslib's EOF block; any gzip reader inflates it"""
    import ctypes as C
    import gzip
    import numpy as np
    L = duckhts_amd.lib()
    L.dhts_bgzf_wrap.restype = C.c_int64
    L.dhts_bgzf_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    rng = np.random.RandomState(3)
    for n in (0, 1, 65279, 65280, 65281, 200000):
        raw = rng.randint(0, 256, n).astype(np.uint8)
        need = L.dhts_bgzf_wrap(raw.ctypes.data, n, None, 0)
        out = np.zeros(need, np.uint8)
        assert L.dhts_bgzf_wrap(raw.ctypes.data, n, out.ctypes.data, need) == need
        b = out.tobytes()
        assert gzip.decompress(b) == raw.tobytes()
        assert b[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
        p, blocks = 0, 0                                    # the BSIZE chain covers the file exactly
        while p < len(b):
            assert b[p:p + 4] == b"\x1f\x8b\x08\x04" and b[p + 12:p + 16] == b"BC\x02\x00"
            p += int.from_bytes(b[p + 16:p + 18], "little") + 1; blocks += 1
        assert p == len(b) and blocks == (n + 65279) // 65280 + 1
