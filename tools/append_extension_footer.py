#!/usr/bin/env python3
"""Turn libduckhts_amd.so into a loadable `duckhts.duckdb_extension`.

DuckDB refuses a shared object without the metadata footer; the reference appends it with
r/Rduckhts/tools/append_extension_metadata.R:17-72 (CMake builds use DuckDB's own script with the same layout).
Layout, 534 bytes after the last byte of the library:

    22 B   start signature: 00 93 04 10 "duckdb_signature" 80 04
    8 x 32 B NUL-padded fields, in file order:
           "", "", "",            three unused fields
           ABI type               "C_STRUCT"  (the extension only uses the duckdb_ext_api_v1 pointer table)
           extension version
           DuckDB (C API) version "v1.2.0"    (the version string duckhts_init_c_api passes to get_api)
           platform               e.g. "linux_amd64"
           "4"                    metadata format version
    256 B  zeros (space for a signature)

usage: append_extension_footer.py [--library-file LIB] [--out-file OUT] [--extension-version V]
                                  [--duckdb-version v1.2.0] [--duckdb-platform linux_amd64] [--abi-type C_STRUCT]
"""
import argparse
import os
import platform as _platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
START_SIGNATURE = bytes([0, 147, 4, 16]) + b"duckdb_signature" + bytes([128, 4])
FOOTER_BYTES = 22 + 8 * 32 + 256


def padded(s: str) -> bytes:
    b = s.encode()[:32]
    return b + b"\0" * (32 - len(b))


def default_platform() -> str:
    arch = {"x86_64": "amd64", "aarch64": "arm64"}.get(_platform.machine(), _platform.machine())
    return f"linux_{arch}"


def footer(extension_version: str, duckdb_version: str = "v1.2.0", duckdb_platform: str = None, abi_type: str = "C_STRUCT") -> bytes:
    fields = ["", "", "", abi_type or "C_STRUCT", extension_version, duckdb_version, duckdb_platform or default_platform(), "4"]
    out = START_SIGNATURE + b"".join(padded(f) for f in fields) + b"\0" * 256
    assert len(out) == FOOTER_BYTES
    return out


def append(library_file: str, out_file: str, **kw) -> int:
    data = open(library_file, "rb").read()
    tmp = out_file + ".tmp"
    with open(tmp, "wb") as f:
        f.write(data)
        f.write(footer(**kw))
    os.replace(tmp, out_file)
    return len(data) + FOOTER_BYTES


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--library-file", default=os.path.join(ROOT, "duckhts_amd", "libduckhts_amd.so"))
    ap.add_argument("--out-file", default=os.path.join(ROOT, "build", "duckhts.duckdb_extension"))
    ap.add_argument("--extension-name", default="duckhts")           # accepted for symmetry with the reference's script; not stored
    ap.add_argument("--extension-version", default="0.2.0-mi355x")
    ap.add_argument("--duckdb-version", default="v1.2.0")
    ap.add_argument("--duckdb-platform", default=None)
    ap.add_argument("--abi-type", default="C_STRUCT")
    a = ap.parse_args(argv)
    os.makedirs(os.path.dirname(os.path.abspath(a.out_file)), exist_ok=True)
    n = append(a.library_file, a.out_file, extension_version=a.extension_version, duckdb_version=a.duckdb_version,
               duckdb_platform=a.duckdb_platform, abi_type=a.abi_type)
    print(f"{a.out_file}: {n} bytes ({FOOTER_BYTES}-byte footer)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
