"""Sanitizer pass over the phase-A kernel text (GPU ASAN is unavailable on the pool, so the kernel source is compiled for the
host by tools/hostsim/run.sh: ASAN + UBSAN, exact-size LDS image, tokens replayed against each block's CRC32/ISIZE trailer).
This is a check of the kernel's indexing, not a product path: nothing in duckhts_amd can reach it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_phase_a_kernel_text_is_sanitizer_clean():
    files = [os.path.join(ROOT, "tests", "golden", f) for f in ("range.bam", "vcf_file.bcf")]
    r = subprocess.run([os.path.join(ROOT, "tools", "hostsim", "run.sh")] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("blocks")]
    assert len(lines) == 4 and all("failed 0 mismatching 0" in l for l in lines), r.stdout      # two files x two LDS layouts


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_wave_kernel_text_is_sanitizer_clean_and_equals_the_lane_kernel(tmp_path):
    """bgzf_huff_wave.hip compiled for the host as it stands (wave phases become loops over 64 lanes; ASAN + UBSAN, exact-size
    LDS image): every block's literal / token / meta output equals the lane-per-block kernel's and replays to the block's
    CRC32 / ISIZE -- dynamic, fixed and stored blocks, many DEFLATE blocks per BGZF block, long literal runs, and damaged payloads
    (where both kernels must reject the same blocks)."""
    import sys
    import zlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamwriter as bw
    import cases
    files = [os.path.join(ROOT, "tests", "golden", f) for f in ("range.bam", "vcf_file.bcf", "bgzf_boundaries3.bam")]
    for k in ("fixed_huffman", "basic_stored", "basic_level1", "basic_level9", "basic_tiny_blocks", "long_record", "empty_blocks"):
        p = tmp_path / f"{k}.bam"; p.write_bytes(cases.ALL_CASES[k]()); files.append(str(p))
    rnd = __import__("random").Random(4)
    noise = bytes(rnd.getrandbits(8) for _ in range(40000))
    text = b"".join(b"read%06d\tACGTACGTTTGACCA\t%d\n" % (i, i * 7919 % 100003) for i in range(2500))
    for name, raw, kw in (("noise", noise, {}), ("text", text, {}), ("runs", (noise[:3000] + b"Q" * 700) * 12, {})):
        p = tmp_path / f"{name}.bgzf"; p.write_bytes(bw.bgzf_file(raw, payload=65280, level=6)); files.append(str(p))
    # dozens of DEFLATE blocks inside one BGZF block (memLevel 1) and sync-flush points (empty stored blocks between them)
    import struct
    for name, ml, flush in (("memlevel1", 1, False), ("syncflush", 8, True)):
        co = zlib.compressobj(6, zlib.DEFLATED, -15, ml); raw = text[:60000]; payload = b""
        for i in range(0, len(raw), 7000):
            payload += co.compress(raw[i:i + 7000]) + (co.flush(zlib.Z_SYNC_FLUSH) if flush else b"")
        payload += co.flush()
        blk = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", len(payload) + 25) + payload + struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw))
        p = tmp_path / f"{name}.bgzf"; p.write_bytes(blk + bw.bgzf_block(b"")); files.append(str(p))
    run = os.path.join(ROOT, "tools", "hostsim", "run_wave.sh")
    r = subprocess.run([run] + files, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if "blocks" in l]
    assert r.returncode == 0 and len(lines) == len(files) and all("failed 0 mismatching 0 differing-from-lane-kernel 0" in l for l in lines), r.stdout + r.stderr
    # damaged payloads: the two kernels agree on which blocks fail and on every block that still decodes
    for seed in (1, 2, 3):
        r = subprocess.run([run, "--flip", "25", str(seed)] + files[:6], capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if "blocks" in l]
        assert len(lines) == 6 and all("differing-from-lane-kernel 0" in l for l in lines), r.stdout + r.stderr
