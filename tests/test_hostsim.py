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
