"""Parity at the size of one bench segment and a slice of the randomized soak, inside the driver-run suite (-m gpu).

* 4,000,000 records (one segment of bench.py's config-2 file: 428 MB of BGZF, 1.3 GB inflated, 20,000 blocks) through read_bam, all 13
  columns, against the oracle column by column (CRC-32 digests of values, lengths and validity: orc_bam_digest / bench.gpu_digest), once as
  one whole-file scan and once with 1,000-block batches;
* tools/soak.py, all nine modes, fixed seeds, a few seconds each (the long runs live in profiles/: this keeps every mode's harness and a
  sample of its seeds under the driver's eyes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NAMES = ["n_rows", "status", "FLAG", "RNAME(id)", "POS", "MAPQ", "RNEXT(id)", "PNEXT", "TLEN", "QNAME.len", "QNAME", "CIGAR.len", "CIGAR", "SEQ.len", "SEQ", "QUAL.len", "QUAL",
         "RG.valid", "RG.len", "READ_GROUP_ID", "SM.valid", "SM.len", "SAMPLE_ID"]


def test_four_million_records_against_the_oracle():
    sys.path.insert(0, ROOT)
    import bench
    import duckhts_amd
    from duckhts_amd import synth
    arr, st = synth.bam_segment(4_000_000, seed=4242)
    data = arr.tobytes()
    want = orc.bam_digest(data)
    assert want[0] == 4_000_000 and want[1] == 0
    for max_blocks in (0, 1000):
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(arr)
            nb = ctx.bgzf_index()
            hdr = ctx.bam_open()
            got = bench.gpu_digest(ctx, hdr, 0, int(nb), False, max_blocks)
        finally:
            ctx.close()
        bad = [NAMES[i] for i in range(23) if got[i] != want[i]]
        assert not bad, (max_blocks, bad)


SOAK_MODES = [[], ["--corrupt"], ["--scans"], ["--surface"], ["--regions"], ["--vcf"], ["--vcfregions"], ["--bgzip"], ["--isize"]]


@pytest.mark.parametrize("mode", SOAK_MODES, ids=lambda m: (m[0][2:] if m else "main"))
def test_soak_slice(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--first", "31000", "--seeds", "400", "--seconds", "6"] + mode,
                       capture_output=True, text=True, timeout=240)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    last = [l for l in r.stdout.splitlines() if l.startswith("soak:")]
    assert last and " 0 with mismatches" in last[-1], tail
    assert int(last[-1].split()[1]) >= 1, tail             # at least one seed ran
