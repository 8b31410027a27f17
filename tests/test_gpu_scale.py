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


def test_four_million_records_as_three_speculative_shards():
    """the same 4 M-record file cut into three block-range shards by compressed bytes (shards 1 and 2 speculate their first record, every
    shard reads into its successor's blocks until its last record completes): the running digest over the shards in file order equals the
    oracle's digest of the whole file, and the shards' first / end offsets chain"""
    sys.path.insert(0, ROOT)
    import bench
    import duckhts_amd
    from duckhts_amd import synth
    arr, st = synth.bam_segment(4_000_000, seed=777)
    want = orc.bam_digest(arr.tobytes())
    assert want[0] == 4_000_000 and want[1] == 0
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(arr)
        nb = int(ctx.bgzf_index())
        hdr = ctx.bam_open()
        coff, _, _, _ = ctx.bgzf_table(nb)
        cuts = [0] + [int(np.searchsorted(coff, coff[-1] * k // 3)) for k in (1, 2)] + [nb]
        acc = bench.DigestAcc(hdr)
        ends = []
        for r in range(3):
            ctx.set_block_range(cuts[r], cuts[r + 1], r > 0)
            first = None
            while True:
                b = ctx.next_batch(3000)
                if b.n_rows and first is None:
                    first = int(b.first_rec_uoff)
                acc.add(ctx, b)
                if b.status != 0:
                    assert b.status == 1, b.status
                    ends.append((first, int(b.end_uoff)))
                    break
        got = acc.result(0)
    finally:
        ctx.close()
    assert all(ends[i][1] == ends[i + 1][0] for i in range(2)), ends            # the 8-byte hand-off of DESIGN section 6
    bad = [NAMES[i] for i in range(23) if got[i] != want[i]]
    assert not bad, bad


def test_device_numa_node_and_producer_binding():
    """the NUMA node of device 0 as sysfs reports it (-1 where the platform gives none) and the producer's binding seen through the
    operator's stage trace: one line per producer saying which node its device hangs off and whether it was bound"""
    import duckhts_amd
    from duckhts_amd import synth
    L = duckhts_amd.lib()
    node = L.dhts_device_numa_node(0)
    assert node >= -1
    assert L.dhts_device_numa_node(9999) == -1
    import tempfile
    host = os.path.join(ROOT, "tests", "minihost", "minihost")
    with tempfile.NamedTemporaryFile(suffix=".bam") as f:
        f.write(synth.bam_file(50000, seed=5)); f.flush()
        r = subprocess.run([host, duckhts_amd.LIB_PATH, "read_bam", f.name], capture_output=True, text=True, env=dict(os.environ, DHTS_TRACE="1"), timeout=300)
    assert r.returncode == 0 and "OK rows=50000" in r.stdout, (r.stdout[-300:], r.stderr[-600:])
    line = [l for l in r.stderr.splitlines() if "on NUMA node" in l]
    assert line and (f"NUMA node {node} " in line[0]) and (("(bound)" in line[0]) == (node >= 0) or "(not bound)" in line[0]), r.stderr[-600:]


SOAK_MODES = [[], ["--corrupt"], ["--scans"], ["--surface"], ["--regions"], ["--vcf"], ["--vcfregions"], ["--bgzip"], ["--isize"]]


@pytest.mark.parametrize("mode", SOAK_MODES, ids=lambda m: (m[0][2:] if m else "main"))
def test_soak_slice(mode):
    # (the two modes whose long runs found defects in rounds 2 and 3 -- --vcf and --scans -- get 30 s, the rest 6 s)
    secs = "30" if mode and mode[0] in ("--vcf", "--scans") else "6"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--first", "31000", "--seeds", "4000", "--seconds", secs] + mode,
                       capture_output=True, text=True, timeout=400)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    last = [l for l in r.stdout.splitlines() if l.startswith("soak:")]
    assert last and " 0 with mismatches" in last[-1], tail
    assert int(last[-1].split()[1]) >= 1, tail             # at least one seed ran
