#!/usr/bin/env python3
"""Index writers: time of one whole-file scan + hts_idx_push on the host, per format.

  python tools/bench_index.py [--records 4000000]

BAI and CSI of a synthetic BAM, CSI of a synthetic BCF (16 samples), TBI and CSI of a bgzipped sites-only VCF written by the device bgzip.
One JSON line each: file size, records, seconds (second call: pools warm), records/s and MB/s of BGZF."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=4000000)
    args = ap.parse_args()
    import duckhts_amd
    from duckhts_amd import synth
    L = duckhts_amd.lib()
    L.dhts_bcf_build_index.restype = C.c_int64; L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
    L.dhts_bam_build_index_csi.restype = C.c_int64; L.dhts_bam_build_index_csi.argtypes = [C.c_void_p, C.c_int]

    def run(what, data, nrec, opener, builder):
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(data); ctx.bgzf_index(); opener(ctx)
            times = []
            for _ in range(3):
                t0 = time.time(); n = builder(ctx); times.append(time.time() - t0)
                assert n > 0, L.dhts_error(ctx.h)
            dt = min(times[1:])
            print(json.dumps({"what": what, "file_bytes": len(data), "records": nrec, "index_bytes": int(n), "seconds": round(dt, 4), "first_call_seconds": round(times[0], 4),
                              "M_records_per_s": round(nrec / dt / 1e6, 2), "MB_per_s_bgzf": round(len(data) / dt / 1e6, 1)}), flush=True)
        finally:
            ctx.close()

    bam = synth.bam_file(args.records, seed=21)                 # (a single segment of the generator is coordinate-sorted)
    run("BAI of a BAM", bam, args.records, lambda c: c.bam_open(), lambda c: L.dhts_bam_build_index_csi(c.h, 0))
    run("CSI (min_shift 14) of a BAM", bam, args.records, lambda c: c.bam_open(), lambda c: L.dhts_bam_build_index_csi(c.h, 14))
    nb = args.records // 8
    bcf = synth.bcf_file(nb, seed=5)
    run("CSI of a BCF (16 samples)", bcf, nb, lambda c: duckhts_amd.BcfScan(c), lambda c: L.dhts_bcf_build_index(c.h, 14))
    # sites-only VCF text, compressed by the device
    import random
    rnd = random.Random(2)
    lines = ["##fileformat=VCFv4.2", "##contig=<ID=chr1,length=248956422>", "##contig=<ID=chr2,length=242193529>", '##INFO=<ID=AC,Number=A,Type=Integer,Description="d">',
             '##INFO=<ID=AF,Number=A,Type=Float,Description="d">', "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    nv = args.records // 2
    for chrom in ("chr1", "chr2"):
        pos = 0
        for _ in range(nv // 2):
            pos += rnd.randint(1, 90)
            lines.append("%s\t%d\t.\t%s\t%s\t50\tPASS\tAC=%d;AF=%.5f" % (chrom, pos, rnd.choice("ACGT"), rnd.choice("ACGT"), rnd.randint(1, 99), rnd.random()))
    raw = ("\n".join(lines) + "\n").encode()
    ctx = duckhts_amd.Context(0)
    vz = ctx.bgzf_compress(raw)
    ctx.close()
    run("TBI of a bgzipped VCF", vz, nv // 2 * 2, lambda c: duckhts_amd.BcfScan(c), lambda c: L.dhts_bcf_build_index(c.h, 0))
    run("CSI (min_shift 14) of a bgzipped VCF", vz, nv // 2 * 2, lambda c: duckhts_amd.BcfScan(c), lambda c: L.dhts_bcf_build_index(c.h, 14))


if __name__ == "__main__":
    main()
