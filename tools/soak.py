#!/usr/bin/env python3
"""soak.py -- randomized GPU-vs-oracle parity over many seeds (not part of the test suite; a way to spend spare GPU minutes).

For every seed: a BAM of random records (random BGZF payload size / zlib level / batch size) through read_bam, its standard-tag and
auxiliary-tag views, a random region query, and a BCF of fuzz records (wide or tidy, random payload / batch) through read_bcf --
each compared column for column with the CPU oracle.  Prints one line per seed and a summary; exit code 1 on any mismatch.
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def damage(data, rnd):
    """flip 1-3 random bits after the first 200 bytes (so that most files still open).  ISIZE fields are left alone here (a damaged ISIZE
    with a good CRC is not damage to htslib, the oracle or -- since round 4 -- the GPU path: --isize covers that)."""
    import struct
    b = bytearray(data)
    isize = set()
    p = 0
    while p + 18 <= len(b) and b[p:p + 4] == b"\x1f\x8b\x08\x04":
        bl = struct.unpack_from("<H", b, p + 16)[0] + 1
        isize.update(range(p + bl - 4, p + bl))
        p += bl
    for _ in range(rnd.randint(1, 3)):
        i = rnd.randrange(min(200, len(b) - 1), len(b))
        if i in isize:
            continue
        b[i] ^= 1 << rnd.randrange(8)
    return bytes(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--seconds", type=float, default=480.0)
    ap.add_argument("--regions", action="store_true", help="random (mostly malformed) region strings against the restatement of hts_parse_region / sam_itr_regarray")
    ap.add_argument("--surface", action="store_true", help="per seed: the DuckDB table functions through the mini host, random projections, chunk-exact against the oracle")
    ap.add_argument("--scans", action="store_true", help="per seed: N-way block-range shards, a self-built BAI driving region queries, the overlap join, a projection mask")
    ap.add_argument("--bgzip", action="store_true", help="per seed: an input of random texture and size through the device compressor; every member checked with zlib, the whole with the library's own inflate")
    ap.add_argument("--vcfregions", action="store_true", help="per seed: a sorted bgzipped VCF with symbolic alleles / SVLEN / END / gVCF LEN, the tabix writer (TBI or CSI), random region queries vs the oracle's tabix interval rule")
    ap.add_argument("--isize", action="store_true", help="per seed: hostile ISIZE trailer values (bit flips, 0xFFFFxxxx, > 64 KiB) on random blocks: htslib never reads ISIZE, so the scan must return what the clean file returns")
    ap.add_argument("--vcf", action="store_true", help="per seed: a VCF TEXT file (sites-only or with samples, plain or BGZF) of lines made from a grammar and then damaged character by character, wide or tidy, random batch sizes")
    ap.add_argument("--corrupt", action="store_true", help="flip 1-3 random bytes of each BAM / BCF file: the rows before the damage and the error sign must still agree")
    args = ap.parse_args()
    import bamwriter as bw  # noqa: F401
    import bcf_cases
    import bcfwriter as W
    import cases
    import duckhts_amd
    import orc
    import region_oracle as ro
    import tag_cases
    t0 = time.time()
    bad = 0
    done = 0
    for seed in range(args.first, args.first + args.seeds):
        if time.time() - t0 > args.seconds:
            break
        rnd = random.Random(seed)
        msgs = []
        if args.isize:
            import struct
            payload = rnd.choice([61, 777, 4000, 65280]); n = rnd.choice([50, 300, 1500])
            data = bytearray(cases.case_basic(payload=payload, level=rnd.choice([0, 1, 6]), seed=seed, n=n))
            clean = orc.bam_read(bytes(data))
            p, blocks = 0, []
            while p + 18 <= len(data) and data[p:p + 4] == b"\x1f\x8b\x08\x04":
                bl = struct.unpack_from("<H", data, p + 16)[0] + 1
                blocks.append((p, bl)); p += bl
            orig = {}
            for _ in range(rnd.randint(1, 3)):
                k = rnd.randrange(1, len(blocks))
                at = blocks[k][0] + blocks[k][1] - 4
                old = struct.unpack_from("<I", data, at)[0]
                orig.setdefault(k, old)
                new = rnd.choice([old ^ (1 << rnd.randrange(32)), 0xFFFF0000 + old, 0xFFFFFFFF, 0x80000000 | old, 65537, 65536, old + 1, rnd.getrandbits(32)])
                struct.pack_into("<I", data, at, new & 0xFFFFFFFF)
            hit = [k for k, o in orig.items() if struct.unpack_from("<I", data, blocks[k][0] + blocks[k][1] - 4)[0] != o]   # (a second hit may restore the value)
            mb = rnd.choice([0, 1, 2, 5])
            got = {}
            try:
                # htslib never reads ISIZE (bgzf.c:793-801: the CRC decides): the file reads like the clean one -- the block table is re-placed
                # from the decoded lengths (round 4; rounds 1-3 ended the stream at such a block)
                got = duckhts_amd.read_bam(bytes(data), max_blocks=mb)
                if got["status"] < 0 or got["n_rows"] != clean["n_rows"]:
                    msgs.append(f"a file with wrong ISIZE fields {sorted(set(hit))} of {len(blocks)} blocks differs from the clean one: status {got['status']} rows {got['n_rows']}/{clean['n_rows']}")
                for kk in duckhts_amd.BAM_COLUMNS:
                    if list(got[kk]) != list(clean[kk][:got["n_rows"]]):
                        msgs.append(f"column {kk} differs from the clean scan")
            except duckhts_amd.DhtsError as e:
                msgs.append(f"raised {e}")
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  (isize: {len(blocks)} blocks, damaged {sorted(set(hit))}, rows {got['n_rows'] if 'n_rows' in got else '?'} of {clean['n_rows']})", flush=True)
            bad += bool(msgs)
            continue
        if args.regions:
            if seed == args.first:
                rdata = cases.case_basic(payload=4000, level=6, seed=5, n=1500)
                rexp = orc.bam_read(rdata)
                rnames = [bytes(x).decode() for x in rexp["ref_names"]]
                main.cache = (rdata, rexp, rnames)
            rdata, rexp, rnames = main.cache
            for _ in range(20):
                toks = []
                for _t in range(rnd.randint(1, 3)):
                    kind = rnd.random()
                    nm = rnd.choice(rnames + ["nosuch", "", "*", ".", "{" + rnd.choice(rnames) + "}", rnd.choice(rnames) + ":1"])
                    if kind < 0.3:
                        toks.append(nm)
                    elif kind < 0.7:
                        a = rnd.choice(["", "1", "1,000", "2k", "5K", "1m", "0", "-5", "1e3", "12x", "99999999999", str(rnd.randrange(1, 30000))])
                        b = rnd.choice(["", "1", "3,500", "20k", "1M", "0", "abc", str(rnd.randrange(1, 60000))])
                        toks.append(f"{nm}:{a}-{b}" if rnd.random() < 0.8 else f"{nm}:{a}")
                    else:
                        toks.append("".join(rnd.choice("chr1:-,{}*.0123456789kKmMeE ") for _c in range(rnd.randint(1, 12))))
                reg = ",".join(toks)
                try:
                    keep = ro.keep_mask(rexp, reg)
                except Exception as e:
                    msgs.append(f"oracle raised on {reg!r}: {e}"); continue
                try:
                    g = duckhts_amd.read_bam(rdata, region=reg)
                    if keep is None:
                        msgs.append(f"region {reg!r}: gpu {g['n_rows']} rows, oracle: no known reference")
                    elif g["n_rows"] != int(keep.sum()) or g["QNAME"] != [q for q, k in zip(rexp["QNAME"], keep) if k]:
                        msgs.append(f"region {reg!r}: {g['n_rows']} rows vs {int(keep.sum())}")
                except duckhts_amd.DhtsError as e:
                    if keep is not None:
                        msgs.append(f"region {reg!r}: gpu raised ({e}), oracle keeps {int(keep.sum())}")
            # the same strings as read_bcf regions (chained single-region scans; unknown ones are skipped)
            if seed == args.first:
                import numpy as np
                bd = W.bcf_bytes(bcf_cases.std_header(), bcf_cases.fuzz_records(7, 1200, len(bcf_cases.SAMPLES)), payload=4000)
                be = orc.bcf_read(bd)
                c3 = duckhts_amd.Context(0); c3.open(bd); c3.bgzf_index()
                bcont = [x.decode() if x else "\x01" for x in duckhts_amd.BcfScan(c3, False).contigs]
                c3.close()
                main.bcache = (bd, be, bcont)
            bd, be, bcont = main.bcache
            for _ in range(6):
                toks = []
                for _t in range(rnd.randint(1, 3)):
                    nm = rnd.choice(bcont + ["nosuch", ".", "{" + rnd.choice(bcont) + "}"])
                    a = rnd.choice(["", "1", "1,000", "2k", "0", "-5", "1e3", "12x", str(rnd.randrange(1, 3000))])
                    b = rnd.choice(["", "1", "3,500", "20k", "abc", str(rnd.randrange(1, 9000))])
                    toks.append(rnd.choice([nm, f"{nm}:{a}-{b}", f"{nm}:{a}", "".join(rnd.choice("12:-,{}.kK ") for _c in range(rnd.randint(1, 8)))]))
                reg = ",".join(toks)
                try:
                    rows_ = ro.bcf_region_rows(be, bcont, reg, 1)
                    want = orc.bcf_take_rows(be, rows_)
                    gotr = duckhts_amd.read_bcf(bd, region=reg)
                    d = orc.bcf_cols_diff(want, gotr)
                    if d is not None:
                        msgs.append(f"bcf region {reg!r}: {d}")
                except Exception as e:
                    msgs.append(f"bcf region {reg!r}: {type(e).__name__} {e}")
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs[:3])}", flush=True)
            bad += bool(msgs)
            continue
        if args.vcf:
            import vcf_text_cases as V
            smp = rnd.random() < 0.5
            hdr = list(V.SHDR if smp else V.HDR)
            if rnd.random() < 0.3:
                hdr[0] = "##fileformat=VCFv4.4"
            specials = "\t;=,:/|.+-eE 0123456789ACGT<>x\r"
            def tok(kind, strict=False):                  # strict: inside FORMAT, where trailing garbage is an error: such tokens are rare there
                risky = (not strict) or rnd.random() < 0.01
                if kind == "int":
                    return rnd.choice([".", "", "-", "+5", str(rnd.randrange(-10, 10 ** rnd.randrange(1, 12)))] + (["12abc", "0x1f"] if risky else []))
                if kind == "flt":
                    return rnd.choice([".", "", ".5", "5.", "%g" % (rnd.random() * 10 ** rnd.randrange(-40, 40)), "%.*f" % (rnd.randrange(0, 18), rnd.random()), "nan", "inf", "-inf", "%d" % rnd.randrange(10 ** 18)] + (["1e", "abc", "0x10"] if risky else ["0x10"]))
                return "".join(rnd.choice("abcXYZ_01. ") for _ in range(rnd.randrange(0, 9)))
            def info():
                parts = []
                for _ in range(rnd.randrange(0, 7)):
                    k = rnd.choice(["DP", "AF", "AC", "MQ", "DB", "SB", "ANN_S", "TAGS", "FV", "NEW" + str(rnd.randrange(4)), "q10", "GT", ""])
                    if rnd.random() < 0.15:
                        parts.append(k)
                    else:
                        kind = {"DP": "int", "AC": "int", "SB": "int", "AF": "flt", "MQ": "flt", "FV": "flt"}.get(k, "str")
                        parts.append(k + "=" + ",".join(tok(kind) for _ in range(rnd.randrange(1, 4))))
                return ";".join(parts) or "."
            def sample(keys):
                vals = []
                for k in keys[:rnd.randrange(1, len(keys) + 1)]:
                    if k == "GT":
                        vals.append(rnd.choice(["0/1", "1|1", "./.", ".", "0", "1/2/3", "10/11"] + (["|0|1", "/1"] if hdr[0].endswith("4.4") or rnd.random() < 0.01 else []) + (["0/x", ""] if rnd.random() < 0.01 else [])))
                    elif k in ("GQ", "AD"):
                        vals.append(",".join(tok("int", True) for _ in range(rnd.randrange(1, 4))))
                    elif k == "GL":
                        vals.append(",".join(tok("flt", True) for _ in range(rnd.randrange(1, 4))))
                    else:
                        vals.append(tok("str").replace(" ", "_"))
                return ":".join(vals)
            lines = []
            for i in range(rnd.choice([20, 200, 1500])):
                l = "\t".join([rnd.choice(["chr1", "chr2", "chrUn" + str(rnd.randrange(3))]), str(rnd.randrange(0, 10 ** rnd.randrange(1, 10))), rnd.choice([".", "rs%d" % i, ""]),
                               rnd.choice(["A", "ACGT", ""]), rnd.choice([".", "T", "T,G", "<DEL>", ",", ""]), tok("flt"), rnd.choice(["PASS", ".", "q10", "q10;s50", "new%d" % rnd.randrange(3), "q10;"] + ([""] if rnd.random() < 0.01 else [])), info()])
                if smp and rnd.random() < 0.95:
                    keys = rnd.choice([["GT", "GQ", "AD", "GL", "FT"], ["GT"], ["GQ", "GT"], ["GT", "NEWF" + str(rnd.randrange(3))], ["GL", "AD"], ["."], ["GT", "GT"]] + ([["FLG"], ["GT", "."]] if rnd.random() < 0.01 else []))
                    l += "\t" + ":".join(keys) + "".join("\t" + sample(keys) for _ in range(3 if rnd.random() < 0.99 else rnd.choice([2, 4])))
                if rnd.random() < 0.004:                    # damage: replace / insert / delete a character
                    b = list(l)
                    for _ in range(rnd.randrange(1, 3)):
                        if not b:
                            break
                        j = rnd.randrange(len(b)); op = rnd.randrange(3)
                        if op == 0:
                            b[j] = rnd.choice(specials)
                        elif op == 1:
                            b.insert(j, rnd.choice(specials))
                        else:
                            del b[j]
                    l = "".join(b).replace("\n", "")
                lines.append(l)
            raw = V.text(lines, hdr=hdr, eol=rnd.choice(["\n", "\n", "\r\n"]), last_eol=rnd.random() < 0.8)
            kind = rnd.random()
            if kind < 0.3:
                data = raw
            elif kind < 0.45:                                # plain (non-BGZF) gzip, one member or several: read through the serial device decoder; the oracle gets the text
                import gzip as _gz
                cuts_ = sorted(rnd.sample(range(1, max(2, len(raw))), rnd.choice([0, 0, 1, 3]))) if len(raw) > 4 else []
                data = b"".join(_gz.compress(raw[a:b], rnd.choice([0, 1, 6, 9])) for a, b in zip([0] + cuts_, cuts_ + [len(raw)]))
            else:
                data = bw.bgzf_file(raw, payload=rnd.choice([500, 3000, 65280]), level=rnd.choice([1, 6]))
            tidy = smp and rnd.random() < 0.3
            try:
                exp = orc.bcf_read(raw if 0.3 <= kind < 0.45 else data, tidy)
                got = duckhts_amd.read_bcf(data, tidy=tidy, max_blocks=rnd.choice([0, 1, 2, 5]))
                d = orc.bcf_cols_diff(exp, got)
                if d is not None:
                    msgs.append("vcf text: " + d)
                    if os.environ.get("SOAK_DUMP"):            # keep the input of a failing seed for tools/dbg (and say where the tables part)
                        os.makedirs(os.environ["SOAK_DUMP"], exist_ok=True)
                        open(os.path.join(os.environ["SOAK_DUMP"], f"vcf_seed_{seed}{'_tidy' if tidy else ''}.bin"), "wb").write(data)
                        try:
                            ce = orc.bcf_col_py(exp["by_name"]["CHROM"]); cg = orc.bcf_col_py(got["by_name"]["CHROM"])
                            k = next((i for i, (a, b) in enumerate(zip(ce, cg)) if a != b), None)
                            msgs.append(f"first CHROM difference at row {k}: oracle {ce[k] if k is not None else None} gpu {cg[k] if k is not None else None}; oracle rows {exp['n_rows']} gpu rows {got['n_rows']}")
                        except Exception as e2:
                            msgs.append(f"(dump: {e2})")
                elif (got["status"] == 1) != (exp["status"] == 0):
                    msgs.append(f"vcf text status {got['status']} vs {exp['status']}")
                if not msgs and exp["n_rows"] > 0 and rnd.random() < 0.5:
                    # a random projection: the text encoder writes only the INFO keys the projected columns read (VcfArgs::info_keep); the columns must be the full read's
                    names = [c_["name"] for c_ in exp["cols"]]
                    want = rnd.sample(names, rnd.randint(1, min(len(names), 6)))
                    sub = duckhts_amd.read_bcf(data, tidy=tidy, columns=[names.index(w) for w in want], max_blocks=rnd.choice([0, 1, 3]))
                    d2 = orc.bcf_cols_diff({"n_rows": exp["n_rows"], "cols": [exp["by_name"][w] for w in want]}, sub)
                    if d2 is not None:
                        msgs.append(f"vcf text projection {want}: " + d2)
                    elif (sub["status"] == 1) != (exp["status"] == 0):
                        msgs.append(f"vcf text projection {want}: status {sub['status']} vs {exp['status']}")
                if not msgs and exp["status"] == 0 and exp["n_rows"] > 0:
                    # block-range shards of the text: every cut reproduces the scan (lines synchronise on the newline)
                    c0 = duckhts_amd.Context(0); c0.open(data); nblk = c0.bgzf_index(); c0.close()
                    if nblk >= 2:
                        ways = rnd.randint(2, min(nblk, 6))
                        cuts = sorted(rnd.sample(range(1, nblk), ways - 1)) if nblk > ways else list(range(1, nblk))
                        cuts = [0] + cuts + [nblk]
                        parts = [duckhts_amd.read_bcf(data, tidy=tidy, block_range=(cuts[r], cuts[r + 1], r > 0), max_blocks=rnd.choice([0, 1, 3])) for r in range(len(cuts) - 1)]
                        if sum(p["n_rows"] for p in parts) != exp["n_rows"] or any(p["status"] != 1 for p in parts):
                            msgs.append(f"text shards {cuts}: rows {[p['n_rows'] for p in parts]} status {[p['status'] for p in parts]} vs {exp['n_rows']}")
                        else:
                            import numpy as np
                            pos = np.concatenate([p["by_name"]["POS"]["fixed"] for p in parts if p["n_rows"]])
                            if not np.array_equal(pos, exp["by_name"]["POS"]["fixed"]):
                                msgs.append(f"text shards {cuts}: POS differs")
            except Exception as e:
                msgs.append(f"vcf text: {type(e).__name__} {str(e)[:200]}")
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  (vcf text: {len(lines)} lines{' samples' if smp else ''}{' tidy' if tidy else ''}, {exp['n_rows'] if 'exp' in dir() else '?'} rows)", flush=True)
            bad += bool(msgs)
            continue
        if args.bgzip:
            # bgzip on the device: inputs of every texture (random, runs, periodic, text, mixed; sizes around the block boundaries) must come back
            # from zlib unchanged, block by block, with the container fields right; bgunzip of the result through the library's own inflate too
            import gzip
            import zlib
            import struct
            kind = rnd.choice(["random", "runs", "period", "text", "mixed", "lowent"])
            size = rnd.choice([0, 1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 4000, 65279, 65280, 65281, 130560, 130561, rnd.randint(0, 400000), rnd.randint(0, 3000000)])
            def gen(k, m):
                if k == "random":
                    return rnd.randbytes(m)
                if k == "runs":
                    out = bytearray()
                    while len(out) < m:
                        out += bytes([rnd.randrange(256)]) * rnd.choice([1, 2, 3, 4, 5, 257, 258, 259, 300, 1000, 70000])
                    return bytes(out[:m])
                if k == "period":
                    per = rnd.choice([1, 2, 3, 4, 5, 7, 8, 255, 256, 32767, 32768, 32769, 40000])
                    return (rnd.randbytes(per) * (m // per + 1))[:m]
                if k == "text":
                    words = [rnd.randbytes(rnd.randint(1, 12)).hex().encode() for _ in range(rnd.choice([5, 50, 2000]))]
                    out = bytearray()
                    while len(out) < m:
                        out += rnd.choice(words) + rnd.choice([b" ", b"\t", b"\n", b"="])
                    return bytes(out[:m])
                if k == "lowent":
                    return bytes(rnd.choice(b"\x00\x00\x00\x01\xff\x90") for _ in range(m))
                out = bytearray()
                while len(out) < m:
                    out += gen(rnd.choice(["random", "runs", "period", "text", "lowent"]), rnd.randint(1, 70000))
                return bytes(out[:m])
            raw = gen(kind, size)
            level = rnd.choice([-1, -1, 6, 1, 9, 0])
            try:
                ctx = duckhts_amd.Context(0)
                try:
                    z = ctx.bgzf_compress(raw, level)
                    if z[-28:] != bw.EOF_BLOCK:
                        msgs.append("no EOF block")
                    p, at = 0, 0
                    while p < len(z) - 28 and not msgs:
                        bs = struct.unpack_from("<H", z, p + 16)[0] + 1
                        if z[p:p + 4] != b"\x1f\x8b\x08\x04" or z[p + 10:p + 16] != b"\x06\x00BC\x02\x00" or bs > 65536:
                            msgs.append(f"bad member header at {p}"); break
                        piece = zlib.decompress(z[p + 18:p + bs - 8], -15)
                        crc, isz = struct.unpack_from("<II", z, p + bs - 8)
                        if piece != raw[at:at + 65280] or isz != len(piece) or crc != zlib.crc32(piece):
                            msgs.append(f"member at {p}: payload / CRC / ISIZE differ"); break
                        at += isz; p += bs
                    if not msgs and (at != len(raw) or p != len(z) - 28):
                        msgs.append(f"covered {at} of {len(raw)} bytes")
                    if not msgs and len(raw):
                        ctx.open(z); nb = ctx.bgzf_index()
                        out, bst = ctx.bgzf_inflate(0, nb, len(raw))
                        if bytes(out) != raw or any(bst):
                            msgs.append("the library's own inflate disagrees")
                finally:
                    ctx.close()
            except Exception as e:
                msgs.append(f"bgzip: {type(e).__name__} {str(e)[:200]}")
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs[:3])}  (bgzip {kind} {size} bytes level {level})", flush=True)
            bad += bool(msgs)
            continue
        if args.vcfregions:
            # sorted bgzipped VCF text with every form the tabix interval rule looks at (symbolic alleles, SVLEN, END, gVCF LEN) -> the index
            # writer (TBI or CSI) -> random regions: rows vs the oracle's tabix rule
            import test_vcf_region as TR
            import vcf_text_cases as V
            hdr = V.SHDR[:-1] + ['##INFO=<ID=END,Number=1,Type=Integer,Description="d">', '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="d">',
                                 '##FORMAT=<ID=LEN,Number=1,Type=Integer,Description="d">'] + (["##contig=<ID=chr3,length=%d>" % rnd.choice([5000, 10 ** 7, 3 * 10 ** 9])] if rnd.random() < 0.5 else []) + [V.SHDR[-1]]
            names = rnd.sample(["chr1", "chr2", "chrUn", "chr3", "1", "HLA-A*01:01"], rnd.randint(1, 4))
            lines = []
            per = rnd.choice([30, 400, 3000])
            for chrom in names:
                pos = rnd.choice([0, 1, 1000])
                for _ in range(per):
                    pos += rnd.choice([0, 1, 1, rnd.randint(1, 50), rnd.randint(1, 20000)])
                    alt = rnd.choice(["T", "T", "T,C", "<DEL>", "<DUP>", "<DUP:TANDEM>", "<INV>", "<CNV>", "<INS>", "<DELX>", "<*>", "T,<NON_REF>", "<DEL>,T,<DUP>", "."])
                    inf = []
                    if rnd.random() < 0.3:
                        inf.append("SVLEN=" + ",".join(rnd.choice([".", "", str(rnd.randint(-30000, 30000)), "5x"]) for _ in range(rnd.randint(1, 3))))
                    if rnd.random() < 0.3:
                        inf.append(rnd.choice(["END=", "XEND=", "END=", "END="]) + rnd.choice([".", str(pos + rnd.randint(-5, 40000)), str(rnd.randint(0, 100)), "0x10", "12abc"]))
                    if rnd.random() < 0.3:
                        inf.append("DP=%d" % rnd.randint(0, 99))
                    rnd.shuffle(inf)
                    fmt = rnd.choice(["GT", "GT:LEN", "LEN:GT", "GT:GQ:LEN"])
                    def smp():
                        v = {"GT": "0/0", "GQ": str(rnd.randint(0, 99)), "LEN": rnd.choice([".", str(rnd.randint(0, 30000)), str(rnd.randint(0, 50))])}
                        ks = fmt.split(":")
                        return ":".join(v[k] for k in ks[:rnd.randint(1, len(ks))])
                    lines.append("\t".join([chrom, str(pos), ".", rnd.choice(["A", "ACGT", "ACGTACGTACGTACGTAC"]), alt, ".", ".", ";".join(inf) or ".", fmt, smp(), smp(), smp()]))
            data = bw.bgzf_file(V.text(lines, hdr=hdr), payload=rnd.choice([500, 3000, 65280]), level=rnd.choice([0, 1, 6]))
            try:
                ms = rnd.choice([0, 0, 14, 10, 16])
                raw, index = TR.build_index(data, ms)
                tidy = rnd.random() < 0.2
                exp = orc.bcf_read(data, tidy)
                reps = exp["n_samples"] if tidy else 1
                chrom = orc.bcf_col_py(exp["by_name"]["CHROM"])[::reps]
                if exp["status"] != 0 or exp["n_rows"] != len(lines) * reps:
                    msgs.append(f"generator: oracle status {exp['status']} rows {exp['n_rows']}")
                t = TR.parse_tabix(raw)
                if [x.decode() for x in t["names"]] != names:
                    msgs.append(f"index names {t['names']} vs {names}")
                for _ in range(4):
                    toks = []
                    for _k in range(rnd.randint(1, 3)):
                        nm = rnd.choice(names + ["nosuch"])
                        if ":" in nm or rnd.random() < 0.1:
                            nm = "{" + nm + "}"
                        b = rnd.randint(1, 300000)
                        toks.append(rnd.choice([nm, "%s:%d-%d" % (nm, b, b + rnd.choice([0, 10, 5000, 10 ** 6])), "%s:%d" % (nm, b), "%s:-%d" % (nm, b), "."]))
                    rg = ",".join(toks)
                    rows = ro.vcf_text_region_rows(exp, chrom, names, rg, reps)
                    got = duckhts_amd.read_bcf(data, tidy=tidy, region=rg, index=index, max_blocks=rnd.choice([0, 1, 3]))
                    d = orc.bcf_cols_diff(orc.bcf_take_rows(exp, rows), got)
                    if d is not None:
                        msgs.append(f"region {rg}: {d}")
            except Exception as e:
                msgs.append(f"vcf regions: {type(e).__name__} {str(e)[:200]}")
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs[:3])}  (vcf regions: {len(lines)} lines on {names}, index min_shift {ms})", flush=True)
            bad += bool(msgs)
            continue
        if args.surface:
            import tempfile
            import test_duckdb_surface as sf
            tmp = tempfile.mkdtemp()
            try:
                n = rnd.choice([50, 300, 1500, 5000])
                data = cases.case_basic(payload=rnd.choice([777, 4000, 65280]), level=rnd.choice([1, 6]), seed=seed, n=n)
                proj = None if rnd.random() < 0.3 else sorted(rnd.sample(range(13), rnd.randint(1, 6)), key=lambda _: rnd.random())
                sf.compare(data, proj=proj, tmp_path=tmp)
                ns = rnd.choice([0, len(bcf_cases.SAMPLES)])
                hdr = bcf_cases.std_header() if ns else bcf_cases.std_header(samples=())
                bdata = W.bcf_bytes(hdr, bcf_cases.fuzz_records(seed, rnd.choice([100, 800, 3000]), ns), payload=rnd.choice([777, 4000, 65280]))
                tidy = rnd.random() < 0.3
                ncol = len(orc.bcf_read(bdata, tidy)["cols"])
                bproj = None if rnd.random() < 0.3 else sorted(rnd.sample(range(ncol), rnd.randint(1, min(6, ncol))), key=lambda _: rnd.random())
                sf.compare_bcf(bdata, tmp, tidy=tidy, proj=bproj)
            except AssertionError as e:
                msgs.append("surface: " + str(e)[:300])
            finally:
                import shutil
                shutil.rmtree(tmp, ignore_errors=True)
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  (surface: bam {n} rows proj {proj}, bcf proj {bproj}{' tidy' if tidy else ''})", flush=True)
            bad += bool(msgs)
            continue
        if args.scans:
            import numpy as np
            from duckhts_amd import synth
            # ---- shards of a random-record file: the union of the ranks' rows is the file, in order ----
            data = cases.case_basic(payload=rnd.choice([300, 777, 4000, 20000]), level=rnd.choice([1, 6]), seed=seed, n=rnd.choice([300, 1500]))
            exp = orc.bam_read(data)
            world = rnd.randint(2, 6)
            rows, qn = 0, []
            for rank in range(world):
                g = duckhts_amd.read_bam(data, shard=(rank, world), max_blocks=rnd.choice([0, 2]))
                rows += g["n_rows"]; qn += g["QNAME"]
            if rows != exp["n_rows"] or qn != exp["QNAME"]:
                msgs.append(f"{world}-way shards: {rows} rows vs {exp['n_rows']}")
            # ---- sorted file: BAI writer -> indexed region == oracle predicate; overlap join; projection ----
            sdata = synth.bam_file(rnd.choice([3000, 20000, 60000]), seed=seed)
            sexp = orc.bam_read(sdata)
            names = [bytes(x).decode() for x in sexp["ref_names"]]
            ctx = duckhts_amd.Context(0)
            try:
                ctx.open(sdata); ctx.bgzf_index(); ctx.bam_open()
                bai = ctx.build_index()
            finally:
                ctx.close()
            for _ in range(2):
                t = rnd.randrange(len(names)); b = rnd.randrange(1, 50_000_000)
                reg = ",".join(f"{names[rnd.randrange(len(names))]}:{b}-{b + rnd.choice([1000, 1_000_000, 30_000_000])}" for _ in range(rnd.randint(1, 3)))
                keep = ro.keep_mask(sexp, reg)
                try:
                    g = duckhts_amd.read_bam(sdata, region=reg, index=bai, max_blocks=rnd.choice([0, 3]))
                    if keep is None or g["n_rows"] != int(keep.sum()) or g["QNAME"] != [q for q, k in zip(sexp["QNAME"], keep) if k]:
                        msgs.append(f"indexed region {reg}")
                except duckhts_amd.DhtsError:
                    if keep is not None and keep.any():
                        msgs.append(f"indexed region {reg} raised")
            ni = rnd.choice([10, 2000])
            tid = np.array([rnd.randrange(len(names)) for _ in range(ni)], np.int32)
            beg = np.array([rnd.randrange(0, 60_000_000) for _ in range(ni)], np.int64)
            end = beg + np.array([rnd.choice([0, 1, 500, 100_000, 10_000_000]) for _ in range(ni)], np.int64)
            eo = ro.overlap_join(sexp, tid, beg, end)
            go = duckhts_amd.read_bam(sdata, overlap=(tid, beg, end), max_blocks=rnd.choice([0, 4]))
            if len(go["OVERLAPS"]) != len(eo) or any(np.sort(a).tolist() != b.tolist() for a, b in zip(go["OVERLAPS"], eo)):
                msgs.append("overlap join")
            # ---- BCF block-range shards: hand-off chain + POS column ----
            ns = len(bcf_cases.SAMPLES)
            bdata = W.bcf_bytes(bcf_cases.std_header(), bcf_cases.fuzz_records(seed, rnd.choice([300, 2500]), ns), payload=rnd.choice([777, 4000, 30000]))
            eb = orc.bcf_read(bdata)
            c2 = duckhts_amd.Context(0); c2.open(bdata); nbb = c2.bgzf_index(); c2.close()
            w2 = rnd.randint(2, 5)
            cuts = [nbb * r // w2 for r in range(w2 + 1)]
            spans, poss = [], []
            for r in range(w2):
                if cuts[r] == cuts[r + 1]:
                    continue
                g = duckhts_amd.read_bcf(bdata, block_range=(cuts[r], cuts[r + 1], r > 0), max_blocks=rnd.choice([0, 2]), columns=["POS"])
                if g["n_rows"]:                     # (a shard of empty blocks, e.g. the EOF marker, owns no record)
                    spans.append((g["first_rec_uoff"], g["end_uoff"], g["n_rows"])); poss.append(g["by_name"]["POS"]["fixed"])
            try:
                tot = duckhts_amd.check_handoff(spans)
            except Exception as e:
                tot = f"hand-off failed: {e}"
            if tot != eb["n_rows"] or not np.array_equal(np.concatenate(poss) if poss else np.zeros(0), eb["by_name"]["POS"]["fixed"]):
                msgs.append(f"bcf {w2}-way shards: {tot} vs {eb['n_rows']}")
            # ---- BCF region := 'a,b' (chained union of single regions), wide or tidy ----
            tidy = rnd.random() < 0.3
            ebt = orc.bcf_read(bdata, tidy)
            c3 = duckhts_amd.Context(0); c3.open(bdata); c3.bgzf_index()
            contigs = [x.decode() if x else "\x01" for x in duckhts_amd.BcfScan(c3, tidy).contigs]
            c3.close()
            pos = eb["by_name"]["POS"]["fixed"].astype(np.int64)
            pmax = int(pos.max()) if len(pos) else 1000
            reg = ",".join(f"{rnd.choice(contigs)}:{b}-{b + rnd.choice([0, 10, 1000, pmax])}" for b in [rnd.randrange(1, pmax + 2) for _ in range(rnd.randint(1, 3))])
            rows_ = ro.bcf_region_rows(ebt, contigs, reg, ebt["n_samples"] if tidy and ebt["n_samples"] else 1)
            want = orc.bcf_take_rows(ebt, rows_)
            gotr = duckhts_amd.read_bcf(bdata, tidy=tidy, region=reg, max_blocks=rnd.choice([0, 2]))
            d = orc.bcf_cols_diff(want, gotr)
            if d is not None:
                msgs.append(f"bcf region {reg}: {d}")
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  (shards {world}-way of {exp['n_rows']} rows; sorted file {sexp['n_rows']} rows, {ni} intervals)", flush=True)
            bad += bool(msgs)
            continue
        # ---- BAM core columns ----
        payload = rnd.choice([61, 300, 777, 4000, 20000, 65280])
        level = rnd.choice([0, 1, 6, 9])
        n = rnd.choice([50, 300, 1500])
        data = cases.case_basic(payload=payload, level=level, seed=seed, n=n)
        if args.corrupt:
            data = damage(data, rnd)
        exp = orc.bam_read(data)
        mb = rnd.choice([0, 1, 2, 5])
        try:
            got = duckhts_amd.read_bam(data, max_blocks=mb)
        except duckhts_amd.DhtsError as e:
            if not args.corrupt:
                raise
            # the header itself is damaged: the oracle must have produced nothing and flagged it
            got = {"n_rows": 0, "status": exp["status"]}     # (the oracle wrapper reports a header failure through its return code)
            for k in duckhts_amd.BAM_COLUMNS:
                got[k] = []
            if exp["n_rows"] != 0:
                msgs.append(f"bam open failed on the GPU ({e}) but the oracle read {exp['n_rows']} rows status {exp['status']}")
        if (got["status"] < 0) != (exp["status"] < 0):
            msgs.append(f"bam status {got['status']} vs {exp['status']}")
        if got["n_rows"] != exp["n_rows"]:
            msgs.append(f"bam n_rows {got['n_rows']} != {exp['n_rows']}")
        else:
            for k in duckhts_amd.BAM_COLUMNS:
                if list(got[k]) != list(exp[k]):
                    msgs.append(f"bam column {k}")
        # region
        names = [bytes(x).decode() for x in exp["ref_names"]]
        if names and exp["n_rows"] and not args.corrupt:
            nm = rnd.choice(names)
            b = rnd.randrange(1, 5000)
            reg = f"{nm}:{b}-{b + rnd.randrange(1, 20000)}" if rnd.random() < 0.7 else nm
            keep = ro.keep_mask(exp, reg)
            try:
                g2 = duckhts_amd.read_bam(data, region=reg, max_blocks=mb)
                if keep is None or g2["n_rows"] != int(keep.sum()) or g2["QNAME"] != [q for q, k in zip(exp["QNAME"], keep) if k]:
                    msgs.append(f"bam region {reg}")
            except duckhts_amd.DhtsError:
                if keep is not None:
                    msgs.append(f"bam region {reg} raised")
        if args.corrupt:
            # BCF under damage, then next seed (the tag views add nothing here)
            ns = rnd.choice([0, len(bcf_cases.SAMPLES)])
            hdr = bcf_cases.std_header() if ns else bcf_cases.std_header(samples=())
            bdata = damage(W.bcf_bytes(hdr, bcf_cases.fuzz_records(seed, rnd.choice([100, 800]), ns), payload=rnd.choice([777, 4000, 65280])), rnd)
            eb = orc.bcf_read(bdata)
            if eb["status"] <= -100:            # header-level failure
                eb = None
            try:
                gb = duckhts_amd.read_bcf(bdata, max_blocks=rnd.choice([0, 1, 3]))
            except duckhts_amd.DhtsError:
                gb = None
            if (eb is None) != (gb is None):
                msgs.append(f"bcf open: oracle {'fails' if eb is None else 'ok'}, gpu {'fails' if gb is None else 'ok'}")
            elif eb is not None:
                d = orc.bcf_cols_diff(eb, gb)
                if d is not None:
                    msgs.append("bcf: " + d)
            done += 1
            print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  (damaged: bam {exp['n_rows']} rows status {exp['status']}, bcf {eb['n_rows'] if eb else 'open fails'})", flush=True)
            bad += bool(msgs)
            continue
        # ---- tags ----
        tdata = tag_cases.fuzz(seed=seed, n=rnd.choice([200, 1500]), payload=rnd.choice([500, 3000, 30000]))
        et = orc.bam_read_std_tags(tdata)
        gt = duckhts_amd.read_bam(tdata, std_tags_cols=list(range(56)), max_blocks=rnd.choice([0, 3]))
        d = orc.bcf_cols_diff(et, gt["tags"])
        if d is not None:
            msgs.append("std tags: " + d)
        excl = rnd.random() < 0.5
        ea = orc.bam_read_aux_map(tdata, excl)
        ga = duckhts_amd.read_bam(tdata, aux_map="exclude_standard" if excl else "all", max_blocks=rnd.choice([0, 4]))
        d = orc.bcf_cols_diff({"n_rows": ea["n_rows"], "cols": ea["cols"]}, ga["aux"])
        if d is not None:
            msgs.append("aux map: " + d)
        # ---- BCF ----
        ns = rnd.choice([0, len(bcf_cases.SAMPLES)])
        hdr = bcf_cases.std_header() if ns else bcf_cases.std_header(samples=())
        recs = bcf_cases.fuzz_records(seed, rnd.choice([100, 800, 2500]), ns)
        tidy = rnd.random() < 0.4
        bdata = W.bcf_bytes(hdr, recs, payload=rnd.choice([777, 4000, 65280]))
        eb = orc.bcf_read(bdata, tidy=tidy)
        gb = duckhts_amd.read_bcf(bdata, tidy=tidy, max_blocks=rnd.choice([0, 1, 3]))
        d = orc.bcf_cols_diff(eb, gb)
        if d is not None:
            msgs.append("bcf: " + d)
        done += 1
        print(f"seed {seed}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  (bam {exp['n_rows']} rows payload {payload} level {level}, bcf {eb['n_rows']} rows{' tidy' if tidy else ''})", flush=True)
        bad += bool(msgs)
    print(f"soak: {done} seeds, {bad} with mismatches, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
