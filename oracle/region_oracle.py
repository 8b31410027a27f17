"""region_oracle.py -- CPU restatement (numpy) of the read_bam region semantics.  TEST INFRASTRUCTURE ONLY (see dhts_oracle.h).

Follows, by reading: src/bam_reader.c:318-348 (comma split with strtok), htslib region.c:177-260 hts_reglist_create ("." / "*" /
unknown names skipped, per-tid sort + merge region.c:100-120), hts.c:3995-4150 hts_parse_region and 3884-3940 hts_parse_decimal with
thousands separators, the overlap test of hts_itr_multi_next hts.c:4575-4592 (end > iv.beg and iv.end > beg) with
beg = pos, end = bam_endpos (sam.c:668-673).  Valid for coordinate-sorted, correctly indexed files, where the index only
accelerates; pinned by test/sql/duckhts.test:139-161, 610-618 (18 / 2 / dedup / index_path).
"""
import re

import numpy as np

POS_MAX = (1 << 63) - 1


def parse_decimal(s):
    """-> (value, rest) like hts_parse_decimal(HTS_PARSE_THOUSANDS_SEP); (None, s) when no digits."""
    m = re.match(r"\s*([+-]?)([0-9,]*)(\.[0-9]*)?", s)
    sign, ip, fp = m.group(1), m.group(2) or "", (m.group(3) or "")[1:]
    digits = ip.replace(",", "") + fp
    if not digits:
        return None, s
    rest = s[m.end():]
    e = 0
    m2 = re.match(r"[eE]([+-]?[0-9]*)", rest)
    if m2:
        e = int(m2.group(1) or 0) if m2.group(1) not in ("", "+", "-") else 0
        rest = rest[m2.end():]
    elif rest[:1] in "kK" and rest[:1]:
        e, rest = 3, rest[1:]
    elif rest[:1] in "mM" and rest[:1]:
        e, rest = 6, rest[1:]
    elif rest[:1] in "gG" and rest[:1]:
        e, rest = 9, rest[1:]
    n = int(digits)
    e -= len(fp)
    n = n * 10 ** e if e >= 0 else n // 10 ** (-e)
    return (-n if sign == "-" else n), rest


def parse_region(names, tok):
    """-> (tid, beg, end) or None (unknown reference / malformed)."""
    def getid(nm):
        return names.index(nm) if nm in names else -1
    if tok.startswith("{"):
        close = tok.find("}")
        if close < 0:
            return None
        name = tok[1:close]
        tid = getid(name)
        if tok[close + 1:close + 2] != ":":
            return (tid, 0, POS_MAX) if tid >= 0 else None
        coords = tok[close + 2:]
    else:
        colon = tok.rfind(":")
        if colon < 0:
            tid = getid(tok)
            return (tid, 0, POS_MAX) if tid >= 0 else None
        if getid(tok) >= 0:
            return None if getid(tok[:colon]) >= 0 else (getid(tok), 0, POS_MAX)
        tid = getid(tok[:colon])
        coords = tok[colon + 1:]
    if tid < 0:
        return None
    v, rest = parse_decimal(coords)
    beg = (0 if v is None else v) - 1
    if beg < 0:
        if beg != -1 and rest[:1] == "-" and coords != "":
            return None
        if rest[:1].isdigit() or rest == "" or rest[:1] == ",":
            return (tid, 0, POS_MAX if beg == -1 else -(beg + 1))
        if beg < -1:
            return None
    if rest == "":
        end = POS_MAX
    elif rest[0] == "-":
        v2, r2 = parse_decimal(rest[1:])
        end = 0 if v2 is None else v2
        if r2 not in ("",) and r2[:1] != ",":
            return None
    else:
        return None
    if end == 0:
        end = POS_MAX
    if beg >= end:
        return None
    return tid, beg, end


def reglist(names, region_string):
    """-> (per-tid merged interval lists, all_flag, nocoor_flag) or None when nothing usable remains."""
    per, allf, noc, usable = {}, False, False, 0
    for tok in region_string.split(","):
        if tok == "":
            continue
        if tok == ".":
            allf, usable = True, usable + 1
        elif tok == "*":
            noc, usable = True, usable + 1
        else:
            r = parse_region(names, tok)
            if r is None:
                continue
            per.setdefault(r[0], []).append((r[1], r[2]))
            usable += 1
    if not usable:
        return None
    for t, v in per.items():
        v.sort()
        out = [list(v[0])]
        for b, e in v[1:]:
            if out[-1][1] < b:
                out.append([b, e])
            elif out[-1][1] < e:
                out[-1][1] = e
        per[t] = out
    return per, allf, noc


def endpos(pos0, flag, cigar):
    """bam_endpos: pos + reference length of the CIGAR (M D N = X), 1 if 0 or the read is unmapped; cigar = text column"""
    rlen = 0
    if not (flag & 4) and cigar != b"*":
        for ln, op in re.findall(rb"([0-9]+)(.)", cigar):
            if op in b"MDN=X":
                rlen += int(ln)
    return pos0 + (rlen or 1)


def keep_mask(table, region_string):
    """table = orc.bam_read(...) result -> boolean numpy mask of the rows read_bam(region := ...) returns (file order).
    A string without any non-empty token ('' or ',,') is no region at all: parse_regions' strtok split yields n_regions = 0 and the
    reader does a plain scan (src/bam_reader.c:319-345, 571).  None = tokens were given but none names a known reference."""
    names = [bytes(x).decode() for x in table["ref_names"]]
    if not any(region_string.split(",")):
        return np.ones(table["n_rows"], bool)
    rl = reglist(names, region_string)
    if rl is None:
        return None
    per, allf, noc = rl
    n = table["n_rows"]
    keep = np.zeros(n, bool)
    for i in range(n):
        tid = int(table["tid"][i])
        if allf:
            keep[i] = True
        elif tid < 0:
            keep[i] = noc
        elif tid in per:
            beg = int(table["POS"][i]) - 1
            end = endpos(beg, int(table["FLAG"][i]), table["CIGAR"][i])
            keep[i] = any(end > b and e > beg for b, e in per[tid])
    return keep


def bcf_region_rows(table, contigs, region_string, tidy_reps=1):
    """read_bcf(region := 'a,b'): chained single-region iterators in the given order (src/bcf_reader.c:1327-1345, 935-953; unknown
    regions are skipped; overlapping regions repeat rows).  Record test of hts_itr_next (hts.c:4287-4300): rid == tid and
    end > beg_q and end_q > beg with beg = pos, end = pos + rlen (bcf_readrec vcf.c:2267-2276).
    table = orc.bcf_read(...) result (its "rec" arrays); -> list of ROW indices in output order."""
    rec = table["rec"]
    out = []
    if not any(region_string.split(",")):            # parse_regions_duckdb (bcf_reader.c:286-328): no non-empty token = no region = plain scan
        return list(range(len(rec["rid"]) * tidy_reps))
    for tok in region_string.split(","):
        if tok == "":
            continue
        if tok == ".":
            idx = np.arange(len(rec["rid"]))
        else:
            r = parse_region(contigs, tok)
            if r is None:
                continue
            tid, b, e = r
            beg, end = rec["pos0"], rec["pos0"] + rec["rlen"]
            idx = np.nonzero((rec["rid"] == tid) & (end > b) & (e > beg))[0]
        for i in idx:
            out.extend(range(int(i) * tidy_reps, int(i) * tidy_reps + tidy_reps))
    return out


def tabix_names(index_bytes):
    """sequence names of a tabix index in index order (tbx.c:552-597 index_load: the 28-byte header + names behind n_ref of a .tbi,
    or in the aux block of a .csi); None when the index has no tabix header"""
    import gzip
    import struct
    d = gzip.decompress(index_bytes) if index_bytes[:2] == b"\x1f\x8b" else bytes(index_bytes)
    if d[:4] == b"TBI\x01":
        m = d[8:]
    elif d[:4] == b"CSI\x01":
        (l_aux,) = struct.unpack_from("<i", d, 12)
        m = d[16:16 + l_aux]
    else:
        return None
    if len(m) < 28:
        return None
    (l_nm,) = struct.unpack_from("<i", m, 24)
    return [x.decode() for x in m[28:28 + l_nm].split(b"\x00")[:-1]]


def vcf_text_region_rows(table, chrom, names, region_string, tidy_reps=1):
    """read_bcf(region := ...) on bgzipped VCF TEXT: tbx_itr_querys per region token (src/bcf_reader.c:938-939, 1327-1345) -- the name is
    looked up among the INDEX's sequences (tbx_name2id), unknown ones give no iterator and are skipped -- then hts_itr_next's test on the
    interval tbx_parse1 computes for the line (tbx.c:96-312): same sequence and end > beg_q and end_q > beg, beg = POS - 1 clamped at 0,
    end = what bcf_oracle.c tbx_vcf_end returns (stored by the oracle as pos0 + rlen of the text record).
    chrom = CHROM per RECORD (bytes); names = tabix_names(index)."""
    rec = table["rec"]
    out = []
    if not any(region_string.split(",")):
        return list(range(len(rec["rid"]) * tidy_reps))
    chrom = np.array([c if c is not None else b"" for c in chrom], dtype=object)
    for tok in region_string.split(","):
        if tok == "":
            continue
        if tok == ".":
            idx = np.arange(len(rec["rid"]))
        else:
            r = parse_region(names, tok)
            if r is None:
                continue
            tid, b, e = r
            beg, end = np.maximum(rec["pos0"], 0), rec["pos0"] + rec["rlen"]
            same = np.array([c == names[tid].encode() for c in chrom], bool) if len(chrom) else np.zeros(0, bool)
            idx = np.nonzero(same & (end > b) & (e > beg))[0]
        for i in idx:
            out.extend(range(int(i) * tidy_reps, int(i) * tidy_reps + tidy_reps))
    return out


def overlap_join(table, tid, beg, end):
    """Interval overlap join of read_bam rows with intervals (tid, beg, end) (half-open, 0-based, tid = header index).
    Semantics = cgranges cr_overlap (ref: third_party/cgranges/cgranges.c:255-297): interval i of the read's contig is reported iff
    st_i < en && st < en_i, where [st, en) = [pos, bam_endpos) of the read (htslib sam.c:668-673).
    table = orc.bam_read(...) result -> list (per row) of sorted interval ids."""
    tid = np.asarray(tid, np.int64); beg = np.asarray(beg, np.int64); end = np.asarray(end, np.int64)
    by_tid = {}
    for t in np.unique(tid):
        idx = np.nonzero(tid == t)[0]
        by_tid[int(t)] = (idx, beg[idx], end[idx])
    out = []
    for i in range(table["n_rows"]):
        t = int(table["tid"][i])
        if t < 0 or t not in by_tid:
            out.append(np.zeros(0, np.int64))
            continue
        st = int(table["POS"][i]) - 1
        en = endpos(st, int(table["FLAG"][i]), table["CIGAR"][i])
        idx, b, e = by_tid[t]
        out.append(np.sort(idx[(b < en) & (st < e)]))
    return out


def _strtoll_whole(f):
    """parse_int64_span_local (src/interval_udf.c:127-139): strtoll(field, &end, 10) with *end == 0, else None (NULL).  C's strtoll:
    white space, one optional sign, decimal digits, saturating at the ends of int64."""
    if not f:
        return None
    i = 0
    while i < len(f) and f[i:i + 1] in (b" ", b"\t", b"\n", b"\v", b"\f", b"\r"):
        i += 1
    neg = False
    if i < len(f) and f[i:i + 1] in (b"+", b"-"):
        neg = f[i:i + 1] == b"-"
        i += 1
    d = f[i:]
    if not d or not d.isdigit() or any(c > 0x39 for c in d):
        return None
    v = -int(d) if neg else int(d)
    return max(-(1 << 63), min((1 << 63) - 1, v))


def bed_rows(text):
    """The rows read_bed returns for BED text (src/interval_udf.c:330-342 next_bed_line over hts_getline: lines lose their '\n' and a
    '\r' in front of it; empty lines and those starting with '#', 'track', 'browser' are skipped, 141-147; fewer than 3 tab-delimited
    fields raises, 358-365; a C string ends at a NUL).  -> list of (chrom bytes, start or None, end or None)."""
    rows = []
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for ln in lines:
        if ln.endswith(b"\r"):
            ln = ln[:-1]
        ln = ln.split(b"\0", 1)[0]
        if not ln or ln[:1] == b"#" or ln[:5] == b"track" or ln[:7] == b"browser":
            continue
        f = ln.split(b"\t")
        if len(f) < 3:
            raise ValueError("read_bed: BED line has fewer than 3 tab-delimited fields")
        rows.append((f[0], _strtoll_whole(f[1]), _strtoll_whole(f[2])))
    return rows


def bed_join_intervals(rows, ref_names):
    """(tid, beg, end) of the join for read_bed rows: chrom -> index in the BAM header (first of equal names), -1 (never matches) when
    the header lacks it or start / end is NULL"""
    first = {}
    for i, nm in enumerate(ref_names):
        first.setdefault(bytes(nm), i)
    tid = [(-1 if (b is None or e is None) else first.get(bytes(c), -1)) for c, b, e in rows]
    return (np.array(tid, np.int32), np.array([b or 0 for _, b, _ in rows], np.int64), np.array([e or 0 for _, _, e in rows], np.int64))


def cgranges_overlap(lib_path, names, tid, beg, end, queries):
    """The REFERENCE's own cgranges (oracle/_ref/libcgranges.so, compiled from /root/reference by oracle/Makefile) through
    ctypes: cr_add every interval with label = its id, cr_index, then cr_overlap per query (name, st, en) -> sorted labels."""
    import ctypes as C
    L = C.CDLL(lib_path)
    L.cr_init.restype = C.c_void_p
    L.cr_add.restype = C.c_void_p
    L.cr_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
    L.cr_index.argtypes = [C.c_void_p]
    L.cr_destroy.argtypes = [C.c_void_p]
    L.cr_overlap.restype = C.c_int64
    L.cr_overlap.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]

    class Intv(C.Structure):                      # cr_intv_t (cgranges.h): x, y:31/rev:1, label
        _fields_ = [("x", C.c_uint64), ("y", C.c_uint32), ("label", C.c_int32)]

    class Cr(C.Structure):                        # cgranges_t head: n_r, m_r, r
        _fields_ = [("n_r", C.c_int64), ("m_r", C.c_int64), ("r", C.POINTER(Intv))]

    cr = L.cr_init()
    for i in range(len(tid)):
        L.cr_add(cr, names[int(tid[i])].encode(), int(beg[i]), int(end[i]), i)
    L.cr_index(cr)
    r = C.cast(cr, C.POINTER(Cr)).contents.r
    b = C.POINTER(C.c_int64)()
    mb = C.c_int64(0)
    out = []
    for name, st, en in queries:
        n = L.cr_overlap(cr, name.encode(), int(st), int(en), C.byref(b), C.byref(mb))
        out.append(np.sort(np.array([r[b[k]].label for k in range(n)], np.int64)))
    C.CDLL(None).free(b)
    L.cr_destroy(cr)
    return out
