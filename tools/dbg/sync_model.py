#!/usr/bin/env python3
"""sync_model.py -- design study for the workgroup-per-block inflate: on real DEFLATE blocks, measure
 (a) how many rounds the self-synchronising sub-stream decode needs for N lanes,
 (b) the critical path (in lockstep symbol iterations) of resolving LZ77 matches when every lane walks its own sub-stream,
 (c) the same when matches are resolved in position batches."""
import collections
import struct
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from deflate_stats import Bits, mkdec, LBASE, LEXT, DBASE, DEXT


def lut(lens, maxbits=15):
    """(len, code)->sym dict to a direct table on `maxbits` reversed bits: idx -> (sym, len) or None"""
    tab = mkdec(lens); t = [None] * (1 << maxbits)
    for (L, code), s in tab.items():
        r = int(format(code, f"0{L}b")[::-1], 2)
        for hi in range(1 << (maxbits - L)):
            t[r | (hi << L)] = (s, L)
    return t


def parse_header(b):
    last = b.take(1); typ = b.take(2)
    if typ == 0:
        return last, typ, None, None
    if typ == 1:
        return last, typ, [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, [5] * 32
    hl = b.take(5) + 257; hd = b.take(5) + 1; hc = b.take(4) + 4
    cl = [0] * 19
    for i in range(hc):
        cl[[16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15][i]] = b.take(3)
    ct = mkdec(cl); lens = []
    while len(lens) < hl + hd:
        code = 0
        for L in range(1, 8):
            code = (code << 1) | b.take(1)
            if (L, code) in ct:
                s = ct[(L, code)]; break
        if s < 16: lens.append(s)
        elif s == 16: lens += [lens[-1]] * (3 + b.take(2))
        elif s == 17: lens += [0] * (3 + b.take(3))
        else: lens += [0] * (11 + b.take(7))
    return last, typ, lens[:hl], lens[hl:hl + hd]


def step(v, pos, lt, dt):
    """decode ONE unit (literal | match | EOB) at bit `pos`; returns (newpos, kind, a, b) or None if invalid"""
    e = lt[(v >> pos) & 0x7fff]
    if e is None: return None
    s, L = e; pos += L
    if s < 256: return pos, 0, s, 0
    if s == 256: return pos, 2, 0, 0
    j = s - 257
    if j >= 29: return None
    ln = LBASE[j] + ((v >> pos) & ((1 << LEXT[j]) - 1)); pos += LEXT[j]
    e = dt[(v >> pos) & 0x7fff]
    if e is None: return None
    d, L = e; pos += L
    if d >= 30: return None
    dist = DBASE[d] + ((v >> pos) & ((1 << DEXT[d]) - 1)); pos += DEXT[d]
    return pos, 1, ln, dist


def study(payload, nlanes, sub_bits_fixed=None):
    b = Bits(payload); total_bits = len(payload) * 8
    res = []
    out_base = 0
    while True:
        last, typ, ll, dl = parse_header(b)
        if typ == 0:
            b.pos = (b.pos + 7) & ~7; ln = b.take(16); b.take(16); b.pos += 8 * ln; out_base += ln
            if last: break
            continue
        lt, dt = lut(ll), lut(dl)
        start = b.pos
        # the true unit sequence
        units = []; pos = start
        while True:
            r = step(b.v, pos, lt, dt)
            units.append((pos,) + r[1:]); pos = r[0]
            if r[1] == 2: break
        end = pos
        true_pos = {u[0] for u in units}
        # ---- (a) sub-stream sync ----
        S = sub_bits_fixed or max(64, -(-(total_bits - start) // nlanes))
        nl_used = -(-(end - start) // S)
        bounds = [start + i * S for i in range(nlanes + 1)]
        starts = list(bounds[:nlanes]); ends = [None] * nlanes
        dirty = [True] * nlanes; rounds = 0; work = 0
        while any(dirty[:nl_used]):
            rounds += 1; new_starts = list(starts); nd = [False] * nlanes
            waves_active = set()
            for i in range(nl_used):
                if not dirty[i]: continue
                waves_active.add(i // 64)
                pos = starts[i]; nsym = 0
                while pos is not None and pos < bounds[i + 1] and pos < total_bits:
                    r = step(b.v, pos, lt, dt); nsym += 1
                    if r is None: pos = None; break
                    pos = r[0]
                    if r[1] == 2: pos = ('eob', pos); break
                e = pos
                if e != ends[i]:
                    ends[i] = e
                    if i + 1 < nlanes and not isinstance(e, tuple) and e is not None:
                        if new_starts[i + 1] != e: new_starts[i + 1] = e; nd[i + 1] = True
            work += len(waves_active)
            starts = new_starts; dirty = nd
        ok = all(starts[i] in true_pos for i in range(nl_used))
        # ---- (b) lockstep critical path: lane = sub-stream with verified start; T[unit] ----
        # ownership: unit k belongs to lane i if starts[i] <= pos < starts[i+1]
        import bisect
        sp = [starts[i] for i in range(nl_used)]
        opos = []; o = out_base
        for u in units:
            opos.append(o); o += 1 if u[1] == 0 else (u[2] if u[1] == 1 else 0)
        # T_fin[k]: iteration at which unit k completes. lane sequential: T >= T_prev_in_lane + 1; match: >= T of the unit producing last source byte + 1
        T = [0] * len(units); lane_of = [bisect.bisect_right(sp, u[0]) - 1 for u in units]
        prev_in_lane = {}
        per_lane_syms = collections.Counter(lane_of)
        for k, u in enumerate(units):
            i = lane_of[k]; t = prev_in_lane.get(i, 0) + 1
            if u[1] == 1:
                t += 1                                   # a match costs two Huffman symbols
                src_end = min(opos[k], opos[k] - u[3] + u[2])  # exclusive end of the source bytes that are not own output
                if src_end > out_base:
                    kk = bisect.bisect_right(opos, src_end - 1) - 1     # unit producing byte src_end-1
                    if lane_of[kk] != i: t = max(t, T[kk] + 1)
            T[k] = t; prev_in_lane[i] = t
        crit = max(T); ideal = max(per_lane_syms.values())
        # ---- (c) position batches: batch of BW output bytes, lane = 32-byte cell; rounds per batch = longest in-batch chain ----
        def batches(bw, cell):
            tot_rounds = 0; nb = 0
            k0 = 0
            lo = out_base
            while lo < o:
                hi = lo + bw
                depth = {}; mx = 0; lane_last = {}
                for k in range(bisect.bisect_left(opos, lo), bisect.bisect_left(opos, hi)):
                    u = units[k]
                    if u[1] != 1: continue
                    cellid = (opos[k] - lo) // cell
                    d = lane_last.get(cellid, 0) + 1
                    src_end = min(opos[k], opos[k] - u[3] + u[2])
                    if src_end > lo:
                        # depends on matches covering source bytes inside this batch: take the max depth of matches overlapping [src, src_end)
                        s0 = opos[k] - u[3]
                        k1 = bisect.bisect_right(opos, max(s0, lo)) - 1; k2 = bisect.bisect_right(opos, src_end - 1) - 1
                        for kk in range(k1, k2 + 1):
                            if units[kk][1] == 1 and kk in depth and (opos[kk] - lo) // cell != cellid: d = max(d, depth[kk] + 1)
                    depth[k] = d; lane_last[cellid] = d; mx = max(mx, d)
                tot_rounds += mx; nb += 1; lo = hi
            return nb, tot_rounds
        res.append(dict(units=len(units), bits=end - start, S=S, lanes_used=nl_used, sync_rounds=rounds, wave_passes=work, ok=ok, crit=crit, ideal=ideal,
                        b8k=batches(8192, 32), b16k=batches(16384, 64), b4k=batches(4096, 16)))
        out_base = o
        b.pos = end
        if last: break
    return res


def main():
    data = open(sys.argv[1], "rb").read(); maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 4; skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    p = 0; n = 0
    while p + 18 <= len(data) and n < maxb + skip:
        bl = struct.unpack_from("<H", data, p + 16)[0] + 1
        if n >= skip and bl > 28:
            for nl in (256, 512):
                for r in study(data[p + 18:p + bl - 8], nl):
                    print(nl, r)
        p += bl; n += 1


if __name__ == "__main__":
    main()
