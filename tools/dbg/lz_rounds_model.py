"""lz_rounds_model.py file.bam [blocks] -- dependency rounds of bgzf_lz_resolve's 64-token batches on a real DEFLATE stream (CPU model):
rounds per batch under the kernel's cell policy, a cheaper 'first pending destination' policy and the exact byte-level bound.  The figures
behind DESIGN 5c (84 % of a batch's matches are ready in the first round)."""
import os, sys, struct, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import deflate_stats as ds

def tokens_of(payload):
    """returns list of (lrun, mlen, dist) + trailing literals"""
    b = ds.Bits(payload); toks = []; run = 0
    while True:
        last = b.take(1); typ = b.take(2)
        if typ == 0:
            b.pos = (b.pos + 7) & ~7; ln = b.take(16); b.take(16); b.pos += 8 * ln; run += ln
        else:
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8; dl = [5] * 32
            else:
                hl = b.take(5) + 257; hd = b.take(5) + 1; hc = b.take(4) + 4
                cl = [0] * 19
                for i in range(hc):
                    cl[[16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15][i]] = b.take(3)
                ct = ds.mkdec(cl); lens = []
                while len(lens) < hl + hd:
                    s, _ = ds.dec(b, ct)
                    if s < 16: lens.append(s)
                    elif s == 16: lens += [lens[-1]] * (3 + b.take(2))
                    elif s == 17: lens += [0] * (3 + b.take(3))
                    else: lens += [0] * (11 + b.take(7))
                ll, dl = lens[:hl], lens[hl:hl + hd]
            lt, dt = ds.mkdec(ll), ds.mkdec(dl)
            while True:
                s, L = ds.dec(b, lt)
                if s < 256: run += 1
                elif s == 256: break
                else:
                    j = s - 257; ln = ds.LBASE[j] + b.take(ds.LEXT[j])
                    d, L2 = ds.dec(b, dt); dist = ds.DBASE[d] + b.take(ds.DEXT[d])
                    while run >= 511: toks.append((511, 0, 0)); run -= 511
                    toks.append((run, ln, dist)); run = 0
        if last: break
    return toks, run

def load(path, maxb, skip):
    data = open(path, 'rb').read(); p = 0; n = 0; out = []
    while p + 18 <= len(data) and n < maxb + skip:
        bl = struct.unpack_from('<H', data, p + 16)[0] + 1
        if n >= skip and bl > 28: out.append(tokens_of(data[p + 18:p + bl - 8]))
        p += bl; n += 1
    return out

def simulate(toks, W, policy):
    """returns (steps, rounds, list of ready counts)"""
    outpos = 0; steps = 0; rounds = 0; nm = 0
    for t0 in range(0, len(toks), W):
        batch = toks[t0:t0 + W]; steps += 1
        # positions
        pos = outpos; items = []
        for (lr, ml, di) in batch:
            md = pos + lr; items.append((md, ml, di)); pos = md + ml
        tot = pos - outpos
        pend = [i for i, (md, ml, di) in enumerate(items) if ml > 0]
        nm += len(pend)
        sh = 4
        while (tot >> sh) > 63: sh += 1
        while pend:
            rounds += 1
            ready = []
            if policy == 'cell':
                owed = 0
                for i in pend:
                    md, ml, di = items[i]; ms = md - di; sp = min(ml, di)
                    lo = (md - outpos) >> sh; hi = (md - outpos + ml - 1) >> sh
                    dmask = ((1 << (hi + 1)) - 1) & ~((1 << lo) - 1)
                    smask = 0
                    if ms + sp > outpos:
                        slo = (max(ms, outpos) - outpos) >> sh; shi = (ms + sp - 1 - outpos) >> sh
                        smask = ((1 << (shi + 1)) - 1) & ~((1 << slo) - 1)
                    if smask & owed == 0: ready.append(i)
                    owed |= dmask
            elif policy == 'first':
                fmd = items[pend[0]][0]
                for k, i in enumerate(pend):
                    md, ml, di = items[i]; ms = md - di; sp = min(ml, di)
                    if k == 0 or ms + sp <= fmd: ready.append(i)
            elif policy == 'exact':
                owed = []
                for i in pend:
                    md, ml, di = items[i]; ms = md - di; sp = min(ml, di)
                    if not any(ms < e and ms + sp > s for (s, e) in owed): ready.append(i)
                    owed.append((md, md + ml))
            rs = set(ready); pend = [i for i in pend if i not in rs]
        outpos = pos
    return steps, rounds, nm

if __name__ == '__main__':
    blocks = load(sys.argv[1], int(sys.argv[2]), 3)
    for W in (64, 128):
        for pol in ('cell', 'first', 'exact'):
            S = R = M = 0
            for toks, tail in blocks:
                s, r, m = simulate(toks, W, pol); S += s; R += r; M += m
            print(f"W={W} {pol:6s}: blocks {len(blocks)} steps/blk {S/len(blocks):.1f} rounds/blk {R/len(blocks):.1f} rounds/step {R/S:.2f} matches/blk {M/len(blocks):.0f}")
