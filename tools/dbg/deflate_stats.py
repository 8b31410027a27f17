#!/usr/bin/env python3
"""deflate_stats.py -- what the DEFLATE streams of a BGZF file look like (sizing input for the inflate kernels):
deflate blocks per BGZF block, header bits, Huffman symbols per block, code-length use, match length / distance spread."""
import collections
import struct
import sys

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits:
    def __init__(self, d):
        self.v = int.from_bytes(d, "little"); self.pos = 0
    def take(self, n):
        r = (self.v >> self.pos) & ((1 << n) - 1); self.pos += n; return r


def mkdec(lens):
    # canonical decode dict: (len, code) -> sym
    cnt = collections.Counter(l for l in lens if l)
    code = 0; first = {}
    for L in range(1, 16):
        code = (code + cnt.get(L - 1, 0)) << 1; first[L] = code
    nxt = dict(first); tab = {}
    for s, l in enumerate(lens):
        if l:
            tab[(l, nxt[l])] = s; nxt[l] += 1
    return tab


def dec(b, tab):
    code = 0
    for L in range(1, 16):
        code = (code << 1) | b.take(1)
        if (L, code) in tab:
            return tab[(L, code)], L
    raise ValueError("bad code")


def inflate_stats(payload, st):
    b = Bits(payload); out = 0; nblk = 0
    while True:
        p0 = b.pos
        last = b.take(1); typ = b.take(2); nblk += 1
        if typ == 0:
            b.pos = (b.pos + 7) & ~7; ln = b.take(16); b.take(16); b.pos += 8 * ln; out += ln; st["stored"] += 1
        else:
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8; dl = [5] * 32; st["fixed"] += 1
            else:
                hl = b.take(5) + 257; hd = b.take(5) + 1; hc = b.take(4) + 4
                cl = [0] * 19
                for i in range(hc):
                    cl[[16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15][i]] = b.take(3)
                ct = mkdec(cl); lens = []; ncl = 0
                while len(lens) < hl + hd:
                    s, _ = dec(b, ct); ncl += 1
                    if s < 16: lens.append(s)
                    elif s == 16: lens += [lens[-1]] * (3 + b.take(2))
                    elif s == 17: lens += [0] * (3 + b.take(3))
                    else: lens += [0] * (11 + b.take(7))
                ll, dl = lens[:hl], lens[hl:hl + hd]
                st["hdr_bits"].append(b.pos - p0); st["cl_syms"].append(ncl)
                st["maxlen_ll"][max(ll)] += 1; st["maxlen_d"][max(dl) if any(dl) else 0] += 1
            lt, dt = mkdec(ll), mkdec(dl)
            nsym = 0; bits0 = b.pos; run = 0; out0 = out
            while True:
                s, L = dec(b, lt); nsym += 1; st["ll_len"][L] += 1
                if s < 256:
                    out += 1; run += 1; st["nlit"] += 1
                elif s == 256:
                    break
                else:
                    st["litrun"][min(run, 20)] += 1; run = 0
                    j = s - 257; ln = LBASE[j] + b.take(LEXT[j])
                    d, L2 = dec(b, dt); nsym += 1; st["d_len"][L2] += 1
                    dist = DBASE[d] + b.take(DEXT[d])
                    st["mlen"][min(ln, 40)] += 1; st["nmatch"] += 1; st["mbytes"] += ln
                    st["dist"][dist.bit_length()] += 1
                    if dist < ln: st["overlap"] += 1
                    out += ln
            st["syms"].append(nsym); st["blk_bits"].append(b.pos - bits0); st["blk_out"].append(out - out0)
        if last:
            break
    st["dblocks"].append(nblk)
    return out


def main():
    data = open(sys.argv[1], "rb").read(); maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 12; skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    st = collections.defaultdict(int)
    for k in ("hdr_bits", "cl_syms", "syms", "blk_bits", "blk_out", "dblocks"): st[k] = []
    for k in ("ll_len", "d_len", "mlen", "dist", "litrun", "maxlen_ll", "maxlen_d"): st[k] = collections.Counter()
    p = 0; n = 0; tot_c = tot_u = 0
    while p + 18 <= len(data) and n < maxb + skip:
        bl = struct.unpack_from("<H", data, p + 16)[0] + 1
        if n >= skip and bl > 28:
            u = inflate_stats(data[p + 18:p + bl - 8], st); tot_c += bl; tot_u += u
        p += bl; n += 1
    print(f"BGZF blocks {len(st['dblocks'])}: comp {tot_c} B, inflated {tot_u} B; deflate blocks per BGZF block {st['dblocks']}")
    print(f"stored {st['stored']} fixed {st['fixed']} dynamic {len(st['hdr_bits'])}; header bits {st['hdr_bits']}; CL symbols {st['cl_syms']}")
    print(f"Huffman symbols per deflate block {st['syms']}\n bits {st['blk_bits']}\n out bytes {st['blk_out']}")
    tl = sum(st["ll_len"].values()); td = sum(st["d_len"].values())
    print("lit/len code length use %:", {k: round(100 * v / tl, 2) for k, v in sorted(st["ll_len"].items())})
    print("dist code length use %:", {k: round(100 * v / td, 2) for k, v in sorted(st["d_len"].items())})
    print("max code length lit/len", dict(st["maxlen_ll"]), "dist", dict(st["maxlen_d"]))
    print(f"literals {st['nlit']} matches {st['nmatch']} (avg len {st['mbytes'] / max(1, st['nmatch']):.1f}, overlapping {st['overlap']})")
    print("match len:", sorted(st["mlen"].items()))
    print("dist bit_length:", sorted(st["dist"].items()))
    print("literal run before a match:", sorted(st["litrun"].items()))


if __name__ == "__main__":
    main()
