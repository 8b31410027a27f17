import sys, os, random, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases, orc, bcf_cases, bcfwriter as W
from soak import damage

def blocks(b):
    p = 0; out = []
    while p + 18 <= len(b):
        if b[p:p+4] != b"\x1f\x8b\x08\x04": break
        bl = struct.unpack_from("<H", b, p + 16)[0] + 1
        out.append((p, bl)); p += bl
    return out

def build(seed):
    rnd = random.Random(seed)
    payload = rnd.choice([61, 300, 777, 4000, 20000, 65280]); level = rnd.choice([0, 1, 6, 9]); n = rnd.choice([50, 300, 1500])
    clean = cases.case_basic(payload=payload, level=level, seed=seed, n=n)
    data = damage(clean, rnd)
    mb = rnd.choice([0, 1, 2, 5])
    ns = rnd.choice([0, len(bcf_cases.SAMPLES)])
    hdr = bcf_cases.std_header() if ns else bcf_cases.std_header(samples=())
    bclean = W.bcf_bytes(hdr, bcf_cases.fuzz_records(seed, rnd.choice([100, 800]), ns), payload=rnd.choice([777, 4000, 65280]))
    bdata = damage(bclean, rnd)
    bmb = rnd.choice([0, 1, 3])
    return dict(payload=payload, level=level, n=n, clean=clean, data=data, mb=mb, bclean=bclean, bdata=bdata, bmb=bmb)

def where(clean, data):
    bl = blocks(clean); res = []
    for i, (a, b) in enumerate(zip(clean, data)):
        if a != b:
            for k, (p, l) in enumerate(bl):
                if p <= i < p + l:
                    off = i - p
                    part = "header" if off < 18 else "crc" if off >= l - 8 and off < l - 4 else "isize" if off >= l - 4 else "payload"
                    res.append((i, k, off, l, part)); break
            else:
                res.append((i, None, None, None, "outside"))
    return res, len(bl)

if __name__ == "__main__":
    for seed in map(int, sys.argv[1:]):
        d = build(seed)
        w, nb = where(d["clean"], d["data"])
        e = orc.bam_read(d["data"]); ec = orc.bam_read(d["clean"])
        print(f"seed {seed}: BAM payload {d['payload']} level {d['level']} n {d['n']} mb {d['mb']} blocks {nb}; flips {w}; oracle rows {e['n_rows']} status {e['status']} (clean {ec['n_rows']})")
        w, nb = where(d["bclean"], d["bdata"])
        eb = orc.bcf_read(d["bdata"])
        print(f"          BCF mb {d['bmb']} blocks {nb}; flips {w}; oracle rows {eb['n_rows']} status {eb['status']}")
