"""replays the two --isize soak seeds that were not reported (round 2 final soak) and prints the per-batch status"""
import os, random, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, duckhts_amd, orc
import ctypes as C
for seed in (2000684, 2001161):
    rnd = random.Random(seed)
    payload = rnd.choice([61, 777, 4000, 65280]); n = rnd.choice([50, 300, 1500])
    data = bytearray(cases.case_basic(payload=payload, level=rnd.choice([0, 1, 6]), seed=seed, n=n))
    p, blocks = 0, []
    while p + 18 <= len(data) and data[p:p + 4] == b"\x1f\x8b\x08\x04":
        bl = struct.unpack_from("<H", data, p + 16)[0] + 1
        blocks.append((p, bl)); p += bl
    for _ in range(rnd.randint(1, 3)):
        k = rnd.randrange(1, len(blocks))
        at = blocks[k][0] + blocks[k][1] - 4
        old = struct.unpack_from("<I", data, at)[0]
        new = rnd.choice([old ^ (1 << rnd.randrange(32)), 0xFFFF0000 + old, 0xFFFFFFFF, 0x80000000 | old, 65537, 65536, old + 1, rnd.getrandbits(32)])
        struct.pack_into("<I", data, at, new & 0xFFFFFFFF)
        print("seed", seed, "block", k, "isize", old, "->", hex(new & 0xffffffff))
    mb = rnd.choice([0, 1, 2, 5])
    for force in (None, "lane", "wave"):
        if force: os.environ["DHTS_PHASE_A"] = force
        ctx = duckhts_amd.Context(0)
        ctx.open(bytes(data)); nb = ctx.bgzf_index()
        coff, clen, isize, st = ctx.bgzf_table(nb)
        print(" phase A", force, "blocks", nb, "isize", list(isize), "index status", st)
        ctx.bam_open()
        while True:
            b = ctx.next_batch(mb)
            print("   batch rows", b.n_rows, "status", b.status)
            if b.status != 0: break
        ctx.close()
