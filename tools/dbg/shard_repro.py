import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, orc, duckhts_amd
import numpy as np
for seed in map(int, sys.argv[1:]):
    rnd = random.Random(seed)
    payload = rnd.choice([300, 777, 4000, 20000]); level = rnd.choice([1, 6]); n = rnd.choice([300, 1500])
    data = cases.case_basic(payload=payload, level=level, seed=seed, n=n)
    exp = orc.bam_read(data)
    world = rnd.randint(2, 6)
    ctx = duckhts_amd.Context(0); ctx.open(data); nb = ctx.bgzf_index(); coff, clen, isize, st = ctx.bgzf_table(nb); ctx.close()
    print(f"seed {seed}: payload {payload} level {level} n {n} blocks {nb} world {world} file {len(data)}")
    tot = 0
    for rank in range(world):
        mb = rnd.choice([0, 2])
        b0, b1 = duckhts_amd.shard_cut(coff, len(data), rank, world)
        g = duckhts_amd.read_bam(data, shard=(rank, world), max_blocks=mb)
        first = g["QNAME"][0] if g["n_rows"] else None
        last = g["QNAME"][-1] if g["n_rows"] else None
        i0 = exp["QNAME"].index(first) if first else None
        print(f"   rank {rank}: blocks [{b0},{b1}) mb {mb} rows {g['n_rows']} status {g['status']} first row = oracle row {i0}, expected start row {tot}")
        tot += g["n_rows"]
    print("   total", tot, "expected", exp["n_rows"])
