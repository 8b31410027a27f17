import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, orc, duckhts_amd, bcf_cases, bcfwriter as W
import numpy as np
from duckhts_amd import synth
for seed in map(int, sys.argv[1:]):
    rnd = random.Random(seed)
    # replay the rng draws of soak --scans up to the BCF part
    rnd.choice([300, 777, 4000, 20000]); rnd.choice([1, 6]); rnd.choice([300, 1500])
    world = rnd.randint(2, 6)
    for rank in range(world): rnd.choice([0, 2])
    nrec = rnd.choice([3000, 20000, 60000])
    sdata = synth.bam_file(nrec, seed=seed); sexp = orc.bam_read(sdata); names = sexp["ref_names"]
    for _ in range(2):
        rnd.randrange(len(names)); rnd.randrange(1, 50_000_000)
        for _ in range(rnd.randint(1, 3)):
            rnd.randrange(len(names)); rnd.choice([1000, 1_000_000, 30_000_000])
        rnd.choice([0, 3])
    ni = rnd.choice([10, 2000])
    for _ in range(ni): rnd.randrange(len(names))
    for _ in range(ni): rnd.randrange(0, 60_000_000)
    for _ in range(ni): rnd.choice([0, 1, 500, 100_000, 10_000_000])
    rnd.choice([0, 4])
    ns = len(bcf_cases.SAMPLES)
    n = rnd.choice([300, 2500]); payload = rnd.choice([777, 4000, 30000])
    bdata = W.bcf_bytes(bcf_cases.std_header(), bcf_cases.fuzz_records(seed, n, ns), payload=payload)
    eb = orc.bcf_read(bdata)
    c2 = duckhts_amd.Context(0); c2.open(bdata); nbb = c2.bgzf_index(); c2.close()
    w2 = rnd.randint(2, 5)
    cuts = [nbb * r // w2 for r in range(w2 + 1)]
    print(f"seed {seed}: bcf n {n} payload {payload} blocks {nbb} cuts {cuts} oracle rows {eb['n_rows']} status {eb['status']}")
    for r in range(w2):
        if cuts[r] == cuts[r + 1]: continue
        mb = rnd.choice([0, 2])
        g = duckhts_amd.read_bcf(bdata, block_range=(cuts[r], cuts[r + 1], r > 0), max_blocks=mb, columns=["POS"])
        print(f"   rank {r}: blocks [{cuts[r]},{cuts[r+1]}) mb {mb} rows {g['n_rows']} status {g['status']} first {g['first_rec_uoff']} end {g['end_uoff']}")
