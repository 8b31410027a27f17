import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, collections
import duckhts_amd, orc
from duckhts_amd import synth
arr, st = synth.bam_segment(300000, seed=42)
d = arr.tobytes()
exp = orc.bam_read(d)
for rep in range(3):
    got = duckhts_amd.read_bam(d)
    for k in duckhts_amd.BAM_COLUMNS:
        bad = [i for i in range(exp['n_rows']) if got[k][i] != exp[k][i]]
        if bad: print(rep, k, 'bad rows', len(bad), bad[:12])
    print('rep', rep, 'done')
