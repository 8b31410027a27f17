import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, collections
import duckhts_amd, orc
from duckhts_amd import synth
arr, st = synth.bam_segment(300000, seed=42)
d = arr.tobytes()
exp = orc.bam_read(d); got = duckhts_amd.read_bam(d)
bad = [i for i in range(exp['n_rows']) if got['SEQ'][i] != exp['SEQ'][i]]
print('bad rows', len(bad), bad[:20])
posc = collections.Counter()
for i in bad[:2000]:
    a,b = got['SEQ'][i], exp['SEQ'][i]
    for k in range(len(b)):
        if a[k]!=b[k]: posc[k%16]+=1
print(posc)
for k in ['QNAME','CIGAR','QUAL','READ_GROUP_ID']:
    print(k, sum(1 for i in range(exp['n_rows']) if got[k][i]!=exp[k][i]))
i = bad[0]
ro = exp['rec_off'][i]
print('rec_off', ro, ro%16, ro%8, ro%4, 'row', i, 'seqoff', (150*i)%16)
print([ (exp['rec_off'][j]%8, (exp['rec_off'][j]+36+len(exp['QNAME'][j])+1+4)%8) for j in bad[:10]])
ok=[j for j in range(100) if j not in bad]
print([ (exp['rec_off'][j]%8) for j in ok[:20]])
