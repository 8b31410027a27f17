"""deflate_sync_distance.py file.bam -- how many bits a DEFLATE decoder started at a random bit needs to fall into step with the true
symbol sequence (one large dynamic block of the file): sizing input for pass 0 of bgzf_huff_decode_wave (HW_SYNC_W)."""
import os, sys, struct, random, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import deflate_stats as ds
data = open(sys.argv[1],'rb').read()
# take BGZF block #5
p=0; n=0
while n < 5:
    bl = struct.unpack_from('<H', data, p+16)[0]+1; p += bl; n += 1
bl = struct.unpack_from('<H', data, p+16)[0]+1
payload = data[p+18:p+bl-8]
b = ds.Bits(payload)
last=b.take(1); typ=b.take(2); assert typ==2
hl=b.take(5)+257; hd=b.take(5)+1; hc=b.take(4)+4
cl=[0]*19
for i in range(hc): cl[[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15][i]]=b.take(3)
ct=ds.mkdec(cl); lens=[]
while len(lens)<hl+hd:
    s,_=ds.dec(b,ct)
    if s<16: lens.append(s)
    elif s==16: lens+=[lens[-1]]*(3+b.take(2))
    elif s==17: lens+=[0]*(3+b.take(3))
    else: lens+=[0]*(11+b.take(7))
ll,dl=lens[:hl],lens[hl:hl+hd]
lt,dt=ds.mkdec(ll),ds.mkdec(dl)
start=b.pos
def unit(b):
    """decode one unit; returns False on invalid/eob"""
    try:
        s,L=ds.dec(b,lt)
    except ValueError: return None
    if s<256: return 'L'
    if s==256: return None
    j=s-257
    if j>=29: return None
    b.take(ds.LEXT[j])
    try: d,L2=ds.dec(b,dt)
    except ValueError: return None
    if d>=30: return None
    b.take(ds.DEXT[d]); return 'M'
# true boundaries
true=set(); b.pos=start; kinds={}
while True:
    true.add(b.pos); q=b.pos
    k=unit(b)
    if k is None: break
    kinds[q]=k
end=max(true); print('units',len(true),'bits',end-start)
random.seed(1); hist=collections.Counter(); fails=[]
T=3000
for t in range(T):
    s0=random.randrange(start, end-4000)
    b.pos=s0; steps=0; ok=False
    while b.pos - s0 < 3000:
        if b.pos in true: ok=True; break
        if unit(b) is None: break
        steps+=1
    if ok: hist[min((b.pos-s0)//128,20)]+=1
    else: fails.append((s0, b.pos-s0))
print('sync distance (bits/128) histogram', sorted(hist.items()))
print('never synced within 3000 bits or invalid:', len(fails), fails[:10])
