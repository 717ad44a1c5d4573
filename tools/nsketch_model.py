#!/usr/bin/env python3
"""The decomposition behind pgx_sketch_n.hip, as a Python model checked against the CPU oracle (CPU only, ~1 minute):

    mm_sketch(read with ambiguous bases) = for every run of unambiguous bases, in order: mm_sketch(run, as a read of its own -- started at
        f-(k-1), f = the position where the run length really reaches k given the k-mer state carried across the break) WITHOUT its last
        element, positions shifted; then the rightmost smallest finite entry among the last w entries of the whole read, if any

6,000 adversarial strings (random / short-period tandem / (AT)n / two-letter, 1..60 breaks incl. runs of breaks and breaks near the
ends), w in {5, 11, 24, 80}, k in {4, 6, 12, 16} -- at k = 4 one k-mer in sixteen is its own reverse complement, which stresses the
stale-k-mer rule.  Result of the committed run: 6000 trials, 0 mismatches.   usage: python tools/nsketch_model.py"""
import sys, numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import oracle_util as U
MM = U.MM_DTYPE if hasattr(U,'MM_DTYPE') else None
INF = (1<<64)-1
CODE = {65:0,67:1,71:2,84:3}
def hash64(key, mask):
    key = (~key + (key << 21)) & mask
    key = key ^ key >> 24
    key = ((key + (key << 3)) + (key << 8)) & mask
    key = key ^ key >> 14
    key = ((key + (key << 2)) + (key << 4)) & mask
    key = key ^ key >> 28
    key = (key + (key << 31)) & mask
    return key
def stream(seq, w, k):
    """entry stream of the true machine: list of (x or INF, pos, strand, run)"""
    mask=(1<<(2*k))-1; top=2*(k-1); fwd=rev=0; run=0; out=[]
    for i,ch in enumerate(seq):
        c=CODE.get(ch,4)
        if c<4:
            fwd=((fwd<<2)|c)&mask; rev=(rev>>2)|((3^c)<<top)
            if fwd==rev: continue
            z=0 if fwd<rev else 1
            run+=1
            if run>=k: out.append((hash64(rev if z else fwd,mask)<<8|k, i, z, run))
            else: out.append((INF,i,0,run))
        else:
            run=0; out.append((INF,i,0,0))
    return out
def end_element(seq,w,k):
    st=stream(seq,w,k)
    win=st[-w:]
    best=None
    for e in win:
        if best is None or e[0]<=best[0]: best=e
    if best is None or best[0]==INF: return None
    return best
def predicted(seq,w,k,rid):
    n=len(seq); mask=(1<<(2*k))-1; top=2*(k-1)
    res=[]
    # segments
    i=0; fwd=rev=0   # N-deleted context carried exactly
    while i<n:
        if CODE.get(seq[i],4)==4: i+=1; continue
        j=i
        while j<n and CODE.get(seq[j],4)<4: j+=1
        # segment [i,j): exact warm-up: find f = first position where run reaches k
        run=0; f=None; p=i; fw,rv=fwd,rev
        while p<j:
            c=CODE[seq[p]]; fw=((fw<<2)|c)&mask; rv=(rv>>2)|((3^c)<<top)
            if fw!=rv:
                run+=1
                if run==k: f=p; break
            p+=1
        # advance the carried context over the whole segment
        for p2 in range(i,j):
            c=CODE[seq[p2]]; fwd=((fwd<<2)|c)&mask; rev=(rev>>2)|((3^c)<<top)
        if f is not None:
            vs=f-(k-1)
            sub=bytes(seq[vs:j])
            o=U.orc_sketch_ascii(sub,w,k,rid)
            o=o[:-1].copy()
            o['y']+=np.uint64(vs<<1)
            res.append(o)
        i=j
    e=end_element(seq,w,k)
    if e is not None:
        a=np.zeros(1,res[0].dtype if res else U.orc_sketch_ascii(b"ACGTACGTACGTACGTACGTACGT",w,k,0).dtype)
        a['x']=e[0]; a['y']=(rid<<32)|(e[1]<<1)|e[2]
        res.append(a)
    if not res: return np.zeros(0,U.orc_sketch_ascii(b"ACGT"*8,w,k,0).dtype)
    return np.concatenate(res)
rng=np.random.default_rng(5)
bad=0; tot=0
for trial in range(6000):
    w=int(rng.choice([5,11,24,80])); k=int(rng.choice([4,6,12,16]))
    L=int(rng.integers(1,1500))
    mode=trial%5
    if mode==0: s=rng.integers(0,4,L)
    elif mode==1:
        per=int(rng.integers(1,8)); s=np.resize(rng.integers(0,4,per),L)
        hit=rng.random(L)<0.02; s=np.where(hit,rng.integers(0,4,L),s)
    elif mode==2: s=np.resize(np.array([0,3]),L); hit=rng.random(L)<0.03; s=np.where(hit,rng.integers(0,4,L),s)
    elif mode==3: s=rng.integers(0,2,L)*3
    else: s=rng.integers(0,4,L)
    a=np.array(list(b"ACGT"),np.uint8)[s]
    nn=int(rng.integers(1,8)) if trial%7 else int(rng.integers(1,max(2,L//10)))
    for _ in range(nn):
        p=int(rng.integers(0,L)); ln=int(rng.integers(1,4)); a[p:p+ln]=ord('N')
    if trial%11==0: a[-int(rng.integers(1,min(L,120)+1)):][:1]=ord('N')
    seq=bytes(a)
    want=U.orc_sketch_ascii(seq,w,k,7)
    got=predicted(seq,w,k,7)
    tot+=1
    if len(want)!=len(got) or not np.array_equal(want,got):
        bad+=1
        if bad<5: print('MISMATCH',trial,w,k,L,len(want),len(got))
print('trials',tot,'mismatches',bad)
