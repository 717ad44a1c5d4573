import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_util as U
from peregrine_amd import formats, simreads
g = simreads.make_genome(4_000_000, 1004, repeat_families=2, repeat_len=6000, repeat_copies=60, tandem=40)
db = simreads.simulate_reads(g, coverage=30.0, seed=42)
pre='/tmp/memo/sd'
formats.write_seqdb(pre, db)
N=8
import concurrent.futures as cf
t=time.time()
with cf.ThreadPoolExecutor(8) as ex:
    list(ex.map(lambda c: U.ref_run("shmr_index","-p",pre,"-t",N,"-c",c,"-m",0,"-o","/tmp/memo/ix"), range(1,N+1)))
    list(ex.map(lambda c: U.ref_run("shmr_overlap","-p",pre,"-l","/tmp/memo/ix-L2","-t",N,"-c",c,"-o","/tmp/memo/ov.%d"%c), range(1,N+1)))
U.ref_run("shmr_overlap","-p",pre,"-l","/tmp/memo/ix-L2","-t",1,"-c",1,"-o","/tmp/memo/ov1")
print('ref time', time.time()-t)
def keys(o):
    p0=((o['y0']&np.uint64(0xFFFFFFFF))>>np.uint64(1)).astype(np.int64); p1=((o['y1']&np.uint64(0xFFFFFFFF))>>np.uint64(1)).astype(np.int64)
    r0=(o['y0']>>np.uint64(32)).astype(np.int64); r1=(o['y1']>>np.uint64(32)).astype(np.int64)
    return np.stack([r0,r1,p0-p1,o['strand0'].astype(np.int64),o['strand1'].astype(np.int64)],1)
allk=[]; tot=0
for c in range(1,N+1):
    o=formats.read_ovlp('/tmp/memo/ov.%d'%c); tot+=len(o); allk.append(keys(o))
k=np.concatenate(allk)
u=np.unique(k,axis=0)
pr=np.unique(np.stack([np.minimum(k[:,0],k[:,1]),np.maximum(k[:,0],k[:,1])],1),axis=0)
o1=formats.read_ovlp('/tmp/memo/ov1')
print('records 8 chunks',tot,'distinct keys',len(u), 'unique read pairs',len(pr),'1-chunk records',len(o1))
# same but ignoring q_off drift: (r0,r1,s0,s1)
u2=np.unique(k[:,[0,1,3,4]],axis=0); print('distinct ordered (r0,r1,strands)',len(u2))
