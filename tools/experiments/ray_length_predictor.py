import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
from oracle import oracle as orc
from raytrace_amd import scene
asm = scene.formula1(width=160, height=96, n_samples=1, detail=0.5, tex_size=64)
objs = scene.finalize(asm)
S = orc.Scene(objs, asm.materials, asm.textures)
cfg = asm.config
prm = orc.make_params(cfg, 0x5EED0001, 16)
rays = orc.primary_rays(prm, 0)
rng = np.random.default_rng(1)
def trace(rays):
    n=len(rays); cnt=np.zeros(n); O=np.zeros(n,np.int32); T=np.zeros(n,np.int32); tt=np.zeros(n,np.float32)
    for i in range(n):
        o,t,tv,u,v,c = S.closest_hit(rays[i:i+1])
        cnt[i]=c[1]; O[i]=o[0]; T[i]=t[0]; tt[i]=tv[0]
    return cnt,O,T,tt
def tri_normal(o,t):
    tri = objs[o][0][t] if isinstance(objs[o], tuple) else objs[o].triangles[t]
    return tri
print(type(objs[0]), getattr(objs[0],'__dict__',{}).keys() if hasattr(objs[0],'__dict__') else objs[0][0].dtype)
def bounce(rays, O, T, tt):
    out=[]; keep=[]
    for i in range(len(rays)):
        if O[i] < 0: continue
        tri = objs[O[i]][0][T[i]]
        p = tri['v']['position']
        nrm = np.cross(p[1]-p[0], p[2]-p[0]); nl=np.linalg.norm(nrm)
        if nl == 0: continue
        nrm/=nl
        d = rays[i,3:6]
        if np.dot(d,nrm) > 0: nrm = -nrm
        while True:
            r = rng.uniform(-1,1,3)
            if r@r < 1: break
        nd = nrm + r; nd/=np.linalg.norm(nd)
        org = rays[i,0:3] + d*tt[i] + nd*1e-4
        out.append(np.concatenate([org,nd])); keep.append(i)
    return np.array(out,dtype=np.float32), np.array(keep)
t0=time.time()
sub = rays[rng.choice(len(rays), 15000, replace=False)]
c0,O,T,tt = trace(sub)
r1,k1 = bounce(sub,O,T,tt)
c1,O1,T1,tt1 = trace(r1)
r2,k2 = bounce(r1,O1,T1,tt1)
c2,O2,T2,tt2 = trace(r2)
r3,k3 = bounce(r2,O2,T2,tt2)
c3,_,_,_ = trace(r3)
from scipy.stats import spearmanr
print('n', len(sub), len(r1), len(r2), len(r3), 'time', time.time()-t0)
print('mean box tests', c0.mean(), c1.mean(), c2.mean(), c3.mean())
print('spearman c0->c1', spearmanr(c0[k1], c1).correlation)
print('spearman c1->c2', spearmanr(c1[k2], c2).correlation)
print('spearman c2->c3', spearmanr(c2[k3], c3).correlation)
# how well does "prev count above median" pick the long rays: fraction of top-10% longest c2 that had c1 above median
thr = np.quantile(c2, 0.9); med = np.median(c1[k2])
print('top10% of c2 with prev above median:', ((c1[k2] > med) & (c2 >= thr)).sum() / (c2 >= thr).sum())
# object of previous hit as predictor: mean c2 by object hit at previous
import collections
by=collections.defaultdict(list)
for j,i in enumerate(k2): by[int(O1[i])].append(c2[j])
print(sorted([(round(np.mean(v)),len(v),o) for o,v in by.items() if len(v)>30], reverse=True)[:12])
for name,prev,cur in (('c0->c1',c0[k1],c1),('c1->c2',c1[k2],c2)):
    for q in (0.9,0.98):
        thr=np.quantile(cur,q)
        for pq in (0.5,0.75):
            pt=np.quantile(prev,pq)
            print(name,'top',round(1-q,2),'of cur: share whose prev is above its',pq,'quantile =',round(((prev>pt)&(cur>=thr)).sum()/(cur>=thr).sum(),3), ' max cur',cur.max(), 'thr',thr)
