"""CPU replay of the stick chain's producer loops (csrc/voxel_sticks.hip, one lane per Gaussian, 64 consecutive Gaussians in lockstep):
per LDS atomic instruction the active lanes, the distinct counters, the largest same-counter and same-bank multiplicities -- on the trained
clouds (gpurun_out/clouds or /tmp/r2_clouds, tests/trained_cloud.py) and on synthetic ones.  Written to find out why the chain's count
and scatter kernels were 3-5 x slower per instance on trained clouds; it cleared LDS contention (the numbers are the same for both kinds of
cloud: profiles/experiments/r05g_sticks_wave_replay.txt) -- the cause was a few Gaussians of thousands of tiles.
    python scripts/sticks_wave_replay.py"""
import numpy as np, sys, torch
sys.path.insert(0,'/root/repo')
from r2_gaussian_amd import scene as S
from tests import trained_cloud as TCl
def cubes(c, G=256):
    xyz = c.xyz.numpy().astype(np.float32); sc = c.scales.numpy().astype(np.float32)
    n=np.array([G]*3); s=np.array([2.,2.,2.],np.float32); dv=(s/n).astype(np.float32); ms=sc.max(1)
    rad=np.ceil((np.float32(3.)*ms)[:,None]/dv[None,:]); pv=(xyz+s/2)/dv; g=(n+7)//8
    lo=np.clip(((pv-rad)/8).astype(np.int32),0,g); hi=np.clip(((pv+rad+7)/8).astype(np.int32),0,g)
    out=(pv+rad<0).any(1)|(pv-rad>n).any(1)
    ext=hi-lo; ext[out]=0
    return lo, ext, g
def analyze(name, c, nw=400, sh=3):
    lo, ext, g = cubes(c)
    P=len(lo); tt=ext.prod(1)
    print("== %s: P %d tpg %.1f ext mean %s" % (name, P, tt[tt>0].mean(), ext[tt>0].mean(0)))
    big = np.sort(tt)[::-1]
    print("  tiles of the largest Gaussians: %s; Gaussians of more than 256 / 4096 tiles: %d / %d" % (big[:6].tolist(), int((tt > 256).sum()), int((tt > 4096).sum())))
    rng=np.random.default_rng(0)
    waves = rng.choice(P//64, size=min(nw,P//64), replace=False)
    tot_steps=0; tot_same=0; tot_bank=0; tot_active=0; tot_distinct=0
    for w in waves:
        ids=np.arange(w*64,(w+1)*64)
        seqs=[]
        for i in ids:
            if tt[i]==0: seqs.append([]); continue
            l=[]
            for z in range(ext[i,2]):
                for y in range(ext[i,1]):
                    t0=((lo[i,2]+z)*g[1]+lo[i,1]+y)*g[0]+lo[i,0]; t1=t0+ext[i,0]-1
                    for st in range(t0>>sh,(t1>>sh)+1): l.append(st)
            seqs.append(l)
        L=max(len(s) for s in seqs)
        for step in range(L):
            a=np.array([s[step] for s in seqs if len(s)>step])
            tot_steps+=1; tot_active+=len(a)
            u,cnt=np.unique(a,return_counts=True)
            tot_same+=cnt.max(); tot_distinct+=len(u)
            b=np.bincount(u%32, minlength=32)   # distinct addresses per bank
            tot_bank+=b.max()
    print("  steps/wave %.1f, active lanes/step %.1f, distinct addresses/step %.1f, max same-address %.2f, max distinct addresses in one bank %.2f" % (
        tot_steps/len(waves), tot_active/tot_steps, tot_distinct/tot_steps, tot_same/tot_steps, tot_bank/tot_steps))
c,_=TCl.load("small", train=False); analyze("trained small", c)
c,_=TCl.load("large", train=False); analyze("trained large", c, nw=200)
analyze("synthetic 50k", S.make_cloud(50000, seed=0))
analyze("synthetic 20k", S.make_cloud(20000, seed=0))
analyze("synthetic 300k", S.make_cloud(300000, seed=0))
