"""tools/chunk_bounds_stat.py [m=5e-3] — many-sphere scene (BASELINE configs[4]'s 1,024 spheres): how many of the 64 kd-leaf chunks
does a ray's LINE touch when the leaves are bounded by spheres (what the kernel tests) and when they are
bounded by axis-aligned boxes, with the safety margin m both need (a bound may only skip spheres whose
REFERENCE discriminant is certainly negative: distance to the line > r + m |v|, m^2 >= the direction's deviation from unit
length + the float error of b*b - 4c)? CPU only, numpy; the kd split is restated from csrc/ptss_api.hip (kdSplit).

Result (round 3): mid-bounce rays (origins on sphere surfaces, uniform directions) touch 5.09 sphere bounds; boxes with the same
margin 5.04 (-1 %), with a margin no analysis could justify (m = 3e-4) 4.23 (-17 %); camera rays 6.97 / 5.88 / 5.38. A box test
costs ~25 instructions per bound against 18 (line-box separating axes d x e_k plus the behind-the-origin test; no reciprocal), and
every lane tests all 64 bounds: +450 instructions per query against at most 0.85 x 800 saved in visits. Not built.

Built instead (second half of round 3; the last lines this tool prints): the distance of the bound's centre from the RAY (one test,
vv - min(dv, 0)^2: 4.67), near-minimal enclosing balls as bounds (3.95; with the leaves improved pair by pair in packScene: 3.73)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cuda-path-tracer-ss_amd'))
import ptss
sc=ptss.Scene('stress'); d=sc.desc
P=np.array([[d.spheres[i].position.x,d.spheres[i].position.y,d.spheres[i].position.z] for i in range(d.numSpheres)],dtype=np.float64)
R=np.array([d.spheres[i].radius for i in range(d.numSpheres)],dtype=np.float64)
print(P.min(0),P.max(0),R.min(),R.max(),R.mean())
CH=16
def kd(idx):
    n=len(idx)
    if n<=CH: return [idx]
    c=P[idx]; ext=c.max(0)-c.min(0); ax=int(np.argmax(ext))
    unit=64 if n>64 else CH
    left=((n//2+unit-1)//unit)*unit
    if left>=n: left-=unit
    if left<=0: return [idx]
    order=sorted(idx,key=lambda i:(P[i,ax],i))
    return kd(order[:left])+kd(order[left:])
leaves=kd(list(range(len(P))))
print(len(leaves),[len(l) for l in leaves][:8])
m=float(sys.argv[1]) if len(sys.argv) > 1 else 5e-3
C=[];Rb=[];LO=[];HI=[]
for l in leaves:
    c=P[l].mean(0); C.append(c); Rb.append((np.linalg.norm(P[l]-c,axis=1)+R[l]).max())
    LO.append((P[l]-R[l][:,None]).min(0)); HI.append((P[l]+R[l][:,None]).max(0))
C=np.array(C);Rb=np.array(Rb);LO=np.array(LO);HI=np.array(HI)
rng=np.random.default_rng(1)
N=20000
# rays: origins on random spheres' surfaces, random directions (mid-bounce like); plus camera rays
k=rng.integers(0,len(P),N); n=rng.normal(size=(N,3)); n/=np.linalg.norm(n,axis=1,keepdims=True)
o=P[k]+n*R[k,None]*(1+1e-4)
dd=rng.normal(size=(N,3)); dd/=np.linalg.norm(dd,axis=1,keepdims=True)
dd=np.where((dd*n).sum(1,keepdims=True)<0,-dd,dd)
def stats(o,dd,label):
    v=o[:,None,:]-C[None]          # N x K x 3
    dv=(v*dd[:,None,:]).sum(2); vv=(v*v).sum(2)
    R2=(Rb**2)*(1+m)**3
    mu=m+m*m
    lineclear=(vv*(1-mu)-(1+2e-5)*dv*dv)>R2
    behind=(dv>0)&((dv*dv*(1-2e-5))>R2+mu*vv)
    sph=~(lineclear|behind)
    # box: SAT line vs inflated AABB
    D=np.linalg.norm(o,axis=1).max()+np.linalg.norm(P,axis=1).max()+R.max()
    delta=m*D
    Cb=(LO+HI)/2; H=(HI-LO)/2+delta
    vb=o[:,None,:]-Cb[None]
    ad=np.abs(dd)[:,None,:]
    cr=np.cross(np.broadcast_to(dd[:,None,:],vb.shape),vb)
    sep=(np.abs(cr[...,0])>H[None,:,1]*ad[...,2]+H[None,:,2]*ad[...,1])|(np.abs(cr[...,1])>H[None,:,0]*ad[...,2]+H[None,:,2]*ad[...,0])|(np.abs(cr[...,2])>H[None,:,0]*ad[...,1]+H[None,:,1]*ad[...,0])
    dvb=(vb*dd[:,None,:]).sum(2)
    beh=(-dvb+(H[None]*ad).sum(2))<0
    box=~(sep|beh)
    # exact: which chunks contain a sphere actually hit (line within r, in front)
    print(label,'delta',delta,'sphere-bound chunks/ray',sph.sum(1).mean(),'box chunks/ray',box.sum(1).mean(),'both',(sph&box).sum(1).mean())
stats(o,dd,'surface rays')
# camera rays
W=200
x=(rng.random(N)-0.5)*2; y=(rng.random(N)-0.5)*2*9/16
dc=np.stack([x,y,-np.ones(N)],1); dc/=np.linalg.norm(dc,axis=1,keepdims=True)
stats(np.zeros((N,3)),dc,'camera rays')


# ---- what was built instead: the RAY's distance, and near-minimal enclosing balls (csrc/ptss_api.hip enclosingBall) -------------
def ball(l):
    l = np.array(l); c = P[l].mean(0); rbest = 1e30; cbest = c
    for it in range(1, 513):
        dist = np.linalg.norm(P[l] - c, axis=1) + R[l]
        j = int(np.argmax(dist))
        if dist[j] < rbest: rbest, cbest = dist[j], c.copy()
        dirv = P[l][j] - c
        c = c + dirv / max(np.linalg.norm(dirv), 1e-30) * dist[j] / (it + 1)
    return cbest, rbest
balls = [ball(l) for l in leaves]
C2 = np.array([c for c, r in balls]); R2 = np.array([r for c, r in balls])
mu = m + m * m
for label, CC, RR in (("mean centres", C, Rb), ("near-minimal balls", C2, R2)):
    v = o[:, None, :] - CC[None]; dv = (v * dd[:, None, :]).sum(2); vv = (v * v).sum(2)
    dvm = np.minimum(dv, 0)
    ray = ~((vv * (1 - mu) - (1 + 2e-5) * dvm * dvm) > RR ** 2 * (1 + m) ** 3)
    print("surface rays, distance from the RAY, %s: %.2f bounds per ray (mean radius %.3f)" % (label, ray.sum(1).mean(), RR.mean()))
