"""Statistics of the attention dropout counter hash (attn.hip drop4; oracle/dropout_hash.py drop4_words): the round-3 murmur
finaliser form (`old`, two 32-bit multiplies = quarter-rate v_mul_lo_u32) against the round-4 form built from 24-bit multiply-adds
(`v3` = what the kernels use).  Keep rate, correlations of samples / keep decisions between neighbouring keys and queries, chi-square of
the sample histogram, per-row and per-column drop-count variance against the binomial, and the avalanche matrix.  CPU only (numpy)."""
import numpy as np
M=np.uint32
def fmix32(h):
    h=h.astype(np.uint32).copy(); h^=h>>M(16); h=(h*M(0x85ebca6b)).astype(np.uint32); h^=h>>M(13); h=(h*M(0xc2b2ae35)).astype(np.uint32); h^=h>>M(16); return h
def old(h):
    w0=fmix32(h); x=w0^(w0>>M(15)); x=(x*M(0x2c1b3c6d)).astype(np.uint32); w1=x^(x>>M(12)); return w0,w1
def mad24(x,y,z):
    return ((x.astype(np.uint64)&0xffffff)*(np.uint64(y)&0xffffff)+z.astype(np.uint64)).astype(np.uint32)
K1,K2,K3=0x85ebcb,0xc2b2af,0x9e3779|1
def new(h):
    h=h.astype(np.uint32)
    x1=mad24(h,K1,h>>M(11))
    x2=x1^(x1>>M(14))
    w0=mad24(x2,K2,x2>>M(9))
    w0=w0^(w0>>M(15))
    w1=mad24(w0,K3,x1)
    w1=w1^(w1>>M(13))
    return w0,w1
def samples(fn, base, Q, K4):
    with np.errstate(over='ignore'):
        q=np.arange(Q,dtype=np.uint32)[:,None]; k=np.arange(K4,dtype=np.uint32)[None,:]
        h=(M(base)+q*M(0x85ebca77)+k*M(0xc2b2ae3d)).astype(np.uint32)
        w0,w1=fn(h)
    s=np.stack([w0&M(0xffff),w0>>M(16),w1&M(0xffff),w1>>M(16)],-1).reshape(Q,K4*4).astype(np.float64)
    return s
def stats(name, fn):
    rng=np.random.default_rng(0)
    res=[]
    for base in rng.integers(0,2**32,size=6):
        s=samples(fn,int(base),1056,264)
        keep=(s>=6554)
        u=s/65536
        # correlations
        c_adj_key=np.corrcoef(u[:,:-1].ravel(),u[:,1:].ravel())[0,1]
        c_adj_q=np.corrcoef(u[:-1].ravel(),u[1:].ravel())[0,1]
        c_k4=np.corrcoef(u[:,:-4].ravel(),u[:,4:].ravel())[0,1]
        ck=np.corrcoef(keep[:,:-1].ravel(),keep[:,1:].ravel())[0,1]
        cq=np.corrcoef(keep[:-1].ravel(),keep[1:].ravel())[0,1]
        # chi2 over 256 buckets
        hist=np.bincount((s.ravel()/256).astype(int),minlength=256); e=s.size/256
        chi=((hist-e)**2/e).sum()
        # row drop-count variance vs binomial
        d=(~keep).sum(1); var_ratio=d.var()/(1056*0.1*0.9)
        res.append((keep.mean(),c_adj_key,c_adj_q,c_k4,ck,cq,chi,var_ratio, (~keep).sum(0).var()/(1056*0.1*0.9)))
    r=np.array(res)
    print(name,'keep %.5f | corr adjkey %.4f adjq %.4f key+4 %.4f | keepcorr k %.4f q %.4f | chi2(255dof) %.0f | rowvar %.3f colvar %.3f'%tuple(np.abs(r).max(0)[[0]].tolist()+np.abs(r[:,1:6]).max(0).tolist()+[r[:,6].max(), r[:,7].mean(), r[:,8].mean()]))
    # avalanche: flip each input bit
    rng=np.random.default_rng(1)
    h=rng.integers(0,2**32,size=200000,dtype=np.uint64).astype(np.uint32)
    w0,w1=fn(h)
    worst=0
    for b in range(32):
        v0,v1=fn(h^M(1<<b))
        for a,bb in ((w0,v0),(w1,v1)):
            d=a^bb
            for ob in range(32):
                p=((d>>M(ob))&M(1)).mean()
                worst=max(worst,abs(p-0.5))
    print('   worst avalanche deviation from 0.5:', worst)
stats('old',old); stats('new',new)

def aval(fn,n=100000):
    rng=np.random.default_rng(1)
    h=rng.integers(0,2**32,size=n,dtype=np.uint64).astype(np.uint32)
    w0,w1=fn(h); worst=0; tot=0; cnt=0
    for b in range(32):
        v0,v1=fn(h^M(1<<b))
        for a,bb in ((w0,v0),(w1,v1)):
            d=a^bb
            for ob in range(32):
                p=((d>>M(ob))&M(1)).mean(); worst=max(worst,abs(p-0.5)); tot+=abs(p-0.5); cnt+=1
    return worst, tot/cnt
def v2(h):
    h=h.astype(np.uint32)
    x1=mad24(h,K1,h>>M(11)); x1=x1^(x1>>M(14))
    x2=mad24(x1,K2,x1>>M(9)); x2=x2^(x2>>M(13))
    w0=mad24(x2,K3,x2>>M(10)); w0=w0^(w0>>M(15))
    w1=mad24(w0,0xa54ff5,x2); w1=w1^(w1>>M(13))
    return w0,w1
def v3(h):   # xor high byte down first so the 24-bit multiplier sees all 32 bits
    h=h.astype(np.uint32)
    x=h^(h>>M(16))
    x1=mad24(x,K1,h>>M(8)); x1=x1^(x1>>M(13))
    w0=mad24(x1,K2,x1>>M(11)); w0=w0^(w0>>M(15))
    w1=mad24(w0,K3,x1>>M(7)); w1=w1^(w1>>M(12))
    return w0,w1
for name,fn in (('old',old),('new',new),('v2',v2),('v3',v3)):
    print(name, aval(fn))
stats('v2',v2); stats('v3',v3)
