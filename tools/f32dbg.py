import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dummynode4graphlearning_amd import ops
DEV="cuda:0"
def rel(a,b): return float((a.double()-b.double()).abs().max()/b.double().abs().max())
for H in (64,128,256):
    rng=np.random.default_rng(H)
    sizes=[0,37,5000,1,9001]; rel_ptr=[0]+list(np.cumsum(sizes)); P=rel_ptr[-1]; R=5
    A=torch.from_numpy(rng.standard_normal((3000,H)).astype(np.float32)).to(DEV); G=torch.from_numpy(rng.standard_normal((2500,H)).astype(np.float32)).to(DEV)
    ia=torch.from_numpy(rng.integers(0,3000,size=P)).to(torch.int32).to(DEV); ig=torch.from_numpy(rng.integers(0,2500,size=P)).to(torch.int32).to(DEV)
    table=ops.make_row_chunks([int(v) for v in rel_ptr],DEV,chunk_rows=2048)
    res={}
    for ex in (True,False):
        ops.F32_EXACT=ex
        res[ex]=ops.rows_wgrad(A,G,table,R,idx_a=ia,idx_g=ig,out_dtype=torch.float32,colsum_of=2)
    print(H,"wgrad split vs exact",rel(res[False][0],res[True][0]),"colsum",rel(res[False][1],res[True][1]))
    # masked variant (dense rows)
    N=4000
    g=torch.randn(N,H,device=DEV); x=torch.randn(N,H,device=DEV); y=torch.randn(N,H,device=DEV)
    t2=ops.make_row_chunks([0,N],DEV,chunk_rows=1024)
    out={}
    for ex in (True,False):
        ops.F32_EXACT=ex
        gm=torch.empty_like(g)
        gw,cs=ops.rows_wgrad(g,x,t2,1,out_dtype=torch.float32,colsum_of=1,mask_a=y,a_out=gm)
        out[ex]=(gw,cs,gm)
    print(H,"masked wgrad",rel(out[False][0],out[True][0]),rel(out[False][1],out[True][1]),rel(out[False][2],out[True][2]), "gm ok", rel(out[False][2], g*(y>0)))
    # transform
    sizes=[0,37,1500,1,33,64]; rp=[0]+[int(v) for v in np.cumsum(sizes)]; Rr=len(sizes); Pp=rp[-1]
    X=torch.randn(700,H,device=DEV); X2=torch.randn(90,H,device=DEV); Wn=torch.randn(Rr,H,H,device=DEV)/H**0.5; bias=torch.randn(Rr,H,device=DEV)
    idx=torch.from_numpy(rng.integers(0,790,size=Pp)).to(torch.int32).to(DEV)
    tiles=ops.make_row_tiles(rp,DEV); mask=torch.randn(Pp,H,device=DEV)
    o={}
    for ex in (True,False):
        ops.F32_EXACT=ex
        o[ex]=(ops.rows_transform(X,Wn,tiles,Pp,idx=idx,X2=X2,bias=bias,relu=True), ops.rows_transform(X,Wn,tiles,Pp,idx=idx,X2=X2,mask_pos=mask))
    print(H,"transform",rel(o[False][0],o[True][0]),rel(o[False][1],o[True][1]))
