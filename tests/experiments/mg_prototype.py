"""Design experiment behind openimpala_amd/csrc/oi_mg.hip: scipy prototype of the 2x2x2 aggregation
multigrid preconditioner on the oracle's system (CPU only, test/experiment infrastructure).
Output of one run is kept in profiles/r01/mg_prototype.txt."""
import sys, time; import os; sys.path.insert(0,os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..'))
import numpy as np, scipy.sparse as sp
from oracle import oracle as o
ph0 = o.read_tiff_phase(os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','golden')+'/SampleData_2Phase_stack_3d_1bit.tif')
def build(ph, pid, d, vlo, vhi):
    mask,_=o.active_mask(ph,pid,d); code=o.codes(mask,d); nz,ny,nx=ph.shape
    free=(code>0); idx=-np.ones(ph.size,np.int64); fidx=np.nonzero(free.reshape(-1))[0]; idx[fidx]=np.arange(fidx.size)
    x0=o.initial_field(mask,d,vlo,vhi); b=-o.apply(code,x0).reshape(-1)[fidx]
    rows=[];cols=[];vals=[]
    f3=free
    I=np.arange(ph.size).reshape(ph.shape)
    for ax in range(3):
        sl1=[slice(None)]*3; sl2=[slice(None)]*3; sl1[ax]=slice(0,-1); sl2[ax]=slice(1,None)
        a=I[tuple(sl1)].reshape(-1); c=I[tuple(sl2)].reshape(-1)
        m=f3.reshape(-1)[a]&f3.reshape(-1)[c]
        rows+= [idx[a[m]], idx[c[m]]]; cols+=[idx[c[m]], idx[a[m]]]; vals+=[-np.ones(m.sum())]*2
    A=sp.coo_matrix((np.concatenate(vals),(np.concatenate(rows),np.concatenate(cols))),shape=(fidx.size,fidx.size)).tocsr()
    A=A+sp.diags(code.reshape(-1)[fidx].astype(float))
    # coordinates of free cells
    k,j,i=np.unravel_index(fidx,ph.shape)
    return A,b,(i,j,k),o.bnorm(mask,d,vlo,vhi)
def agg(coords, shape):
    i,j,k=coords; ci,cj,ck=i//2,j//2,k//2
    cs=((shape[2]+1)//2,(shape[1]+1)//2,(shape[0]+1)//2)
    key=ci+cs[0]*(cj+cs[1]*ck)
    u,inv=np.unique(key,return_inverse=True)
    P=sp.csr_matrix((np.ones(inv.size),(np.arange(inv.size),inv)),shape=(inv.size,u.size))
    ck2=u//(cs[0]*cs[1]); r=u%(cs[0]*cs[1]); cj2=r//cs[0]; ci2=r%cs[0]
    return P,(ci2,cj2,ck2),(cs[2],cs[1],cs[0])
def levels(A,coords,shape,minsize=200):
    L=[dict(A=A,D=A.diagonal())]
    while A.shape[0]>minsize and len(L)<10:
        P,coords,shape=agg(coords,shape)
        A=(P.T@A@P).tocsr(); L[-1]['P']=P; L.append(dict(A=A,D=A.diagonal()))
    import scipy.sparse.linalg as spl
    L[-1]['solve']=spl.factorized(L[-1]['A'].tocsc())
    return L
def vcycle(L,l,b,omega,nu,scale):
    lev=L[l]
    if 'solve' in lev: return lev['solve'](b)
    A,D=lev['A'],lev['D']
    x=omega*b/D
    for _ in range(nu-1): x+=omega*(b-A@x)/D
    r=b-A@x
    xc=vcycle(L,l+1,lev['P'].T@r,omega,nu,scale)
    x+=scale*(lev['P']@xc)
    for _ in range(nu): x+=omega*(b-A@x)/D
    return x
def pcg(A,b,bn,eps,M,maxit=5000):
    x=np.zeros_like(b); r=b.copy(); z=M(r); p=z.copy(); rz=r@z; it=0
    while np.sqrt(r@r)/bn>eps and it<maxit:
        q=A@p; a=rz/(p@q); x+=a*p; r-=a*q; z=M(r); rzn=r@z; p=z+(rzn/rz)*p; rz=rzn; it+=1
    return x,it
for rep in (1,2):
    ph=np.tile(ph0,(rep,rep,rep)) if rep>1 else ph0
    if rep>1: ph=ph[:160,:160,:160].copy()
    A,b,coords,bn=build(ph,1,0,-1.,1.)
    t=time.time(); L=levels(A,coords,ph.shape); print('shape',ph.shape,'n',A.shape[0],'levels',[l['A'].shape[0] for l in L],'setup %.1fs'%(time.time()-t))
    for eps in (1e-6,1e-9):
        x,it0=pcg(A,b,bn,eps,lambda r:r/L[0]['D'])
        res=[('jacobi',it0)]
        for nu,omega,scale in ((1,0.8,1.0),(1,0.8,1.6),(2,0.8,1.6),(2,0.7,1.8)):
            x,it=pcg(A,b,bn,eps,lambda r:vcycle(L,0,r,omega,nu,scale))
            res.append(((nu,omega,scale),it))
        print(' eps',eps,res)
print("---- coarsest level by Jacobi sweeps, other structures ----")
def vcycle2(L,l,b,omega,nu,scale,ncoarse):
    lev=L[l]; A,D=lev['A'],lev['D']
    if 'P' not in lev:
        x=omega*b/D
        for _ in range(ncoarse-1): x+=omega*(b-A@x)/D
        return x
    x=omega*b/D
    for _ in range(nu-1): x+=omega*(b-A@x)/D
    r=b-A@x
    xc=vcycle2(L,l+1,lev['P'].T@r,omega,nu,scale,ncoarse)
    x+=scale*(lev['P']@xc)
    for _ in range(nu): x+=omega*(b-A@x)/D
    return x
cases=[('sample p1 x',ph0,1,0,-1.,1.),('sample p0 z',ph0,0,2,0.,1.),('bernoulli 96',o.gen_bernoulli((96,96,96),20251031,0.5),1,0,0.,1.),('blobs 128',o.gen_blobs((128,128,128),7,12,23996),1,0,0.,1.)]
for name,ph,pid,d,vlo,vhi in cases:
    A,b,coords,bn=build(ph,pid,d,vlo,vhi)
    L=levels(A,coords,ph.shape,minsize=600)
    x,it0=pcg(A,b,bn,1e-6,lambda r:r/L[0]['D'])
    out=[('jacobi',it0)]
    for nu,omega,scale,nc in ((1,0.8,1.0,16),(1,0.8,1.5,16),(1,0.8,1.5,4),(2,0.8,1.5,16)):
        x,it=pcg(A,b,bn,1e-6,lambda r:vcycle2(L,0,r,omega,nu,scale,nc))
        out.append(((nu,omega,scale,nc),it))
    print(name,'n',A.shape[0],'levels',[l['A'].shape[0] for l in L],out)
print("---- W-cycle / smoothing variants ----")
def cyc(L,l,b,omega,scale,gamma,ncoarse):
    lev=L[l]; A,D=lev['A'],lev['D']
    if 'P' not in lev:
        x=omega*b/D
        for _ in range(ncoarse-1): x+=omega*(b-A@x)/D
        return x
    x=omega*b/D
    r=b-A@x
    bc=lev['P'].T@r
    xc=cyc(L,l+1,bc,omega,scale,gamma,ncoarse)
    for _ in range(gamma-1):   # extra coarse visits: refine xc
        rc=bc-L[l+1]['A']@xc
        xc=xc+cyc(L,l+1,rc,omega,scale,gamma,ncoarse)
    x+=scale*(lev['P']@xc)
    x+=omega*(b-A@x)/D
    return x
for name,ph,pid,d,vlo,vhi in cases[:1]+cases[3:]:
    A,b,coords,bn=build(ph,pid,d,vlo,vhi)
    L=levels(A,coords,ph.shape,minsize=600)
    out=[]
    for gamma,scale in ((1,1.5),(2,1.5),(2,1.0),(2,1.2)):
        x,it=pcg(A,b,bn,1e-6,lambda r:cyc(L,0,r,0.8,scale,gamma,16))
        out.append(((gamma,scale),it))
    print(name,out)

print("---- hybrid cycle: gamma=2 on the first nW coarse levels only ----")
def hyb(L,l,b,omega,scale,nW,ncoarse):
    lev=L[l]; A,D=lev['A'],lev['D']
    if 'P' not in lev:
        x=omega*b/D
        for _ in range(ncoarse-1): x+=omega*(b-A@x)/D
        return x
    x=omega*b/D
    r=b-A@x
    bc=lev['P'].T@r
    xc=hyb(L,l+1,bc,omega,scale,nW,ncoarse)
    if l<nW:
        rc=bc-L[l+1]['A']@xc
        xc=xc+hyb(L,l+1,rc,omega,scale,nW,ncoarse)
    x+=scale*(lev['P']@xc)
    x+=omega*(b-A@x)/D
    return x
for name,ph,pid,d,vlo,vhi in cases[:1]+cases[3:]:
    A,b,coords,bn=build(ph,pid,d,vlo,vhi)
    L=levels(A,coords,ph.shape,minsize=600)
    out=[]
    for nW in (0,1,2,3,9):
        x,it=pcg(A,b,bn,1e-6,lambda r:hyb(L,0,r,0.8,1.5,nW,16))
        out.append((nW,it))
    print(name,'(nW, iterations):',out)
