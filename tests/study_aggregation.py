"""Offline experiment (CPU, scipy.sparse, the 10 000-vertex C3 mesh): preconditioned-CG iteration counts of the NICP normal
equations with the shipped coarse spaces (12 global modes, 66 polynomial columns) against AGGREGATION spaces -- the 6 zero-
stiffness modes restricted to aggregates of s Morton-consecutive vertices (+ the 6 global parameters), exact coarse solve.
Uses the sparse oracle, so it lives with the tests (the only place allowed to import oracle/); a study tool, not collected:
    python tests/study_aggregation.py 100 4e-8 1e-3 1024,256,64,16        (aggregate sizes)
    python tests/study_aggregation.py 100 4e-8 1e-3 64 2,3,4              (instead: full polynomial spaces of these degrees)
Results: profiles/r3_nicp_studies.txt, DESIGN.md section 3.3."""
import sys,time; sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'oracle'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import icp_oracle as orc, icp_oracle_sparse as ors, gen_large_mesh as glm
nx=int(sys.argv[1]); lam=float(sys.argv[2]); tol=float(sys.argv[3]) if len(sys.argv)>3 else 1e-3
d=glm.inputs(nx); v=d["v"]; n=len(v); S=12
C=(0.1,1000.,100.)
P=d["P0"].copy()
# a mid-solve state: a couple of LM steps would be better; perturb slightly so R != I
J=ors.d_min_func_sparse(P,d["nbr"],d["wt"],v,d["edges"],d["nn"],C,False)
r=ors.min_func_v(P,d["nbr"],d["wt"],v,d["edges"],d["nn"],C,False)
d2=np.asarray(J.multiply(J).sum(0)).reshape(-1); d2[d2<=0]=1
A=(J.T@J+sp.diags(lam*d2)).tocsr()
g=-(J.T@r)
# block jacobi
Ab=A[:n*S,:n*S].tobsr(blocksize=(S,S))
# extract diagonal blocks
diag=np.zeros((n,S,S))
indptr,indices,data=Ab.indptr,Ab.indices,Ab.data
for i in range(n):
    for k in range(indptr[i],indptr[i+1]):
        if indices[k]==i: diag[i]=data[k]
Binv=np.linalg.inv(diag)
tail=1.0/A.diagonal()[n*S:]
def applyB(x):
    y=np.empty_like(x)
    y[:n*S]=np.einsum('nij,nj->ni',Binv,x[:n*S].reshape(n,S)).reshape(-1)
    y[n*S:]=tail*x[n*S:]
    return y
def zmodes_all():
    Z=np.zeros((n,6,12))
    Pm=P[:n*S].reshape(n,S)
    rows=[Pm[:,0:3],Pm[:,3:6],Pm[:,6:9],v+Pm[:,9:12]]
    for rr,x in enumerate(rows):
        o=rr*3
        Z[:,0,o]=-x[:,1]; Z[:,0,o+1]=x[:,0]
        Z[:,1,o+1]=-x[:,2]; Z[:,1,o+2]=x[:,1]
        Z[:,2,o]=x[:,2]; Z[:,2,o+2]=-x[:,0]
    Z[:,3,9]=Z[:,4,10]=Z[:,5,11]=1.0
    return Z
ZM=zmodes_all()
def coarse_from_weights(W):
    # W: [n, m] sparse or dense weights of m scalar functions; columns = 6 modes x m + 6 globals
    W=sp.csr_matrix(W); m=W.shape[1]
    rows=[];cols=[];vals=[]
    Wc=W.tocoo()
    for k in range(6):
        for e in range(12):
            z=ZM[Wc.row,k,e]
            nzm=z!=0
            rows.append(Wc.row[nzm]*S+e); cols.append(Wc.col[nzm]*6+k); vals.append(Wc.data[nzm]*z[nzm])
    Z=sp.coo_matrix((np.concatenate(vals),(np.concatenate(rows),np.concatenate(cols))),shape=(n*S+6,m*6+6)).tolil()
    for k in range(6): Z[n*S+k,m*6+k]=1.0
    return Z.tocsc()
def pcg(Z,maxit=20000):
    if Z is not None:
        E=(Z.T@(A@Z)).toarray()
        Einv=np.linalg.pinv(E,rcond=1e-13,hermitian=True)
    def M(rv):
        z=applyB(rv)
        if Z is not None: z=z+Z@(Einv@(Z.T@rv))
        return z
    x=np.zeros_like(g); rv=g.copy(); z=M(rv); p=z.copy(); rz=rv@z; rr0=rv@rv
    for it in range(1,maxit+1):
        Ap=A@p; al=rz/(p@Ap); x+=al*p; rv-=al*Ap
        if rv@rv<=tol*tol*rr0: return it
        z=M(rv); rzn=rv@z; p=z+(rzn/rz)*p; rz=rzn
    return maxit
c=v-v.mean(0); c=c/np.abs(c).max()
one=np.ones((n,1))
poly=np.stack([np.ones(n),c[:,0],c[:,1],c[:,2],c[:,0]**2,c[:,1]**2,c[:,2]**2,c[:,0]*c[:,1],c[:,1]*c[:,2],c[:,0]*c[:,2]],1)
def morton_order():
    q=((v-v.min(0))/(v.max(0)-v.min(0)+1e-12)*1023).astype(np.int64)
    def spread(x):
        out=np.zeros_like(x)
        for b in range(10): out|=((x>>b)&1)<<(3*b)
        return out
    code=spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
    return np.argsort(code,kind="stable")
mo=morton_order()
def agg(size):
    m=(n+size-1)//size
    W=sp.coo_matrix((np.ones(n),(mo,np.arange(n)//size)),shape=(n,m))
    return W
import itertools
def monomials(deg):
    cols=[]
    for d in range(deg+1):
        for ex in itertools.product(range(d+1),repeat=3):
            if sum(ex)==d: cols.append(c[:,0]**ex[0]*c[:,1]**ex[1]*c[:,2]**ex[2])
    return np.stack(cols,1)
t=time.time()
print("n",n,"lam",lam,"tol",tol,flush=True)
if len(sys.argv)>5:
    for deg in [int(a) for a in sys.argv[5].split(",")]:
        W=monomials(deg)
        print("poly degree %d (%d functions, %d cols):"%(deg,W.shape[1],W.shape[1]*6+6),pcg(coarse_from_weights(W)),"%.0fs"%(time.time()-t),flush=True)
    sys.exit(0)
print("plain 12:",pcg(coarse_from_weights(one)),"%.0fs"%(time.time()-t),flush=True)
print("poly 66:",pcg(coarse_from_weights(poly)),"%.0fs"%(time.time()-t),flush=True)
for size in [int(a) for a in sys.argv[4].split(",")]:
    W=agg(size)
    print("agg %d (%d aggregates, %d cols):"%(size,W.shape[1],W.shape[1]*6+6),pcg(coarse_from_weights(W)),"%.0fs"%(time.time()-t),flush=True)
