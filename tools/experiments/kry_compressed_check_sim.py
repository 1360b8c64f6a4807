"""Would a compressed (thick-restart style) projected problem make the block-Lanczos convergence checks cheaper?
CPU simulation in numpy (N = 6 000, P = 20, Neig = 64, blocks of 16, full re-orthogonalisation): Rayleigh-Ritz on
span{k kept Ritz vectors of an earlier check, the blocks since} keeps the Lanczos relation exact, so its residual
estimates are exact -- and they track the full check's within a factor 2 -- but the directions dropped at the
compression never come back (new blocks are orthogonal to the whole old basis): the converged set then misses wanted
eigenvalues (error of the wanted set 1e-4 ... 1e-6 of lambda_1 unless the compression happens when the residual is
already < 1e-6). Usable as an estimator only; with the final full eigensolve still needed the saving at C4 is < 3 %.
Not built. Run: python tools/experiments/kry_compressed_check_sim.py"""
import numpy as np, scipy.linalg as sl, time
rng=np.random.default_rng(1)
N,P,Neig,bs=6000,20,64,16
X=rng.standard_normal((N,P)); X=(X-X.mean(0))/X.std(0,ddof=1)
sq=(X*X).sum(1); K=np.exp(-(sq[:,None]+sq[None,:]-2*X@X.T)/P)
# block Lanczos with full reorth
Q=np.zeros((N,0)); 
B0=rng.standard_normal((N,bs)); Qj,_=np.linalg.qr(B0)
blocks=[Qj]; A=[];Bs=[]
def build_T(A,Bs):
    j=len(A); T=np.zeros((j*bs,j*bs))
    for i in range(j):
        T[i*bs:(i+1)*bs,i*bs:(i+1)*bs]=A[i]
        if i+1<j:
            T[(i+1)*bs:(i+2)*bs,i*bs:(i+1)*bs]=Bs[i]; T[i*bs:(i+1)*bs,(i+1)*bs:(i+2)*bs]=Bs[i].T
    return T
prev=None
hist=[]
for step in range(1,60):
    Qj=blocks[-1]
    W=K@Qj
    Aj=Qj.T@W; Aj=(Aj+Aj.T)/2
    A.append(Aj)
    Qall=np.hstack(blocks)
    W=W-Qall@(Qall.T@W); W=W-Qall@(Qall.T@W)
    Qn,R=np.linalg.qr(W)
    Bs.append(R)
    # check
    T=build_T(A,Bs[:-1])
    th,S=np.linalg.eigh(T); th=th[::-1]; S=S[:,::-1]
    res_full=np.linalg.norm(R@S[-bs:,:Neig],axis=0)   # residual norms of top Neig
    hist.append((step,T.shape[0],res_full.max()/th[0]))
    blocks.append(Qn)
# now emulate compressed checks: first check at step s1, keep k vectors, then check at later steps
def compressed(s1,s2,k):
    T1=build_T(A[:s1],Bs[:s1-1]); th1,S1=np.linalg.eigh(T1); th1=th1[::-1][:k]; S1=S1[:,::-1][:,:k]
    # H on span{S1, blocks s1+1..s2}
    m=(s2-s1)*bs
    H=np.zeros((k+m,k+m)); H[:k,:k]=np.diag(th1)
    C=S1[-bs:,:].T@Bs[s1-1].T    # coupling k x bs  (T[(s1-1)blk, s1 blk] = Bs[s1-1].T)
    H[:k,k:k+bs]=C; H[k:k+bs,:k]=C.T
    Tt=build_T(A[s1:s2],Bs[s1:s2-1]); H[k:,k:]=Tt
    th,S=np.linalg.eigh(H); th=th[::-1]; S=S[:,::-1]
    res=np.linalg.norm(Bs[s2-1]@S[-bs:,:Neig],axis=0)
    return res.max()/th[0], th[:Neig]
for (s,d,r) in hist:
    if s%2==0 or r<1e-6: print(s,d,'%.2e'%r)
conv=[s for s,d,r in hist if r<1e-10][0]
print('full converges at step',conv)
for s1 in (conv-8,conv-5,conv-3):
    for k in (Neig, int(1.5*Neig), 2*Neig, 3*Neig):
        out=[]
        for s2 in range(s1+1,conv+4):
            r,th=compressed(s1,s2,k); out.append('%d:%.1e'%(s2,r))
        print('first check',s1,'keep',k,' '.join(out))

print("---- repeated compression: first check at step s1, then every `every` steps, keep k")
evals_true=np.linalg.eigvalsh(K)[::-1][:Neig]
def run_chain(s1,every,k,final_tol=1e-10,maxstep=58):
    # state: Z (dim x k) coefficient matrix of kept Ritz vectors in the Lanczos basis, th (k)
    T1=build_T(A[:s1],Bs[:s1-1]); th,S=np.linalg.eigh(T1); th=th[::-1]; S=S[:,::-1]
    res=np.linalg.norm(Bs[s1-1]@S[-bs:,:Neig],axis=0).max()/th[0]
    log=[(s1,T1.shape[0],res)]
    Z=S[:,:k]; thk=th[:k]; last=s1
    s=s1
    while res>final_tol and s<maxstep:
        s2=min(s+every,maxstep)
        m=(s2-last)*bs
        H=np.zeros((k+m,k+m)); H[:k,:k]=np.diag(thk)
        C=Z[-bs:,:].T@Bs[last-1].T
        H[:k,k:k+bs]=C; H[k:k+bs,:k]=C.T
        H[k:,k:]=build_T(A[last:s2],Bs[last:s2-1])
        th,S=np.linalg.eigh(H); th=th[::-1]; S=S[:,::-1]
        res=np.linalg.norm(Bs[s2-1]@S[-bs:,:Neig],axis=0).max()/th[0]
        log.append((s2,k+m,res))
        # new Z: [Z 0;0 I] S[:, :k]
        Zn=np.vstack([Z@S[:k,:k], S[k:,:k]])
        Z=Zn; thk=th[:k]; last=s2; s=s2
    err=np.abs(thk[:Neig]-evals_true).max()/evals_true[0]
    return log,err
for s1,every,k in ((6,4,64),(6,4,96),(6,2,96),(8,3,128),(12,4,96),(20,3,64),(6,1,96)):
    log,err=run_chain(s1,every,k)
    print('s1',s1,'every',every,'keep',k,'-> converged at step',log[-1][0],'checks',len(log),'dims',[d for _,d,_ in log],'eig err %.1e'%err)
    print('     ',' '.join('%d:%.1e'%(s,r) for s,_,r in log))
print("---- eigenvalue error of the wanted set vs keep")
Tf=build_T(A[:29],Bs[:28]); thf=np.linalg.eigvalsh(Tf)[::-1][:Neig]
print('full at step 29: eig err %.1e'%(np.abs(thf-evals_true).max()/evals_true[0]))
for s1,every,k in ((20,3,64),(20,3,96),(20,3,128),(20,3,192),(20,3,256),(14,5,128),(14,5,192),(14,5,224),(24,2,64),(24,2,96),(24,2,128),(26,1,64),(26,1,96)):
    log,err=run_chain(s1,every,k)
    print('s1',s1,'every',every,'keep',k,'-> converged at step',log[-1][0],'checks',len(log),'dims',[d for _,d,_ in log],'eig err %.1e'%err)
