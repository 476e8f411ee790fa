"""Dev-only NumPy prototype of the blocked two-sided Jacobi scheme used by
precondition_amd/csrc/eigh.hip (parameter study: block width, inner sweeps,
polish). Not used by the product or the tests."""
import numpy as np, time
F=np.float32
def round_robin(m):
    # m even: returns list of rounds, each list of (i,j) pairs
    idx=list(range(m)); rounds=[]
    for r in range(m-1):
        rounds.append([(min(idx[k],idx[m-1-k]),max(idx[k],idx[m-1-k])) for k in range(m//2)])
        idx=[idx[0]]+[idx[-1]]+idx[1:-1]
    return rounds
def inner_jacobi(S, sweeps):
    # cyclic jacobi on small symmetric S (float32), returns Q with Q^T S Q ~ more diagonal
    m=S.shape[0]; S=S.copy(); Q=np.eye(m,dtype=F)
    rr=round_robin(m)
    for sw in range(sweeps):
        for pairs in rr:
            p=np.array([a for a,b in pairs]); q=np.array([b for a,b in pairs])
            app=S[p,p]; aqq=S[q,q]; apq=S[p,q]
            with np.errstate(all='ignore'):
                tau=(aqq-app)/(F(2)*apq)
                t=np.sign(tau)/(np.abs(tau)+np.sqrt(F(1)+tau*tau))
                t=np.where(tau==0, F(1), t)
            small = np.abs(apq) <= F(1e-30)
            t=np.where(small, F(0), t).astype(F)
            c=(F(1)/np.sqrt(F(1)+t*t)).astype(F); s=(t*c).astype(F)
            # rows
            Sp=S[p,:].copy(); Sq=S[q,:].copy()
            S[p,:]=c[:,None]*Sp - s[:,None]*Sq; S[q,:]=s[:,None]*Sp + c[:,None]*Sq
            Sp=S[:,p].copy(); Sq=S[:,q].copy()
            S[:,p]=c[None,:]*Sp - s[None,:]*Sq; S[:,q]=s[None,:]*Sp + c[None,:]*Sq
            Qp=Q[:,p].copy(); Qq=Q[:,q].copy()
            Q[:,p]=c[None,:]*Qp - s[None,:]*Qq; Q[:,q]=s[None,:]*Qp + c[None,:]*Qq
    return Q
def block_jacobi(A, b=64, inner=1, max_sweeps=12, tol=1e-6, verbose=True):
    n=A.shape[0]; A=A.astype(F).copy(); V=np.eye(n,dtype=F); nb=n//b
    rr=round_robin(nb); nrm=np.linalg.norm(A)
    hist=[]
    for sw in range(max_sweeps):
        off2=0.0
        for pairs in rr:
            Qs=[]
            for (I,J) in pairs:
                idx=np.r_[I*b:(I+1)*b, J*b:(J+1)*b]
                S=A[np.ix_(idx,idx)]
                off2+=2*float(np.sum(S[:b,b:].astype(np.float64)**2))
                Qs.append((idx, inner_jacobi(S, inner)))
            for idx,Q in Qs: A[idx,:]=Q.T@A[idx,:]
            for idx,Q in Qs: A[:,idx]=A[:,idx]@Q; V[:,idx]=V[:,idx]@Q
        offd=np.sqrt(off2)/nrm
        hist.append(offd)
        if verbose: print('sweep',sw,'off(blocks)/|A|',offd, 'full off', np.linalg.norm(A-np.diag(np.diag(A)))/nrm)
        if offd<tol: break
    return np.diag(A).copy(), V, hist
if __name__=='__main__':
    import sys
    n=int(sys.argv[1]); b=int(sys.argv[2]); inner=int(sys.argv[3])
    g=np.random.default_rng(0).standard_normal((n,4*n)).astype(F); A=(g@g.T).astype(F)
    A=A+F(1e-6*np.linalg.eigvalsh(A.astype(np.float64)).max())*np.eye(n,dtype=F)
    t=time.time(); e,V,h=block_jacobi(A,b,inner); print('time',time.time()-t)
    w,U=np.linalg.eigh(A)
    p=2
    val=(V*(e**(-1.0/p)).astype(F))@V.T
    ref=(U*(w**(-1.0/p)))@U.T
    ref64w,ref64U=np.linalg.eigh(A.astype(np.float64)); ref64=(ref64U*ref64w**(-1.0/p))@ref64U.T
    print('val vs lapack32', np.linalg.norm(val-ref)/np.linalg.norm(ref), 'val vs f64', np.linalg.norm(val-ref64)/np.linalg.norm(ref64), 'lapack32 vs f64', np.linalg.norm(ref-ref64)/np.linalg.norm(ref64))
    print('orth', np.abs(V.T@V-np.eye(n)).max(), 'resid', np.abs(V.T@A@V-np.diag(e)).max(), 'lapack resid', np.abs(U.T@A@U-np.diag(w)).max())
