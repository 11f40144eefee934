"""NumPy emulation of the blocked tridiagonalisation as eig_dense.hip implements it (k_tdb_column /
k_tdb_symv / k_tdb_panel_end: LAPACK dlatrd with the reductions deferred to launch boundaries).
Formula-level check of the device algorithm: eigenvalues of T and Q T Q^T against the input.
   python tests/probes/blocked_tridiag_emulation.py"""
import numpy as np
def blocked_td(A, nb=32):
    A=A.copy(); n=A.shape[0]
    d=np.zeros(n); e=np.zeros(n); taus=np.zeros(n)
    P1=np.zeros((n,2*nb))   # [V | W]
    refl=np.zeros((n,n))
    j0=0
    while j0<n:
        nbk=min(nb,n-j0)
        wprime=np.zeros(n); c_prev=0.0; tau_prev=0.0; part2=0.0
        for jj in range(nbk):
            j=j0+jj
            # ---- K1
            if jj>0:
                c_prev=0.5*tau_prev*part2
                # finalize W[:, jj-1] rows >= j
                P1[j:,nb+jj-1]=wprime[j:]-c_prev*P1[j:,jj-1]
            Vj=P1[j,:nb].copy(); Wj=P1[j,nb:].copy()
            a=np.zeros(n)
            for r in range(j,n):
                a[r]=A[j,r]-np.dot(P1[r,:jj],Wj[:jj])-np.dot(P1[r,nb:nb+jj],Vj[:jj])
            norm2=np.sum(a[j+2:]**2)
            pv=P1[j+1:,:nb].T@a[j+1:] ; pw=P1[j+1:,nb:].T@a[j+1:]
            pv[jj:]=0; pw[jj:]=0
            d[j]=a[j]
            if j==n-1: break
            # ---- K2
            alpha=a[j+1]
            tau=0.0; scale=0.0; beta=alpha
            if norm2>0:
                beta=-np.copysign(np.sqrt(alpha*alpha+norm2),alpha)
                tau=(beta-alpha)/beta; scale=1.0/(alpha-beta)
            v=np.zeros(n); v[j+1]=1.0; v[j+2:]=scale*a[j+2:]
            e[j]=beta; taus[j]=tau
            Vtv=P1[j+1,:nb]+scale*(pv-P1[j+1,:nb]*alpha)
            Wtv=P1[j+1,nb:]+scale*(pw-P1[j+1,nb:]*alpha)
            Vtv[jj:]=0; Wtv[jj:]=0
            y=A[j+1:,j+1:]@v[j+1:]
            corr=P1[j+1:,:nb]@Wtv+P1[j+1:,nb:]@Vtv   # uses only k<jj since others zeroed
            wp=tau*(y-corr)
            wprime[:]=0; wprime[j+1:]=wp
            part2=np.dot(wp,v[j+1:])
            P1[:,jj]=0; P1[j+1:,jj]=v[j+1:]
            refl[j,j+1:]=v[j+1:]
            tau_prev=tau
        else:
            pass
        j1=j0+nbk
        if j1<n:
            # K3: finalize last W column, build P2, syr2k
            c=0.5*tau_prev*part2
            P1[j1:,nb+nbk-1]=wprime[j1:]-c*P1[j1:,nbk-1]
            V=P1[j1:,:nbk]; W=P1[j1:,nb:nb+nbk]
            A[j1:,j1:]-= V@W.T+W@V.T
        j0=j1
    return d,e,taus,refl
rng=np.random.default_rng(0)
for n,nb in ((5,2),(7,4),(33,8),(100,32),(257,32),(64,32)):
    M=rng.standard_normal((n,n)); M=M+M.T
    d,e,taus,refl=blocked_td(M,nb)
    T=np.diag(d)+np.diag(e[:n-1],1)+np.diag(e[:n-1],-1)
    ev=np.linalg.eigvalsh(T); ref=np.linalg.eigvalsh(M)
    # check Q
    Q=np.eye(n)
    for j in range(n-1):
        v=refl[j].copy(); H=np.eye(n)-taus[j]*np.outer(v,v); Q=Q@H
    print(n,nb,"eig err %.2e"%np.abs(ev-ref).max(), "QTQ^T err %.2e"%np.abs(Q@T@Q.T-M).max())
