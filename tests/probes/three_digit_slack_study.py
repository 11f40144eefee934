"""CPU study (round 5): what dropping the low x low digit product of the matrix-free
Diffuse's pruner would cost.  T3 = 65536 hh + 256 (hl + lh) omits ll_ij = sum_k l_ik l_jk, so the
proven slack must widen by a bound on |ll_ij|: 128 * L_i (L_i = sum_k |l_ik|; "3a") or, by
Cauchy-Schwarz, Lambda_i * Lambda_max (Lambda_i = ||l_i||_2; "3b").  Prints, on refined affinities
of the BASELINE configurations, the relative slack and the candidates per row under the shipped
four-product bound and under both three-product bounds, and how many j lie within eps of each
row maximum of S (the candidate count explodes between eps = 1e-4 and 3e-4: where the wider
slacks land).   python tests/probes/three_digit_slack_study.py a|b|c
(oracle = test infrastructure: this file lives under tests/)"""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle'))
import spectral_oracle as so
def refined_A(x, p=0.95):
    a = so.affinity(x)
    a = so.crop_diagonal(a); a = so.gaussian_blur(a, 1.0)
    a = so.row_wise_threshold(a, p, so.THRESHOLD_ROW_MAX if hasattr(so,'THRESHOLD_ROW_MAX') else 0, 0.01) if False else so.row_wise_threshold(a, p_percentile=p)
    a = so.symmetrize(a)
    return a
def study(name, x, p=0.95):
    t=time.time()
    A = refined_A(x, p); n = A.shape[0]
    amax = np.abs(A).max(); sigma = 32639.0/amax
    q = np.rint(A*sigma); h = np.floor((q+128)/256); l = q-256*h
    R = np.abs(q).sum(1); L = np.abs(l).sum(1); Lam = np.sqrt((l*l).sum(1))
    T = q@q.T
    ll = l@l.T
    T3 = T-ll
    S = A@A.T
    M = T.max(1); M3 = T3.max(1)
    sl4 = (R+R.max()) + 0.52*n
    sl3a = sl4 + 2*128*L
    sl3b = sl4 + 2*Lam*Lam.max()
    c4 = (T >= (M-sl4)[:,None]).sum(1)
    c3a = (T3 >= (M3-sl3a)[:,None]).sum(1)
    c3b = (T3 >= (M3-sl3b)[:,None]).sum(1)
    # check correctness: true argmax in candidate set
    jm = S.argmax(1)
    ok4 = (T[np.arange(n),jm] >= M-sl4).all(); ok3 = (T3[np.arange(n),jm] >= M3-sl3a).all(); ok3b=(T3[np.arange(n),jm] >= M3-sl3b).all()
    print(name, 'n',n,'p',p, 'rel slack4 %.2e 3a %.2e 3b %.2e'%( (sl4/M).mean(), (sl3a/M3).mean(), (sl3b/M3).mean()),
          'cands mean/max 4: %.2f/%d  3a: %.2f/%d  3b: %.2f/%d'%(c4.mean(),c4.max(),c3a.mean(),c3a.max(),c3b.mean(),c3b.max()),
          'rows>8: %d %d %d'%((c4>8).sum(),(c3a>8).sum(),(c3b>8).sum()), ok4, ok3, ok3b, 'actual |ll| max %.2e'%np.abs(ll).max(), '%.1fs'%(time.time()-t))
    # how many j within eps of max of S
    Sm = S.max(1)
    for eps in (1e-4,3e-4,1e-3,3e-3,1e-2):
        c = (S >= ((1-eps)*Sm)[:,None]).sum(1)
        print('   eps %.0e: mean %.2f max %d rows>8 %d'%(eps,c.mean(),c.max(),(c>8).sum()))
if __name__=='__main__':
    which = sys.argv[1]
    if which=='a':
        study('blobs2048', so.blobs(2048,128,4,seed=2048))
        study('blobs4096', so.blobs(4096,256,8,seed=4096))
        for p in (0.55,0.75):
            study('blobs4096', so.blobs(4096,256,8,seed=4096), p)
    elif which=='b':
        study('blobs8192', so.blobs(8192,256,8,seed=0))
    elif which=='c':
        rng=np.random.default_rng(512)
        ns = rng.integers(300,3001,512); ks=rng.integers(2,8,512)
        for i in (0,1,2,3,5,8):
            if ns[i]>=1536: study('utt%d'%i, so.blobs(int(ns[i]),256,int(ks[i]),seed=i))
        study('hard_iid', so.hard_inputs('iid',2048,256,2048) )
