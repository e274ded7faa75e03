import numpy as np, sys
d=np.fromfile(sys.argv[1],dtype=np.int64).reshape(-1,3)
t=d[:,0]; lev=d[:,1]; cand=d[:,2]
w=t.reshape(-1,64).max(1); wl=lev.reshape(-1,64).max(1); wc=cand.reshape(-1,64).max(1)
print('lanes: levels hist', np.bincount(lev))
print('cand per lane mean/median/p99/max', cand.mean(), np.median(cand), np.percentile(cand,99), cand.max())
print('waves: time mean/median/p90/p99/max', w.mean(), np.median(w), np.percentile(w,90), np.percentile(w,99), w.max())
print('waves: maxlevel hist', np.bincount(wl))
print('waves: max cand mean', wc.mean(), 'sum lane cand/64 mean', cand.reshape(-1,64).sum(1).mean()/64)
for L in range(1,wl.max()+1):
    m=wl==L
    if m.any(): print(' level',L,'waves',m.sum(),'mean time',w[m].mean(),'mean maxcand',wc[m].mean())
blk=w.reshape(-1,4).max(1)
print('block time mean', blk.mean(), 'max', blk.max(), 'sum/1024 SIMD*4..', blk.sum())
