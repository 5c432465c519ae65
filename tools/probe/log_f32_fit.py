import numpy as np, mpmath as mp, sys
mp.mp.dps=40
def lawson(g, wt, a, b, ncoef, N=4000, iters=600):
    k=np.arange(N); z=(a+b)/2+(b-a)/2*np.cos(np.pi*(k+0.5)/N)
    gz=np.array([float(g(mp.mpf(float(x)))) for x in z]); w=np.array([float(wt(mp.mpf(float(x)))) for x in z])
    V=np.vander(z, ncoef, increasing=True)
    lam=np.ones(N)/N; best=None
    for it in range(iters):
        sw=np.sqrt(lam)*w
        c,*_=np.linalg.lstsq(V*sw[:,None], gz*sw, rcond=None)
        e=np.abs(w*(gz-V@c)); m=e.max()
        if best is None or m<best[0]: best=(m,c.copy())
        lam=lam*e; lam/=lam.sum()
    return best
lo,hi=float(mp.sqrt(0.5))-1, float(mp.sqrt(2))-1
def g(f):
    if abs(f)<mp.mpf('1e-8'): return mp.mpf(1)/3
    return (mp.log1p(f)-f+f*f/2)/f**3
def wt(f):
    # error of f^3*P relative to |log1p(f)|
    if abs(f)<mp.mpf('1e-8'): return mp.mpf(0)
    return abs(f**3/mp.log1p(f))
for nc in (7,8,9,10):
    m,c=lawson(g,wt,lo,hi,nc)
    print(nc,"rel err",m,"ulp",m/2**-24); print("  ",[float(np.float32(x)).hex() for x in c])
