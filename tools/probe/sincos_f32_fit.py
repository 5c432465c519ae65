import numpy as np, mpmath as mp, sys
mp.mp.dps=40
def lawson(g, wt, a, b, ncoef, N=3000, iters=400):
    # minimise max |wt(z) * (g(z) - p(z))| over [a,b]
    k=np.arange(N); z=(a+b)/2+(b-a)/2*np.cos(np.pi*(k+0.5)/N)
    gz=np.array([float(g(mp.mpf(x))) for x in z]); w=np.array([float(wt(x)) for x in z])
    V=np.vander(z, ncoef, increasing=True)
    lam=np.ones(N)/N
    best=None
    for it in range(iters):
        sw=np.sqrt(lam)*w
        c,*_=np.linalg.lstsq(V*sw[:,None], gz*sw, rcond=None)
        e=np.abs(w*(gz-V@c)); m=e.max()
        if best is None or m<best[0]: best=(m,c.copy())
        lam=lam*e; lam/=lam.sum()
    return best
R=float(sys.argv[1]) if len(sys.argv)>1 else 0.80
def gs(z):
    if z<mp.mpf('1e-20'): return mp.mpf(-1)/6
    r=mp.sqrt(z); return (mp.sin(r)/r-1)/z
def gc(z):
    if z<mp.mpf('1e-20'): return mp.mpf(1)/24
    r=mp.sqrt(z); return (mp.cos(r)-1+z/2)/(z*z)
for nc in (3,4):
    m,c=lawson(gs, lambda z: z, 0.0, R*R, nc)
    print("sin ncoef",nc,"max rel err",m,"ulp(2^-24)",m/2**-24); print("  ",[float(np.float32(x)).hex() for x in c], list(c))
for nc in (2,3):
    m,c=lawson(gc, lambda z: z*z/float(mp.cos(mp.sqrt(z))), 0.0, R*R, nc)
    print("cos ncoef",nc,"max rel err",m,"ulp",m/2**-24); print("  ",[float(np.float32(x)).hex() for x in c], list(c))
