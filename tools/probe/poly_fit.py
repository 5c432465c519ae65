"""DEV TOOL: minimax (Remez) polynomials for the f32 inner terms of pow / log (elementwise.hip: exp_q, log1p_q).
   Q(t) = (e^t − 1 − t)/t² on |t| ≤ ln2/2 and q(u) = (log1p(u) − u + u²/2)/u³ on |u| ≤ 2^-7; prints the coefficients rounded
   to f32 and the error they leave in the result.   python tools/probe/poly_fit.py"""
import numpy as np, mpmath as mp
mp.mp.dps=40
def remez(f, a, b, deg, iters=30):
    # simple Remez exchange for absolute error, using mp for function and longdouble linear algebra
    n=deg+2
    xs=[ (a+b)/2 + (b-a)/2*mp.cos(mp.pi*(n-1-i)/(n-1)) for i in range(n)]
    for _ in range(iters):
        A=mp.matrix(n,n); rhs=mp.matrix(n,1)
        for i,x in enumerate(xs):
            for k in range(deg+1): A[i,k]=x**k
            A[i,deg+1]=(-1)**i
            rhs[i]=f(x)
        sol=mp.lu_solve(A,rhs)
        c=[sol[k] for k in range(deg+1)]; E=sol[deg+1]
        # find extrema of error on fine grid
        N=4000
        grid=[a+(b-a)*i/N for i in range(N+1)]
        err=[f(x)-sum(c[k]*x**k for k in range(deg+1)) for x in grid]
        # pick local extrema with alternating signs
        ext=[]
        for i in range(N+1):
            l=err[i-1] if i>0 else None; r=err[i+1] if i<N else None
            e=err[i]
            if (l is None or abs(e)>=abs(l) or (e>0)!=(l>0)) and (r is None or abs(e)>=abs(r) or (e>0)!=(r>0)):
                if ext and (err[ext[-1]]>0)==(e>0):
                    if abs(e)>abs(err[ext[-1]]): ext[-1]=i
                else: ext.append(i)
        if len(ext)<n: break
        # keep n largest consecutive
        while len(ext)>n:
            if abs(err[ext[0]])<abs(err[ext[-1]]): ext.pop(0)
            else: ext.pop()
        xs=[grid[i] for i in ext]
    maxerr=max(abs(e) for e in err)
    return [float(x) for x in c], float(maxerr)
def Qf(t):
    t=mp.mpf(t)
    if abs(t)<mp.mpf('1e-12'): return mp.mpf(1)/2
    return (mp.exp(t)-1-t)/t**2
h=float(mp.log(2)/2)*1.0001
for deg in (3,4,5):
    c,e=remez(Qf,-h,h,deg)
    print('Q deg',deg,'max err in Q',e,'=> rel err in exp ~ t^2*e =',e*h*h, 'log2', np.log2(e*h*h))
    print('  ',[float(np.float32(x)).hex() for x in c], c)
def qf(u):
    u=mp.mpf(u)
    if abs(u)<mp.mpf('1e-12'): return mp.mpf(1)/3
    return (mp.log1p(u)-u+u*u/2)/u**3
for deg in (1,2):
    c,e=remez(qf,-2.0**-7,2.0**-7,deg)
    print('q deg',deg,'max err in q',e,'=> rel to u: u^2*e =',e*2.0**-14,'log2',np.log2(e*2.0**-14))
    print('  ',[float(np.float32(x)).hex() for x in c], c)
