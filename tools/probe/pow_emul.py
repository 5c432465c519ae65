import numpy as np, math
F=np.float32; D=np.float64
# table
RC=np.empty(128); LC=np.empty(128)
for j in range(128):
    c=1+(j+0.5)/128
    big = j>=53
    cp = c/2 if big else c
    rc = 1.0/cp
    if j==0 or j==127: rc=1.0
    RC[j]=rc
    # high precision log2 via mpmath-free: use math.log2 on exact double rc (error ~1ulp f64)
    LC[j]=-math.log2(rc)
LOG2E=float.fromhex('0x1.71547652b82fep+0'); LN2=float.fromhex('0x1.62e42fefa39efp-1')
def powm(x,y):
    xd=x.astype(D); yd=y.astype(D)
    bits=xd.view(np.uint64); hi=(bits>>np.uint64(32)).astype(np.int64)
    e=((hi>>20)&0x7ff)-1023
    j=(hi>>13)&127
    big=j>=53
    mbits=(bits&np.uint64(0x000fffffffffffff))|np.where(big,np.uint64(0x3fe0000000000000),np.uint64(0x3ff0000000000000))
    m=mbits.view(D)
    e=e+big
    u=m*RC[j]-1.0   # fma exact-ish; emulate: m*RC has rounding. use higher precision
    u=(m.astype(np.longdouble)*RC[j].astype(np.longdouble)-1).astype(D)
    uf=u.astype(F)
    q3=uf*F(float.fromhex('0x1.999f5p-3'))+F(float.fromhex('-0x1.0002p-2')); q3=uf*q3+F(float.fromhex('0x1.555556p-2'))  # log1p_q
    u2=u*u
    t=u2*(-0.5)+u
    l1p=(u2*u)*q3.astype(D)+t
    L=l1p*LOG2E+(e.astype(D)+LC[j])
    w=np.clip(yd*L,-2000,2000)
    kd=np.rint(w); r=w-kd
    tt=r*LN2; z=tt*tt; tf=tt.astype(F)
    q=F(float.fromhex('0x1.a17e0cp-13'))  # exp_q: degree-5 minimax (tools/probe/poly_fit.py)
    for c in ('0x1.6d4328p-10','0x1.1110acp-7','0x1.5554eap-5','0x1.555556p-3','0x1p-1'):
        q=tf*q+F(float.fromhex(c))
    p=z*q.astype(D)+(1.0+tt)
    with np.errstate(over='ignore',under='ignore'):
        res=np.ldexp(p,kd.astype(np.int64)).astype(F)
    return res
def ordv(a):
    i=a.view(np.int32).astype(np.int64); return np.where(i<0,-(i&0x7fffffff),i)
rng=np.random.default_rng(3)
n=1<<22
for name,(x,y) in {
 "generic":(np.abs(rng.standard_normal(n)).astype(F)*10+F(1e-30), (rng.standard_normal(n)*3).astype(F)),
 "wide":((2.0**rng.uniform(-126,127,n)).astype(F),(rng.uniform(-1.2,1.2,n)).astype(F)),
 "near1":((1+rng.uniform(-1e-3,1e-3,n)).astype(F),(2.0**rng.uniform(0,16,n)*rng.choice([-1,1],n)).astype(F)),
 "near1b":((1+rng.uniform(-6e-7,6e-7,n)).astype(F),(2.0**rng.uniform(10,30,n)*rng.choice([-1,1],n)).astype(F)),
 "denorm":((2.0**rng.uniform(-149,-120,n)).astype(F),(rng.uniform(-0.9,0.9,n)).astype(F)),
 "bigexp":((2.0**rng.uniform(-3,3,n)).astype(F),(rng.uniform(-60,60,n)).astype(F)),
}.items():
    got=powm(x,y)
    with np.errstate(over='ignore',under='ignore',invalid='ignore'):
        exp=np.power(x.astype(D),y.astype(D)).astype(F)
    ok = np.isfinite(exp)
    d=np.abs(ordv(got)-ordv(exp))
    print(name,"max ulp",d.max(),"frac>0",(d>0).mean(), "n inf/0", (~ok).sum(), (exp==0).sum())
