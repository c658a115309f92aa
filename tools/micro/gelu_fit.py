"""Development script: the polynomial behind srv_gelu2 (csrc/srv_kernels.hip) - GELU's even part E(t) = t (Phi(t) - 1/2) as t^2 P(t^2) on
[0, A], minimax by reweighted least squares, evaluated in f32 the way the kernel does (Horner of fmas, the |x| > A tail as (|x| - A) / 2).
Prints fit and f32 evaluation errors for a few (A, degree) and the coefficients of the one in use (A = 4, 7 coefficients)."""
import numpy as np, math
erf=np.vectorize(math.erf)
def Phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def gelu(x): return x*Phi(x)
f32=lambda a: np.asarray(a,np.float64).astype(np.float32)
def fitE(A,n):
    t=np.cos(np.pi*(np.arange(6000)+0.5)/6000)*A; t=t[t>0]
    y=(Phi(t)-0.5)/t
    X=np.stack([t**(2*k) for k in range(n)],1)
    w=np.ones_like(t)
    for it in range(200):
        c=np.linalg.lstsq(X*w[:,None],y*w,rcond=None)[0]
        e=np.abs((X@c-y)*t*t)
        w=w*(1+2*e/e.max())
    return c,e.max()
def evalE32(c,A,x):
    x=f32(x); ax=np.abs(x); t=np.minimum(ax,np.float32(A)); t2=t*t
    q=np.full_like(t,np.float32(c[-1]))
    for k in range(len(c)-2,-1,-1): q=(q.astype(np.float64)*t2+np.float32(c[k])).astype(np.float32)
    tail=ax-t
    e=(t2.astype(np.float64)*q).astype(np.float32)
    y=(x.astype(np.float64)*0.5+e).astype(np.float32)
    y=(tail.astype(np.float64)*0.5+y).astype(np.float32)
    return y.astype(np.float64)
xs=np.linspace(-12,12,800001); g=gelu(xs)
for A in (3.5,3.75,4.0,4.25,4.5):
    for n in (5,6,7,8):
        c,e=fitE(A,n); y=evalE32(c,A,xs); err=np.abs(y-g)
        print("A=%.2f n=%d fit %.1e  max %.2e at %.2f"%(A,n,e,err.max(),xs[err.argmax()]))
c,e=fitE(4.0,7); print(repr(c))
