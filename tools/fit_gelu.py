"""Fit of the GEMM epilogues' one-transcendental erf-GELU (kjarni_amd/csrc/device_utils.h, gelu_erf_fast): P(a) = -log2 Phi(-a) on [0, 6] by
weighted Chebyshev least squares with a few reweighting rounds, printed in the power basis; max |error| of the f32 evaluation
over |x| <= 12 beside the Abramowitz & Stegun form of rounds 1-4."""
import numpy as np
from scipy.special import erfc, log_ndtr
from numpy.polynomial import chebyshev as Ch
A=6.0
a=np.linspace(0,A,200001)
# P(a) = -log2(0.5*erfc(a/sqrt2)) = -log_ndtr(-a)/ln2
P=-log_ndtr(-a)/np.log(2)
g=a*np.exp2(-P)
def fit(deg, iters=40):
    w=np.maximum(g*np.log(2),1e-12)
    t=2*a/A-1
    ww=w.copy()
    for it in range(iters):
        c=Ch.chebfit(t,P,deg,w=ww)
        err=np.abs((Ch.chebval(t,c)-P)*w)
        ww=ww*(1+ 4*err/err.max())**1  # push weight to where error is large
        ww/=ww.max()
    return c
def eval32(c_pow, x):
    # gelu via f32 Horner in a
    x=x.astype(np.float32); ax=np.minimum(np.abs(x),np.float32(A))
    p=np.full_like(ax,np.float32(c_pow[-1]))
    for k in range(len(c_pow)-2,-1,-1):
        p=(p*ax+np.float32(c_pow[k])).astype(np.float32)  # not fused, pessimistic
    e=np.exp2(-p.astype(np.float64)).astype(np.float32)
    gg=(ax*e).astype(np.float32)
    return (np.maximum(x,0)-gg).astype(np.float32)
xs=np.concatenate([np.linspace(-12,12,2000001), np.random.default_rng(0).standard_normal(1000000)*3]).astype(np.float32)
from scipy.special import erf
ref=0.5*xs.astype(np.float64)*(1+erf(xs.astype(np.float64)/np.sqrt(2)))
for deg in (6,7,8,9,10):
    c=fit(deg)
    # convert cheb (in t) to power basis in a
    pt=Ch.cheb2poly(c)            # poly in t
    # t = 2a/A - 1
    from numpy.polynomial import polynomial as Pl
    pa=np.zeros(1)
    base=np.array([-1.0,2.0/A])
    acc=np.array([1.0])
    for k,ck in enumerate(pt):
        pa=Pl.polyadd(pa,ck*acc); acc=Pl.polymul(acc,base)
    got=eval32(pa,xs)
    err=np.abs(got-ref)
    i=err.argmax()
    print(deg, "max abs err", err.max(), "at x=", xs[i], " coeffs", [float(np.float32(v)) for v in pa])
# current A&S for comparison
def as_gelu(x):
    x=x.astype(np.float32); ax=np.abs(x); z=(ax*np.float32(0.7071067811865475)).astype(np.float32)
    t=(np.float32(1)/(np.float32(0.3275911)*z+np.float32(1))).astype(np.float32)
    p=np.float32(1.061405429)*t+np.float32(-1.453152027); p=p*t+np.float32(1.421413741); p=p*t+np.float32(-0.284496736); p=p*t+np.float32(0.254829592); p=p*t
    e=np.exp2((-z*z*np.float32(1.4426950408889634)).astype(np.float64)).astype(np.float32)
    ea=(np.float32(1)-p*e).astype(np.float32)
    return ((ax*ea+x)*np.float32(0.5)).astype(np.float32)
print("A&S now: max abs err", np.abs(as_gelu(xs)-ref).max())
