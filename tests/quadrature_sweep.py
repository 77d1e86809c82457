"""Value and gradient against the oracle over Gauss-Hermite orders 1 ... 512 (the largest the C-ABI takes) for the quadrature-capable
likelihoods, both dtypes.  A script for the GPU box (the oracle is the checker): python tests/quadrature_sweep.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle","approximategps.jl_amd","tests"): sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT",ROOT), p))
import numpy as np, svgp_oracle as o
from approxgp import _ffi
from helpers import device_model, rel
ctx=_ffi.Context(0)
for lik in (o.LIK_BERNOULLI_LOGISTIC, o.LIK_GAUSSIAN, o.LIK_POISSON_EXP, o.LIK_GAMMA_EXP, o.LIK_BERNOULLI_NORMCDF):
  for qn in (1,2,3,31,64,129,200,400,512):
    for dtype in (np.float64, np.float32):
        x,y,sva,s2=o.synth_problem(900+qn,700,60,3,lik=lik,dtype=dtype)
        try:
            ref,gref=o.elbo_grad(sva,x,y,lik=lik,sigma2=s2,quadrature_n=qn)
        except Exception as e:
            print("oracle exc",lik,qn,repr(e)[:80]); continue
        m=device_model(ctx,sva,dtype=dtype,lik=lik,sigma2=s2,quadrature_n=qn); d=_ffi.DeviceData(ctx,x,y,dtype)
        try:
            v=m.elbo(d)[0]; vg,_,g=m.elbo_grad(d)
            eL=np.abs(np.asarray(g["Lq"],dtype=np.float64)-gref["Lq"]).max()/np.abs(gref["Lq"]).max()
            flag = "" if (rel(v,ref) < (1e-8 if dtype==np.float64 else 1e-4) and eL < (1e-6 if dtype==np.float64 else 5e-3)) else "  <<<<"
            print(f"lik {lik} qn {qn:3d} {dtype.__name__}: value {rel(v,ref):.1e} grad-value {rel(vg,ref):.1e} Lq {eL:.1e}{flag}")
        except Exception as e:
            print("device exc",lik,qn,dtype.__name__,repr(e)[:120])
        m.free(); d.free()
