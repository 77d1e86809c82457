import sys, os
R=os.getcwd()
for p in ("approximategps.jl_amd","oracle","tests"): sys.path.insert(0, os.path.join(R,p))
import torch, numpy as np, svgp_oracle as o
from approxgp import _ffi
from helpers import device_model
ctx=_ffi.Context(0)
for M,N,d in ((2100,3000,4),(2048,20000,16)):
  for jit in (1e-3,1e-2):
    x,y,sva,s2=o.synth_problem(88,N,M,d,dtype=np.float32,jitter=jit, family=o.KERNEL_MATERN52 if d==16 else o.KERNEL_SE)
    model=device_model(ctx,sva,dtype=np.float32,sigma2=s2); data=_ffi.DeviceData(ctx,x,y,np.float32)
    vr,gr=o.elbo_grad(sva,x,y,sigma2=s2,num_data=2.0*N); v,_,g=model.elbo_grad(data,0,N,2.0*N)
    e=lambda a,b: float(np.abs(np.asarray(a,dtype=float).reshape(np.shape(b),order="F")-np.asarray(b)).max()/max(np.abs(np.asarray(b)).max(),1e-12))
    print(M,N,d,"jitter",jit,"val",abs(v-vr)/abs(vr),{k:round(e(g[k],gr[k]),6) for k in ("m","Lq","inv_lengthscale","z")}, "var", abs(g["variance"]-gr["variance"])/abs(gr["variance"]))
    model.free(); data.free()
