import sys, os, faulthandler; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, _native
from oracle import oracle
faulthandler.enable()
ctx=_native.default_ctx(0)
for (N,K,L,dt) in [(1024,512,8,np.float32),(256,128,8,np.float32),(256,128,4,np.float32),(64,32,8,np.float64),(64,32,2,np.float64),(64,3,2,np.float64)]:
    rs=np.random.RandomState(L)
    mask=np.zeros(N,dtype=int); mask[rs.permutation(N)[:K]]=1
    code=_native.Code(ctx,mask)
    llr=(rs.standard_normal((3,N))*2.5).astype(dt)
    print('try',N,K,L,dt.__name__,flush=True)
    info,pm,uh=code.scl_decode(llr,L,want_list=(dt is np.float64))
    print(' ok',code.last_launch()['kernel'],flush=True)
