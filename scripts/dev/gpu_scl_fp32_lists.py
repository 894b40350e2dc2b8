# fp32 list decoding with the whole sorted list and its metrics as outputs: register kernels against the LDS-staged kernel
import sys, os; sys.path[:0]=['polar-codes_amd','.']
os.environ['POLAR_EXPERIMENT']='1'
import numpy as np, _native, functions as pcf
from oracle import oracle
ctx=_native.default_ctx(0)
for N,K,L in [(64,32,8),(256,128,4),(512,256,8),(1024,512,8),(1024,512,2),(128,64,32),(512,256,32),(1024,512,16),(2048,1024,8),(2048,1024,16)]:
    mask=pcf.degaBuild(N,K,2.0); code=_native.Code(ctx,mask)
    rs=np.random.RandomState(N+L); B=400
    x=oracle.polar_encode(rs.randint(0,2,size=(B,K)),mask).reshape(B,N)
    llr=code.channel_llr(x,1.0/(K/N*10**0.1),31,0,np.float32)
    llr[:40]=np.round(llr[:40]); llr[40:44]*=200.0
    os.environ.pop('POLAR_SCL_KERNEL',None)
    a,apm,auh=code.scl_decode(llr,L,want_list=True); k=code.last_launch()['kernel']
    os.environ['POLAR_SCL_KERNEL']='lds'
    b,bpm,buh=code.scl_decode(llr,L,want_list=True); k2=code.last_launch()['kernel']
    print(N,K,L,k,'vs',k2,'bits differ',int(np.any(a!=b,axis=1).sum()),'lists differ',int(np.any(auh!=buh,axis=(1,2)).sum()),'metrics differ',int(np.any(~((apm==bpm)|(np.isnan(apm)&np.isnan(bpm))),axis=1).sum()),flush=True)
