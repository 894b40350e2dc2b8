# rates of the calls that still run LDS-staged kernels
import sys, os; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
def frames(N,K,nb,dt,tdt):
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,1,0,nb); ctx.sync(); return code,llr,info,hat
for N,nb in [(2048,1<<15),(4096,1<<14),(16384,1<<12)]:
    code,llr,info,hat=frames(N,N//2,nb,np.float64,torch.float64)
    for fast in (True,False):
        for _ in range(2): code.sc_decode_dev(llr.data_ptr(),np.float64,hat.data_ptr(),nb,fast=fast)
        li=code.last_launch(); print('SC fp64',N,'fast' if fast else 'every leaf',li['kernel'],'%.2f M frames/s'%(nb/(li['last_ms']*1e-3)/1e6),flush=True)
    code,llr,info,hat=frames(N,N//2,nb,np.float32,torch.float32)
    for _ in range(2): code.sc_decode_dev(llr.data_ptr(),np.float32,hat.data_ptr(),nb,fast=False)
    li=code.last_launch(); print('SC fp32',N,'every leaf',li['kernel'],'%.2f M frames/s'%(nb/(li['last_ms']*1e-3)/1e6),flush=True)
for N,L,nb in [(2048,4,1<<14),(2048,2,1<<14),(1024,1,1<<16)]:
    code,llr,info,hat=frames(N,N//2,nb,np.float32,torch.float32)
    for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb)
    li=code.last_launch(); print('SCL fp32',N,'L',L,li['kernel'],'%.2f M frames/s'%(nb/(li['last_ms']*1e-3)/1e6),flush=True)
