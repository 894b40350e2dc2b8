# list decoding of codes longer than the register kernels serve: rate of whatever kernel the library picks
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
for N,K,L,nb in [(2048,1024,8,1<<15),(2048,1024,32,1<<14),(4096,2048,8,1<<14),(1024,512,8,1<<16)]:
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb); ctx.sync()
    ms=[]
    for _ in range(2):
        code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb); ms.append(code.last_launch()['last_ms'])
    ctx.sync(); fe=int((hat!=info).any(dim=1).sum())
    print(N,K,'L',L,code.last_launch()['kernel'],'%.3f M frames/s'%(nb/(min(ms)*1e-3)/1e6),'fer %.3g'%(fe/nb),flush=True)
