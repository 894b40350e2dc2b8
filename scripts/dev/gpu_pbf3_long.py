# three simultaneous flips on a code longer than the register search serves: (2048, 1024 + CRC-24), LDS-staged runs, search in pieces
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
N,K=2048,1048
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.set_crc(0x1800063); code.set_reliability(pcf.GA(N,K,2.0))
for snr,nb,T in [(2.5,1<<14,3),(2.0,1<<13,3),(2.0,1<<13,2)]:
    n0=Channel('BPSK',snr,'SNRb',1024/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    runs=torch.zeros(nb,dtype=torch.int32,device='cuda')
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb,crc=True); ctx.sync()
    t0=time.time(); code.pbf_decode_dev(llr.data_ptr(),np.float32,T,hat.data_ptr(),nb,runs_ptr=runs.data_ptr()); ctx.sync(); dt=time.time()-t0
    fe=int((hat!=info).any(dim=1).sum())
    print('N',N,'T',T,'snr',snr,'frames',nb,'%.3f s'%dt,'%.1f k frames/s'%(nb/dt/1e3),code.last_launch()['kernel'],'fer %.3g'%(fe/nb),'runs mean %.1f max %d total %.3g -> %.1f M runs/s'%(runs.float().mean().item(),int(runs.max()),runs.double().sum().item(),runs.double().sum().item()/dt/1e6),flush=True)
