# the bench step with and without a host synchronisation after every step: is the gap between kernel time and step time host latency?
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
N,K,B=1024,512,10_000_000
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.specialize()
n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
llr=torch.empty((B,N),dtype=torch.float32,device='cuda'); info=torch.empty((B,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,B); ctx.sync()
cnt=torch.zeros((64,3),dtype=torch.int64,device='cuda')
def run(K_,sync_each,timing):
    ctx.enable_timing(timing)
    for i in range(3):
        code.sc_decode_count_dev(llr.data_ptr(),np.float32,info.data_ptr(),B,cnt[i].data_ptr(),fast=True,info_hat_ptr=hat.data_ptr())
    ctx.sync(); t0=time.perf_counter()
    for i in range(K_):
        code.sc_decode_count_dev(llr.data_ptr(),np.float32,info.data_ptr(),B,cnt[3+i].data_ptr(),fast=True,info_hat_ptr=hat.data_ptr())
        if sync_each: ctx.sync()
    ctx.sync(); dt=time.perf_counter()-t0
    print('steps',K_,'sync each' if sync_each else 'one sync','timing bracket' if timing else 'no bracket','%.3f ms per step'%(1e3*dt/K_),'kernel',code.last_launch()['kernel'],'%.3f ms'%code.last_launch()['last_ms'],flush=True)
for _ in range(2):
    run(20,True,True); run(20,True,False); run(20,False,False)
