# one-kernel Monte-Carlo loop: counters against the staged pipeline, then its rate
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
os.environ['POLAR_EXPERIMENT']='1'
import numpy as np, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
for (N,K) in [(1024,512),(512,256),(128,64)]:
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.specialize(); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    a=code.mc_run(_native.DEC_SC,1,np.float32,n0,2024,0,(1<<20)+4096); k=code.last_launch()['kernel']
    os.environ['POLAR_MC_KERNEL']='staged'
    b=code.mc_run(_native.DEC_SC,1,np.float32,n0,2024,0,(1<<20)+4096); k2=code.last_launch()['kernel']
    del os.environ['POLAR_MC_KERNEL']
    n=1<<25 if N<1024 else 1<<24
    best=1e9
    for _ in range(3):
        t0=time.perf_counter(); c=code.mc_run(_native.DEC_SC,1,np.float32,n0,2024,0,n); best=min(best,time.perf_counter()-t0)
    print(N,K,k,'counters',a.tolist(),'staged',k2,b.tolist(),'same' if a.tolist()==b.tolist() else 'DIFFERENT','M frames/s %.1f'%(n/best/1e6),flush=True)
