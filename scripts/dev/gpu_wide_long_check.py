# the two-lanes-per-path list decoder at N = 2048 (or argv[1] = 4096, where only L = 8 has an LDS-staged counterpart): L = 16 against the LDS-staged kernel, L = 32 (which the LDS-staged kernel has
# no room for) against the CPU oracle on a few frames; rates
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
os.environ['POLAR_EXPERIMENT']='1'
import numpy as np, torch, _native, functions as pcf
from channel import Channel
from oracle import oracle
ctx=_native.default_ctx(0); ctx.enable_timing(True)
N=int(sys.argv[1]) if len(sys.argv)>1 else 2048; K=N//2
mask=pcf.degaBuild(N,K,2.0)
code=_native.Code(ctx,mask); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
for L,dt,tdt,nb in [(8,np.float32,torch.float32,1<<14),(8,np.float64,torch.float64,1<<13),(16,np.float32,torch.float32,1<<14),(32,np.float32,torch.float32,1<<14),(32,np.float64,torch.float64,1<<13)]:
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.zeros_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,1,0,nb); ctx.sync()
    ms=[]
    for _ in range(2):
        code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ms.append(code.last_launch()['last_ms'])
    k=code.last_launch()['kernel']; ctx.sync()
    fe=int((hat!=info).any(dim=1).sum())
    msg='%d %d L %d %s %s %.3f M frames/s fer %.3g'%(N,K,L,dt.__name__,k,nb/(min(ms)*1e-3)/1e6,fe/nb)
    if L<=16 and not (N==4096 and L==16):
        os.environ['POLAR_SCL_KERNEL']='lds'; nchk=1<<11
        ref=torch.zeros((nchk,code.Kw),dtype=torch.int32,device='cuda'); code.scl_decode_dev(llr.data_ptr(),dt,L,ref.data_ptr(),nchk); k2=code.last_launch()['kernel']; ctx.sync()
        del os.environ['POLAR_SCL_KERNEL']
        msg+=' | vs %s differ %d / %d'%(k2,int((ref!=hat[:nchk]).any(dim=1).sum()),nchk)
    else:
        nchk=24; h=hat[:nchk].cpu().numpy(); l=llr[:nchk].cpu().numpy(); bad=0
        for f in range(nchk):
            uh,pm,cnt=oracle.scl_decode(l[f],mask,L)
            bits=uh[0][np.asarray(mask)==1]
            got=np.unpackbits(h[f].view(np.uint8),bitorder='little')[:K]
            bad+=int(not np.array_equal(bits,got))
        msg+=' | vs CPU oracle differ %d / %d'%(bad,nchk)
    print(msg,flush=True)
