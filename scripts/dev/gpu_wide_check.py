# wide list decoder on a long batch (workgroups loop over many frames): run-to-run identity and identity with the LDS kernel
import sys, os; sys.path[:0]=['polar-codes_amd','.']
os.environ['POLAR_EXPERIMENT']='1'
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
for (N,K,L,nb,dt) in [(1024,512,32,1<<15,np.float32),(512,256,32,1<<15,np.float32),(1024,512,16,1<<14,np.float32),(1024,512,32,1<<13,np.float64)]:
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    tdt=torch.float64 if dt is np.float64 else torch.float32
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda')
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,1,0,nb)
    outs=[]
    for r in range(3):
        hat=torch.zeros_like(info); code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ctx.sync(); outs.append(hat)
    k=code.last_launch()['kernel']
    os.environ['POLAR_SCL_KERNEL']='lds'
    ref=torch.zeros_like(info); code.scl_decode_dev(llr.data_ptr(),dt,L,ref.data_ptr(),nb); ctx.sync(); k2=code.last_launch()['kernel']
    del os.environ['POLAR_SCL_KERNEL']
    d01=int((outs[0]!=outs[1]).any(dim=1).sum()); d02=int((outs[0]!=outs[2]).any(dim=1).sum()); dr=int((outs[0]!=ref).any(dim=1).sum())
    fe=int((outs[0]!=info).any(dim=1).sum()); fer=int((ref!=info).any(dim=1).sum())
    print(N,K,L,dt.__name__,k,'run-to-run differ',d01,d02,'vs',k2,'differ',dr,'fe',fe,'fe(lds)',fer,flush=True)
