# CRC-aided and genie-aided list decoding at N = 2048 / 4096 on the two-lanes-per-path kernel against the LDS-staged kernel
import sys, os; sys.path[:0]=['polar-codes_amd','.']
os.environ['POLAR_EXPERIMENT']='1'
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
for N,K,L,dt,tdt,snr in [(2048,1048,8,np.float32,torch.float32,1.5),(2048,1048,16,np.float64,torch.float64,1.5),(4096,2072,8,np.float32,torch.float32,1.25)]:
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.set_crc(0x1800063)
    nb=1<<12
    n0=Channel('BPSK',snr,'SNRb',(K-24)/N).noisePower
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.zeros_like(info); ref=torch.zeros_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,3,0,nb,crc=True); ctx.sync()
    for mode in ('plain','crc','genie'):
        def run(dst):
            if mode=='plain': code.scl_decode_dev(llr.data_ptr(),dt,L,dst.data_ptr(),nb)
            elif mode=='crc': code.scl_decode_dev(llr.data_ptr(),dt,L,dst.data_ptr(),nb,crc=True)
            else: code.scl_decode_genie_dev(llr.data_ptr(),dt,L,info.data_ptr(),dst.data_ptr(),nb)
            ctx.sync(); return code.last_launch()['kernel']
        k=run(hat); os.environ['POLAR_SCL_KERNEL']='lds'; k2=run(ref); del os.environ['POLAR_SCL_KERNEL']
        print(N,K,L,dt.__name__,mode,k,'vs',k2,'differ',int((ref!=hat).any(dim=1).sum()),'/',nb,'frame errors',int((hat!=info).any(dim=1).sum()),flush=True)
