# CRC-aided and genie-aided list decoding, (1024,536) + CRC-24, L = 8: rate of the register kernel and equality with the LDS-staged one
import sys, os; sys.path[:0]=['polar-codes_amd','.']
os.environ['POLAR_EXPERIMENT']='1'
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
N,K,L=1024,536,8
for dt,tdt in [(np.float32,torch.float32),(np.float64,torch.float64)]:
  code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.set_crc(0x1800063)
  for snr,nb in [(2.0,1<<19),(1.0,1<<17)]:
    if dt is np.float64: nb>>=2
    n0=Channel('BPSK',snr,'SNRb',512/N).noisePower
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,1,0,nb,crc=True); ctx.sync()
    for mode in ('crc','genie'):
        ms=[]
        for _ in range(3):
            if mode=='crc': code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb,crc=True)
            else: code.scl_decode_genie_dev(llr.data_ptr(),dt,L,info.data_ptr(),hat.data_ptr(),nb)
            ms.append(code.last_launch()['last_ms'])
        k=code.last_launch()['kernel']; ctx.sync()
        nchk=1<<13
        os.environ['POLAR_SCL_KERNEL']='lds'
        ref=torch.empty((nchk,code.Kw),dtype=torch.int32,device='cuda')
        if mode=='crc': code.scl_decode_dev(llr.data_ptr(),dt,L,ref.data_ptr(),nchk,crc=True)
        else: code.scl_decode_genie_dev(llr.data_ptr(),dt,L,info.data_ptr(),ref.data_ptr(),nchk)
        k2=code.last_launch()['kernel']; del os.environ['POLAR_SCL_KERNEL']; ctx.sync()
        bad=int((ref!=hat[:nchk]).any(dim=1).sum()); fe=int((hat!=info).any(dim=1).sum())
        print(dt.__name__,mode,'snr',snr,'%.2f M frames/s'%(nb/(min(ms)*1e-3)/1e6),k,'vs',k2,'differ',bad,'/',nchk,'fer %.4g'%(fe/nb),flush=True)
