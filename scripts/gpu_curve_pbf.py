# FER of SC, progressive bit flipping with one, two and three flips and CRC-aided SCL-8 on the code cn/simulator.py sets up:
# (1024, 512 + CRC-24), dega dSNR 2; frames per point in argv[1] (default 2^16); output: gpurun_out/curve_pbf.txt
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
nb=int(sys.argv[1]) if len(sys.argv)>1 else 1<<16
ctx=_native.default_ctx(0)
N,K=1024,536
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.set_crc(0x1800063); code.set_reliability(pcf.GA(N,K,2.0))
out=open('gpurun_out/curve_pbf.txt','w')
def say(*a):
    line=' '.join(str(x) for x in a); print(line,flush=True); out.write(line+'\n'); out.flush()
say('# (1024,536) dega dSNR 2, CRC-24 0x1800063, 512 message bits, BPSK-AWGN, fp32,',nb,'frames per point; FER = frames whose message bits differ')
say('# Eb/N0  SC  PBF-1 (mean runs)  PBF-2 (mean runs)  PBF-3 (mean runs, seconds)  CA-SCL-8')
for snr in (1.5,2.0,2.5,3.0):
    n0=Channel('BPSK',snr,'SNRb',512/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    runs=torch.zeros(nb,dtype=torch.int32,device='cuda')
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,7,0,nb,crc=True); ctx.sync()
    def fer():
        c=torch.zeros(3,dtype=torch.int64,device='cuda'); code.count_errors_dev(info.data_ptr(),hat.data_ptr(),nb,c.data_ptr(),crc=True); ctx.sync(); return c.cpu().tolist()[1]/nb
    row=['%.1f'%snr]
    code.sc_decode_dev(llr.data_ptr(),np.float32,hat.data_ptr(),nb,fast=True); row.append('%.3e'%fer())
    for T in (1,2,3):
        t0=time.time(); code.pbf_decode_dev(llr.data_ptr(),np.float32,T,hat.data_ptr(),nb,runs_ptr=runs.data_ptr()); ctx.sync(); dt=time.time()-t0
        row.append('%.3e (%.1f%s)'%(fer(),runs.float().mean().item(),', %.2f s'%dt if T==3 else ''))
    code.scl_decode_dev(llr.data_ptr(),np.float32,8,hat.data_ptr(),nb,crc=True); row.append('%.3e'%fer())
    say('  '.join(row))
