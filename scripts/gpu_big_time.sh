# long-code SC kernel at (16384,8192): rate only (POLAR_HIP_LIB selects an experimental build)
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 120 python3 - <<'PY' || exit 1
import sys, os; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
for (N,K,nb) in [(16384,8192,1<<15),(4096,2048,1<<17),(65536,30000,1<<14),(2048,1000,1<<18),(8192,4096,1<<16),(32768,16384,1<<14)]:
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); hat=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda')
    code.gen_frames_dev(None,llr.data_ptr(),np.float32,n0,1,0,nb)
    ms=[]
    for _ in range(3):
        code.sc_decode_dev(llr.data_ptr(),np.float32,hat.data_ptr(),nb,fast=True); ms.append(code.last_launch()['last_ms'])
    li=code.last_launch(); print(os.environ.get('POLAR_HIP_LIB','')[-12:],N,K,li['kernel'],'grid',li['grid_blocks'],'Mframes/s %.2f'%(nb/(min(ms)*1e-3)/1e6),'ms %.2f'%min(ms),flush=True)
PY
