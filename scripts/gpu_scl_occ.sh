# register SCL kernel (L = 8) at 8 (default), 4 and 2 wavefronts per CU: how much of its time is latency
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for PAD in 0 36000 72000; do
POLAR_SCL_LDSPAD=$PAD timeout -k 10 200 python3 - <<'PY'
import sys, os; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
N,K,L,nb=1024,512,8,1<<18
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb)
for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb)
li=code.last_launch(); print('pad',os.environ.get('POLAR_SCL_LDSPAD'),'grid',li['grid_blocks'],'frames/s %.0f'%(nb/(li['last_ms']*1e-3)))
PY
done
