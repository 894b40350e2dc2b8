# fp64 list decoder rates (kernel time from the events); POLAR_HIP_LIB selects an experimental build,
# POLAR_SCL_KERNEL=lds the LDS-staged kernel
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python3 - <<'PY' || exit 1
import sys, os; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
for (N,K,L,snr,nb,dt) in [(1024,512,8,2.0,1<<18,np.float64),(1024,512,8,1.0,1<<17,np.float64),(512,256,8,2.0,1<<18,np.float64),(512,256,32,2.0,1<<16,np.float64),(512,256,16,2.0,1<<16,np.float64),(1024,512,32,2.0,1<<15,np.float64),(256,128,4,2.0,1<<14,np.float64)]:
    code=_native.Code(ctx,pcf.rmPolarBuild(N,K,2.0) if (N,L)==(512,32) else pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',snr,'SNRb',K/N).noisePower
    tdt=torch.float64 if dt is np.float64 else torch.float32
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,1,0,nb)
    ms=[]
    for _ in range(3):
        code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ms.append(code.last_launch()['last_ms'])
    li=code.last_launch(); cnt=torch.zeros(3,dtype=torch.int64,device='cuda'); code.count_errors_dev(info.data_ptr(),hat.data_ptr(),nb,cnt.data_ptr()); ctx.sync()
    print(os.environ.get('POLAR_HIP_LIB','')[-12:] or 'build',os.environ.get('POLAR_SCL_KERNEL',''),N,K,L,dt.__name__,'Mframes/s %.3f'%(nb/(min(ms)*1e-3)/1e6),'ms %.2f'%min(ms),'fe',cnt[1].item(),'be',cnt[2].item(), li['kernel'],'grid',li['grid_blocks'],'lds',li['lds_bytes_per_block'], flush=True)
PY
