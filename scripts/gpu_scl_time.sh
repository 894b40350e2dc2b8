# list decoder rates (kernel time from the events); POLAR_HIP_LIB selects an experimental build
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python3 - <<'PY' || exit 1
import sys, os; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
for (N,K,L,snr,nb) in [(1024,512,8,2.0,1<<19),(1024,512,8,1.0,1<<18),(1024,512,8,3.0,1<<18),(512,256,8,2.0,1<<18),(512,256,32,2.0,1<<16),(1024,512,32,2.0,1<<15)]:
    code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',snr,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb)
    ms=[]
    for _ in range(3):
        code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb); ms.append(code.last_launch()['last_ms'])
    li=code.last_launch(); cnt=torch.zeros(3,dtype=torch.int64,device='cuda'); code.count_errors_dev(info.data_ptr(),hat.data_ptr(),nb,cnt.data_ptr()); ctx.sync()
    print(os.environ.get('POLAR_HIP_LIB','')[-10:] or 'build',N,K,L,snr,'Mframes/s %.2f'%(nb/(min(ms)*1e-3)/1e6),'ms %.2f'%min(ms),'fe',cnt[1].item(),'be',cnt[2].item(), li['kernel'], flush=True)
PY
