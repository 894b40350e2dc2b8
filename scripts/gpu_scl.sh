export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_scl.py tests/test_gpu_curves.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_scl.log 2>&1; echo EXIT $? >> gpurun_out/t_scl.log; tail -3 gpurun_out/t_scl.log
timeout -k 10 300 python3 - <<'PY'
import sys,time; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
for (N,K,L,cons,nb) in [(1024,512,8,'dega',1<<18),(512,256,32,'rmpolar',1<<15),(1024,512,4,'dega',1<<18),(1024,512,2,'dega',1<<18)]:
    mask = pcf.degaBuild(N,K,2.0) if cons=='dega' else pcf.rmPolarBuild(N,K,2.0)
    code=_native.Code(ctx,mask); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb)
    for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb)
    li=code.last_launch(); cnt=torch.zeros(3,dtype=torch.int64,device='cuda'); code.count_errors_dev(info.data_ptr(),hat.data_ptr(),nb,cnt.data_ptr()); ctx.sync()
    print(N,K,L,cons,'frames/s %.0f'%(nb/(li['last_ms']*1e-3)),'ms %.2f'%li['last_ms'],'fer %.5f'%(cnt[1].item()/nb), li['kernel'], li['lds_bytes_per_block'])
PY
