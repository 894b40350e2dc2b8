# two-flip search worked on in (frame, first flip) pieces: parity tests, then the rate against one wavefront per frame
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 500 python -u -m pytest tests/test_flip.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_flip.log 2>&1 || { tail -30 gpurun_out/t_flip.log; exit 1; }
tail -2 gpurun_out/t_flip.log
for V in 0 1; do
POLAR_PBF_SPLIT=$V timeout -k 10 200 python3 - <<'PY' || exit 1
import sys, os, time; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
N,K,poly=1024,536,0x1800063
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); code.set_crc(poly); code.set_reliability(pcf.GA(N,K,2.0)); code.specialize()
n0=Channel('BPSK',2.0,'SNRb',512/N).noisePower
for nb in (1<<15, 1<<18):
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    rn=torch.zeros(nb,dtype=torch.int32,device='cuda')
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb,crc=True)
    for T in (1,2):
        best=1e9
        for _ in range(2):
            ctx.sync(); t=time.perf_counter()
            code.pbf_decode_dev(llr.data_ptr(),np.float32,T,hat.data_ptr(),nb,runs_ptr=rn.data_ptr()); ctx.sync()
            best=min(best,time.perf_counter()-t)
        cnt=torch.zeros(3,dtype=torch.int64,device='cuda'); code.count_errors_dev(info.data_ptr(),hat.data_ptr(),nb,cnt.data_ptr(),crc=True); ctx.sync()
        print('split',os.environ['POLAR_PBF_SPLIT'],'frames',nb,'T',T,'kframes/s %.1f'%(nb/best/1e3),'ms %.1f'%(best*1e3),'fe',cnt[1].item(),'runs mean %.1f max %d total %d'%(rn.float().mean().item(),rn.max().item(),rn.sum().item()),code.last_launch()['kernel'],flush=True)
PY
done
