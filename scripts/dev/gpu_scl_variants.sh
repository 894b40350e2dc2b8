# A/B of register list decoder builds (scripts/dev/build_scl_variant.sh): equality with the LDS-staged kernel, then rate.
# usage: gpu_scl_variants.sh NAME [NAME ...]
export POLAR_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for V in "$@"; do
POLAR_HIP_LIB=$R/polar-codes_amd/libpolar_hip_$V.so VARIANT=$V timeout -k 10 240 python3 - <<'PY' || exit 1
import sys, os; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
N,K,L=1024,512,8
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0))
out=[os.environ['VARIANT']]
for snr,nb in [(2.0,1<<19),(1.0,1<<17),(3.0,1<<17)]:
    n0=Channel('BPSK',snr,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,nb)
    ms=[]
    for _ in range(3):
        code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb); ms.append(code.last_launch()['last_ms'])
    kern=code.last_launch()['kernel']
    nchk=1<<14
    os.environ['POLAR_SCL_KERNEL']='lds'
    ref=torch.empty((nchk,code.Kw),dtype=torch.int32,device='cuda')
    code.scl_decode_dev(llr.data_ptr(),np.float32,L,ref.data_ptr(),nchk); k2=code.last_launch()['kernel']
    del os.environ['POLAR_SCL_KERNEL']
    ctx.sync()
    bad=int((ref!=hat[:nchk]).any(dim=1).sum().item())
    out.append('snr %.1f %.2f Mf/s (%.2f ms) differ %d/%d'%(snr,nb/(min(ms)*1e-3)/1e6,min(ms),bad,nchk))
print(' | '.join(out), kern, k2, 'grid', code.last_launch()['grid_blocks'], flush=True)
PY
done
