# A/B of wide list decoder builds (scripts/dev/build_variant.sh kernels_scl_wide NAME ...): equality with the LDS-staged kernel, then rate.
# usage: gpu_wide_variants.sh NAME [NAME ...]
export POLAR_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for V in "$@"; do
POLAR_HIP_LIB=$R/polar-codes_amd/libpolar_hip_$V.so VARIANT=$V timeout -k 10 240 python3 - <<'PY' || exit 1
import sys, os, ctypes; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
L_=_native.lib()
N,K,L=512,256,32
code=_native.Code(ctx,pcf.rmPolarBuild(N,K,2.0))
for dt,tdt in [(np.float32,torch.float32),(np.float64,torch.float64)]:
  out=[os.environ['VARIANT'],dt.__name__]
  for snr,nb in [(2.0,1<<17),(1.0,1<<15),(3.0,1<<15)]:
    n0=Channel('BPSK',snr,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),dt,n0,1,0,nb); ctx.sync()
    ms=[]
    for _ in range(3):
        code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ms.append(code.last_launch()['last_ms'])
    kern=code.last_launch()['kernel']
    st=''
    if hasattr(L_,'polar_debug_wide_stats'):
        buf=(ctypes.c_ulonglong*8)(); L_.polar_debug_wide_stats(buf,1)
        code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ctx.sync(); L_.polar_debug_wide_stats(buf,1)
        st=' ranked/frame %.1f in-order %.1f with contenders %.1f'%(buf[0]/nb,buf[1]/nb,buf[2]/nb)
    nchk=1<<12
    os.environ['POLAR_SCL_KERNEL']='lds'
    ref=torch.empty((nchk,code.Kw),dtype=torch.int32,device='cuda')
    code.scl_decode_dev(llr.data_ptr(),dt,L,ref.data_ptr(),nchk); k2=code.last_launch()['kernel']
    del os.environ['POLAR_SCL_KERNEL']
    ctx.sync()
    bad=int((ref!=hat[:nchk]).any(dim=1).sum().item())
    out.append('snr %.1f %.2f Mf/s (%.2f ms) differ %d/%d%s'%(snr,nb/(min(ms)*1e-3)/1e6,min(ms),bad,nchk,st))
  print(' | '.join(out), kern, k2, flush=True)
PY
done
