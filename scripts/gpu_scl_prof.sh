# where the register list decoder spends its time: cycle counters per category (library built with `make prof`)
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for V in ${VARIANTS:-reg1 reg}; do
POLAR_HIP_LIB=${PROF_LIB:-$R/polar-codes_amd/libpolar_hip_prof.so} POLAR_SCL_KERNEL=$V timeout -k 10 200 python3 - <<'PY' || exit 1
import sys, os, ctypes; sys.path[:0]=['polar-codes_amd','.']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
L_=_native.lib()
N,K,L,snr,nb=int(os.environ.get('PN','1024')),int(os.environ.get('PK','512')),8,2.0,1<<18
dt=np.float64 if os.environ.get('PDT','f32')=='f64' else np.float32; tdt=torch.float64 if dt is np.float64 else torch.float32
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',snr,'SNRb',K/N).noisePower
llr=torch.empty((nb,N),dtype=tdt,device='cuda'); hat=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda')
code.gen_frames_dev(None,llr.data_ptr(),dt,n0,1,0,nb)
code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ctx.sync()
buf=(ctypes.c_ulonglong*40)(); L_.polar_debug_scl_profile(buf,1)
code.scl_decode_dev(llr.data_ptr(),dt,L,hat.data_ptr(),nb); ctx.sync()
li=code.last_launch(); L_.polar_debug_scl_profile(buf,1)
v=np.array(list(buf),dtype=np.float64)/nb
names={i:'f/g level %d'%i for i in range(4,11)}
print('all-information nodes of a full list tested / decided as a whole, per frame: '+', '.join('%d leaves %.1f / %.1f'%(8<<i,v[24+2*i],v[25+2*i]) for i in range(4))+'; 8-leaf nodes not decided: LLR small %.1f, metric range %.1f, other extension %.1f, order %.1f'%(v[32],v[33],v[34],v[35])); v[24:]=0
print('information leaves of a full list that reach the ranking, per frame, %.1f, of which nothing moved %.1f; decided before any metric %.1f (not: some LLR small %.1f, another extension could survive %.1f, order %.1f)'%(v[22],v[23],v[0],v[1],v[2],v[3])); v[22]=v[23]=v[0]=v[1]=v[2]=v[3]=0; names[11]='combine'; names[21]='output'
for i in range(12,21): names[i]='node8 with %d info'%(i-12)
tot=v.sum()
print(os.environ.get('POLAR_SCL_KERNEL'),li['kernel'],'Mframes/s %.2f'%(nb/(li['last_ms']*1e-3)/1e6),'cycles/frame %.0f'%tot)
m8=np.packbits(np.asarray(code and pcf.degaBuild(N,K,2.0),dtype=np.uint8).reshape(-1,8),axis=1,bitorder='little').ravel()
cnt=np.bincount([bin(x).count('1') for x in m8],minlength=9)
for i in sorted(names):
    if v[i]>0:
        extra=''
        if 12<=i<=20 and cnt[i-12]: extra=' (%d nodes, %.0f cycles each)'%(cnt[i-12],v[i]/cnt[i-12])
        print('  %-22s %9.0f  %5.1f%%%s'%(names[i],v[i],100*v[i]/tot,extra))
PY
done
