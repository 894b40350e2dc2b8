# long-code SC kernel at (16384,8192): rate, then HBM traffic per frame (FETCH_SIZE / WRITE_SIZE in separate passes)
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
cat > /tmp/big1.py <<PY
import sys; sys.path[:0]=['$R/polar-codes_amd','$R']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); ctx.enable_timing(True)
N,K,nb=16384,8192,1<<15
code=_native.Code(ctx,pcf.degaBuild(N,K,2.0)); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); hat=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda')
code.gen_frames_dev(None,llr.data_ptr(),np.float32,n0,1,0,nb)
ms=[]
for _ in range(3):
    code.sc_decode_dev(llr.data_ptr(),np.float32,hat.data_ptr(),nb,fast=True); ms.append(code.last_launch()['last_ms'])
li=code.last_launch(); print(li['kernel'],'grid',li['grid_blocks'],'Mframes/s %.2f'%(nb/(min(ms)*1e-3)/1e6),'ms %.2f'%min(ms),'checksum',int(hat.sum().item()),flush=True)
PY
timeout -k 10 120 python3 /tmp/big1.py || exit 1
for P in FETCH_SIZE WRITE_SIZE; do
  D=$R/gpurun_out/pmcbig_$P; rm -rf $D
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $D -- python3 /tmp/big1.py > $D.log 2>&1 || { echo FAIL $P; tail -5 $D.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
m=collections.defaultdict(list)
for d in sorted(glob.glob('$R/gpurun_out/pmcbig_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        if 'sc_big_kernel' in r['Kernel_Name']: m[r['Counter_Name']].append(float(r['Counter_Value']))
f=sum(m['FETCH_SIZE'])/len(m['FETCH_SIZE']); w=sum(m['WRITE_SIZE'])/len(m['WRITE_SIZE'])
print('per frame: read %.0f B (2 x FETCH_SIZE), written %.0f B, total %.0f B'%(2*f*1024/32768, w*1024/32768, (2*f+w)*1024/32768))
PY
