# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and L2 hit / miss of the secondary kernels: the long-code SC kernel
# at (16384,8192) and the list decoders.  Prints bytes per frame against the algorithmic figure; json in gpurun_out/traffic_legs.json
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
cat > /tmp/legs.py <<PY
import sys; sys.path[:0]=['$R/polar-codes_amd','$R']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
def run(mask, L, nb, reps=2):
    N=len(mask); K=int(mask.sum()); code=_native.Code(ctx,mask); n0=Channel('BPSK',2.0,'SNRb',K/N).noisePower
    llr=torch.empty((nb,N),dtype=torch.float32,device='cuda'); hat=torch.empty((nb,code.Kw),dtype=torch.int32,device='cuda')
    code.gen_frames_dev(None,llr.data_ptr(),np.float32,n0,1,0,nb)
    for _ in range(reps):
        if L: code.scl_decode_dev(llr.data_ptr(),np.float32,L,hat.data_ptr(),nb)
        else: code.sc_decode_dev(llr.data_ptr(),np.float32,hat.data_ptr(),nb,fast=True)
    ctx.sync()
run(pcf.degaBuild(16384,8192,2.0), 0, 1<<15)
run(pcf.degaBuild(1024,512,2.0), 8, 1<<19)
run(pcf.rmPolarBuild(512,256,2.0), 32, 1<<17)
PY
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  D=$R/gpurun_out/pmclegs_$(echo $P | cut -d" " -f1); rm -rf $D
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $D -- python3 /tmp/legs.py > $D.log 2>&1 || { echo FAIL $P; tail -5 $D.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob('$R/gpurun_out/pmclegs_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        k=r['Kernel_Name']
        for name in ('sc_big_kernel','scl_reg_kernel','scl_wide_kernel'):
            if name in k: agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
frames={'sc_big_kernel':1<<15,'scl_reg_kernel':1<<19,'scl_wide_kernel':1<<17}
alg={'sc_big_kernel':4*16384+1024,'scl_reg_kernel':4*1024+64,'scl_wide_kernel':4*512+32}
out={}
for name,c in agg.items():
    m={k:sum(v)/len(v) for k,v in c.items()}
    per=(2*m['FETCH_SIZE']+m['WRITE_SIZE'])*1024/frames[name]   # gfx950 tallies 128-byte read requests at 64 bytes
    out[name]={'bytes_per_frame':per,'algorithmic_bytes_per_frame':alg[name],'ratio':per/alg[name],'FETCH_SIZE_KiB':m['FETCH_SIZE'],'WRITE_SIZE_KiB':m['WRITE_SIZE'],
               'TCC_HIT':m.get('TCC_HIT_sum'),'TCC_MISS':m.get('TCC_MISS_sum'),'TCC_REQ':m.get('TCC_REQ_sum'),'frames_per_launch':frames[name]}
    print(name,'HBM bytes/frame %.0f = %.2f x algorithmic %d; L2 hit %.3g miss %.3g'%(per,per/alg[name],alg[name],m.get('TCC_HIT_sum',0),m.get('TCC_MISS_sum',0)))
json.dump(out,open('$R/gpurun_out/traffic_legs.json','w'),indent=1)
PY
