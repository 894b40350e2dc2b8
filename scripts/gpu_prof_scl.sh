# rocprof stats + SQ counters for the SCL L=8 leg
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
cat > /tmp/scl_only.py <<PY
import sys; sys.path[:0]=['$R/polar-codes_amd','$R']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0); mask=pcf.degaBuild(1024,512,2.0); code=_native.Code(ctx,mask)
n0=Channel('BPSK',2.0,'SNRb',0.5).noisePower; nb=1<<20
llr=torch.empty((nb,1024),dtype=torch.float32,device='cuda'); hat=torch.empty((nb,16),dtype=torch.int32,device='cuda')
code.gen_frames_dev(None,llr.data_ptr(),np.float32,n0,1,0,nb)
for _ in range(3): code.scl_decode_dev(llr.data_ptr(),np.float32,8,hat.data_ptr(),nb)
ctx.sync()
PY
rm -rf $R/gpurun_out/prof_scl; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scl -- python3 /tmp/scl_only.py > $R/gpurun_out/prof_scl.log 2>&1 || { tail -5 $R/gpurun_out/prof_scl.log; exit 1; }
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  D=$R/gpurun_out/pmcscl_$(echo $P | cut -d" " -f1); rm -rf $D
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $D -- python3 /tmp/scl_only.py > $D.log 2>&1 || { echo FAIL $P; exit 1; }
done
cat $R/gpurun_out/prof_scl/*/*kernel_stats.csv | head -4
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$R/gpurun_out/pmcscl_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if 'scl_reg_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k,'avg=%.4g'%(sum(v)/len(v)),'per_frame=%.1f'%(sum(v)/len(v)/(1<<20)))
PY
