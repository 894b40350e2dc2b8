# rocprof kernel stats, SQ counters and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the secondary kernels:
# list decoders (L = 8 and L = 32), long-code SC kernel, flip search.  Summaries: gpurun_out/legs_*.{csv,txt,json}
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
cat > /tmp/legs.py <<PY
import sys; sys.path[:0]=['$R/polar-codes_amd','$R']
import numpy as np, torch, _native, functions as pcf
from channel import Channel
ctx=_native.default_ctx(0)
def frames(mask, nb, snr=2.0, crc=False, Kmsg=None, tdt=torch.float32):
    N=len(mask); K=int(mask.sum()); code=_native.Code(ctx,mask)
    n0=Channel('BPSK',snr,'SNRb',(Kmsg or K)/N).noisePower
    llr=torch.empty((nb,N),dtype=tdt,device='cuda'); info=torch.zeros((nb,code.Kw),dtype=torch.int32,device='cuda'); hat=torch.empty_like(info)
    return code, n0, llr, info, hat
code,n0,llr,info,hat=frames(pcf.degaBuild(1024,512,2.0),1<<19); code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,1<<19)
for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float32,8,hat.data_ptr(),1<<19)
code,n0,llr,info,hat=frames(pcf.rmPolarBuild(512,256,2.0),1<<17); code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,1<<17)
for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float32,32,hat.data_ptr(),1<<17)
code,n0,llr,info,hat=frames(pcf.degaBuild(16384,8192,2.0),1<<15); code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,1<<15)
for _ in range(2): code.sc_decode_dev(llr.data_ptr(),np.float32,hat.data_ptr(),1<<15,fast=True)
code,n0,llr,info,hat=frames(pcf.degaBuild(1024,536,2.0),1<<18,Kmsg=512); code.set_crc(0x1800063); code.set_reliability(pcf.GA(1024,536,2.0))
code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float32,n0,1,0,1<<18,crc=True)
for _ in range(2): code.pbf_decode_dev(llr.data_ptr(),np.float32,2,hat.data_ptr(),1<<18)
# the reference's arithmetic (fp64), the calls bench.py times: every-leaf SC (information bits out, as in the bench leg),
# list decoder with the sorted list's metrics, the two-lanes-per-path list decoder with L = 32
code,n0,llr,info,hat=frames(pcf.degaBuild(1024,512,2.0),1<<17,tdt=torch.float64); code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float64,n0,1,0,1<<17)
pm=torch.empty((1<<17,8),dtype=torch.float64,device='cuda')
for _ in range(2): code.sc_decode_dev(llr.data_ptr(),np.float64,hat.data_ptr(),1<<17,fast=False)
for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float64,8,hat.data_ptr(),1<<17,pm_ptr=pm.data_ptr())
code,n0,llr,info,hat=frames(pcf.rmPolarBuild(512,256,2.0),1<<16,tdt=torch.float64); code.gen_frames_dev(info.data_ptr(),llr.data_ptr(),np.float64,n0,1,0,1<<16)
for _ in range(2): code.scl_decode_dev(llr.data_ptr(),np.float64,32,hat.data_ptr(),1<<16)
# the one-kernel Monte-Carlo loop of a specialised code
code=_native.Code(ctx,pcf.degaBuild(1024,512,2.0)); code.specialize()
for _ in range(2): code.mc_run(_native.DEC_SC,1,np.float32,Channel('BPSK',2.0,'SNRb',0.5).noisePower,2024,0,1<<24)
ctx.sync()
PY
rm -rf $R/gpurun_out/prof_legs; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_legs -- python3 /tmp/legs.py > $R/gpurun_out/prof_legs.log 2>&1 || { tail -5 $R/gpurun_out/prof_legs.log; exit 1; }
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$R/gpurun_out/pmclegs_$(echo $P | cut -d" " -f1); rm -rf $D
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $D -- python3 /tmp/legs.py > $D.log 2>&1 || { echo FAIL $P; tail -5 $D.log; exit 1; }
done
cp $R/gpurun_out/prof_legs/*/*kernel_stats.csv $R/gpurun_out/legs_kernel_stats.csv
python3 - <<PY
import csv,glob,collections,json
legs={'scl_reg_kernel<10, 8, float>':(1<<19,4*1024+64,'(1024,512) L=8'),'scl_reg_kernel<10, 8, double>':(1<<17,8*1024+64,'(1024,512) L=8 fp64, sorted metrics out'),
      'sc_reg_full_kernel<double, 10>':(1<<17,8*1024+64,'(1024,512) SC fp64, every leaf, information bits out (the call bench.py times)'),
      'scl_wide_kernel<9, 32, float>':(1<<17,4*512+32,'rmpolar (512,256) L=32'),'scl_wide_kernel<9, 32, double>':(1<<16,8*512+32,'rmpolar (512,256) L=32 fp64'),
      'sc_static_mc_kernel':(1<<24,None,'(1024,512) one-kernel Monte-Carlo loop, 2^24 frames'),'sc_big_kernel':(1<<15,4*16384+1024,'(16384,8192) SC'),
      'pbf_reg_kernel<float, 10, 0>':(1<<18,None,'(1024,536) CRC-24 flip search, single flips, 2^18 frames in the batch'),
      'pbf_reg_kernel<float, 10, 1>':(1<<18,None,'same, pair search')}
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob('$R/gpurun_out/pmclegs_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        for name in legs:
            if name in r['Kernel_Name'].replace('(anonymous namespace)::',''): agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
with open('$R/gpurun_out/legs_sq_counters.txt','w') as f:
    f.write('scripts/gpu_prof_legs.sh: counters per launch (average over launches) and per frame of the batch\\n')
    for name,c in agg.items():
        nb,alg,what=legs[name]
        if not nb: continue
        f.write('\\n%s  %s  (%d frames per launch)\\n'%(name,what,nb))
        m={k:sum(v)/len(v) for k,v in c.items()}
        for k in sorted(m): f.write('  %-22s %.4g  per_frame=%.1f\\n'%(k,m[k],m[k]/nb))
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            per=(2*m['FETCH_SIZE']+m['WRITE_SIZE'])*1024/nb   # gfx950 tallies 128-byte read requests at 64 bytes
            f.write('  HBM bytes per frame    %.0f%s\\n'%(per,' = %.2f x algorithmic %.0f'%(per/alg,alg) if alg else ''))
            out[name]={'what':what,'bytes_per_frame':per,'algorithmic_bytes_per_frame':alg,'FETCH_SIZE_KiB':m['FETCH_SIZE'],'WRITE_SIZE_KiB':m['WRITE_SIZE'],'frames_per_launch':nb}
json.dump(out,open('$R/gpurun_out/traffic_legs.json','w'),indent=1)
print(open('$R/gpurun_out/legs_sq_counters.txt').read())
PY
