# usage: bash scripts/gpu_pmc.sh  -> per-dispatch counters of the SC kernel under gpurun_out/pmc_*
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
A="--no-cpu --no-extras --frames 4000000 --steps 1 --warmup 0"
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_IFETCH GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  D=$R/gpurun_out/pmc_$(echo $P | cut -d" " -f1)
  rm -rf $D
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $D -- python3 $R/bench.py $A > $D.log 2>&1 || { echo FAIL $P; tail -5 $D.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$R/gpurun_out/pmc_*/vm/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if any(k in r['Kernel_Name'] for k in ('sc_reg_kernel','sc_decode_kernel','sc_static_kernel')):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, 'n=%d'%len(v), 'avg=%.4g'%(sum(v)/len(v)), 'per_pass(1M)=%.1f'%(sum(v)/len(v)/1e6))
PY
