# usage: bash scripts/gpu_pmc3.sh  -> issue / wait counters of the SC kernel (4M frames per launch) under gpurun_out/pmc3_*
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
A="--no-cpu --no-extras --frames 4000000 --steps 1 --warmup 0"
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  D=$R/gpurun_out/pmc3_$(echo $P | cut -d" " -f1)
  rm -rf $D
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $D -- python3 $R/bench.py $A > $D.log 2>&1 || { echo FAIL $P; tail -5 $D.log; }
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$R/gpurun_out/pmc3_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if any(k in r['Kernel_Name'] for k in '${POLAR_PMC_KERNELS:-sc_reg_kernel,sc_static_kernel}'.split(',')):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, 'n=%d'%len(v), 'avg=%.4g'%(sum(v)/len(v)), 'per wave pass (1M)=%.1f'%(sum(v)/len(v)/1e6))
PY
