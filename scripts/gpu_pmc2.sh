# usage: bash scripts/gpu_pmc2.sh  -> latency / branch counters of the SC kernel under gpurun_out/pmc2_*
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
A="--no-cpu --no-extras --frames 4000000 --steps 1 --warmup 0"
for P in "SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  D=$R/gpurun_out/pmc2_$(echo $P | cut -d" " -f1)
  rm -rf $D
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $D -- python3 $R/bench.py $A > $D.log 2>&1 || { echo FAIL $P; tail -5 $D.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$R/gpurun_out/pmc2_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if any(k in r['Kernel_Name'] for k in ('sc_reg_kernel','sc_static_kernel')):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, 'n=%d'%len(v), 'avg=%.4g'%(sum(v)/len(v)), 'per_pass(1M)=%.1f'%(sum(v)/len(v)/1e6))
PY
