# specialised SC decoder: channel pairs kept in registers x pairs kept in LDS x target occupancy (workgroups of 4 waves
# per CU = waves per SIMD)
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
for CFG in "16 16 4" "12 16 4" "12 20 4" "8 16 4" "8 24 4" "20 12 4" "4 16 5" "8 12 5" "16 16 4"; do
  set -- $CFG
  POLAR_STATIC_KEEP=$1 POLAR_STATIC_KEEP_LDS=$2 POLAR_STATIC_WAVES=$3 timeout -k 10 120 python3 bench.py --no-cpu --no-extras > gpurun_out/ss.log 2>&1 || { tail -3 gpurun_out/ss.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ss.log').read().strip().splitlines()[-1]); print('keep regs $1 lds $2 waves $3:', round(d['frames_per_s']/1e6,1), 'M frames/s, kernel', round(d['roofline']['kernel_ms'],2), 'ms, grid', d['roofline']['grid_blocks'], 'lds', d['roofline']['lds_bytes_per_block'], 'frame errors', d['counters'][1], d['specialize_error'])"
done
