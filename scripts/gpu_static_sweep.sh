# specialised SC decoder: kept channel pairs x target occupancy
for CFG in "0 4" "0 5" "0 6" "8 4" "8 5" "16 4" "16 5" "32 3" "32 4" "24 4"; do
  set -- $CFG
  POLAR_STATIC_KEEP=$1 POLAR_STATIC_WAVES=$2 timeout -k 10 120 python3 bench.py --no-cpu --no-extras > gpurun_out/ss.log 2>&1 || { tail -3 gpurun_out/ss.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ss.log').read().strip().splitlines()[-1]); print('keep $1 waves $2', round(d['frames_per_s']/1e6,1), round(d['roofline']['kernel_ms'],2), d['roofline']['grid_blocks'], d['counters'][1])"
done
