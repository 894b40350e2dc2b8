# specialised SC decoder: 4-byte channel loads (WIDE 0) against 16-byte loads staged through LDS, frames in flight per group
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_sc.py -m gpu -x -q -k "specialised or fused" --timeout 200 --timeout-method thread > gpurun_out/t_static.log 2>&1 || { tail -30 gpurun_out/t_static.log; exit 1; }
tail -1 gpurun_out/t_static.log
for CFG in "0 2" "1 1" "1 2" "1 4"; do
  set -- $CFG
  POLAR_STATIC_WIDE=$1 POLAR_STATIC_WIDE_GROUP=$2 timeout -k 10 120 python3 bench.py --no-cpu --no-extras > gpurun_out/ss.log 2>&1 || { tail -3 gpurun_out/ss.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ss.log').read().strip().splitlines()[-1]); print('wide $1 group $2:', round(d['frames_per_s']/1e6,1), 'M frames/s, kernel', round(d['roofline']['kernel_ms'],3), 'ms, frac', round(d['roofline']['frac'],3), 'grid', d['roofline']['grid_blocks'], 'lds', d['roofline']['lds_bytes_per_block'], 'frame errors', d['counters'][1], d['specialize_error'])"
done
