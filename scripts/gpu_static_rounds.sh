# specialised SC decoder: workgroups per resident slot (grid = CUs x workgroups per CU x rounds), 10^7 frames
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
for R in ${ROUNDS_LIST:-32 64 128 256 1024 32}; do
  POLAR_STATIC_ROUNDS=$R timeout -k 10 120 python3 bench.py --no-cpu --no-extras --steps 5 > gpurun_out/sr.log 2>&1 || { tail -3 gpurun_out/sr.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/sr.log').read().strip().splitlines()[-1]); print('rounds $R:', round(d['frames_per_s']/1e6,1), 'M frames/s, step', round(d['ms_per_step'],3), 'ms, kernel', round(d['roofline']['kernel_ms'],3), 'ms, grid', d['roofline']['grid_blocks'], 'frame errors', d['counters'][1])"
done
