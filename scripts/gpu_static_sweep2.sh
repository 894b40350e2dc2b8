# specialised SC decoder: channel pairs kept in registers / in LDS, each setting twice and interleaved so that drift shows
export POLAR_EXPERIMENT=1
for CFG in "16 16 4" "20 12 4" "24 8 4" "18 14 4" "16 16 4" "20 12 4" "24 8 4" "18 14 4"; do
  set -- $CFG
  POLAR_STATIC_KEEP=$1 POLAR_STATIC_KEEP_LDS=$2 POLAR_STATIC_WAVES=$3 timeout -k 10 120 python3 bench.py --no-cpu --no-extras --steps 5 > gpurun_out/ss.log 2>&1 || { tail -3 gpurun_out/ss.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ss.log').read().strip().splitlines()[-1]); print('keep regs $1 lds $2 waves $3:', round(d['frames_per_s']/1e6,1), 'M frames/s, step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'ms, grid', d['roofline']['grid_blocks'], 'lds', d['roofline']['lds_bytes_per_block'], d['specialize_error'])"
done
