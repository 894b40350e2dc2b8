# SC kernel throughput against occupancy (unused LDS per wavefront limits the waves per CU)
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
for PAD in 0 13000 19000 26000 39000 79000; do
  POLAR_SC_LDSPAD=$PAD timeout -k 10 120 python3 bench.py --no-cpu --no-extras --frames 4000000 > gpurun_out/occ_$PAD.log 2>&1 || exit 1
  python3 -c "
import json; d=json.loads(open('gpurun_out/occ_$PAD.log').read().strip().splitlines()[-1]); print($PAD, 160*1024//max($PAD,1), d['frames_per_s']/1e6, d['roofline']['kernel_ms'], d['roofline']['grid_blocks'])"
done
