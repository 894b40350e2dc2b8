# Two builds of the library side by side on one box (box-to-box spread is +-3 %): polar-codes_amd/libpolar_ab_base.so is a build of
# the tree to compare against (make, then copy libpolar_hip.so under that name), the in-tree library is the other; three
# alternating runs of the 10^7-frame step, then the lengths-and-rates table for both.
export POLAR_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2 3; do
for V in base new; do
  if [ $V = base ]; then export POLAR_HIP_LIB=$R/polar-codes_amd/libpolar_ab_base.so; else unset POLAR_HIP_LIB; fi
  timeout -k 10 120 python3 bench.py --no-cpu --no-extras --steps 5 > gpurun_out/ab.log 2>&1 || { tail -3 gpurun_out/ab.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1]); print('$V rep $rep:', round(d['frames_per_s']/1e6,1), 'M frames/s, step', round(d['ms_per_step'],3), 'ms, kernel', round(d['roofline']['kernel_ms'],3))"
done; done
for V in base new; do
  if [ $V = base ]; then export POLAR_HIP_LIB=$R/polar-codes_amd/libpolar_ab_base.so; else unset POLAR_HIP_LIB; fi
  echo $V; timeout -k 10 200 python3 -u scripts/gpu_sizes.py 2>&1 | grep -v amdgpu
done
