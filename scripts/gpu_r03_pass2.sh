# second half of a measurement pass (the first is scripts/gpu_full.sh): secondary kernels, SC issue counters, two-rank rehearsal,
# list decoder rates.  tools/refresh_profiles.py collects what both halves leave in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash scripts/gpu_prof_legs.sh > gpurun_out/prof_legs_run.txt 2>&1 || { tail -20 gpurun_out/prof_legs_run.txt; exit 1; }
tail -3 gpurun_out/prof_legs_run.txt
POLAR_PMC_KERNELS=sc_static_kernel bash scripts/gpu_pmc3.sh > gpurun_out/sc_sq_counters.txt 2>&1 || { tail -20 gpurun_out/sc_sq_counters.txt; exit 1; }
tail -3 gpurun_out/sc_sq_counters.txt
cd $R
POLAR_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --frames 4000000 --no-cpu > gpurun_out/bench_n2_gloo.json 2> gpurun_out/bench_n2_gloo.err || { tail -20 gpurun_out/bench_n2_gloo.err; exit 1; }
bash scripts/gpu_scl_time.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/scl_time.txt; bash scripts/gpu_scl64_time.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/scl64_time.txt
timeout -k 10 300 python3 scripts/gpu_mc_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/mc_time.txt
cat gpurun_out/scl_time.txt gpurun_out/scl64_time.txt gpurun_out/mc_time.txt
# cycle / event profile of the register list decoder (a second library built with -DPOLAR_SCL_PROFILE: scripts/dev/build_scl_variant.sh sclprof -DPOLAR_SCL_PROFILE)
if [ -f $R/polar-codes_amd/libpolar_hip_sclprof.so ]; then
  PROF_LIB=$R/polar-codes_amd/libpolar_hip_sclprof.so VARIANTS=reg bash scripts/gpu_scl_prof.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/scl_cycle_profile.txt; tail -25 gpurun_out/scl_cycle_profile.txt
fi
