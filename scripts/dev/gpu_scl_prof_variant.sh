# cycle / event profile of one register list decoder build made with -DPOLAR_SCL_PROFILE (scripts/dev/build_scl_variant.sh)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
PROF_LIB=$R/polar-codes_amd/libpolar_hip_$1.so VARIANTS=reg bash scripts/gpu_scl_prof.sh
