# list decoder on the GPU: its parity tests, the error paths, the edge cases, then its rates
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_error_paths.py tests/test_gpu_scl.py tests/test_cascl.py tests/test_genie.py tests/test_gpu_edge_cases.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_scl_all.log 2>&1; rc=$?; tail -15 gpurun_out/t_scl_all.log; [ $rc -eq 0 ] || exit $rc
bash scripts/gpu_scl_time.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/scl_time.txt
