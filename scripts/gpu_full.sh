# Full GPU check: whole gpu test suite, smoke, bench (JSON), rocprof stats of the bench, PMC traffic passes.
export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/tests_gpu.log 2>&1; echo EXIT $? >> gpurun_out/tests_gpu.log; tail -3 gpurun_out/tests_gpu.log
grep -q "EXIT 0" gpurun_out/tests_gpu.log || exit 1
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
tail -c 600 gpurun_out/bench.json; echo
cd /tmp
rm -rf $R/gpurun_out/prof_stats; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu --no-extras > $R/gpurun_out/prof_stats.log 2>&1 || { tail -5 $R/gpurun_out/prof_stats.log; exit 1; }
for P in "FETCH_SIZE" "WRITE_SIZE"; do D=$R/gpurun_out/pmcfull_$P; rm -rf $D; timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $D -- python3 $R/bench.py --no-cpu --no-extras --steps 1 --warmup 0 > $D.log 2>&1 || { echo FAIL $P; exit 1; }; done
echo done
