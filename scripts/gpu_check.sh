export POLAR_EXPERIMENT=1  # the library reads its kernel switches only with this set
timeout -k 10 300 python -u -m pytest tests/test_gpu_sc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t3.log 2>&1; echo EXIT $? >> gpurun_out/t3.log; tail -3 gpurun_out/t3.log; timeout -k 10 300 python3 bench.py --no-cpu --no-extras > gpurun_out/bench_v2.log 2>&1; python3 -c "
import json; d=json.loads(open('gpurun_out/bench_v2.log').read().strip().splitlines()[-1]); print(d['frames_per_s'], d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['counters'])"
