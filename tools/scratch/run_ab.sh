set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python tools/microbench.py --sizes 65536 --epws 0
timeout -k 10 200 python tools/microbench_env_sources.py 2>&1 | tail -2 | cut -c1-100
timeout -k 10 200 python tools/microbench_observers.py 2>&1 | grep -v amdgpu.ids
