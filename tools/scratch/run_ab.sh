timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python tools/scratch/pes_parts.py 2>&1 | grep -v amdgpu
timeout -k 10 200 python tools/microbench_multi_map.py 2>&1 | grep -v amdgpu | tail -2
