for v in 16 8 4 2 1; do
  export LLE_PARTIAL_EPW=$v
  echo "== partial epw $v"
  timeout -k 10 200 python tools/microbench_observers.py 2>&1 | grep partial
done
unset LLE_PARTIAL_EPW
timeout -k 10 600 python -m pytest tests/test_gpu_observers.py tests/test_gpu_env.py tests/test_gpu_multi_map.py -x -q -m gpu 2>&1 | tail -3
