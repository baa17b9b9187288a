for v in 4 2 1; do
  export LLE_STEP_WPW=$v
  echo "== waves per workgroup $v"
  timeout -k 10 200 python tools/microbench_multi_map.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-120
  timeout -k 10 200 python tools/microbench.py --sizes 65536 --epws 0 2>&1 | grep -v amdgpu
done
