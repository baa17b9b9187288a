for r in 1 2 3; do
for v in base swt; do
  if [ $v = base ]; then unset LLE_HIP_LIB; else export LLE_HIP_LIB=$PWD/lle_amd/liblle_hip_$v.so; fi
  echo "state stores $v: $(timeout -k 10 200 python tools/microbench.py --sizes 65536,524288 --epws 0 2>&1 | grep -v amdgpu | tr '\n' '|')"
done; done
