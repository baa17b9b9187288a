for r in 1 2; do
for v in base sc1nt sc0sc1nt; do
  if [ $v = base ]; then unset LLE_HIP_LIB; else export LLE_HIP_LIB=$PWD/lle_amd/liblle_hip_$v.so; fi
  echo "stores $v: $(timeout -k 10 200 python tools/microbench.py --sizes 16384,65536,131072 --epws 0 2>&1 | grep -v amdgpu | tr '\n' '|')"
done; done
