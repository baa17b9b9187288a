cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
rm -rf $O; mkdir -p $O
for v in A B; do
if [ $v = A ]; then export LLE_HIP_LIB=$R/lle_amd/liblle_hip_A.so; else unset LLE_HIP_LIB; fi
for mode in obs noobs; do
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/$v$mode -- python3 $R/tools/pmc_step.py $mode > /dev/null 2> $O/$v$mode.err; echo "$v $mode rc=$?"
done; done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_sq"
for v in "AB":
  for mode in ("obs","noobs"):
    f=glob.glob(f"{root}/{v}{mode}/**/*counter_collection.csv", recursive=True)
    if not f: print(v,mode,"no file"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, mode, {k: round(sorted(x)[len(x)//2]/4096) for k,x in sorted(acc.items())})
PY
