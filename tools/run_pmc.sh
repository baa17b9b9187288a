# SQ counters of the default step kernel (level 6, 65 536 envs), with and without the observation stream, one PMC pass
# each (no other trace domains); medians per wavefront go to gpurun_out/pmc_sq/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
rm -rf $O; mkdir -p $O
for mode in obs noobs; do
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/$mode -- python3 $R/tools/pmc_step.py $mode > /dev/null 2> $O/$mode.err; echo "$mode rc=$?"
done
python3 - <<'PY' | tee $O/summary.txt
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_sq"
for mode in ("obs","noobs"):
    f=glob.glob(f"{root}/{mode}/**/*counter_collection.csv", recursive=True)
    if not f: print(mode,"no file"); continue
    acc=collections.defaultdict(list); name=None
    for r in csv.DictReader(open(f[0])):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"]
    print(f"{'with' if mode=='obs' else 'without'} the observation stream ({name}), median over dispatches, per wavefront (/4096):")
    for k,x in sorted(acc.items()): print(f"   {k:22s} {sorted(x)[len(x)//2]/4096:10.0f}")
PY
