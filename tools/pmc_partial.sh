#!/bin/bash
# usage: pmc_partial.sh <tag> [env assignments...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES; do
  env "$@" rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_partial/$tag/$c -- python3 $GRAFT_REPO_ROOT/tools/lle_prof.py target partial -k 7 --iters 40 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob,collections
tag="$tag"
out={}
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES".split():
    tot=n=0
    for f in glob.glob(f"gpurun_out/pmc_partial/{tag}/{c}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "partial_lanes" in r["Kernel_Name"] and r["Counter_Name"]==c:
                tot+=float(r["Counter_Value"]); n+=1
    out[c]=tot/max(n,1)
dur=[]
for f in glob.glob(f"gpurun_out/pmc_partial/{tag}/SQ_WAVES/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "partial_lanes" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
w=out["SQ_WAVES"] or 1
print(tag, {k:round(v/w,1) for k,v in out.items()}, "waves",w, "dur_us", round(sum(dur)/max(len(dur),1),2), "launches", len(dur))
PY
