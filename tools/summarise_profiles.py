"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof/) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md and profiles/traffic.json (read by bench.py)."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def one(pattern):
    files = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


stats = one("stats/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
step = max((r for r in rows if "step_kernel" in r["Name"] or ("world_kernel" in r["Name"] and ", 0>" in r["Name"])), key=lambda r: float(r["TotalDurationNs"]))
trace = list(csv.DictReader(open(one("stats/*/*_kernel_trace.csv"))))
step_rows = [r for r in trace if r["Kernel_Name"] == step["Name"]]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step_rows]


def pmc(kind):
    f = one(f"pmc_{kind}/*/*_counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == step["Name"]]
    return statistics.median(vals), len(vals)


w_kb, nw = pmc("write")
f_kb, nf = pmc("fetch")
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide
# coalesced read stream -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
traffic = (2 * f_kb + w_kb) * 1024
bench = json.load(open(os.path.join(SRC, "bench.json")))
bench_prof = json.load(open(os.path.join(SRC, "bench_under_rocprof.json")))
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
json.dump({"hbm_bytes_per_launch": traffic, "fetch_kib_raw": f_kb, "write_kib": w_kb, "fetch_correction": 2.0,
           "kernel": step["Name"], "source": f"profiles/{tag}_summary.md"}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
g = step_rows[len(step_rows) // 2]
with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary ({tag}): `python bench.py` on one MI355X\n\n")
    f.write("Command (tools/collect_profiles.sh): `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-fused` (single-step launches only); "
            "PMC counters in separate passes (`--pmc WRITE_SIZE`, `--pmc FETCH_SIZE`).\n\n")
    f.write("## Kernel stats (`--kernel-trace --stats`)\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |\n")
    f.write(f"\nStep kernel: grid {g['Grid_Size_X']} work-items, workgroup {g['Workgroup_Size_X']}, LDS {g['LDS_Block_Size']} B, VGPR {g['VGPR_Count']}, SGPR {g['SGPR_Count']}, "
            f"scratch {g['Scratch_Size']}; median duration {statistics.median(dur)/1e3:.2f} us over {len(dur)} dispatches.\n\n")
    f.write("## HBM traffic of the step kernel (PMC, per launch)\n\n")
    f.write(f"- WRITE_SIZE median {w_kb:.1f} KiB over {nw} dispatches = {w_kb*1024/1e6:.2f} MB\n")
    f.write(f"- FETCH_SIZE median {f_kb:.1f} KiB over {nf} dispatches, x2 gfx950 correction = {2*f_kb*1024/1e6:.2f} MB\n")
    f.write(f"- traffic = {traffic/1e6:.2f} MB per launch vs algorithmic {algo/1e6:.2f} MB (1937 B x 65536 envs): ratio {traffic/algo:.3f}\n\n")
    f.write("## bench.py lines of the same session\n\n")
    f.write("Un-profiled:\n```json\n" + json.dumps(bench) + "\n```\n\nUnder rocprofv3 --kernel-trace --stats:\n```json\n" + json.dumps(bench_prof) + "\n```\n")
    avg_us = float(step["AverageNs"]) / 1e3
    f.write(f"\nbench.py event-timed average launch {bench_prof['roofline']['kernel_ms']*1e3:.2f} us vs rocprofv3 average kernel duration {avg_us:.2f} us "
            f"(same process).\n")
print(open(os.path.join(ROOT, "profiles", f"{tag}_summary.md")).read())
