"""gpurun_out/prof_obs/ (tools/collect_observer_profiles.sh) -> profiles/<tag>_observers_kernel_stats.csv,
profiles/<tag>_env_kernel_stats.csv and profiles/<tag>_observers_summary.md."""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_obs")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
with open(os.path.join(ROOT, "profiles", f"{tag}_observers_summary.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats ({tag}): observation builders and BatchedLLE.step, one MI355X, 65 536 envs\n\n")
    for name, script in (("observers", "tools/microbench_observers.py"), ("env", "tools/microbench_env.py")):
        stats = sorted(glob.glob(os.path.join(SRC, name, "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
        shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_{name}_kernel_stats.csv"))
        f.write(f"## `python3 {script}`\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
        for r in csv.DictReader(open(stats)):
            if float(r["Percentage"]) >= 0.05:
                f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |\n")
        f.write("\nProgram output (launch-to-launch event timing; under the profiler the host-bound small launches read higher than in DESIGN.md):\n```\n" + open(os.path.join(SRC, f"{name}.log")).read() + "```\n\n")
print(open(os.path.join(ROOT, "profiles", f"{tag}_observers_summary.md")).read()[:6000])
