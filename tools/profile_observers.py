"""profile_observers.py -- rocprofv3 evidence for the observation builders (GPU box: `gpurun -- python3 tools/profile_observers.py r02`).

  1. `rocprofv3 --kernel-trace --stats -- python3 tools/lle_prof.py observers`     kernel durations of every builder
  2. `rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 tools/lle_prof.py target partial -k 7` with
     LLE_PARTIAL_PROJECT=0 (window kernel) and =1 (projection kernel): instructions per wavefront, before / after
Writes gpurun_out/<tag>_observers_summary.md (copy it to profiles/)."""
import csv
import glob
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "prof_obs")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
PROF = os.path.join(ROOT, "tools", "lle_prof.py")
COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]


def run(cmd, log, env=None):
    print("+", " ".join(cmd), flush=True)
    with open(os.path.join(OUT, log), "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", **(env or {})), stdout=subprocess.PIPE, stderr=f, text=True, timeout=900)


def latest(pattern):
    files = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def main():
    subprocess.run(["rm", "-rf", OUT])
    os.makedirs(OUT, exist_ok=True)
    res = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(OUT, "stats"), "--", sys.executable, PROF, "observers"],
              "stats.err")
    lines = [f"# rocprofv3 ({tag}): the observation builders, one MI355X, 65 536 envs", "",
             "## `rocprofv3 --kernel-trace --stats -- python3 tools/lle_prof.py observers`", "",
             "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(latest("stats/**/*_kernel_stats.csv"))):
        if float(r["Percentage"]) >= 0.05:
            lines.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
    lines += ["", "Program output (launch-to-launch event timing; a kernel name covers level 6 and config 5):", "```", res.stdout.strip(), "```", "",
              "## SQ counters of the partial k x k kernels", "",
              "`rocprofv3 --pmc " + " ".join(COUNTERS) + " --kernel-trace -- python3 tools/lle_prof.py target partial -k K [--cfg5]` with "
              "`LLE_PARTIAL_KERNEL=window` / `project` / unset (the lane-per-(env, observer) kernel, round 3); medians over the dispatches, "
              "per ENVIRONMENT (counter / 65 536) -- the kernels differ in environments per wavefront.", "",
              "| map | k | kernel | waves | " + " | ".join(c.replace("SQ_", "") + " / env" for c in COUNTERS[1:]) + " |", "|---|---|---|---|" + "---|" * (len(COUNTERS) - 1)]
    n_envs = 65536
    for label, extra in (("level 6", []), ("config 5", ["--cfg5"])):
        for k in ("3", "7"):
            for which, name in (("window", "partial_observe_kernel"), ("project", "partial_project_kernel"), ("", "partial_lanes_kernel")):
                d = os.path.join(OUT, f"pmc_{label.replace(' ', '')}_{k}_{which or 'lanes'}")
                run(["rocprofv3", "--pmc"] + COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, PROF, "target", "partial",
                     "-k", k, "--iters", "20"] + extra, os.path.basename(d) + ".err", env={"LLE_PARTIAL_KERNEL": which} if which else {})
                acc = {}
                f = latest(os.path.basename(d) + "/**/*_counter_collection.csv")
                for r in (csv.DictReader(open(f)) if f else []):
                    if name in r["Kernel_Name"]:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if not acc:
                    lines.append(f"| {label} | {k} | `{name}` | (no dispatches found) |")
                    continue
                med = {kk: statistics.median(v) for kk, v in acc.items()}
                lines.append(f"| {label} | {k}x{k} | `{name}` | {med.get('SQ_WAVES', 0):.0f} | " +
                             " | ".join(f"{med.get(c, float('nan')) / n_envs:.1f}" for c in COUNTERS[1:]) + " |")
    text = "\n".join(lines) + "\n"
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_observers_summary.md"), "w").write(text)
    print(text[-2500:])


if __name__ == "__main__":
    main()
