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
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
              "## SQ counters of partial 7x7 on level 6: window kernel vs projection kernel", "",
              "`rocprofv3 --pmc " + " ".join(COUNTERS) + " --kernel-trace -- python3 tools/lle_prof.py target partial -k 7` with `LLE_PARTIAL_PROJECT=0` / `1`; "
              "medians over the dispatches, per wavefront (the window kernel runs 8 envs per wavefront, the projection 16).", "",
              "| kernel | waves | " + " | ".join(c.replace("SQ_", "") for c in COUNTERS[1:]) + " | VALU per env |", "|---|---|" + "---|" * len(COUNTERS)]
    for proj, name, epw in (("0", "partial_observe_kernel", 8), ("1", "partial_project_kernel", 16)):
        d = os.path.join(OUT, f"pmc{proj}")
        run(["rocprofv3", "--pmc"] + COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, PROF, "target", "partial", "-k", "7", "--iters", "20"],
            f"pmc{proj}.err", env={"LLE_PARTIAL_PROJECT": proj})
        acc = {}
        for r in csv.DictReader(open(latest(f"pmc{proj}/**/*_counter_collection.csv"))):
            if name in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if not acc:
            lines.append(f"| `{name}` | (no dispatches found) |")
            continue
        med = {k: statistics.median(v) for k, v in acc.items()}
        waves = med.get("SQ_WAVES", 1.0) or 1.0
        per = [med.get(c, float("nan")) / waves for c in COUNTERS[1:]]
        lines.append(f"| `{name}` | {waves:.0f} | " + " | ".join(f"{v:.0f}" for v in per) + f" | {per[0] / epw:.0f} |")
    text = "\n".join(lines) + "\n"
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_observers_summary.md"), "w").write(text)
    print(text[-2500:])


if __name__ == "__main__":
    main()
