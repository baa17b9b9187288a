"""profile_round.py -- the round's measured evidence in one go (runs on the GPU box: `gpurun -- python3 tools/profile_round.py r02`).

  1. `python3 bench.py`                                               -> gpurun_out/prof/bench.json
  2. `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`       -> per (kernel, grid) launch durations
  3. `rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python3 bench.py`  -> bytes written per launch (L2 -> fabric)
  4. `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py`  -> bytes fetched per launch (x2 on gfx950)
     (counters in passes of their own, with --kernel-trace only: MI355X_MICROARCH.md, HBM / rocprofv3 section)
and writes profiles/<tag>_summary.md, profiles/<tag>_kernel_stats.csv and profiles/traffic.json (read by bench.py).

bench.py runs the headline workload (level 6 x 65 536), the same kernel at 262 144 envs (rows of one launch larger than the
Infinity Cache), cfg2 (level 1 x 4 096), cfg5 (32x32, 8 agents x 65 536) and the fused rollouts, so one profiled run covers
every number of the bench line; launches are told apart by (kernel name, grid size)."""
import csv
import glob
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ENV = dict(os.environ, TMPDIR="/tmp")
BENCH = os.path.join(ROOT, "bench.py")
# (label, kernel-name fragment, grid size in work-items = waves x 64): the single-step launches of the bench line
# (label, kernel-name fragment, environments, algorithmic bytes): the single-step launches of the bench line.  The grid is environments /
# (environments per wavefront) x 64 work-items; since round 5 the batches autotune their environments per wavefront at construction, so the
# grid is whichever of those sizes the kernel was launched with most often.
WORKLOADS = [
    ("cfg3 level 6 x 65 536 (headline)", "step_kernel<4, 4, 6, true, 3>", 65536, 1937 * 65536),  # MODE 6: row heads first
    # (the same kernel also serves 65 536-env batches of other bench blocks -- incremental rows, widened rows -- at 16 / 8 per wavefront: this row's
    #  launches are told apart by grids those cannot produce)
    ("level 6 x 262 144 (rows > Infinity Cache)", "step_kernel<4, 4, 0, true, 3>", 262144, 1937 * 262144, (16, 8)),
    ("cfg2 level 1 x 4 096", "step_kernel<1, 4, 0, true, 0>", 4096, 953 * 4096),
    ("cfg5 32x32 8 agents x 65 536", "step_kernel<8, 8, 0, false, -1>", 65536, 20617 * 65536),
    # BatchedLLE.step with randomize_lasers (bench `lle_step`): per-env sources, row heads first (MODE 8), the fused outputs on top
    ("LLE.step + randomize_lasers, level 6 x 65 536 (MODE 8)", "step_kernel<4, 4, 8, true, 3>", 65536, (1937 + 16 * 4 + 4 + 20 + 1) * 65536),
]
# the step with the rows widened at the store (round 5): profiled in passes of their own (`tools/lle_prof.py target step --obs-dtype ...`: the kernel's
# name is MODE 0's, which the bench line also launches with int8 rows)
WIDE = [("level 6 x 65 536, fp16 rows", "float16", 2), ("level 6 x 65 536, bf16 rows", "bfloat16", 2), ("level 6 x 65 536, fp32 rows", "float32", 4)]


def run(cmd, log):
    print("+", " ".join(cmd), flush=True)
    with open(os.path.join(OUT, log), "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=ENV, stdout=subprocess.PIPE, stderr=f, text=True, timeout=1100)


def latest(pattern):
    files = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def main():
    os.makedirs(OUT, exist_ok=True)
    for d in ("stats", "pmc_write", "pmc_fetch"):
        subprocess.run(["rm", "-rf", os.path.join(OUT, d)])
    res = run([sys.executable, BENCH], "bench.err")
    bench = json.loads(res.stdout.strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(OUT, "bench.json"), "w"))
    light = ["--no-cpu-baseline", "--steps", "200", "--sustained-steps", "0", "--config-steps", "60"]
    res = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(OUT, "stats"), "--", sys.executable, BENCH] + light,
              "stats.err")
    bench_prof = json.loads(res.stdout.strip().splitlines()[-1])
    for kind, counter in (("write", "WRITE_SIZE"), ("fetch", "FETCH_SIZE")):
        run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(OUT, f"pmc_{kind}"), "--",
             sys.executable, BENCH] + light + ["--no-fused", "--steps", "50", "--no-multi-map"], f"pmc_{kind}.err")  # (see bench.py --no-multi-map)

    trace = list(csv.DictReader(open(latest("stats/**/*_kernel_trace.csv"))))
    stats = list(csv.DictReader(open(latest("stats/**/*_kernel_stats.csv"))))

    def pmc(kind, frag, grid):
        rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(latest(f"pmc_{kind}/**/*_counter_collection.csv")))
                if frag in r["Kernel_Name"] and int(r["Grid_Size"]) == grid]
        return (statistics.median(rows), len(rows)) if rows else (None, 0)

    lines = [f"# rocprofv3 summary ({tag}): `python3 bench.py` on one MI355X", "",
             "Collected by `tools/profile_round.py` (one gpurun call): bench.py un-profiled; `rocprofv3 --kernel-trace --stats -- python3 bench.py "
             + " ".join(light) + "`; `--pmc WRITE_SIZE` and `--pmc FETCH_SIZE` in passes of their own.", "",
             "## Launches of the bench line, by (kernel, grid)", "",
             "(A (kernel, grid) pair is launched by several blocks of the bench line -- the cfg5 kernel also by its incremental-rows and consumer-loop blocks, the 262 144-env "
             "kernel with the alternating and the one-directional walk -- so for every row but the headline the MEDIAN is the figure of the block that names it; the headline "
             "kernel's launches are all the same workload.)", "",
             "| workload | kernel | launches | avg us | median us | min us | algorithmic MB | algorithmic GB/s at avg | WRITE_SIZE MB | 2 x FETCH_SIZE MB | traffic / algorithmic |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    traffic = {}
    csv_rows = [["workload", "kernel", "grid", "launches", "avg_ns", "median_ns", "min_ns", "max_ns", "write_bytes", "fetch_bytes_corrected"]]
    import collections
    for label, frag, n_envs, algo, *only in WORKLOADS:
        # the grid this batch's launches used most (the autotune's trials at neighbouring sizes are a few dozen launches each)
        lanes = int(frag.split("<")[1].split(",")[0])
        grids = collections.Counter(int(r["Grid_Size_X"]) for r in trace if frag in r["Kernel_Name"])
        cands = [n_envs // e * 64 for e in (only[0] if only else (64, 32, 16, 8, 4, 2, 1)) if e <= 64 // lanes]
        grid = max(cands, key=lambda g: grids.get(g, 0)) if cands else 0
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if frag in r["Kernel_Name"] and int(r["Grid_Size_X"]) == grid]
        if not d:
            lines.append(f"| {label} | `{frag}` | 0 | - | - | - | - | - | - | - | - |")
            continue
        w, _ = pmc("write", frag, grid)
        fch, _ = pmc("fetch", frag, grid)
        wb = w * 1024 if w is not None else None               # KiB -> bytes (exact for 16-B-per-lane streaming stores)
        fb = 2 * fch * 1024 if fch is not None else None       # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
        tr = (wb + fb) if wb is not None and fb is not None else None
        avg = statistics.mean(d)
        lines.append(f"| {label} | `{frag}` | {len(d)} | {avg/1e3:.2f} | {statistics.median(d)/1e3:.2f} | {min(d)/1e3:.2f} | {algo/1e6:.1f} | {algo/avg:.0f} | "
                     f"{(wb or 0)/1e6:.1f} | {(fb or 0)/1e6:.1f} | {(tr/algo if tr else float('nan')):.3f} |")
        csv_rows.append([label, frag, grid, len(d), round(avg), statistics.median(d), min(d), max(d), wb, fb])
        traffic[label] = tr
    # ---- the widened steps, each in three passes of its own (kernel trace, WRITE_SIZE, FETCH_SIZE)
    lines += ["", "## The step with the rows widened at the store (`lle_batch_options.obs_dtype`; level 6 x 65 536, `tools/lle_prof.py target step --obs-dtype`)", "",
              "| rows | launches | avg us | min us | bytes the launch writes (rows x element size + small outputs) MB | GB/s at avg | of 8 TB/s | WRITE_SIZE MB | 2 x FETCH_SIZE MB |", "|---|---|---|---|---|---|---|---|---|"]
    LP = os.path.join(ROOT, "tools", "lle_prof.py")
    for label, dt, es in WIDE:
        vals = {}
        for kind, extra in (("trace", ["--kernel-trace", "--stats"]), ("write", ["--pmc", "WRITE_SIZE", "--kernel-trace"]), ("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"])):
            sub = os.path.join(OUT, f"wide_{dt}_{kind}")
            subprocess.run(["rm", "-rf", sub])
            run(["rocprofv3"] + extra + ["--output-format", "csv", "-d", sub, "--", sys.executable, LP, "target", "step", "--obs-dtype", dt, "--iters", "300"], f"wide_{dt}_{kind}.err")
            if kind == "trace":
                tr_rows = list(csv.DictReader(open(latest(f"wide_{dt}_trace/**/*_kernel_trace.csv"))))
                dd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr_rows if "step_kernel<4, 4, 0, true, 3>" in r["Kernel_Name"]][20:]
                vals["d"] = dd
            else:
                rows_ = [float(r["Counter_Value"]) for r in csv.DictReader(open(latest(f"wide_{dt}_{kind}/**/*_counter_collection.csv"))) if "step_kernel<4, 4, 0, true, 3>" in r["Kernel_Name"]]
                vals[kind] = statistics.median(rows_) * 1024 if rows_ else None
        dd = vals["d"]
        wbytes = 65536 * (1920 * es + (1937 - 1872))
        avg = statistics.mean(dd)
        lines.append(f"| {label} | {len(dd)} | {avg/1e3:.2f} | {min(dd)/1e3:.2f} | {wbytes/1e6:.1f} | {wbytes/avg:.0f} | {wbytes/avg/8000:.3f} | {(vals['write'] or 0)/1e6:.1f} | {2*(vals['fetch'] or 0)/1e6:.1f} |")
        csv_rows.append([label, "step_kernel<4, 4, 0, true, 3>", 0, len(dd), round(avg), statistics.median(dd), min(dd), max(dd), vals["write"], 2 * (vals["fetch"] or 0)])
    lines += ["", "## Kernel stats of the profiled run (`--stats`; a kernel name covers every grid it ran with)", "",
              "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    for r in stats:
        if float(r["Percentage"]) >= 0.05:
            lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
    g = next(r for r in trace if WORKLOADS[0][1] in r["Kernel_Name"])
    lines += ["", f"Headline kernel: workgroup {g['Workgroup_Size_X']}, LDS {g['LDS_Block_Size']} B (static; the dynamic size is in the bench line), VGPR {g['VGPR_Count']}, "
              f"SGPR {g['SGPR_Count']}, scratch {g['Scratch_Size']}.", "",
              "## bench.py lines of the same session", "", "Un-profiled:", "```json", json.dumps(bench), "```", "",
              "Under rocprofv3 --kernel-trace --stats:", "```json", json.dumps(bench_prof), "```", ""]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(csv_rows)
    json.dump({"hbm_bytes_per_launch": traffic.get(WORKLOADS[0][0]), "hbm_regime_bytes_per_launch": traffic.get(WORKLOADS[1][0]),
               "cfg2_bytes_per_launch": traffic.get(WORKLOADS[2][0]), "cfg5_bytes_per_launch": traffic.get(WORKLOADS[3][0]),
               "fetch_correction": 2.0, "unit": "bytes (WRITE_SIZE + 2 x FETCH_SIZE, KiB counters x 1024)", "source": f"profiles/{tag}_summary.md"},
              open(os.path.join(ROOT, "gpurun_out", f"{tag}_traffic.json"), "w"), indent=1)
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
