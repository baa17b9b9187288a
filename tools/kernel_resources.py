"""Register / scratch / occupancy table of every kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).

    python3 tools/kernel_resources.py [--out profiles/rNN_kernel_resources.md] [files ...]

Compiles the translation units of lle_amd/csrc (default: all .hip files) for gfx950 without linking -- no GPU needed --
and writes one row per kernel: VGPRs, SGPRs, scratch bytes per lane, spills, occupancy."""
import argparse
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lle_amd", "csrc")
FIELDS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout
    return [re.sub(r"\(lle::BatchPtrs.*", "", ln).replace("void lle::", "") for ln in out.splitlines()]


def analyse(path):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
           "-c", path, "-o", os.devnull]
    err = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[m.group(1).strip()] = m.group(2)
    return os.path.basename(path), rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--out", default=None)
    ap.add_argument("--jobs", type=int, default=8)
    args = ap.parse_args()
    files = [os.path.abspath(f) for f in args.files] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with concurrent.futures.ThreadPoolExecutor(args.jobs) as ex:
        results = list(ex.map(analyse, files))
    lines = ["# Kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)", "",
             "`step_kernel<G, LM, MODE, ML1, LX>`: G lanes per env, LM beam masks per lane (LM >= 8: kept in the LDS record, not in registers), MODE 0 single step (6: with the row heads ahead of "
             "the state machine) / 1 fused rollout / 2-3 general rollout (several maps / per-env sources) / 4-5 general single step (7 / 8: 4 / 5 with the row heads), ML1 = at most one laser "
             "layer per cell, LX = exact source count.", ""]
    total = spilled = 0
    for fname, rows in results:
        if not rows:
            continue
        names = demangle([r["name"] for r in rows])
        lines += [f"## {fname} ({len(rows)} kernels)", "", "| kernel | VGPRs | SGPRs | scratch B/lane | SGPR spill | VGPR spill | waves/SIMD |", "|---|---|---|---|---|---|---|"]
        for r, nm in zip(rows, names):
            total += 1
            scratch = int(r.get("ScratchSize [bytes/lane]", 0))
            spilled += scratch > 0
            lines.append(f"| `{nm}` | {r.get('VGPRs')} | {r.get('TotalSGPRs')} | {scratch} | {r.get('SGPRs Spill')} | {r.get('VGPRs Spill')} | {r.get('Occupancy [waves/SIMD]')} |")
        lines.append("")
    lines.insert(2, f"{total} kernels, {spilled} with scratch.\n")
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    scratchy = [ln for ln in lines if ln.startswith("| `") and int(ln.split("|")[4]) > 0]
    print(f"{total} kernels, {spilled} with scratch")
    for ln in scratchy:
        print(ln)


if __name__ == "__main__":
    sys.exit(main())
