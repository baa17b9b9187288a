"""Scan the gfx950 ISA of the kernels for a register-allocator copy placed AHEAD of the exec-restoring `s_or_b64 exec, exec, ...` at the top of a
join block: such a copy runs under the divergent region's partial exec mask, so lanes that skipped the region keep a stale value (round 4: this
miscompile of hipcc 7.2 made `step_kernel<4,4,4,false,-1>` store through a stale pointer, profiles/r04_pes_tax.md).  Usage:
    python tools/isa_exec_copy_scan.py [file.hip ...]        (default: every kernel translation unit of lle_amd/csrc)
Exit code 1 when a suspicious block is found."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lle_amd", "csrc")
COPY = re.compile(r"^\t(v_mov_b32|v_mov_b64|v_accvgpr_\w+|v_pk_mov_b32)(_e32|_e64)?\s")
SMOV = re.compile(r"^\t(s_mov_b32|s_mov_b64|v_readlane_b32|v_writelane_b32|s_nop)\s")
RESTORE = re.compile(r"^\ts_or_b64 exec, exec, ")


def asm_of(path):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", path, "-o", "-"],
                         cwd=SRC, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    return out.stdout


def scan(text):
    hits, kernel = [], None
    lines = text.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        if not re.match(r"^\.LBB\d+_\d+:", l):
            continue
        copies, j = [], i + 1
        while j < len(lines) and (COPY.match(lines[j]) or SMOV.match(lines[j]) or lines[j].startswith(";") or not lines[j].strip()):
            if COPY.match(lines[j]):
                copies.append(lines[j].strip())
            j += 1
        if copies and j < len(lines) and RESTORE.match(lines[j]):
            hits.append((kernel, l.split(":")[0], copies, lines[j].strip()))
    return hits


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))
    bad = 0
    with ThreadPoolExecutor(max_workers=8) as ex:
        for f, text in zip(files, ex.map(asm_of, files)):
            hits = scan(text)
            demangled = subprocess.run(["c++filt"], input="\n".join(h[0] or "?" for h in hits), capture_output=True, text=True).stdout.split("\n")
            print(f"{f}: {len(hits)} suspicious block(s)", flush=True)
            for h, name in zip(hits, demangled):
                print(f"   {name.split('(')[0]}  {h[1]}: {'; '.join(h[2])}  |  {h[3]}")
            bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
