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


def kernel_names(hits):
    """Demangled, in the spelling of lle_debug_launched: 'step_kernel<2,4,1,true,0>'."""
    dem = subprocess.run(["c++filt"], input="\n".join(h[0] or "?" for h in hits), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"^void ", "", d.split("(")[0]).replace("lle::", "").replace(", ", ",") for d in dem[:len(hits)]]


def classify(text, hit):
    """The SHAPE of a hit, for the allowlist of tests/test_isa_scan.py (tests/golden/isa_exec_copy_allowlist.json):
      exec_zero_only       the block is entered only by `s_cbranch_scc1` behind `s_cmp_eq_u64 exec, 0` (never by falling through): the copy
                           runs with no lane active and writes nothing;
      uniform_under_saved_mask  the block saves its mask (`s_mov_b64 s[a:b], exec`) and every copy broadcasts a SCALAR register: wave-uniform values
                           handed to the lanes of the region, whose consumer runs under that saved mask again;
      region_assign_b32    ONE 32-bit VGPR-to-VGPR copy: a source-level assignment inside the divergent region (the argument is per kernel);
      other                anything else -- the round-4 miscompile was a 64-bit VGPR pair (a pointer) copied this way."""
    kernel, label, copies, _restore = hit
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next((i for i in range(start + 1, len(lines)) if re.match(r"^_Z\w+:", lines[i])), len(lines))
    body = lines[start:end]
    at = next(i for i, l in enumerate(body) if l.startswith(label + ":"))
    refs = [i for i, l in enumerate(body) if re.search(r"\s" + re.escape(label) + r"\s*$", l) and not l.startswith(label)]
    prev = next((body[i].strip() for i in range(at - 1, -1, -1) if body[i].strip() and not body[i].lstrip().startswith(";")), "")
    falls_through = not prev.startswith(("s_branch", "s_endpgm", "s_setpc"))
    if refs and not falls_through and all(body[i].strip().startswith("s_cbranch_scc1") and
                                          any("s_cmp_eq_u64 exec, 0" in body[j] for j in range(max(0, i - 4), i)) for i in refs):
        return "exec_zero_only"
    block = []
    for i in range(at + 1, len(body)):
        if RESTORE.match(body[i]):
            break
        block.append(body[i].strip())
    if any(re.match(r"s_mov_b64 s\[\d+:\d+\], exec$", b) for b in block) and all(re.search(r",\s*s(\d+|\[\d+:\d+\])$", c) for c in copies):
        return "uniform_under_saved_mask"
    if len(copies) == 1 and re.match(r"v_mov_b32_e32 v\d+, v\d+$", copies[0]):
        return "region_assign_b32"
    return "other"


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
