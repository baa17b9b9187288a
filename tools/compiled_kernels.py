"""Names of the kernels COMPILED into lle_amd/liblle_hip.so (the gfx950 code objects of its offload bundles), in the spelling of
lle_debug_reachable / lle_debug_launched ("step_kernel<4,4,6,true,3>").

    python3 tools/compiled_kernels.py            # one name per line

tests/test_capi.py compares this list with lle_debug_reachable(): a kernel the dispatch cannot reach is a kernel no test can launch."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def compiled_kernels(lib=os.path.join(ROOT, "lle_amd", "liblle_hip.so")):
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        mangled = set()
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", os.path.join(tmp, f)], stdout=subprocess.PIPE, text=True, check=True).stdout
            for line in out.splitlines():
                cols = line.split()
                if len(cols) >= 8 and cols[3] == "OBJECT" and cols[7].endswith(".kd"):  # one kernel descriptor per kernel
                    mangled.add(cols[7][:-3])
    mangled = sorted(mangled)
    dem = subprocess.run(["c++filt"], input="\n".join(mangled), stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    names = set()
    for d in dem:
        m = re.match(r"_ZN3lle(\d+)", d)  # (binutils' c++filt does not know _Float16 arguments: a plain lle::name stays mangled)
        if m:
            k = m.end()
            d = "lle::" + d[k:k + int(m.group(1))] + "("
        d = re.sub(r"^void ", "", d)
        d = d.replace("lle::", "")
        m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)\(", d)
        names.add((m.group(1) if m else d).replace(", ", ","))
    return sorted(names)


if __name__ == "__main__":
    print("\n".join(compiled_kernels()))
    sys.exit(0)
