"""Builds liblle_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_native(force=False, verbose=False):
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building liblle_hip.so failed (hipcc --offload-arch=gfx950)")
    return os.path.join(_HERE, "liblle_hip.so")
