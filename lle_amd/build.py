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


def build_c_example(verbose=False, name="c_abi_rollout"):
    """examples/c_abi_rollout.c, examples/c_abi_multi_gpu.c: plain-C hosts over include/lle_hip.h and the HIP runtime (gcc, no
    torch, no Python)."""
    root = os.path.dirname(_HERE)
    src, out = os.path.join(root, "examples", name + ".c"), os.path.join(root, "examples", name)
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
           src, "-o", out, "-L" + _HERE, "-llle_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../lle_amd", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError(f"building examples/{name} failed")
    return out
