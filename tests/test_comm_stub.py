"""The one-process / N-GPU communicator path of the C ABI (lle_comm_create_all + the *_group reductions, lle_amd/csrc/comm.cpp) walked
end to end on the CPU against a stubbed librccl (LLE_RCCL_LIB) and a preloaded stand-in for the five HIP calls it makes.

RCCL refuses two ranks on one device and the GPU box has one GPU, so N = 2 has never run on hardware ("N > 1 over RCCL has
never executed anywhere"): until an 8-GPU node runs bench.py --gpus 8, this pins the control flow -- group start / end pairing, every
rank's call posted inside one group, results on every rank, and the failure paths fixed in round 3 (commit 335c7f7: nothing leaks when
ncclCommInitAll, a later hipMalloc or an ncclAllReduce inside the group fails).  What it cannot show is RCCL itself: "unmeasured"."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "rccl_stub")

CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.environ["LLE_ROOT"])
from lle_amd import _capi
L = _capi.lib()
rccl = C.CDLL(os.environ["LLE_RCCL_LIB"], mode=C.RTLD_GLOBAL)
hip = C.CDLL(os.environ["FAKE_HIP_LIB"], mode=C.RTLD_GLOBAL)
case = sys.argv[1]
vp = C.c_void_p
comms = (vp * 2)()
rc = L.lle_comm_create_all(comms, 2, None)
if case == "dup":
    rc = L.lle_comm_create_all(comms, 2, (C.c_int * 2)(1, 1))
    assert rc == -2 and b"once" in L.lle_last_error(), (rc, L.lle_last_error())
    print("ok"); sys.exit(0)
if case in ("initall", "malloc2"):
    assert rc == -3, rc
    assert comms[0] is None and comms[1] is None
    assert rccl.fake_rccl_live_comms() == 0 and hip.fake_hip_live_allocations() == 0, (rccl.fake_rccl_live_comms(), hip.fake_hip_live_allocations())
    assert hip.fake_hip_current_device() == 0   # the caller's device is put back
    print("ok"); sys.exit(0)
assert rc == 0, (rc, L.lle_last_error())
r, n = C.c_int(-1), C.c_int(-1)
for k in range(2):
    assert L.lle_comm_rank(comms[k], C.byref(r), C.byref(n)) == 0 and (r.value, n.value) == (k, 2)
bufs = [(C.c_int64 * 6)(*( [10 * (k + 1) + e for e in range(3)] + [0, 0, 0])) for k in range(2)]   # (3 values + the stub's scratch)
ptrs = (vp * 2)(*[C.cast(b, vp) for b in bufs])
if case == "allreduce2":
    rc = L.lle_comm_allreduce_i64_group(comms, ptrs, None, 2, 3, 0)
    assert rc == -3 and b"ncclAllReduce" in L.lle_last_error(), (rc, L.lle_last_error())
    assert rccl.fake_rccl_group_depth() == 0     # the group was closed on the failure path
    assert list(bufs[0])[:3] == [10, 11, 12]     # nothing was reduced
else:
    assert L.lle_comm_allreduce_i64_group(comms, ptrs, None, 2, 3, 0) == 0          # LLE_COMM_SUM
    assert [list(b)[:3] for b in bufs] == [[30, 32, 34]] * 2
    for k in range(2):
        bufs[k][0] = 5 - 9 * k
    assert L.lle_comm_allreduce_i64_group(comms, ptrs, None, 2, 3, 1) == 0          # LLE_COMM_MAX
    assert [list(b)[:3] for b in bufs] == [[5, 32, 34]] * 2
    assert L.lle_comm_allreduce_i64_group(comms, ptrs, None, 1, 3, 0) == -2          # not every rank of the communicator
    one = (C.c_int64 * 2)(41, 0)
    assert L.lle_comm_allreduce_i64(comms[1], C.cast(one, vp), 1, 0, None) == 0 and one[0] == 41   # (a lone call outside a group: the stub reduces over what was posted)
    assert hip.fake_hip_current_device() == 0
for k in range(2):
    L.lle_comm_free(comms[k])
assert rccl.fake_rccl_live_comms() == 0 and hip.fake_hip_live_allocations() == 0
print("ok")
'''


def _build():
    out = {}
    for name in ("fake_rccl", "fake_hip"):
        so = os.path.join(STUB, f"lib{name}.so")
        src = os.path.join(STUB, name + ".c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-o", so, src])
        out[name] = so
    return out


def _run(case, **env):
    libs = _build()
    e = dict(os.environ, LLE_ROOT=ROOT, LLE_RCCL_LIB=libs["fake_rccl"], FAKE_HIP_LIB=libs["fake_hip"], LD_PRELOAD=libs["fake_hip"], FAKE_HIP_DEVICES="2", **env)
    res = subprocess.run([sys.executable, "-c", CHILD, case], env=e, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), (case, res.stdout[-2000:], res.stderr[-4000:])


def test_create_all_and_group_reductions_with_two_ranks():
    _run("happy")


def test_a_device_twice_is_refused():
    _run("dup")


def test_comm_init_all_failure_leaves_nothing_behind():
    _run("initall", FAKE_RCCL_FAIL="initall")


def test_a_later_allocation_failure_unwinds_every_rank():
    _run("malloc2", FAKE_HIP_FAIL_MALLOC="2")


def test_a_failing_all_reduce_inside_the_group_closes_the_group():
    _run("allreduce2", FAKE_RCCL_FAIL="allreduce2")
