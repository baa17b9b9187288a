"""Tripwire for the hipcc miscompile of round 4 (profiles/r04_pes_tax.md section 3): a live-range split copy placed AHEAD of the
exec-restoring `s_or_b64 exec, exec, ...` of a divergent join runs under the region's partial mask.  tools/isa_exec_copy_scan.py lists join
blocks of that shape; the general single-step kernels without heads (MODE 4 / 5: the instantiations that sat at the register cap) must have
none -- the rollout modes carry legitimate phi copies of that shape and are not asserted on."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    spec = importlib.util.spec_from_file_location("isa_exec_copy_scan", os.path.join(ROOT, "tools", "isa_exec_copy_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scanner_recognises_the_pattern():
    scan = _scanner().scan
    bad = "_ZN3lle4kernEv:\n.LBB32_247:\n\ts_or_b64 exec, exec, s[2:3]\n.LBB32_248:\n\tv_mov_b64_e32 v[38:39], v[64:65]\n\ts_mov_b32 s24, s46\n\ts_or_b64 exec, exec, s[20:21]\n\tv_or_b32_dpp v0, v31, v31\n"
    hits = scan(bad)
    assert len(hits) == 1 and hits[0][1] == ".LBB32_248" and "v[38:39]" in hits[0][2][0]
    good = "_ZN3lle4kernEv:\n.LBB1_2:\n\ts_or_b64 exec, exec, s[20:21]\n\tv_mov_b64_e32 v[38:39], v[64:65]\n"
    assert scan(good) == []


def test_general_single_step_kernels_have_no_copy_ahead_of_an_exec_restore():
    mod = _scanner()
    from concurrent.futures import ThreadPoolExecutor
    files = ["step_mode4.hip", "step_mode5.hip"]
    with ThreadPoolExecutor(max_workers=2) as ex:
        for f, text in zip(files, ex.map(mod.asm_of, files)):
            hits = mod.scan(text)
            assert hits == [], (f, hits[:3])
