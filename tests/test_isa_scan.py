"""Tripwire for the hipcc miscompile of round 4 (profiles/r04_pes_tax.md section 3): a live-range split copy placed AHEAD of the
exec-restoring `s_or_b64 exec, exec, ...` of a divergent join runs under the region's partial mask, and the lanes that skipped the region
keep a stale value (step_kernel<4,4,4,false,-1> stored through a stale pointer).  tools/isa_exec_copy_scan.py lists join blocks of that
shape in the gfx950 ISA of EVERY kernel translation unit; each hit must be on the audited allowlist
(tests/golden/isa_exec_copy_allowlist.json: kernel, shape, and the argument why the copied register is dead or uniform for the lanes
outside the region).  A hit in another kernel, or of another shape in a listed kernel, fails: read the block, then either fix the source
(step_kernel.hpp LLE_ENV_LATE is how the round-4 case was fixed) or add the audited entry."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    spec = importlib.util.spec_from_file_location("isa_exec_copy_scan", os.path.join(ROOT, "tools", "isa_exec_copy_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scanner_recognises_the_pattern():
    mod = _scanner()
    bad = "_ZN3lle4kernEv:\n.LBB32_247:\n\ts_or_b64 exec, exec, s[2:3]\n.LBB32_248:\n\tv_mov_b64_e32 v[38:39], v[64:65]\n\ts_mov_b32 s24, s46\n\ts_or_b64 exec, exec, s[20:21]\n\tv_or_b32_dpp v0, v31, v31\n"
    hits = mod.scan(bad)
    assert len(hits) == 1 and hits[0][1] == ".LBB32_248" and "v[38:39]" in hits[0][2][0]
    assert mod.classify(bad, hits[0]) == "other"  # (the round-4 fault: a 64-bit VGPR pair, no saved mask, reached by falling through)
    good = "_ZN3lle4kernEv:\n.LBB1_2:\n\ts_or_b64 exec, exec, s[20:21]\n\tv_mov_b64_e32 v[38:39], v[64:65]\n"
    assert mod.scan(good) == []
    # the three audited shapes
    zero = ("_ZN3lle4kernEv:\n\ts_cmp_eq_u64 exec, 0\n\tv_mov_b32_e32 v32, 0\n\ts_cbranch_scc1 .LBB0_9\n\ts_branch .LBB0_3\n.LBB0_9:\n"
            "\tv_mov_b64_e32 v[0:1], 0\n\ts_or_b64 exec, exec, s[96:97]\n")
    assert [mod.classify(zero, h) for h in mod.scan(zero)] == ["exec_zero_only"]
    saved = "_ZN3lle4kernEv:\n\tv_nop\n.LBB0_9:\n\ts_mov_b64 s[0:1], exec\n\tv_mov_b32_e32 v17, s61\n\tv_mov_b64_e32 v[4:5], s[60:61]\n\ts_or_b64 exec, exec, s[58:59]\n"
    assert [mod.classify(saved, h) for h in mod.scan(saved)] == ["uniform_under_saved_mask"]
    assign = "_ZN3lle4kernEv:\n\ts_branch .LBB0_9\n.LBB0_9:\n\tv_mov_b32_e32 v49, v2\n\ts_or_b64 exec, exec, s[54:55]\n"
    assert [mod.classify(assign, h) for h in mod.scan(assign)] == ["region_assign_b32"]


def test_every_translation_unit_against_the_audited_allowlist():
    mod = _scanner()
    from concurrent.futures import ThreadPoolExecutor
    with open(os.path.join(ROOT, "tests", "golden", "isa_exec_copy_allowlist.json")) as f:
        allowed = {(e["kernel"], e["shape"]) for e in json.load(f)["entries"]}
    files = sorted(f for f in os.listdir(mod.SRC) if f.endswith(".hip"))
    assert len(files) == 12 and "kernels.hip" in files and "observers.hip" in files and all(f"step_mode{m}.hip" in files for m in range(10))
    seen, unknown = set(), []
    with ThreadPoolExecutor(max_workers=8) as ex:
        for f, text in zip(files, ex.map(mod.asm_of, files)):
            hits = mod.scan(text)
            for h, name in zip(hits, mod.kernel_names(hits)):
                key = (name, mod.classify(text, h))
                if key in allowed and key not in seen:
                    seen.add(key)
                else:  # (a SECOND block of an audited shape in the same kernel is a new block: audit it)
                    unknown.append((f,) + key + (h[1], h[2]))
    assert not unknown, f"{len(unknown)} join block(s) with a copy ahead of the exec restore that nobody has audited: {unknown[:4]}"
