"""lle_prof.py -- the measurement tools of this repo behind one command line (GPU box only).

    python3 tools/lle_prof.py <command> [options]

  step        full step / state machine only / observation only, per batch size and envs-per-wave
  logic       variants of the state-machine-only launch (sampled, explicit STAY, dead envs, invalid actions, floor)
  configs     the BASELINE.json configurations (cfg2, cfg3, cfg5 and neighbours): us per step, GB/s
  hbm         the step kernel past the Infinity Cache: batch size x row alignment x store policy
  rollout     lle_batch_rollout: steps per launch x ring slots
  observers   the other observation builders (observers.hip)
  env         BatchedLLE.step end to end and its pieces
  sources     step with per-environment sources vs the default path
  multimap    config 5 with one map vs 64 distinct maps
  graph       HIP-graph replay of single-step launches vs plain launches
  pipe        state machine and observation as two kernels on two streams (timing prototype)
  streams     the batch split over k HIP streams
  placement   several arenas of the same batch side by side: does the write rate depend on where a buffer lands?
  heads       row heads (lle_map_set_head_lines): us per launch by head size, over maps and batch sizes
  stamps      in-kernel s_memrealtime timeline of the step kernel (--fine: per-quarter view of a stamped MODE-0 build)
  target      a fixed workload for `rocprofv3 --pmc ... -- python3 tools/lle_prof.py target ...` (step / noobs / partial / cfg5 / hbm)

Every timing is launch-to-launch over HIP events on torch's current stream (the stream every launch uses)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lle_amd import BatchedLLE, BatchedWorld, Map, _capi, mapgen  # noqa: E402


_warmed = [False]


def timeit(fn, iters=100, warm=10):
    """us per call, launch-to-launch.  The first measurement of a process runs `fn` for a second first: a fresh box
    ramps its clocks up over the first few hundred milliseconds (config 5: 260 us per step cold, 224 warm)."""
    if not _warmed[0]:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
        _warmed[0] = True
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def algo_bytes(m):
    """algorithmic bytes per env-step."""
    A, G, L = m.n_agents, m.n_gems, m.n_sources
    S = 2 * A + 3 * ((A + 7) // 8) + (G + 7) // 8 + 4 * L
    return m.obs_bytes + 2 * S + A + A + (1 + 2 * A)


def the_map(args):
    if getattr(args, "cfg5", False):
        return Map(mapgen.config5(0), row_align=getattr(args, "row_align", None))
    return Map(level=args.level, row_align=getattr(args, "row_align", None))


def stepper(bw, **kw):
    def fn():
        bw.step(sample=True, auto_reset=True, seed=1, **kw)
    return fn


def ints(s):
    return [int(x) for x in s.split(",") if x]


# ------------------------------------------------------------------------------------------------ commands
def cmd_step(args):
    for n in ints(args.sizes):
        for epw in ints(args.epws):
            bw = BatchedWorld(the_map(args), n, envs_per_wave=epw or None)
            B = algo_bytes(bw.map)
            r = {"full": timeit(stepper(bw)), "logic": timeit(stepper(bw, write_obs=False)), "obs": timeit(bw.observe)}
            print(f"n={n} epw={epw}: full {r['full']:.2f} us ({B*n/r['full']/1e3:.0f} GB/s)  logic-only {r['logic']:.2f} us  obs-only {r['obs']:.2f} us "
                  f"({bw.map.obs_bytes*n/r['obs']/1e3:.0f} GB/s)", flush=True)
            del bw


def cmd_logic(args):
    n = ints(args.sizes)[0]
    for epw in (32, 64):
        bw = BatchedWorld(the_map(args), n, envs_per_wave=epw)
        A = bw.map.n_agents
        stay = torch.full((n, A), 4, dtype=torch.uint8, device="cuda")
        inv = torch.full((n, A), 7, dtype=torch.uint8, device="cuda")
        e0 = torch.empty(1, device="cuda")
        bw.reset(); r1 = timeit(stepper(bw, write_obs=False))
        bw.reset(); r3 = timeit(lambda: bw.step(stay, write_obs=False))
        bw.reset(); r2 = timeit(lambda: bw.step(sample=True, seed=1, write_obs=False))  # all agents die eventually -> mostly STAY-only lanes
        bw.reset(); r4 = timeit(lambda: bw.step(inv, write_obs=False))                  # invalid action: checks only, no step
        r5 = timeit(lambda: e0.add_(1))                                                 # a trivial torch kernel: the launch-to-launch floor
        print(f"n={n} epw={epw}: trivial-kernel floor {r5:.2f} us | sample+autoreset {r1:.2f} us | explicit STAY (fresh envs, no deaths) {r3:.2f} us | "
              f"sample no-reset (mostly dead) {r2:.2f} us | invalid actions (no step) {r4:.2f} us", flush=True)


def cmd_configs(args):
    al = args.row_align
    for name, mp, n in (("config2 level1 n=4096", Map(level=1, row_align=al), 4096), ("level1 n=65536", Map(level=1, row_align=al), 65536),
                        ("config3 level6 n=65536", Map(level=6, row_align=al), 65536), ("config5 32x32x8 n=65536", Map(mapgen.config5(0), row_align=al), 65536),
                        ("config5 32x32x8 n=16384", Map(mapgen.config5(0), row_align=al), 16384)):
        bw = BatchedWorld(mp, n)
        us = timeit(stepper(bw), iters=args.iters)
        B = algo_bytes(mp)
        print(f"{name}: {us:.1f} us/step, {n/us:.1f} M env-steps/s, {B} B/env-step -> {B*n/us/1e3:.0f} GB/s ({B*n/us/1e3/80:.1f} % of 8 TB/s)  {bw.kernel_info()}", flush=True)
        del bw


def cmd_hbm(args):
    """Past the 256 MB Infinity Cache the rows go to DRAM: which pitch and which store policy get the most out of it."""
    for n in ints(args.sizes):
        for align in ints(args.aligns):
            for policy in args.policies.split(","):
                if policy in ("0", "1"):
                    os.environ["LLE_WRITE_THROUGH"] = policy
                    __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
                elif policy != "keep":  # "keep": whatever the caller's environment says
                    os.environ.pop("LLE_WRITE_THROUGH", None)
                    __import__("lle_amd")._capi.refresh_tuning()
                m = Map(mapgen.config5(0), row_align=align) if args.cfg5 else Map(level=args.level, row_align=align)
                bw = BatchedWorld(m, n)
                us = timeit(stepper(bw), iters=args.iters)
                B = algo_bytes(m)
                print(f"n={n} pitch={m.obs_stride} (align {align}) write_through={policy}: {us:.2f} us/step  algorithmic {B*n/us/1e3:.0f} GB/s "
                      f"({B*n/us/1e3/80:.1f} %)  rows written {m.obs_stride*n/us/1e3:.0f} GB/s  rows/launch {m.obs_stride*n/1e6:.0f} MB", flush=True)
                del bw
                torch.cuda.empty_cache()


def cmd_rollout(args):
    n = ints(args.sizes)[0]
    for rep in range(2):
        for T, R in ((8, 8), (16, 8), (16, 16), (32, 8), (64, 8), (16, 4), (16, 2)):
            bw = BatchedWorld(the_map(args), n)
            ring = bw.make_ring(R) if R else None
            us = timeit(lambda: bw.rollout(T, auto_reset=True, seed=1, ring=ring, ring_pos=bw.t), iters=24, warm=3) / T
            B = algo_bytes(bw.map)
            print(f"n={n} steps/launch={T} ring={R} ({R*n*bw.map.obs_stride/1e6:.0f} MB): {us:.2f} us/step  {B*n/us/1e3:.0f} GB/s", flush=True)
            del bw, ring


def cmd_observers(args):
    n = ints(args.sizes)[0]
    for label, m in (("level 6", Map(level=6, row_align=args.row_align)), ("config5 32x32", Map(mapgen.config5(0), row_align=args.row_align))):
        bw = BatchedWorld(m, n)
        for t in range(8):
            bw.step(sample=True, auto_reset=True, seed=1, t=t)
        A = bw.map.n_agents
        for name, kind, param in (("layered (view kernel)", _capi.LLE_OBS_LAYERED, 0), ("layered-padded-2", _capi.LLE_OBS_LAYERED_PADDED, 2),
                                  ("perspective", _capi.LLE_OBS_PERSPECTIVE, 0), ("partial3x3", _capi.LLE_OBS_PARTIAL, 3),
                                  ("partial5x5", _capi.LLE_OBS_PARTIAL, 5), ("partial7x7", _capi.LLE_OBS_PARTIAL, 7),
                                  ("state", _capi.LLE_OBS_STATE, 0), ("normalized-state", _capi.LLE_OBS_NORMALIZED_STATE, 0)):
            d = bw.obs_desc(kind, param)
            buf = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device="cuda")
            buf = buf[(-buf.data_ptr()) % 256:][: int(d.bytes)]
            us = timeit(lambda: bw.observe_as(kind, param, out=buf), iters=50, warm=5)
            print(f"{label:14s} {name:22s} {d.bytes/1e6:8.1f} MB  {us:8.2f} us  {d.bytes/us/1e3:6.0f} GB/s", flush=True)
            del buf
        out = torch.empty((n, A, 5), dtype=torch.uint8, device="cuda")
        for wl in (True, False):
            us = timeit(lambda: bw.available_actions(wl, out=out), iters=50, warm=5)
            bound = bw.bound_available_actions(wl)
            print(f"{label:14s} available_actions(walkable_lasers={wl})  {us:8.2f} us   bound call {timeit(bound, iters=200, warm=20):8.2f} us", flush=True)
        for name, kind in (("state", _capi.LLE_OBS_STATE), ("partial3x3", _capi.LLE_OBS_PARTIAL)):
            bound = bw.bound_observer(kind, 3 if kind == _capi.LLE_OBS_PARTIAL else 0)
            print(f"{label:14s} bound_observer({name})  {timeit(bound, iters=200, warm=20):8.2f} us", flush=True)
        st, rw, av = (torch.empty((n, 3 * A + bw.map.n_gems), device="cuda"), torch.empty((n, 1), device="cuda"),
                      torch.empty((n, A, 5), dtype=torch.uint8, device="cuda"))
        us = timeit(lambda: bw.env_outputs(state=st, reward=rw, available=av), iters=200, warm=20)
        bound = bw.bound_env_outputs(state=st, reward=rw, available=av)
        print(f"{label:14s} env_outputs(state, reward, available)  {us:8.2f} us   bound call {timeit(bound, iters=200, warm=20):8.2f} us", flush=True)
        us = timeit(bw.observe, iters=50, warm=5)
        print(f"{label:14s} observe() (layered, world_kernel)  {us:8.2f} us  {bw.map.obs_bytes*n/us/1e3:6.0f} GB/s", flush=True)
        del bw
        torch.cuda.empty_cache()


def cmd_partial(args):
    """partial k x k: the three kernels (LLE_PARTIAL_KERNEL) and, for the lane kernel, envs per batch / batches per wavefront / store policy"""
    n = ints(args.sizes)[0]
    for label, m in (("level 6", Map(level=6)), ("config5 32x32", Map(mapgen.config5(0)))):
        bw = BatchedWorld(m, n)
        for t in range(8):
            bw.step(sample=True, auto_reset=True, seed=1, t=t)
        for k in (3, 5, 7):
            d = bw.obs_desc(_capi.LLE_OBS_PARTIAL, k)
            buf = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device="cuda")
            buf = buf[(-buf.data_ptr()) % 256:][: int(d.bytes)]
            ref = None
            variants = [dict(LLE_PARTIAL_KERNEL="window"), dict(LLE_PARTIAL_KERNEL="project"), dict()]
            if args.sweep:
                variants += [dict(LLE_PARTIAL_E=str(e)) for e in (1, 2, 4, 8, 16)] + [dict(LLE_PARTIAL_BATCHES=str(b)) for b in (1, 2, 4)]
                variants += [dict(LLE_PARTIAL_WT="1")]
            for env in variants:
                for key in ("LLE_PARTIAL_KERNEL", "LLE_PARTIAL_E", "LLE_PARTIAL_BATCHES", "LLE_PARTIAL_WT"):
                    os.environ.pop(key, None)
                    __import__("lle_amd")._capi.refresh_tuning()
                os.environ.update(env)
                _capi.refresh_tuning()
                try:
                    us = timeit(lambda: bw.observe_as(_capi.LLE_OBS_PARTIAL, k, out=buf), iters=50, warm=5)
                except RuntimeError as e:  # (a forced E the launcher cannot use falls back silently; a hard failure is reported)
                    print(f"{label:14s} {k}x{k} {env}: {e}", flush=True)
                    continue
                got = buf.clone()
                same = "" if ref is None else ("  == window" if torch.equal(got, ref) else "  DIFFERS from the window kernel")
                if ref is None:
                    ref = got
                print(f"{label:14s} {k}x{k} {str(env or 'lanes (default)'):44s} {d.bytes/1e6:7.1f} MB {us:8.2f} us {d.bytes/us/1e3:6.0f} GB/s{same}", flush=True)
            del buf
        del bw
        torch.cuda.empty_cache()


def cmd_env(args):
    n = ints(args.sizes)[0]
    for kw in (dict(), dict(walkable_lasers=False), dict(multi_objective=True), dict(obs_type="partial7x7"), dict(randomize_lasers=True)):
        env = BatchedLLE(Map(level=args.level), n, **kw)
        env.reset()
        w = env.world
        acts = torch.full((n, env.n_agents), 4, dtype=torch.uint8, device=w.device)
        t = lambda f: timeit(f, iters=200, warm=20)  # noqa: E731
        fused = f"{t(lambda: env.step(acts, auto_reset=True, fused=True)):.1f}" if env.walkable_lasers else "n/a"
        print(f"{kw}: step(auto_reset) {t(lambda: env.step(acts, auto_reset=True)):.1f} us (one launch: {fused}) | world.step {t(lambda: w.step(acts, auto_reset=True)):.1f}"
              f" | get_state {t(env.get_state):.1f} | reward {t(env.reward):.1f} | done {t(lambda: env.done):.1f}"
              f" | available_actions {t(env.available_actions):.1f} | get_observation {t(env.get_observation):.1f}", flush=True)


def cmd_sources(args):
    n = ints(args.sizes)[0]
    for pes in (False, True):
        bw = BatchedWorld(the_map(args), n)
        if pes:
            g = torch.Generator().manual_seed(0)
            bw.set_sources(torch.randint(0, bw.map.n_agents, (n, bw.map.n_sources), generator=g, dtype=torch.uint8))
        us = timeit(stepper(bw))
        print(f"n={n} per_env_sources={pes}: {us:.2f} us per step ({algo_bytes(bw.map)*n/us/1e3:.0f} GB/s)  stats={bw.stats()}", flush=True)


def cmd_multimap(args):
    n = ints(args.sizes)[0]
    for label, maps in (("one map", mapgen.config5(0)), ("64 maps x 1024 envs", [mapgen.generate(seed=s) for s in range(64)])):
        bw = BatchedWorld(maps, n)
        us = timeit(stepper(bw), iters=50, warm=5)
        print(f"config 5, {label}: {us:.1f} us per step ({20617*n/us/1e3:.0f} GB/s)  kernel {bw.kernel_info()}", flush=True)
        del bw


def cmd_graph(args):
    n = ints(args.sizes)[0]
    bw = BatchedWorld(the_map(args), n)
    K = 20

    def steps():
        for _ in range(K):
            bw.step(sample=True, auto_reset=True, seed=1)
    print(f"plain launches: {timeit(steps, iters=10, warm=1) / K:.2f} us per step", flush=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        steps()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            steps()
        torch.cuda.synchronize()
        print(f"graph replay:   {timeit(g.replay, iters=10, warm=1) / K:.2f} us per step", flush=True)


def cmd_pipe(args):
    n = ints(args.sizes)[0]
    bw = BatchedWorld(the_map(args), n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    evs, evo = [torch.cuda.Event() for _ in range(4)], [torch.cuda.Event() for _ in range(4)]

    def run(steps, t0, lag):
        for t in range(t0, t0 + steps):
            with torch.cuda.stream(s1):
                if lag and t - t0 >= lag:
                    s1.wait_event(evo[(t - lag) % 4])  # record buffer reuse: observer of step t-lag is done
                bw.step(sample=True, auto_reset=True, seed=1, t=t, write_obs=False)
                evs[t % 4].record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(evs[t % 4])
                bw.observe()
                evo[t % 4].record(s2)

    def fused(steps, t0):
        for t in range(t0, t0 + steps):
            bw.step(sample=True, auto_reset=True, seed=1, t=t)
    B = algo_bytes(bw.map)
    for name, f in (("fused single kernel", fused), ("2 streams, lag 2", lambda k, t0: run(k, t0, 2)),
                    ("2 streams, lag 1", lambda k, t0: run(k, t0, 1)), ("2 streams, no reuse wait", lambda k, t0: run(k, t0, 0))):
        f(20, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f(400, 20)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400 * 1e6
        print(f"{name}: {dt:.2f} us per step of {n} envs ({B*n/dt/1e3:.0f} GB/s)", flush=True)


def cmd_streams(args):
    n = ints(args.sizes)[0]
    for k in (1, 2, 4):
        parts = [BatchedWorld(the_map(args), n // k) for _ in range(k)]
        streams = [torch.cuda.Stream() for _ in range(k)]

        def run(steps, t0):
            for t in range(t0, t0 + steps):
                for i, (p, s) in enumerate(zip(parts, streams)):
                    with torch.cuda.stream(s):
                        p.step(sample=True, auto_reset=True, seed=1, t=t, env_offset=i * (n // k))
        run(20, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(200, 20)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200 * 1e6
        print(f"streams={k}: {dt:.2f} us per step of {n} envs ({algo_bytes(parts[0].map)*n/dt/1e3:.0f} GB/s)", flush=True)


def cmd_placement(args):
    """Four arenas of one batch shape, each timed twice: GB-sized buffers show a write rate that depends on where the
    allocation landed (config 5: 222-225 us on some arenas, 252-257 on others, stable for the life of the buffer)."""
    n = ints(args.sizes)[0]
    m = the_map(args)
    bws = [BatchedWorld(m, n) for _ in range(4)]
    for rnd in range(2):
        for i, bw in enumerate(bws):
            us = timeit(stepper(bw), iters=60, warm=5)
            print(f"round {rnd} arena {i}: base {bw._base.data_ptr():#x} obs {bw.obs_rows.data_ptr():#x}: {us:.1f} us/step "
                  f"({algo_bytes(m)*n/us/1e3:.0f} GB/s)", flush=True)


def cmd_stamps(args):
    """In-kernel timeline: per-wave s_memrealtime stamps (10 ns ticks) of lle_batch_step_stamped (MODE 1 build of the step
    kernel).  --fine expects the one-off diagnostic build that stamps in MODE 0 with 16 slots per wave (NOTEBOOK.md section 4)."""
    import numpy as np
    n = ints(args.sizes)[0]
    slots = 16 if args.fine else 8
    for epw in ([0] if args.fine else ints(args.epws)):
        bw = BatchedWorld([the_map(args), the_map(args)] if getattr(args, "general", False) else the_map(args), n, envs_per_wave=epw or None)
        per = bw.kernel_info()["envs_per_wave"]
        nb = (n + per - 1) // per
        stamps = torch.zeros(nb, slots, dtype=torch.int64, device="cuda")
        if getattr(args, "pes", False):
            g = torch.Generator(device="cuda").manual_seed(1)
            bw.set_sources(colours=torch.randint(0, bw.map.n_agents, (n, bw.map.n_sources), generator=g, device="cuda", dtype=torch.uint8))
        for t in range(30):
            bw.step(sample=True, auto_reset=True, seed=1, t=t)
        torch.cuda.synchronize()
        if getattr(args, "outputs", None):
            import ctypes as C
            A, G = bw.map.n_agents, bw.map.n_gems
            every = dict(state=torch.empty((n, 3 * A + G), device="cuda"), reward=torch.empty(n, device="cuda"), done=torch.empty(n, dtype=torch.uint8, device="cuda"),
                         available=torch.empty((n, A, 5), dtype=torch.uint8, device="cuda"))
            names = [k for k in args.outputs.split(",") if k != "none"]
            part = [k for k in names if k.startswith("partial")]
            extra = {}
            if part:  # e.g. partial7: the step launch writes the partial 7 x 7 observation (MODE 9)
                kk = int(part[0][len("partial"):])
                extra = dict(partial=bw.partial_buffer(kk)[0], partial_k=kk)
            eo = bw.make_env_outputs(**{k: every[k] for k in names if k not in part}, **extra)
            for t in range(30, 40):
                bw.step(sample=True, auto_reset=True, seed=1, t=t, env_out=eo)
            torch.cuda.synchronize()
            L = C.CDLL(os.environ["LLE_HIP_LIB"])
            L.lle_debug_set_stamps.argtypes = [C.c_void_p]
            L.lle_debug_set_stamps(stamps.data_ptr())
            bw.step(sample=True, auto_reset=True, seed=1, t=40, env_out=eo)
            torch.cuda.synchronize()
            L.lle_debug_set_stamps(None)
        else:
            assert _capi.lib().lle_batch_step_stamped(bw.h, 3, 1, 30, stamps.data_ptr(), bw._stream()) == 0
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().astype("float64") * 0.01  # us
        t0 = s[:, 0].min()
        if args.fine:
            order = [(0, "entry"), (7, "rows requested+written"), (1, "barrier+template copy"), (2, "state in registers"), (11, "loop top"),
                     (8, "sampled"), (9, "checked+conflicts"), (10, "passes done"), (3, "state machine done"), (4, "records in LDS"),
                     (5, "obs issued"), (6, "drained")]
            for q in range(4):
                sl = slice(q * nb // 4, (q + 1) * nb // 4)
                print(f"quarter {q}: " + "  ".join(f"{name} {np.percentile(s[sl, i] - t0, 50):.2f}" for i, name in order))
            continue
        names = ["entry", "tables in LDS", "state requested", "logic done", "records in LDS", "obs stores issued", "drained", "own table rows copied"]
        print(f"n={n} epw={per} waves={nb}: us since the first wave's entry; p10 / p50 / p90 / p99 / max over waves")
        for i in [0, 7, 1, 2, 3, 4, 5, 6]:
            col = s[:, i] - t0
            print(f"   {names[i]:22s} " + " ".join(f"{np.percentile(col, p):7.2f}" for p in (10, 50, 90, 99)) + f" {col.max():7.2f}")
        d = s[:, 6] - t0
        for q in range(8):
            sl = slice(q * nb // 8, (q + 1) * nb // 8)
            print(f"   waves [{sl.start},{sl.stop}): drained p50 {np.percentile(d[sl], 50):6.2f} max {d[sl].max():6.2f}; logic done p50 "
                  f"{np.percentile(s[sl, 3] - t0, 50):6.2f}; entry p50 {np.percentile(s[sl, 0] - t0, 50):6.2f}")


def cmd_heads(args):
    """Row heads (lle_map_set_head_lines): us per launch by head size, over maps and batch sizes (heads forced on)."""
    os.environ["LLE_ROW_HEADS"] = "1"
    __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
    cases = [("level 6", lambda: Map(level=6), [16384, 32768, 65536, 131072, 262144]), ("level 5", lambda: Map(level=5), [65536]),
             ("level 3", lambda: Map(level=3), [65536, 131072]),
             ("generated 16x16, 8 agents", lambda: Map(mapgen.generate(16, 16, 8, 4, 4, seed=1), row_align=128), [16384, 32768]),
             ("generated 12x12, 12 agents", lambda: Map(mapgen.generate(12, 12, 12, 4, 4, seed=1), row_align=128), [8192, 16384])]
    for name, mk, sizes in cases:
        m0 = mk()
        auto = m0.row_head[1] // 128
        m0.set_head_lines(8)
        print(f"{name}: {m0.n_agents} agents, rows of {m0.obs_stride} B, longest static run {m0.row_head}, automatic head {auto} lines", flush=True)
        for n in sizes:
            out = []
            for h in (0, 1, 2, 3, 4, 6, 8):
                m = mk()
                m.set_head_lines(h)
                if h and m.row_head[1] < h * 128:
                    break
                bw = BatchedWorld(m, n)
                out.append(f"{h}:{timeit(stepper(bw), iters=200):.2f}")
                del bw
            print(f"  n={n}: " + "  ".join(out), flush=True)


def cmd_target(args):
    """A fixed workload to put behind `rocprofv3 ... --` (kernel trace or one --pmc pass)."""
    what = args.what
    n = ints(args.sizes)[0]
    if what == "partial":
        bw = BatchedWorld(the_map(args), n)
        for t in range(20):
            bw.step(sample=True, auto_reset=True, seed=1, t=t)
        for _ in range(args.iters):
            bw.observe_as(_capi.LLE_OBS_PARTIAL, args.k)
    elif what == "perspective":
        bw = BatchedWorld(the_map(args), n)
        for t in range(20):
            bw.step(sample=True, auto_reset=True, seed=1, t=t)
        d = bw.obs_desc(_capi.LLE_OBS_PERSPECTIVE, 0)
        buf = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device="cuda")
        buf = buf[(-buf.data_ptr()) % 256:][: int(d.bytes)]
        for _ in range(args.iters):
            bw.observe_as(_capi.LLE_OBS_PERSPECTIVE, 0, out=buf)
    elif what == "pes":  # per-environment sources (LLE.step with randomize_lasers): step_kernel MODE 8 / 5
        bw = BatchedWorld(the_map(args), n)
        g = torch.Generator().manual_seed(0)
        bw.set_sources(torch.randint(0, bw.map.n_agents, (n, bw.map.n_sources), generator=g, dtype=torch.uint8))
        for t in range(args.iters):
            bw.step(sample=True, auto_reset=True, seed=1, t=t, recolour_resets=True)
    else:  # step | noobs | cfg5 | hbm
        m = Map(mapgen.config5(0), row_align=args.row_align) if what == "cfg5" else the_map(args)
        bw = BatchedWorld(m, 262144 if what == "hbm" and n == 65536 else n, obs_dtype=getattr(args, "obs_dtype", None))
        for t in range(args.iters):
            bw.step(sample=True, auto_reset=True, seed=1, t=t, write_obs=what != "noobs")
    torch.cuda.synchronize()
    print(f"target {what}: done ({bw.kernel_info()})", flush=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    cmds = {"step": cmd_step, "logic": cmd_logic, "configs": cmd_configs, "hbm": cmd_hbm, "rollout": cmd_rollout, "observers": cmd_observers,
            "partial": cmd_partial, "env": cmd_env, "sources": cmd_sources, "multimap": cmd_multimap, "graph": cmd_graph, "pipe": cmd_pipe, "streams": cmd_streams,
            "placement": cmd_placement, "stamps": cmd_stamps, "heads": cmd_heads, "target": cmd_target}
    for name, fn in cmds.items():
        p = sub.add_parser(name, help=(fn.__doc__ or "").strip().split("\n")[0])
        p.add_argument("--level", type=int, default=6)
        p.add_argument("--cfg5", action="store_true", help="the generated 32x32 / 8 agents / 8 lasers map instead of a level")
        p.add_argument("--sizes", default="65536,262144" if name in ("step",) else ("65536,131072,196608,262144" if name == "hbm" else "65536"))
        p.add_argument("--epws", default="0,32" if name == "step" else "0")
        p.add_argument("--row-align", type=int, default=None)
        p.add_argument("--iters", type=int, default=50 if name in ("configs", "hbm") else 60)
        if name == "hbm":
            p.add_argument("--aligns", default="16,128")
            p.add_argument("--policies", default="auto,0,1", help="LLE_WRITE_THROUGH settings: auto, 0 (plain), 1 (sc1)")
        if name == "partial":
            p.add_argument("--sweep", action="store_true", help="also envs per batch, batches per wavefront and the store policy of the lane kernel")
        if name == "stamps":
            p.add_argument("--fine", action="store_true")
            p.add_argument("--pes", action="store_true", help="per-environment sources (random colours): the MODE 3 build")
            p.add_argument("--outputs", default=None, help="comma list of fused LLE.step outputs (state,reward,done,available) or 'none': lle_batch_step_outputs, diagnostic build only")
            p.add_argument("--general", action="store_true", help="two copies of the map, half of the environments each: the general (several maps) build")
        if name == "target":
            p.add_argument("what", choices=["step", "noobs", "partial", "perspective", "cfg5", "hbm", "pes"])
            p.add_argument("-k", type=int, default=7, help="window of the partial observer")
            p.add_argument("--obs-dtype", default=None, help="element type of the rows (step / hbm / cfg5): int8 (default), float16, bfloat16, float32")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "lle_prof.py needs an MI355X"
    cmds[args.cmd](args)


if __name__ == "__main__":
    main()
