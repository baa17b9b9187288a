"""Decomposes the step kernel's time on the GPU: full step, logic only (no obs), obs only, for several batch sizes and
envs-per-wave settings.  Prints one line per variant (us per launch, algorithmic GB/s)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--sizes", type=str, default="65536,262144")
    ap.add_argument("--epws", type=str, default="0,32")
    args = ap.parse_args()
    text = Map(level=args.level)
    for n in [int(x) for x in args.sizes.split(",")]:
        for epw in [int(x) for x in args.epws.split(",")]:
            bw = BatchedWorld(text, n, envs_per_wave=epw or None)  # 0 = default step kernel (one lane per agent)
            B = 1937 if args.level == 6 else bw.map.obs_bytes
            t = [0]

            def full():
                bw.step(sample=True, auto_reset=True, seed=1, t=t[0]); t[0] += 1

            def noobs():
                bw.step(sample=True, auto_reset=True, seed=1, t=t[0], write_obs=False); t[0] += 1

            def obsonly():
                bw.observe()

            r = {k: timeit(f) for k, f in (("full", full), ("logic", noobs), ("obs", obsonly))}
            print(f"n={n} epw={epw}: full {r['full']:.2f} us ({B*n/r['full']/1e3:.0f} GB/s)  logic-only {r['logic']:.2f} us  obs-only {r['obs']:.2f} us "
                  f"({bw.map.obs_bytes*n/r['obs']/1e3:.0f} GB/s)", flush=True)
            del bw


if __name__ == "__main__":
    main()
