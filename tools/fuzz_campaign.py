"""A longer run of the sanitized differential fuzz (tests/test_lanes_fuzz_sanitized.py) than the CPU suite affords: the host build of
the lane-per-agent state machine (step_lanes.hpp, the step kernel's source) against the oracle under ASan + UBSan, on freshly
generated small maps (exits moving every 16 steps included).  CPU only.
Usage: python tools/fuzz_campaign.py [first_seed] [n_seeds] [maps_per_seed] [jobs]"""
import concurrent.futures
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_lanes_fuzz_sanitized as fz  # noqa: E402


def one(args):
    seed, count, engine = args
    maps = fz.generated_maps(count, seed=seed)
    with tempfile.TemporaryDirectory() as d:
        out = fz.run(fz.build(), maps, os.path.join(d, "maps.txt"), envs=12, steps=48, engine=engine)
    return seed, engine, {k: int(v) for k, v in out.items()}


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    jobs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    fz.build()
    total = {}
    work = [(first + s, count, engine) for s in range(n) for engine in (1, 2)]
    with concurrent.futures.ProcessPoolExecutor(jobs) as ex:
        for seed, engine, out in ex.map(one, work):
            print(f"seed {seed} engine {engine}: {out}", flush=True)
            for k, v in out.items():
                total[k] = total.get(k, 0) + v
    print(f"TOTAL over {n} seeds x 2 engines x {count} maps: {total}")


if __name__ == "__main__":
    main()
