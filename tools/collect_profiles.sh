#!/bin/bash
# Runs on the GPU box (through gpurun): bench + rocprofv3 kernel-trace/stats + separate PMC passes (WRITE_SIZE, FETCH_SIZE).
# Raw outputs go to gpurun_out/prof/; tools/summarise_profiles.py turns them into the committed files under profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-fused > $OUT/bench_under_rocprof.json 2> $OUT/stats.err; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-fused --steps 50 > /dev/null 2> $OUT/pmc_write.err; echo "pmc write rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-fused --steps 50 > /dev/null 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?"
cat $OUT/bench.json
