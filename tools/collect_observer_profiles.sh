#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace/stats of the observation builders and of BatchedLLE.step
# (tools/microbench_observers.py, tools/microbench_env.py).  Raw output: gpurun_out/prof_obs/; committed summary:
# profiles/<tag>_observers_kernel_stats.csv (tools/summarise_observer_profiles.py).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_obs
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/observers -- python3 $ROOT/tools/microbench_observers.py > $OUT/observers.log 2> $OUT/observers.err; echo "observers rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env -- python3 $ROOT/tools/microbench_env.py > $OUT/env.log 2> $OUT/env.err; echo "env rc=$?"
cat $OUT/observers.log $OUT/env.log
