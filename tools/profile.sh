#!/bin/bash
# Profiling recipe for the headline step (run on the GPU box from the repo root):
#   bash tools/profile.sh <tag>
# Writes rocprofv3 kernel-trace stats and (separate passes) the FETCH_SIZE /
# WRITE_SIZE counters under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
