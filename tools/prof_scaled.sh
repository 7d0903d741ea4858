#!/usr/bin/env bash
# Kernel stats of the scaled-batch mode alone.  Usage: tools/prof_scaled.sh <tag> [bench args]
set -uo pipefail
TAG=${1:-rXX}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_scaled
rocprofv3 --kernel-trace --stats -d /tmp/p_scaled -o k -- python3 $R/bench.py --only-scaled-batch "$@" > $OUT/${TAG}_scaled_bench.log 2>&1
db=$(find /tmp/p_scaled -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py "$db" $OUT/${TAG}_scaled_kernel_stats.txt \
  "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --only-scaled-batch $*"
cp "$db" $OUT/${TAG}_scaled.db
