#!/usr/bin/env bash
# Kernel stats + chunk timeline of one bench config.  Usage: tools/prof_cfg.sh <tag> <bench args...>
set -uo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/p_$TAG -o k -- python3 $R/bench.py "$@" > $OUT/${TAG}_bench.log 2>&1
db=$(find /tmp/p_$TAG -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py "$db" $OUT/${TAG}_kernel_stats.txt "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py $*"
python3 $R/tools/chunk_timeline.py "$db" $OUT/${TAG}_chunk_timeline.txt > /dev/null
