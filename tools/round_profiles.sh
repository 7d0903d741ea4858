#!/usr/bin/env bash
# Collect the per-round evidence on a GPU box: kernel stats of the default bench
# (cfg5) and of cfg3 (MDNN), PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs),
# chunk timelines.  Usage: tools/round_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
set -uo pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o k -- python3 $R/bench.py "$@" > $OUT/${TAG}_${name}_bench.log 2>&1
  local db=$(find /tmp/p_$name -name "*.db" | head -1)
  python3 $R/tools/rocprof_summary.py "$db" $OUT/${TAG}_${name}_kernel_stats.txt \
    "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py $*"
  python3 $R/tools/chunk_timeline.py "$db" $OUT/${TAG}_${name}_chunk_timeline.txt
}
run_pmc() {     # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/c_${name}_$ctr
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_${name}_$ctr -o c -- python3 $R/bench.py "$@" > /dev/null 2>&1
  local db=$(find /tmp/c_${name}_$ctr -name "*.db" | head -1)
  python3 $R/tools/pmc_dump.py "$db" $OUT/${TAG}_${name}_pmc_$ctr.txt \
    "rocprofv3 --pmc $ctr --kernel-trace -- python bench.py $*"
}
run_stats cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-scaled-batch
run_stats cfg3 --config cfg3 --pairs 20000 --steps 1 --warmup 1 --no-cpu-baseline --no-scaled-batch
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc cfg5 $c --pairs 5000 --steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch
  run_pmc cfg3 $c --config cfg3 --pairs 5000 --steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch
done
