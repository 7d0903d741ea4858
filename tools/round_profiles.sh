#!/usr/bin/env bash
# Collect the per-round evidence on a GPU box (writes gpurun_out/<tag>_*; copy what is to be
# judged into profiles/):
#   - kernel stats (rocprofv3 --kernel-trace --stats) + chunk timeline of the default bench (cfg5)
#     and of cfg3 (MDNN on cross-correlation factor rows), and of the scaled-batch mode alone
#   - PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, --kernel-trace only) for cfg5 and cfg3
#   - kernel stats + PMC of the summarizers at 50k / 100k trajectories (tools/summarizer_bench.py)
#   - SQ_VALU_MFMA_BUSY_CYCLES / SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / GRBM_GUI_ACTIVE for cfg5, cfg3, the
#     streamed kernel and the scaled-batch fit (MFMA-busy fractions, LDS conflict rates)
# Every file starts with "# head: <commit>": the commit the profile was taken on (the GPU box has no
# .git: pass it -- tools/round_profiles.sh <tag> $(git rev-parse --short HEAD)); bench.pmc_traffic
# refuses a profile whose commit is not an ancestor of HEAD.
# Usage: tools/round_profiles.sh <tag> <commit>
set -uo pipefail
TAG=${1:-rXX}
HEADSHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
CSRC=$(python3 "$R/tools/csrc_hash.py")
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, program, args...
  local name=$1 prog=$2; shift 2
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o k -- python3 $R/$prog "$@" > $OUT/${TAG}_${name}_run.log 2>&1
  local db=$(find /tmp/p_$name -name "*.db" | head -1)
  python3 $R/tools/rocprof_summary.py "$db" $OUT/${TAG}_${name}_kernel_stats.txt \
    "$TAG: rocprofv3 --kernel-trace --stats -- python $prog $*"
  sed -i "1i # head: $HEADSHA\n# csrc: $CSRC" $OUT/${TAG}_${name}_kernel_stats.txt
  if [ "$prog" = bench.py ]; then
    python3 $R/tools/chunk_timeline.py "$db" $OUT/${TAG}_${name}_chunk_timeline.txt > /dev/null 2>&1 || true
    [ -f $OUT/${TAG}_${name}_chunk_timeline.txt ] && sed -i "1i # head: $HEADSHA\n# csrc: $CSRC" $OUT/${TAG}_${name}_chunk_timeline.txt
  fi
}
run_pmc() {     # name, counter, program, args...
  local name=$1 ctr=$2 prog=$3; shift 3
  rm -rf /tmp/c_${name}_$ctr
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_${name}_$ctr -o c -- python3 $R/$prog "$@" > /dev/null 2>&1
  local db=$(find /tmp/c_${name}_$ctr -name "*.db" | head -1)
  python3 $R/tools/pmc_dump.py "$db" $OUT/${TAG}_${name}_pmc_$ctr.txt \
    "rocprofv3 --pmc $ctr --kernel-trace -- python $prog $*"
  sed -i "1i # head: $HEADSHA\n# csrc: $CSRC" $OUT/${TAG}_${name}_pmc_$ctr.txt
}
B5="--steps 1 --warmup 1 --no-cpu-baseline --no-scaled-batch --no-per-config --no-largest-size"
run_stats cfg5 bench.py $B5
run_stats cfg3 bench.py --config cfg3 --pairs 20000 $B5
run_stats scaled bench.py --only-scaled-batch
run_stats summarizers tools/summarizer_bench.py
# first layers that do not fit the chip: the streamed kernel (cfg/anymal.yaml, cfg/shadow_hand_more.yaml)
run_stats anymal bench.py --config anymal_yaml --pairs 5000 $B5
run_stats shadow_more bench.py --config shadow_more --pairs 5000 $B5
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc anymal $c bench.py --config anymal_yaml --pairs 2000 --steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch --no-per-config --no-largest-size
  run_pmc shadow_more $c bench.py --config shadow_more --pairs 2000 --steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch --no-per-config --no-largest-size
  run_pmc cfg5 $c bench.py --pairs 5000 --steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch --no-per-config --no-largest-size
  run_pmc cfg3 $c bench.py --config cfg3 --pairs 5000 --steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch --no-per-config --no-largest-size
  run_pmc summarizers $c tools/summarizer_bench.py
done
BQ="--steps 1 --warmup 0 --no-cpu-baseline --no-scaled-batch --no-per-config --no-largest-size"
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE; do
  run_pmc cfg5 $c bench.py --pairs 5000 $BQ
  run_pmc cfg3 $c bench.py --config cfg3 --pairs 5000 $BQ
  run_pmc shadow_more $c bench.py --config shadow_more --pairs 2000 $BQ
  run_pmc scaled $c bench.py --only-scaled-batch
done
