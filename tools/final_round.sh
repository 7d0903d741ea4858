#!/usr/bin/env bash
# Everything the round's DESIGN.md / README.md numbers are taken from, in one call on a GPU box:
#   tools/final_round.sh <tag> <commit>  -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/;
#   the bench line: see the end of this file)
# <commit>: the commit of the code (the box has no .git); stamped into every file as "# head: <commit>".
set -uo pipefail
TAG=${1:-rXX}
HEADSHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
CSRC=$(python3 "$R/tools/csrc_hash.py")
mkdir -p "$OUT"
cd "$R"
python3 tools/yaml_configs_bench.py > "$OUT/${TAG}_yaml_configs.txt" 2>&1
for c in anymal_yaml shadow_more; do
  n=${c%_yaml}
  WIDE_DETAIL=1 python3 tools/persist_stream_prof.py $c 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_${n}_update_timeline.txt"
done
PER_UPDATE=1 LAZY=1 python3 tools/persist_mdnn_prof.py cfg3 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_cfg3_update_timeline.txt"
python3 tools/persist_prof.py 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_cfg5_update_timeline.txt"
python3 tools/summarizer_bench.py 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_summarizer_bench.txt"
[ -x tools/micro/bin/fill_bench ] && tools/micro/bin/fill_bench > "$OUT/${TAG}_fill_bench.txt" 2>&1
python3 tools/gemm_scaled_sweep.py heldout 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_gemm_heldout_sweep.txt"
BSIG_PROF_T0=50 python3 tools/persist_prof.py 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_cfg5_update_timeline_warm.txt"
# a data-parallel rank on a 1-rank RCCL group: resident across the exchange (the default there) and one launch per update
DPQ="--steps 3 --warmup 1 --no-per-config --no-largest-size --no-cpu-baseline --no-scaled-batch"
{ echo "BSIG_DP_RESIDENT=1 BENCH_FORCE_DP=1 python bench.py $DPQ  (rank resident across the exchange: opt-in since round 6)"
  BSIG_DP_RESIDENT=1 BSIG_DP_XR_TRACE=1 BENCH_FORCE_DP=1 python3 bench.py $DPQ 2> "$OUT/${TAG}_dp_resident_stderr.log"
  grep "comm_xr:" "$OUT/${TAG}_dp_resident_stderr.log" | head -2
  grep "time-out bits" "$OUT/${TAG}_dp_resident_stderr.log" | tail -1 | sed "s/.*time-out bits so far://" | tr ' ' '\n' | grep -v "^$" | sort | uniq -c | sed "s/^/  time-out bit of the resident calls (count value): /"
  echo "BSIG_DP_RESIDENT=0 BENCH_FORCE_DP=1 python bench.py $DPQ  (one launch + all-reduce per update)"
  BSIG_DP_RESIDENT=0 BENCH_FORCE_DP=1 python3 bench.py $DPQ 2>/dev/null
  echo "BSIG_DP_RESIDENT=1 BENCH_FORCE_DP=1 python bench.py --config cfg3 --pairs 20000 $DPQ  (MDNN kernel, resident across the exchange)"
  BSIG_DP_RESIDENT=1 BENCH_FORCE_DP=1 python3 bench.py --config cfg3 --pairs 20000 $DPQ 2>/dev/null
  echo "BSIG_DP_RESIDENT=0 BENCH_FORCE_DP=1 python bench.py --config cfg3 --pairs 20000 $DPQ  (MDNN kernel: one launch + all-reduce per update)"
  BSIG_DP_RESIDENT=0 BENCH_FORCE_DP=1 python3 bench.py --config cfg3 --pairs 20000 $DPQ 2>/dev/null
} > "$OUT/${TAG}_dp_1rank.txt"
rm -f "$OUT/${TAG}_dp_resident_stderr.log"
BSIG_PROF_T0=50 python3 tools/persist_xr_prof.py 2>/dev/null | grep -v "^stamped\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > "$OUT/${TAG}_cfg5_dp_resident_timeline.txt"
python3 tools/persist_xr_prof.py cfg3 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > "$OUT/${TAG}_cfg3_dp_resident_timeline.txt"
python3 tools/persist_prof.py cfg2 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_cfg2_update_timeline.txt"
for f in bench_line.json yaml_configs.txt anymal_update_timeline.txt shadow_more_update_timeline.txt cfg3_update_timeline.txt \
         cfg5_update_timeline.txt cfg5_update_timeline_warm.txt cfg2_update_timeline.txt dp_1rank.txt cfg5_dp_resident_timeline.txt cfg3_dp_resident_timeline.txt summarizer_bench.txt fill_bench.txt gemm_heldout_sweep.txt; do
  [ -f "$OUT/${TAG}_$f" ] && [ "${f##*.}" = txt ] && sed -i "1i # head: $HEADSHA\n# csrc: $CSRC" "$OUT/${TAG}_$f"
done
bash tools/round_profiles.sh "$TAG" "$HEADSHA"
# The bench line is NOT taken here: bench.pmc_traffic wants this pass's counters in profiles/ first (and
# a line taken right behind the counter passes once read a scaled-batch fit at 0.49 of the peak).  Copy the files into profiles/, then, in a call of its own:
#   python bench.py > gpurun_out/<tag>_bench_line.json
