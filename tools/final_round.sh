#!/usr/bin/env bash
# Everything the round's DESIGN.md / README.md numbers are taken from, in one call on a GPU box:
#   tools/final_round.sh <tag> <commit>  -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# <commit>: the commit of the code (the box has no .git); stamped into every file as "# head: <commit>".
set -uo pipefail
TAG=${1:-rXX}
HEADSHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
CSRC=$(python3 "$R/tools/csrc_hash.py")
mkdir -p "$OUT"
cd "$R"
python3 bench.py > "$OUT/${TAG}_bench_line.json" 2> "$OUT/${TAG}_bench_stderr.log"
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
python3 tools/persist_prof.py cfg2 2>&1 | grep -v "amdgpu.ids" > "$OUT/${TAG}_cfg2_update_timeline.txt"
for f in bench_line.json yaml_configs.txt anymal_update_timeline.txt shadow_more_update_timeline.txt cfg3_update_timeline.txt \
         cfg5_update_timeline.txt cfg5_update_timeline_warm.txt cfg2_update_timeline.txt summarizer_bench.txt fill_bench.txt gemm_heldout_sweep.txt; do
  [ -f "$OUT/${TAG}_$f" ] && [ "${f##*.}" = txt ] && sed -i "1i # head: $HEADSHA\n# csrc: $CSRC" "$OUT/${TAG}_$f"
done
bash tools/round_profiles.sh "$TAG" "$HEADSHA"
