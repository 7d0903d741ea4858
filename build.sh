#!/usr/bin/env bash
# Build libbsig_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
# An object is rebuilt when the CONTENT of its source, of any csrc header, of include/bsig.h or the
# flags changed (a stamp next to the object; file times do not survive every transport).
set -euo pipefail
cd "$(dirname "$0")"
SRC=bayes_sim_ig_amd/csrc
OUT=${BSIG_OUT_DIR:-bayes_sim_ig_amd/lib}
mkdir -p "$OUT" "$OUT/obj"
HHASH=$(python3 - <<'PY'
h = 0xcbf29ce484222325
for b in open('include/bsig.h', 'rb').read():
    h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
print('0x%016xull' % h)
PY
)
FLAGS="${BSIG_EXTRA_FLAGS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DBSIG_HEADER_HASH=$HHASH"
# the whole-width GEMM tiles (measured slower than the planner's picks; 3.5 minutes of
# compile time each) are built on request only: BSIG_BUILD_WIDE_TILES=1 ./build.sh, then BSIG_GEMM_WIDE_TILE=1
WIDE_TILES=""
if [ "${BSIG_BUILD_WIDE_TILES:-0}" = "1" ]; then FLAGS="$FLAGS -DBSIG_WITH_WIDE_TILES"; WIDE_TILES="gemm_tile_128x288 gemm_tile_288x128"; fi
COMMON=$(cat "$SRC"/*.h include/bsig.h | sha256sum | cut -d' ' -f1)
pids=()
build_one() {  # name, source, extra flags
  local stamp="$OUT/obj/$1.stamp"
  local want="$(echo "$FLAGS $3 $COMMON" | cat - "$2" | sha256sum | cut -d' ' -f1)"
  if [ ! -f "$OUT/obj/$1.o" ] || [ ! -f "$stamp" ] || [ "$(cat "$stamp")" != "$want" ]; then
    ( hipcc $FLAGS $3 -c "$2" -o "$OUT/obj/$1.o" && echo "$want" > "$stamp" ) &
    pids+=($!)
  fi
}
[ -n "$WIDE_TILES" ] || rm -f "$OUT"/obj/gemm_tile_128x288.* "$OUT"/obj/gemm_tile_288x128.*
rm -f "$OUT"/obj/fit_persistent_v1.*      # (retired in round 6: an object of an older build must not be linked)
for f in summarizers gemm_f32 gemm_tile_64 gemm_tile_128 gemm_tile_128x32 gemm_tile_128x64 gemm_tile_128x96 gemm_tile_96x128 gemm_lean_64 gemm_lean_128 gemm_lean_128x32 gemm_lean_128x64 gemm_lean_128x96 gemm_lean_96x128 gemm_wide $WIDE_TILES mdn_head flat_ops estimator fit_persistent fit_persistent_mdnn fit_persistent_mdnn_stream; do
  build_one "$f" "$SRC/$f.hip" ""
done
for f in api comm; do
  build_one "$f" "$SRC/$f.cpp" "-x hip"
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ $rc -eq 0 ] || { echo "build failed" >&2; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbsig_hip.so" "$OUT"/obj/*.o -ldl
echo "built $OUT/libbsig_hip.so"
