#!/usr/bin/env bash
# Build libbsig_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
SRC=bayes_sim_ig_amd/csrc
OUT=bayes_sim_ig_amd/lib
mkdir -p "$OUT" "$OUT/obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in summarizers gemm_f32 gemm_tile_64 gemm_tile_128 gemm_tile_128x32 gemm_tile_128x64 gemm_tile_128x96 gemm_tile_96x128 mdn_head flat_ops estimator fit_persistent fit_persistent_mdnn; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/obj/$f.o" ] || [ -n "$(find "$SRC" -name '*.h' -newer "$OUT/obj/$f.o")" ] || [ include/bsig.h -nt "$OUT/obj/$f.o" ]; then
    hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/obj/$f.o" &
    pids+=($!)
  fi
done
for f in api comm; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$SRC/$f.cpp" -nt "$OUT/obj/$f.o" ] || [ include/bsig.h -nt "$OUT/obj/$f.o" ] || [ "$SRC/common.h" -nt "$OUT/obj/$f.o" ]; then
    hipcc $FLAGS -x hip -c "$SRC/$f.cpp" -o "$OUT/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbsig_hip.so" "$OUT"/obj/*.o -ldl
echo "built $OUT/libbsig_hip.so"
