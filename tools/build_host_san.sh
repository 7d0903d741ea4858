#!/usr/bin/env bash
# Build of libbsig_hip's sources with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST side
# only (-fno-gpu-sanitize: GPU sanitizers are not available on the MI355X pool; device code is
# compiled as usual, at -O1), linked with tests/host/host_san_main.cpp into
# build/host_san/host_san_test: argument checks, parameter layouts, GEMM planners, persistent-kernel
# geometry, plan binding and the external-exchange communicator run under the sanitizers (no GPU
# needed: nothing is launched).  Objects are rebuilt when their sources' content changes.
set -euo pipefail
cd "$(dirname "$0")/.."
SRC=bayes_sim_ig_amd/csrc
OUT=build/host_san
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O1 -std=c++17 -fPIC -DBSIG_HOST_SAN_BUILD -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -fno-omit-frame-pointer -w"
COMMON=$(cat "$SRC"/*.h include/bsig.h | sha256sum | cut -d' ' -f1)
pids=()
build_one() {  # name, source
  local stamp="$OUT/$1.stamp"
  local want="$(echo "$FLAGS $COMMON" | cat - "$2" | sha256sum | cut -d' ' -f1)"
  if [ ! -f "$OUT/$1.o" ] || [ ! -f "$stamp" ] || [ "$(cat "$stamp")" != "$want" ]; then
    ( hipcc $FLAGS -x hip -c "$2" -o "$OUT/$1.o" && echo "$want" > "$stamp" ) &
    pids+=($!)
  fi
}
for f in "$SRC"/*.hip "$SRC"/*.cpp; do
  b=$(basename "$f"); build_one "${b%.*}" "$f"
done
build_one host_san_main tests/host/host_san_main.cpp
# objects of sources that no longer exist (a retired file's object would be linked by the glob below)
for o in "$OUT"/*.o; do
  b=$(basename "$o" .o)
  [ "$b" = host_san_main ] || [ -f "$SRC/$b.hip" ] || [ -f "$SRC/$b.cpp" ] || rm -f "$OUT/$b.o" "$OUT/$b.stamp"
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ $rc -eq 0 ] || { echo "host sanitizer build failed" >&2; exit 1; }
hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -o "$OUT/host_san_test" "$OUT"/*.o -ldl
echo "built $OUT/host_san_test"
