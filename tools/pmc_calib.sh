#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE of kernels with a known byte count (tools/micro/pmc_calib.hip)
set -uo pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/pmc_calib $R/tools/micro/pmc_calib.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/cal_$c -o c -- /tmp/pmc_calib > /dev/null 2>&1
  db=$(find /tmp/cal_$c -name "*.db" | head -1)
  python3 $R/tools/pmc_dump.py "$db" $OUT/${TAG}_pmc_calib_$c.txt "rocprofv3 --pmc $c --kernel-trace -- pmc_calib (1 GiB = 1048576 KB per kernel)"
done
