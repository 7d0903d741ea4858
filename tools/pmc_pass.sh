#!/usr/bin/env bash
# One rocprofv3 PMC pass per counter (separate runs, --kernel-trace only) over a command;
# per-kernel averages into gpurun_out/<tag>_pmc_<counter>.txt.
# Usage: tools/pmc_pass.sh <tag> "<counter> <counter> ..." <program.py> [args...]
set -uo pipefail
TAG=$1; CTRS=$2; PROG=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in $CTRS; do
  rm -rf /tmp/c_${TAG}_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/c_${TAG}_$c -o c -- python3 $R/$PROG "$@" > /dev/null 2>&1
  db=$(find /tmp/c_${TAG}_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python3 $R/tools/pmc_dump.py "$db" $R/gpurun_out/${TAG}_pmc_$c.txt "rocprofv3 --pmc $c --kernel-trace -- python $PROG $*"
  else
    echo "no database for $c" > $R/gpurun_out/${TAG}_pmc_$c.txt
  fi
done
