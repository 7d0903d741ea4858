#!/usr/bin/env bash
# N runs of the 1-rank data-parallel bench with the resident exchange; per run: the resident calls, the
# indices of the calls whose time-out bit came up, the probe line.  Usage: tools/micro/xr_soak.sh [runs] [bench args...]
N=${1:-6}; shift || true
for i in $(seq 1 $N); do
  BSIG_DP_XR_TRACE=1 BENCH_FORCE_DP=1 python3 bench.py --steps 3 --warmup 1 --no-per-config --no-largest-size --no-cpu-baseline --no-scaled-batch "$@" 2> /tmp/xr_soak_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i:', round(d['value']), 'pairs/s, resident calls', d['config']['rank_resident_calls'], end='; ')"
  grep "time-out bits" /tmp/xr_soak_err.txt | tail -1 | sed "s/.*time-out bits so far://" | python3 -c "
import sys
v = sys.stdin.read().split(); print('time-out bit at calls', [i for i, x in enumerate(v) if x not in ('0', '-1')], 'of', len(v))"
  grep "comm_xr:" /tmp/xr_soak_err.txt | head -2 | tr '\n' ' '; echo
done
