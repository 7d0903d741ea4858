#!/usr/bin/env bash
# How many workgroups may a kernel on the exchange stream have while a resident launch holds the chip?
# The stand-in for a peer's contribution (BSIG_DEBUG_GRAD_EXCHANGE_SCALE: a kernel that scales the
# gradient buffer between the all-reduce and the write) with grids of 1 .. 256 workgroups of 1024 threads;
# per grid: the time-out bits of the resident calls of a 4-chunk fit (0: ran, 2: timed out) and pairs/s.
for g in 1 4 8 9 12 16 32 64 256; do
  BSIG_DEBUG_GRAD_EXCHANGE_GRID=$g BSIG_DEBUG_GRAD_EXCHANGE_SCALE=3 BSIG_DP_XR_TRACE=1 BENCH_FORCE_DP=1 \
    python3 bench.py --pairs 4000 --steps 1 --warmup 1 --no-per-config --no-largest-size --no-cpu-baseline --no-scaled-batch 2> /tmp/xg_err.txt \
    | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid %4d: %7d pairs/s, resident calls %d;' % ($g, d['value'], d['config']['rank_resident_calls']), end=' ')"
  grep "time-out bits" /tmp/xg_err.txt | tail -1 | sed "s/.*time-out bits so far:/time-out bits/"
done
