#!/usr/bin/env python3
"""What sits between two kernels of a chunk?  Reads a rocprofv3 rocpd database recorded with
--kernel-trace --memory-copy-trace --hip-trace and lists, for the steady-state windows between
fit_begin_kernel and the persistent update kernel, the memory copies and HIP API calls in between
(average start offset and duration, us)."""
import collections
import sqlite3
import sys


def cols(con, t):
    return [r[1] for r in con.execute("pragma table_info('%s')" % t)]


def main(db):
    con = sqlite3.connect(db)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kc = cols(con, 'kernels')
    name_col = 'name' if 'name' in kc else 'kernel_name'
    ks = list(con.execute('select %s, start, end from kernels order by start' % name_col))
    wins = []
    for i, (n, s, e) in enumerate(ks[:-1]):
        if 'fit_begin_kernel' in n and 'updates_kernel' in ks[i + 1][0]:
            wins.append((e, ks[i + 1][1]))
    wins = wins[len(wins) // 4:]
    print('%d windows fit_begin_kernel -> update kernel, mean gap %.1f us' % (len(wins), sum(b - a for a, b in wins) / len(wins) / 1e3))
    for t in ('memory_copies', 'memory_allocations', 'regions', 'regions_and_samples'):
        if t not in tables:
            continue
        c = cols(con, t)
        ncol = 'name' if 'name' in c else None
        if not ncol or 'start' not in c:
            continue
        rows = list(con.execute('select %s, start, end from %s order by start' % (ncol, t)))
        acc = collections.OrderedDict()
        for a, b in wins:
            for n, s, e in rows:
                if s >= a - 200000 and s <= b:
                    k = n[:60]
                    d = acc.setdefault(k, [0, 0.0, 0.0])
                    d[0] += 1; d[1] += (s - a) / 1e3; d[2] += (e - s) / 1e3
        print('-- %s (calls per window, mean start after fit_begin_kernel ended, mean duration)' % t)
        for k, (n, so, du) in acc.items():
            print('   %-60s %5.2f  %8.1f us  %8.1f us' % (k, n / len(wins), so / n, du / n))


if __name__ == '__main__':
    main(sys.argv[1])
