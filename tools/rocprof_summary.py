#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 --kernel-trace --stats run
(rocpd sqlite output) as a text table for profiles/."""
import sqlite3
import sys


def main(db, out=None, title=''):
    con = sqlite3.connect(db)
    rows = list(con.execute('select name, total_calls, total_duration, average, percentage '
                            'from top_kernels'))
    lines = ['# %s' % title, '# source: %s (rocprofv3 --kernel-trace --stats)' % db,
             '%-90s %8s %14s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', '%')]
    for name, calls, tot, avg, pct in rows:
        lines.append('%-90s %8d %14.1f %10.3f %7.2f' % (name[:90], calls, tot / 1e3 if tot > 1e7 else tot,
                                                      avg / 1e3 if avg > 1e5 else avg, pct))
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None,
         sys.argv[3] if len(sys.argv) > 3 else '')
