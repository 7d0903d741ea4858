#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd sqlite db."""
import sqlite3
import sys


def main(db, out=None, title=''):
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = [n for n in names if n.lower() in ('counters_collection', 'pmc_events', 'counters')]
    lines = ['# %s' % title, '# tables: %s' % ', '.join(n for n in names if 'pmc' in n.lower() or 'counter' in n.lower())]
    if 'counters_collection' in names:
        cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
        lines.append('# counters_collection columns: %s' % ', '.join(cols))
        kcol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else cols[0])
        ccol = 'counter_name' if 'counter_name' in cols else None
        vcol = 'value' if 'value' in cols else ('counter_value' if 'counter_value' in cols else None)
        if ccol and vcol:
            q = ('select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by 1, 2 '
                 'order by 5 desc limit 40' % (kcol, ccol, vcol, vcol))
            lines.append('%-80s %-12s %8s %16s %18s' % ('kernel', 'counter', 'launches', 'avg/launch', 'total'))
            for k, c, n, a, t in cur.execute(q):
                lines.append('%-80s %-12s %8d %16.1f %18.1f' % (str(k)[:80], c, n, a, t))
    text = '\n'.join(lines) + '\n'
    (open(out, 'w') if out else sys.stdout).write(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else '')
