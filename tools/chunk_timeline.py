#!/usr/bin/env python3
"""Where does one run_training chunk spend its time?  Reads a rocprofv3
--kernel-trace rocpd database and prints, for the steady-state chunks (windows
between consecutive fit_begin_kernel launches), the average busy time per
kernel, the idle time between kernels and the window length."""
import collections
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = list(con.execute('select %s, start, end from kernels order by start' % name_col))
    begins = [i for i, r in enumerate(rows) if 'fit_begin_kernel' in r[0]]
    wins = [(begins[i], begins[i + 1]) for i in range(len(begins) - 1)]
    wins = [w for w in wins if rows[w[1]][1] - rows[w[0]][1] < 5e7]   # drop gaps between phases
    # the most common window shape = the bench's main loop
    shape = collections.Counter(w[1] - w[0] for w in wins).most_common(1)[0][0]
    wins = [w for w in wins if w[1] - w[0] == shape]
    busy = collections.OrderedDict()
    calls = collections.Counter()
    idle = 0.0
    total = 0.0
    gaps = collections.OrderedDict()
    for a, b in wins:
        total += rows[b][1] - rows[a][1]
        last_end = rows[a][1]
        for name, s, e in rows[a:b]:
            short = name.split('(')[0].replace('void ', '')[:70]
            busy[short] = busy.get(short, 0.0) + (e - s)
            calls[short] += 1
            idle += max(0.0, s - last_end)
            gaps['gap before ' + short] = gaps.get('gap before ' + short, 0.0) + max(0.0, s - last_end)
            last_end = max(last_end, e)
        idle += max(0.0, rows[b][1] - last_end)
        gaps['gap before the next fit_begin_kernel'] = gaps.get('gap before the next fit_begin_kernel', 0.0) + max(0.0, rows[b][1] - last_end)
    n = float(len(wins))
    lines = ['# %d chunks of %d kernel launches; averages per chunk' % (len(wins), shape),
             '%-72s %7s %10s' % ('kernel', 'calls', 'busy_us')]
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        lines.append('%-72s %7.1f %10.1f' % (k, calls[k] / n, v / n / 1e3))
    lines.append('%-72s %7s %10.1f' % ('idle between kernels', '', idle / n / 1e3))
    for k, v in gaps.items():
        lines.append('%-72s %7s %10.1f' % ('  ' + k[:68], '', v / n / 1e3))
    lines.append('%-72s %7s %10.1f' % ('chunk (begin to begin)', '', total / n / 1e3))
    if wins:
        g = sorted(sum(max(0.0, rows[i + 1][1] - max(r[2] for r in rows[a:i + 1])) for i in range(a, b)) / 1e3 for a, b in wins)
        lines.append('# idle per chunk over the %d chunks: min %.1f, median %.1f, max %.1f us; sorted: %s' % (
            len(g), g[0], g[len(g) // 2], g[-1], ' '.join('%.0f' % x for x in g)))
    if wins:                        # one window in full: offsets from its fit_begin_kernel, us
        a, b = wins[len(wins) // 2]
        lines.append('# one chunk: start offset / duration (us)')
        for name, s_, e in rows[a:b + 1]:
            lines.append('#   %9.1f %9.1f  %s' % ((s_ - rows[a][1]) / 1e3, (e - s_) / 1e3, name.split('(')[0].replace('void ', '')[:70]))
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    sys.stdout.write(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
