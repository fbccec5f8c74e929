"""Timeline of ONE training step from a rocprofv3 (rocpd sqlite) kernel trace of an EAGER bench run:
every dispatch between two consecutive launches of the step's first kernel, with duration and the idle gap before it.

    python profiles/step_timeline.py gpurun_out/<dir>/<x>_results.db [first-kernel-substring] [which-step]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.replace('vargp::', '')[:70]


def timeline(path, first='prologue_kernel', which=-2):
    db = sqlite3.connect(path)
    rows = list(db.execute('select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) from kernels order by start'))
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    a, b = idx[which], idx[which + 1] if which + 1 < 0 or which + 1 < len(idx) else len(rows)
    if which == -1:
        b = len(rows)
    step = rows[a:b]
    t0 = step[0][1]
    busy = 0
    out = []
    prev_end = t0
    for name, s, e, wgs in step:
        out.append((short(name), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, wgs))
        busy += e - s
        prev_end = max(prev_end, e)
    wall = (prev_end - t0) / 1e3
    return out, wall, busy / 1e3


if __name__ == '__main__':
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else 'prologue_kernel'
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    out, wall, busy = timeline(path, first, which)
    print(f'{"kernel":70s} {"t0 us":>9s} {"dur us":>9s} {"gap us":>8s} {"wgs":>7s}')
    for name, t, d, g, w in out:
        print(f'{name:70s} {t:9.1f} {d:9.1f} {g:8.1f} {w:7d}')
    print(f'launches {len(out)}   wall {wall:.1f} us   kernel time {busy:.1f} us')
