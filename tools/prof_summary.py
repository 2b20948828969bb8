#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/.../*_results.db) into the text files kept under profiles/.

  python tools/prof_summary.py stats <db> <out.txt>            per-kernel calls / total / avg / min / max (us)
  python tools/prof_summary.py pmc   <db> <out.txt>            per-kernel average counter value per launch
"""
import sqlite3
import sys


def short(name):
    import re
    name = name.replace('(anonymous namespace)::', '')
    m = re.search(r'rocprim::\w+::detail::(?:trampoline_kernel<rocprim::\w+::detail::wrapped_)?(\w+?)(?:_config)?<', name)
    if m:
        vals = 'pairs' if re.search(r'unsigned long long, unsigned (int|long long)>', name) else 'keys'
        return 'rocprim::' + m.group(1) + ('/' + vals if 'sort' in m.group(1) else '')
    name = name.split('(')[0]
    return name.replace('pav::', '')


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc").fetchall()
    merged = {}
    for n, c, t, a, mn, mx in rows:                       # template instances of one library kernel are listed together
        k = short(n)
        if k in merged:
            m = merged[k]
            merged[k] = (k, m[1] + c, m[2] + t, (m[2] + t) / (m[1] + c), min(m[4], mn), max(m[5], mx))
        else:
            merged[k] = (k, c, t, a, mn, mx)
    rows = sorted(merged.values(), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows) or 1
    with open(out, 'w') as fh:
        fh.write(f'# rocprofv3 --kernel-trace --stats  ({db})\n')
        fh.write(f'{"kernel":34s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}\n')
        for n, c, t, a, mn, mx in rows:
            fh.write(f'{n:34s} {c:6d} {t / 1e3:12.1f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * t / tot:6.2f}\n')
    print(open(out).read())


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
        "group by kernel_name, counter_name order by avg(value) desc").fetchall()
    with open(out, 'w') as fh:
        fh.write(f'# rocprofv3 --pmc  ({db}); FETCH_SIZE / WRITE_SIZE are in KiB per dispatch\n')
        fh.write(f'{"kernel":34s} {"counter":12s} {"launches":>8s} {"avg":>16s} {"min":>16s} {"max":>16s}\n')
        for n, c, k, a, mn, mx in rows:
            fh.write(f'{short(n):34s} {c:12s} {k:8d} {a:16.1f} {mn:16.1f} {mx:16.1f}\n')
    print(open(out).read())


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
