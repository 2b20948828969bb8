#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/.../*_results.db) into the text files kept under profiles/.

  python tools/prof_summary.py stats <db> <out.txt>            per-kernel calls / total / avg / min / max (us)
  python tools/prof_summary.py pmc   <db> <out.txt>            per-kernel average counter value per launch
  python tools/prof_summary.py pmcjson <fetch.db> <write.db> <bench.json> <out.json>   per-kernel FETCH_SIZE / WRITE_SIZE averages
                                                               (KiB per dispatch) keyed to the workload of the bench line
  python tools/prof_summary.py sq    <db> <out.txt>            per-kernel SQ counters of one --pmc pass as fractions of the wave cycles
                                                               (issuing / parked / issue-stalled) and of the LDS-array cycles
  python tools/prof_summary.py timeline <db> <out.txt> [ms]    kernels (and copies) of the last [ms] of the trace in start order,
                                                               with the idle gap in front of each (us)
  python tools/prof_summary.py busy <db> <out.txt> <t0_ms> <t1_ms>   device occupancy of the window [t0, t1) ms after the first
                                                               kernel: wall time, time with at least one kernel running, sum of
                                                               kernel durations, per-kernel totals; windows are found with
                                                               `busy <db> <out> scan <width_ms> [kernel]` (the window of that width
                                                               with the most launches of `kernel`, default k_kmer_lds: the two-lane region)
"""
import sqlite3
import sys


def provenance():
    """The source the profiled library was built from (pav_amd._lib.source_fingerprint): bench.py quotes a committed counter
    only when the file that defines the kernel is unchanged since."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pav_amd import _lib
    fp = _lib.source_fingerprint()
    fp.pop('kernel_file', None)
    return fp


def short(name):
    import re
    name = name.replace('(anonymous namespace)::', '')
    m = re.search(r'rocprim::\w+::detail::(?:trampoline_kernel<rocprim::\w+::detail::wrapped_)?(\w+?)(?:_config)?<', name)
    if m:
        vals = 'pairs' if re.search(r'unsigned long long, unsigned (int|long long)>', name) else 'keys'
        return 'rocprim::' + m.group(1) + ('/' + vals if 'sort' in m.group(1) else '')
    name = name.split('(')[0].replace('pav::', '')
    # the two instances of the walk carry the names the library's own profile (pav_prof_*) and bench.py use
    return {'void walk_emit<1>': 'walk_indel', 'void walk_emit<2>': 'walk_snv'}.get(name, name)


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


def pmcjson(fetch_db, write_db, bench_json, out):
    import json

    def per_kernel(db, counter):
        cur = sqlite3.connect(db).cursor()
        acc = {}
        for n, v in cur.execute('select kernel_name, value from counters_collection where counter_name = ?', (counter,)):
            a = acc.setdefault(short(n), [0.0, 0])
            a[0] += v
            a[1] += 1
        return {k: round(t / c, 1) for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0] / kv[1][1])}
    with open(bench_json) as fh:
        line = json.loads(fh.read().strip().splitlines()[-1])
    doc = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around bench.py, MI355X; values are KiB per '
                     'dispatch, averaged over the dispatches of a kernel (tools/prof_summary.py pmcjson)',
           'workload': {'seed': line['config']['seed'], 'scale': line['config']['scale'],
                        'aligned_bp_per_gpu': line['config']['aligned_bp_per_gpu'], 'name': line['config']['workload']},
           'gfx950_note': 'FETCH_SIZE counts 128 B requests as 64 B for 16 B/lane streaming reads: x2 for pack_kernel (calibrated '
                          'against the 3.08 GB ASCII arena) and verify_kernel (16 B/lane windows of the 2-bit planes); '
                          'other kernels are reported raw',
           'fetch_kib': per_kernel(fetch_db, 'FETCH_SIZE'), 'write_kib': per_kernel(write_db, 'WRITE_SIZE')}
    doc['provenance'] = provenance()
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(open(out).read())


def sq(db, out):
    cur = sqlite3.connect(db).cursor()
    per = {}
    for n, c, k, a in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                  "group by kernel_name, counter_name"):
        per.setdefault(short(n), {})[c] = (k, a)
    cols = ['SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS']
    rows = sorted(per.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', (0, 0))[1])
    with open(out, 'w') as fh:
        fh.write(f'# rocprofv3 --pmc (one SQ pass, {db}); averages per launch.  wave_cyc = SQ_WAVE_CYCLES (quad-cycles summed over\n'
                 '# the waves); issue / parked / stall / stall_lds = SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY (s_waitcnt, barrier) /\n'
                 '# SQ_WAIT_INST_ANY / SQ_WAIT_INST_LDS as fractions of wave_cyc; lds_cyc = SQ_LDS_IDX_ACTIVE (LDS-array cycles),\n'
                 '# conflict = SQ_LDS_BANK_CONFLICT / lds_cyc, unaligned = SQ_LDS_UNALIGNED_STALL / lds_cyc.\n')
        fh.write(f'{"kernel":34s} {"launches":>8s} {"wave_cyc":>14s} {"issue":>7s} {"parked":>7s} {"stall":>7s} {"stall_lds":>9s} '
                 f'{"lds_cyc":>14s} {"conflict":>8s} {"unaligned":>9s}\n')
        for n, c in rows:
            k, wc = c.get('SQ_WAVE_CYCLES', (0, 0.0))
            lds = c.get('SQ_LDS_IDX_ACTIVE', (0, 0.0))[1]
            fr = ['%7.3f' % (c.get(x, (0, 0.0))[1] / wc) if wc else '%7s' % '-' for x in cols]
            lf = ['%8.3f' % (c.get(x, (0, 0.0))[1] / lds) if lds else '%8s' % '-' for x in ('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_UNALIGNED_STALL')]
            fh.write(f'{n:34s} {k:8d} {wc:14.0f} {fr[0]} {fr[1]} {fr[2]} {fr[3]:>9s} {lds:14.0f} {lf[0]} {lf[1]:>9s}\n')
    print(open(out).read())


def ldsjson(db, out, lanes='6'):
    """LDS counters of one --pmc pass (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_INSTS_LDS, SQ_LDS_ADDR_CONFLICT) per kernel,
    averaged per launch, as the JSON bench.py reads for its LDS roofline (roofline.timed_region.*.lds)."""
    import json
    cur = sqlite3.connect(db).cursor()
    per = {}
    for n, c, k, a in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                  "group by kernel_name, counter_name"):
        per.setdefault(short(n), {})[c] = (k, a)
    doc = {'source': f'rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES around bench.py --lanes {lanes} '
                     '(tools/scripts/profile_round.sh); averages per launch, summed over the CUs',
           'lanes': int(lanes), 'kernels': {}}
    for k, c in sorted(per.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', (0, 0.0))[1]):
        act = c.get('SQ_LDS_IDX_ACTIVE', (0, 0.0))[1]
        if act <= 0:
            continue
        doc['kernels'][k] = {'launches': c['SQ_LDS_IDX_ACTIVE'][0], 'lds_idx_active': round(act, 1),
                             'bank_conflict_share': round(c.get('SQ_LDS_BANK_CONFLICT', (0, 0.0))[1] / act, 4),
                             'insts_lds': round(c.get('SQ_INSTS_LDS', (0, 0.0))[1], 1),
                             'wave_cycles': round(c.get('SQ_WAVE_CYCLES', (0, 0.0))[1], 1)}
    doc['provenance'] = provenance()
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(open(out).read())


def timeline(db, out, last_ms=30.0):
    cur = sqlite3.connect(db).cursor()
    ev = [(s, e, short(n)) for n, s, e in cur.execute('select name, start, end from kernels')]
    try:
        ev += [(s, e, 'copy:%s:%d' % (n, b)) for n, s, e, b in cur.execute('select name, start, end, size from memory_copies')]
    except sqlite3.Error:
        pass
    ev.sort()
    t_end = max(e for _, e, _ in ev)
    ev = [x for x in ev if x[0] >= t_end - last_ms * 1e6]
    t0, busy_end = ev[0][0], ev[0][0]
    with open(out, 'w') as fh:
        fh.write(f'# kernels of the last {last_ms} ms of {db}: start offset, duration, idle gap before (us)\n')
        for s, e, n in ev:
            gap = max(0, s - busy_end)
            fh.write(f'{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap / 1e3:8.1f}  {n}\n')
            busy_end = max(busy_end, e)
    print(open(out).read())


def busy(db, out, a, b, mark='k_kmer_lds'):
    cur = sqlite3.connect(db).cursor()
    ev = sorted((s, e, short(n)) for n, s, e in cur.execute('select name, start, end from kernels'))
    t_first = ev[0][0]
    if a == 'scan':                                         # the window of width b with the most launches of walk_snv: steady state
        width = float(b) * 1e6
        marks = [s for s, _, n in ev if n == mark]
        best, lo = (0, marks[0]), 0
        for hi in range(len(marks)):
            while marks[hi] - marks[lo] > width:
                lo += 1
            if hi - lo + 1 > best[0]:
                best = (hi - lo + 1, marks[lo])
        w0, w1 = best[1], best[1] + width
    else:
        w0, w1 = t_first + float(a) * 1e6, t_first + float(b) * 1e6
    sel = [(max(s, w0), min(e, w1), n) for s, e, n in ev if e > w0 and s < w1]
    union, end = 0.0, w0
    for s, e, _ in sel:
        if e > end:
            union += e - max(s, end)
            end = e
    per = {}
    for s, e, n in sel:
        t = per.setdefault(n, [0, 0.0])
        t[0] += 1
        t[1] += e - s
    total = sum(v[1] for v in per.values())
    packs = per.get('tok_tiles', [0, 0.0])[0]              # one per pav_cigar_call = one per pass
    with open(out, 'w') as fh:
        fh.write(f'# device occupancy, window of {(w1 - w0) / 1e6:.2f} ms starting {(w0 - t_first) / 1e6:.2f} ms after the first kernel ({db})\n')
        fh.write(f'# passes in the window (tok_tiles launches: one per CIGAR-call): {packs}  ->  {(w1 - w0) / 1e6 / max(1, packs):.3f} ms of wall time per pass\n')
        fh.write(f'wall_ms {(w1 - w0) / 1e6:.3f}   busy_ms (>= 1 kernel running) {union / 1e6:.3f} = {union / (w1 - w0):.3f} of wall   '
                 f'sum_of_kernel_ms {total / 1e6:.3f} = {total / (w1 - w0):.3f} of wall (kernels of the lanes overlap)\n')
        fh.write(f'{"kernel":34s} {"launches":>8s} {"total_ms":>10s} {"per_pass_ms":>12s}\n')
        for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fh.write(f'{n:34s} {c:8d} {t / 1e6:10.3f} {t / 1e6 / max(1, packs):12.4f}\n')
    print(open(out).read())


if __name__ == '__main__':
    if sys.argv[1] == 'busy':
        busy(*sys.argv[2:7])
    elif sys.argv[1] == 'pmcjson':
        pmcjson(*sys.argv[2:6])
    elif sys.argv[1] == 'ldsjson':
        ldsjson(*sys.argv[2:5])
    elif sys.argv[1] == 'timeline':
        timeline(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 30.0)
    else:
        {'stats': stats, 'pmc': pmc, 'sq': sq}[sys.argv[1]](sys.argv[2], sys.argv[3])
