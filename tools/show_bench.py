#!/usr/bin/env python3
"""Short view of a bench.py JSON line:  python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
p = r['path']
print(f"value {d['value']} {d['unit']}  ms_per_step {d['ms_per_step']}  lanes {d['config']['lanes_per_gpu']}  repeats {d.get('repeats', {}).get('ms_per_step_all')}")
print(f"sum_kernel_ms {p['sum_kernel_ms_per_step']}  single lane {p.get('single_lane')}")
print(f"dominant {r['kernel']} {r['avg_kernel_ms']} ms frac {r['frac']}")
k = r['kernels_ms']
print('kernels:', ' '.join(f'{n}={v}' for n, v in sorted(k.items(), key=lambda kv: -kv[1])))
if d.get('cigar_only'):
    c = d['cigar_only']
    print(f"cigar_only {c['value']} Gbp/s {c['ms_per_step']} ms")
if d.get('inv_scan'):
    i = d['inv_scan']
    print('inv_scan device ms', i['device_ms_per_step'], 'flag device ms', i['flagging']['device_ms_per_step'], 'calls', i['calls'], i['near_tie_guard'])
