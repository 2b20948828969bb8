#!/usr/bin/env python3
"""Compare tuning builds of the library on the bench haplotype (GPU box):  python tools/bench_variants.py [--steps 30] name ...
Every pav_amd/lib/variants/libpav_amd_<name>.so (tools/build_variant.sh) - and `base`, the product build - runs
tools/prof_step.py --plain --kernels in a process of its own; prints one line per build: one-lane ms per step, the sum of the kernel
times, and the kernels that differ from `base` by more than 3 %."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(name, steps):
    env = dict(os.environ)
    if name != 'base':
        env['PAV_AMD_LIB'] = os.path.join(ROOT, 'pav_amd', 'lib', 'variants', f'libpav_amd_{name}.so')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'prof_step.py'), '--no-build', '--plain', '--kernels', '--steps', str(steps)],
                         env=env, capture_output=True, text=True)
    ms, kern = None, None
    for ln in out.stdout.splitlines():
        if ln.startswith('ms per step'):
            ms = float(ln.split(':')[1])
        if ln.startswith('KERNELS '):
            kern = json.loads(ln[8:])
    if kern is None:
        print(name, 'FAILED', out.stderr[-2000:])
    return ms, kern


def main():
    steps = 30
    names = [a for a in sys.argv[1:] if not a.startswith('--')]
    if '--steps' in sys.argv:
        steps = int(sys.argv[sys.argv.index('--steps') + 1])
        names = [n for n in names if n != str(steps)]
    base_ms, base = run('base', steps)
    print(f'base      one lane {base_ms:.3f} ms/step  kernels {base["sum_ms_per_step"]:.3f} ms/step', flush=True)
    for name in names:
        ms, k = run(name, steps)
        if k is None:
            continue
        diff = {n: (base['ms_per_step'].get(n), v) for n, v in k['ms_per_step'].items()
                if base['ms_per_step'].get(n) is None or abs(v - base['ms_per_step'][n]) > 0.03 * max(v, base['ms_per_step'][n], 1e-9)}
        watch = os.environ.get('PAV_VARIANT_WATCH')
        if watch:
            print(f'{name:9s} {watch}: ' + ', '.join(f'{w} {k["ms_per_step"].get(w)} ms in {k.get("launches_per_step", {}).get(w)} launches' for w in watch.split(',')), flush=True)
        print(f'{name:9s} one lane {ms:.3f} ms/step  kernels {k["sum_ms_per_step"]:.3f} ms/step  ' +
              ' '.join(f'{n}: {a} -> {b}' for n, (a, b) in sorted(diff.items(), key=lambda kv: -abs((kv[1][0] or 0) - kv[1][1]))[:10]), flush=True)


if __name__ == '__main__':
    main()
