#!/usr/bin/env python3
"""One-lane step time of tuning builds against the product build, interleaved and repeated (a single run moves by +-5 % with the
host):  python tools/ab_step.py [--repeats 5] [--steps 60] name ...   -> min / median ms per step per build."""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(name, steps):
    env = dict(os.environ)
    if name.startswith('env:'):                                   # the product build with an environment switch, e.g. env:PAV_LIFT_COPY=1
        k, v = name[4:].split('=', 1)
        env[k] = v
    elif name != 'base':
        env['PAV_AMD_LIB'] = os.path.join(ROOT, 'pav_amd', 'lib', 'variants', f'libpav_amd_{name}.so')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'prof_step.py'), '--no-build', '--plain', '--steps', str(steps)], env=env, capture_output=True, text=True)
    for ln in out.stdout.splitlines():
        if ln.startswith('ms per step'):
            return float(ln.split(':')[1])
    raise SystemExit(out.stderr[-2000:])


def main():
    args = sys.argv[1:]
    repeats, steps = 5, 60
    if '--repeats' in args:
        i = args.index('--repeats'); repeats = int(args[i + 1]); del args[i:i + 2]
    if '--steps' in args:
        i = args.index('--steps'); steps = int(args[i + 1]); del args[i:i + 2]
    names = ['base'] + args
    res = {n: [] for n in names}
    for _ in range(repeats):
        for n in names:
            res[n].append(run(n, steps))
    for n in names:
        print(f'{n:10s} min {min(res[n]):.3f}  median {statistics.median(res[n]):.3f}  all {" ".join("%.3f" % v for v in res[n])}', flush=True)


if __name__ == '__main__':
    main()
