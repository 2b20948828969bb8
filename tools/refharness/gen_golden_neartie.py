#!/usr/bin/env python3
"""
Golden vectors for the near-tie guard of the density scan (include/pav_amd.h "Near-tie guard"): inputs *constructed* so that
each float decision of scripts/density.py is taken by a hair, and the table the *reference itself* (scripts/density.py ->
scipy gaussian_kde, run unmodified as a subprocess exactly as pavlib/inv.py:249-266 runs it) produces for them.
Build container only.

Output (committed): tests/golden/den_neartie/<case>.npz with
    ref, tig            uint8 ASCII sequences (the whole records are the regions)
    params              JSON: k, staterunsmooth, staterundelta, what the case is about, the row(s) concerned
    INDEX, STATE_MER, STATE, KERN_FWD, KERN_FWDREV, KERN_REV     the reference's table (exact float64)

Cases
    argmax_search   FWD | REV | FWD blocks of m1, m2, m3 rows; (m1, m2) searched (closed form, then confirmed with the
                    scalar oracle) so that at one table row KERN_FWD and KERN_REV differ by 1e-12 .. 1e-9 relative:
                    below what closed-form run sums resolve, well above what scipy's order resolves.
    argmax_mirror   [FWDREV a][FWD m][FWDREV 21][REV m][FWDREV a]: mirror symmetric about the centre row, where KERN_FWD ==
                    KERN_REV mathematically; the reference's STATE there is decided by rounding (reported as unresolved).
    delta_above / delta_below
                    --staterundelta set 1e-10 (relative) above / below the reference's own max |delta KERN| of one quiet
                    window: the window is interpolated in one table and evaluated in the other.
"""

import base64
import json
import os
import pickle
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refenv  # noqa: E402

refenv.setup()
from oracle import oracle  # noqa: E402  (search aid only: the committed tables come from the reference)
from scipy.special import erf, erfc  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'den_neartie')
ACGT = np.frombuffer(b'ACGT', dtype=np.uint8)
K = 31


def revcomp(a):
    comp = np.zeros(256, dtype=np.uint8)
    comp[ACGT] = ACGT[::-1]
    return comp[a][::-1].copy()


def write_fa(path, name, seq):
    with open(path, 'w') as fh:
        fh.write(f'>{name}\n')
        s = seq.tobytes().decode()
        for i in range(0, len(s), 100):
            fh.write(s[i:i + 100] + '\n')
    with open(path + '.fai', 'w') as fh:
        fh.write(f'{name}\t{len(seq)}\t{len(name) + 2}\t100\t101\n')


def reference_density(tmp, ref, tig, srs=20, delta=None):
    """scripts/density.py as pavlib/inv.py:249-266 spawns it (plus --staterundelta when given)."""
    os.makedirs(tmp, exist_ok=True)
    write_fa(os.path.join(tmp, 'ref.fa'), 'chrN', ref)
    write_fa(os.path.join(tmp, 'tig.fa'), 'tigN', tig)
    args = ['python3', os.path.join(refenv.REFERENCE, 'scripts', 'density.py'),
            '--tigregion', f'tigN:1-{len(tig)}', '--refregion', f'chrN:1-{len(ref)}',
            '--ref', os.path.join(tmp, 'ref.fa'), '--tig', os.path.join(tmp, 'tig.fa'),
            '-k', str(K), '-t', '1', '-r', 'false', '--staterunsmooth', str(srs)]
    if delta is not None:
        args += ['--staterundelta', repr(float(delta))]
    p = subprocess.run(args, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return pickle.loads(base64.b64decode(p.stdout))


def save(case, ref, tig, df, params):
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, case + '.npz'), ref=ref, tig=tig, params=np.array(json.dumps(params)),
                        INDEX=df['INDEX'].to_numpy(np.int64), STATE_MER=df['STATE_MER'].to_numpy(np.int8),
                        STATE=df['STATE'].to_numpy(np.int8), KERN_FWD=df['KERN_FWD'].to_numpy(np.float64),
                        KERN_FWDREV=df['KERN_FWDREV'].to_numpy(np.float64), KERN_REV=df['KERN_REV'].to_numpy(np.float64))


def margins(df):
    kk = np.stack([df['KERN_FWD'].to_numpy(), df['KERN_FWDREV'].to_numpy(), df['KERN_REV'].to_numpy()])
    srt = np.sort(kk, axis=0)
    return (srt[2] - srt[1]) / np.maximum(srt[2], 1e-300)


# ---- closed form of the density of a run of consecutive integers (Euler-Maclaurin; search aid) ---------------------------
def run_density(a, b, x, h):
    """sum_{i=a..b} exp(-((i - x)/h)^2 / 2) / (sqrt(2 pi) h), vectorised."""
    ua, ub = (a - x) / h, (b - x) / h
    fa, fb = np.exp(-ua * ua / 2), np.exp(-ub * ub / 2)
    s = np.sqrt(0.5)
    integ = np.where(ua >= 0, erfc(ua * s) - erfc(ub * s), np.where(ub <= 0, erfc(-ub * s) - erfc(-ua * s), erf(ub * s) - erf(ua * s)))
    integ = integ * h * np.sqrt(np.pi / 2)
    d1 = (ua * fa - ub * fb) / h
    d3 = (ua * (ua * ua - 3) * fa - ub * (ub * ub - 3) * fb) / h ** 3
    return (integ + 0.5 * (fa + fb) + d1 / 12 - d3 / 720) / (np.sqrt(2 * np.pi) * h)


def blocks_fwd_rev(rng, m1, m2):
    """ref holds F and G; tig = F + revcomp(G): m1 FWD rows then m2 REV rows (junction k-mers are novel and dropped)."""
    f = ACGT[rng.integers(0, 4, m1 + K - 1)]
    g = ACGT[rng.integers(0, 4, m2 + K - 1)]
    spacer = ACGT[rng.integers(0, 4, 200)]
    return np.concatenate([f, spacer, g]), np.concatenate([f, revcomp(g)])


def blocks_fwd_rev_fwd(rng, m1, m2, m3):
    """tig = F1 + revcomp(G) + F3: m1 FWD rows, m2 REV rows, m3 FWD rows."""
    f1, g, f3 = (ACGT[rng.integers(0, 4, m + K - 1)] for m in (m1, m2, m3))
    sp = [ACGT[rng.integers(0, 4, 200)] for _ in range(2)]
    return np.concatenate([f1, sp[0], g, sp[1], f3]), np.concatenate([f1, revcomp(g), f3])


def case_argmax_search(rng):
    """FWD | REV | FWD with a REV block only 2-3 FWD bandwidths long: the far FWD arm still contributes 0.1-1 % at the first
    block boundary, which moves the row where KERN_FWD crosses KERN_REV off the half-way point between two rows (with two
    blocks, or a long REV block, the crossing is pinned there by symmetry and no table row comes close)."""
    m1 = np.arange(2500, 4500)[:, None].astype(np.float64)
    m2 = np.arange(600, 1600)[None, :].astype(np.float64)

    def sums(a, b):                                 # sum i, sum i^2 over a <= i < b
        return (b * (b - 1) - a * (a - 1)) / 2, ((b - 1) * b * (2 * b - 1) - (a - 1) * a * (2 * a - 1)) / 6
    best = []
    for m3 in range(1500, 3500, 100):
        n = m1 + m2 + m3
        bw = n ** -0.2
        mf = m1 + m3
        a1, a2 = sums(0, m1)
        b1, b2 = sums(m1 + m2, n)
        h0 = np.sqrt((mf * (a2 + b2) - (a1 + b1) ** 2) / (mf * (mf - 1))) * bw
        h2 = np.sqrt(m2 * (m2 + 1) / 12) * bw
        for dx in range(0, 4):                      # table rows behind the first block boundary
            x = m1 + dx
            kf = run_density(0.0, m1 - 1, x, h0) + run_density(m1 + m2, n - 1, x, h0)
            kr = run_density(m1, m1 + m2 - 1, x, h2)
            rel = np.abs(kf - kr) / np.maximum(kf, kr)
            for i, j in zip(*np.nonzero(rel < 3e-9)):
                best.append((float(rel[i, j]), int(m1[i, 0]), int(m2[0, j]), m3, int(x[i, 0])))
    best.sort()
    print('argmax_search: closed-form candidates', len(best), best[:5])
    for rel, a, b, c, x in best:
        for _ in range(40):                         # a junction k-mer can match by chance (spacer base == next contig base)
            ref, tig = blocks_fwd_rev_fwd(rng, a, b, c)
            o = oracle.density(ref, tig, False)
            if o['status'] == 0 and o['n'] == a + b + c and o['state_count'] == [a + c, 0, b]:
                break
        else:
            continue
        kf, kr = o['KERN_FWD'][x], o['KERN_REV'][x]
        m = abs(kf - kr) / max(kf, kr)
        print(f'  m1={a} m2={b} m3={c} row={x}: closed form {rel:.3e}, oracle {m:.3e}')
        if 2e-12 < m < 5e-10:
            return ref, tig, x, m
    raise SystemExit('no candidate confirmed')


def case_argmax_mirror(rng, m=3000, a=400):
    p1, p2, pc = (ACGT[rng.integers(0, 4, ln + K - 1)] for ln in (a, a, 21))
    f, g = ACGT[rng.integers(0, 4, m + K - 1)], ACGT[rng.integers(0, 4, m + K - 1)]
    sp = [ACGT[rng.integers(0, 4, 100)] for _ in range(8)]
    # ref: the palindromic-state blocks in both orientations, F, G
    ref = np.concatenate([p1, sp[0], revcomp(p1), sp[1], p2, sp[2], revcomp(p2), sp[3], pc, sp[4], revcomp(pc), sp[5], f, sp[6], g])
    tig = np.concatenate([p1, f, pc, revcomp(g), p2])
    return ref, tig, a + m + 10


def main():
    rng = np.random.default_rng(20260101)
    tmp = os.path.join('/tmp', 'pav_neartie')

    ref, tig, row, m = case_argmax_search(rng)
    df = reference_density(tmp, ref, tig)
    mg = margins(df)
    print('argmax_search: reference margin at row', row, mg[row], 'STATE', int(df['STATE'].iloc[row]), 'min margin', mg.min())
    assert 1e-12 < mg[row] < 1e-9
    save('argmax_search', ref, tig, df, {'k': K, 'staterunsmooth': 20, 'staterundelta': 0.005, 'row': int(row),
                                         'what': 'arg-max margin %.3e at the row' % mg[row]})

    for _ in range(40):
        ref, tig, row = case_argmax_mirror(rng)
        if oracle.density(ref, tig, False)['state_count'] == [3000, 2 * 400 + 21, 3000]:
            break
    df = reference_density(tmp, ref, tig)
    mg = margins(df)
    print('argmax_mirror: rows', df.shape[0], 'reference margin at centre row', row, mg[row], 'STATE', int(df['STATE'].iloc[row]),
          'STATE_MER', int(df['STATE_MER'].iloc[row]))
    assert mg[row] < 1e-13 and df.shape[0] == 2 * 3000 + 2 * 400 + 21
    save('argmax_mirror', ref, tig, df, {'k': K, 'staterunsmooth': 20, 'staterundelta': 0.005, 'row': int(row),
                                         'what': 'mirror-symmetric layout: exact tie of KERN_FWD and KERN_REV at the row'})

    # density_change by a hair: one quiet window of a FWD | REV table
    for _ in range(40):
        ref, tig = blocks_fwd_rev(rng, 4100, 3700)
        if oracle.density(ref, tig, False)['state_count'] == [4100, 0, 3700]:
            break
    df0 = reference_density(tmp, ref, tig)
    kk = np.stack([df0[c].to_numpy() for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')])
    sm, st = df0['STATE_MER'].to_numpy(), df0['STATE'].to_numpy()
    n = df0.shape[0]
    pick = None
    for a in range(0, n - 21, 20):
        b = a + 20
        if len(set(sm[a:b + 1])) > 1 or st[a] != st[b] or kk[:, [a, b]].max() > 0.99:
            continue
        d = np.max(np.abs(kk[:, a] - kk[:, b]))
        if 0.002 < d < 0.02:
            pick = (a, b, float(d))
            break
    assert pick, 'no quiet window found'
    a, b, d = pick
    print('delta: window', a, b, 'max |delta KERN| =', repr(d))
    for name, delta in (('delta_above', d * (1 + 1e-10)), ('delta_below', d * (1 - 1e-10))):
        df = reference_density(tmp, ref, tig, delta=delta)
        inner = kk[:, a + 1:b]
        got = np.stack([df[c].to_numpy() for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')])[:, a + 1:b]
        print(' ', name, repr(delta), 'inner rows differ from the default table by', float(np.max(np.abs(got - inner))))
        save(name, ref, tig, df, {'k': K, 'staterunsmooth': 20, 'staterundelta': float(delta), 'window': [int(a), int(b)],
                                  'what': 'max |delta KERN| of the window is %r' % d})


if __name__ == '__main__':
    main()
