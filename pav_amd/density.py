"""
K-mer density for calling inversions - host mirror of ``pavlib/density.py`` (rl_encoder) and of the command-line
contract of ``scripts/density.py``; the arithmetic runs on the GPU (csrc/density.hip).
"""

import numpy as np
import pandas as pd

from . import _lib

# Defaults of scripts/density.py's command line (:438-477) and MAX_REF_KMER_COUNT (:47)
DEFAULT_MIN_INFORMATIVE = 2000
DEFAULT_MIN_STATE_COUNT = 20
DEFAULT_DENSITY_SMOOTH = 1
DEFAULT_STATE_RUN_SMOOTH = 20
DEFAULT_STATE_RUN_DELTA = 0.005
MAX_REF_KMER_COUNT = 100

DENSITY_COLUMNS = ['INDEX', 'STATE_MER', 'STATE', 'KERN_FWD', 'KERN_FWDREV', 'KERN_REV', 'KMER']   # density.py:341


def den_params(k=31, min_informative=DEFAULT_MIN_INFORMATIVE, min_state_count=DEFAULT_MIN_STATE_COUNT,
               den_smooth=DEFAULT_DENSITY_SMOOTH, state_run_delta=DEFAULT_STATE_RUN_DELTA,
               max_ref_kmer_count=MAX_REF_KMER_COUNT, kde_mode=None, kmer_mode=None, guard_rel=0.0, guard_cap=0):
    """``kde_mode``: ``_lib.KDE_RUNS`` (default; closed-form sums over runs of consecutive k-mers) or
    ``_lib.KDE_DIRECT`` (one exp per pair in scipy's accumulation order).  Env ``PAV_KDE_MODE=direct`` forces the latter.
    ``kmer_mode``: ``_lib.KMER_LDS`` (default; reference k-mer sets partitioned into LDS tables) or ``_lib.KMER_HBM``
    (one hash table per region in HBM); same results.  Env ``PAV_KMER_HBM=1`` forces the latter inside the library.
    ``guard_rel``: near-tie guard of the float decisions (include/pav_amd.h): 0 = the default margin 1e-9, negative = off;
    ``guard_cap``: capacity of its re-evaluation list (0 = default)."""
    if kde_mode is None:
        import os
        kde_mode = _lib.KDE_DIRECT if os.environ.get('PAV_KDE_MODE', '').lower() == 'direct' else _lib.KDE_RUNS
    if kmer_mode is None:
        kmer_mode = _lib.KMER_LDS
    return _lib.DenParams(int(k), int(min_informative), int(min_state_count), float(den_smooth), float(state_run_delta),
                          int(max_ref_kmer_count), int(kde_mode), int(kmer_mode), int(guard_cap), float(guard_rel))


def table_frame(cols, finalised=True, extra=None):
    """Density table as the reference's DataFrame (scripts/density.py:340-342): indexed by INDEX; an un-finalised
    table (fewer than --mininf informative k-mers) keeps the early column set of :157-163 with STATE = -1.
    ``extra``: further columns appended in order (FLANK / MATCH of pavlib/inv.py:522-555).  Built in one shot."""
    index = cols['INDEX'].astype(np.int64)
    # KMER: k <= 31 fits int64 like the reference column; with k = 32 pandas gives the reference's list of Python integers the dtype
    # uint64 as soon as one of them needs the top bit (pd.DataFrame(tig_mer_stream), scripts/density.py:164) - which a region of
    # thousands of 32-mers always has; the text of the table is the same digits either way
    kmer = cols['KMER']
    kmer = kmer.astype(np.int64) if (kmer.shape[0] == 0 or int(kmer.max()) < (1 << 63)) else kmer.astype(np.uint64)
    if finalised:
        data = {'INDEX': index, 'STATE_MER': cols['STATE_MER'].astype(np.int64), 'STATE': cols['STATE'].astype(np.int64),
                'KERN_FWD': cols['KERN_FWD'], 'KERN_FWDREV': cols['KERN_FWDREV'], 'KERN_REV': cols['KERN_REV'],
                'KMER': kmer}
    else:
        data = {'KMER': kmer, 'INDEX': index, 'STATE': cols['STATE'].astype(np.int64),
                'STATE_MER': cols['STATE_MER'].astype(np.int64)}
    if extra:
        data.update(extra)
    return pd.DataFrame(data, index=pd.Index(index, name='INDEX'), copy=False)


def rl_encoder(df, state_col='STATE'):
    """
    Count consecutive states and track the INDEX range of each run (pavlib/density.py:330-361).

    :param df: Dataframe of states with INDEX and the state column.
    :param state_col: "STATE" (kernel-density max state) or "STATE_MER" (raw k-mer state).

    :return: Iterator of (state, count, pos, end) tuples.
    """
    states = df[state_col].to_numpy()
    index = df['INDEX'].to_numpy()
    n = states.shape[0]
    if n == 0:
        return
    heads = np.flatnonzero(np.concatenate(([True], states[1:] != states[:-1])))
    ends = np.concatenate((heads[1:], [n])) - 1
    for h, e in zip(heads.tolist(), ends.tolist()):
        yield (int(states[h]), e - h + 1, int(index[h]), int(index[e]))
