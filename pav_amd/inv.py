"""
Routines for calling inversions - host mirror of ``pavlib/inv.py`` for the MI355X path.

``scan_for_inv`` keeps the reference's signature, return type, log lines and soft/hard failure behaviour
(pavlib/inv.py:149-454); the per-iteration ``scripts/density.py`` subprocess is replaced by one call into
``libpav_amd.so`` (k-mer states, compaction, Gaussian KDE, interpolation, arg-max, run-length encoding on the GPU).
``scan_for_inv_batch`` runs many flagged regions in lock-step so that every scan iteration is one batched device
call (regions are independent: SURVEY.md section 8(e)); its results are identical to calling ``scan_for_inv``
region by region.
"""

import os

import numpy as np

from . import _lib, density, seq
from .fasta import open_fasta, read_fai

#
# Constants (pavlib/inv.py:19-40)
#

INITIAL_EXPAND = 4000      # Expand the flagged region by this much before starting.
EXPAND_FACTOR = 1.5        # Expand by this factor while searching
MAX_REGION_SIZE = 1200000  # Maximum region size
MIN_INFORMATIVE_KMERS = 2000
MIN_KMER_STATE_COUNT = 20
DENSITY_SMOOTH_FACTOR = 1
MIN_INV_KMER_RUN = 100     # States must have a continuous run of this many strictly inverted k-mers
MIN_QRY_REF_PROP = 0.6     # The contig and reference region sizes must be within this factor (reciprocal)
DEFAULT_MIN_EXP_COUNT = 1
DEFAULT_STATE_RUN_SMOOTH = 20
CALL_SOURCE = 'FLAG-DEN'
ERR_INV_FAIL = 125         # pavlib/constants.py:55

# KMER_LOC_STATE[in-upstream, in-dnstream] (pavlib/inv.py:46-51) as the codes pav_density_annotate returns
_MATCH_TEXT = np.array(['', 'SAME', 'OTHER', None], dtype=object)
_FLANK_TEXT = np.array(['', 'UP', 'DN'], dtype=object)


class InvCall:
    """An inversion call with the data supporting it (same attributes as pavlib.inv.InvCall, inv.py:54-118)."""

    def __init__(self, region_ref_outer, region_ref_inner, region_tig_outer, region_tig_inner, region_ref_discovery,
                 region_tig_discovery, region_flag, df):
        self.region_ref_outer = region_ref_outer
        self.region_ref_inner = region_ref_inner
        self.region_tig_outer = region_tig_outer
        self.region_tig_inner = region_tig_inner
        self.region_ref_discovery = region_ref_discovery
        self.region_tig_discovery = region_tig_discovery
        self.region_flag = region_flag
        self._df = df                       # DataFrame, or a zero-argument builder evaluated on first access
        self.native_table = None            # (context, region number, scan generation) when the native driver made the call
        self.svlen = len(region_ref_outer)
        self.id = '{}-{}-INV-{}'.format(region_ref_outer.chrom, region_ref_outer.pos + 1, self.svlen)

    @property
    def df(self):
        """Density table (INDEX, STATE_MER, STATE, KERN_*, KMER, FLANK, MATCH).  The columns are copied off the device
        when the call is made; the DataFrame itself is assembled on first access."""
        if callable(self._df):
            self._df = self._df()
        return self._df

    @df.setter
    def df(self, value):
        self._df = value

    def __repr__(self):
        return self.id


class _NativeBatch:
    """What the calls of one native scan share: the result records of the regions with something to report (columns as
    plain lists), the record names, the flagged regions, and where the density tables live."""

    def __init__(self, sub, hit, names, region_flags, table_of):
        self.sub, self.hit, self.names, self.region_flags, self.table_of = sub, hit, names, region_flags, table_of
        ro = sub['ref_outer']
        self.ro_seq, self.ro_pos, self.ro_end = ro['seq_id'].tolist(), ro['pos'].tolist(), ro['end'].tolist()


class _NativeInvCall(InvCall):
    """InvCall made by the native driver.  Everything is decoded from the driver's result record on first access (a batch of
    a thousand regions yields a hundred calls with six regions each; building them eagerly was most of the Python time of a
    scan): the object starts as (batch, row in the batch, region number)."""

    _FIELDS = {'region_ref_outer': ('ref_outer', 0, False), 'region_ref_inner': ('ref_inner', 0, False),
               'region_tig_outer': ('tig_outer', 1, True), 'region_tig_inner': ('tig_inner', 1, True),
               'region_ref_discovery': ('ref_discovery', 0, False), 'region_tig_discovery': ('tig_discovery', 1, True)}

    def __init__(self, batch, q, i):     # noqa: super().__init__ not called: the attributes stay lazy
        self._batch, self._q, self._i = batch, q, i

    def __getattr__(self, name):                             # reached only while the attribute has not been built yet
        if name in ('_batch', '_q', '_i'):
            raise AttributeError(name)
        b, q = self._batch, self._q
        spec = _NativeInvCall._FIELDS.get(name)
        if spec is not None:
            g = b.sub[q][spec[0]]
            n_aln, aln = g['n_aln'].tolist(), g['aln_index'].tolist()
            value = seq.Region(b.names[spec[1]][int(g['seq_id'])], int(g['pos']), int(g['end']),
                               is_rev=bool(g['is_rev']) if spec[2] else False,
                               pos_aln_index=(tuple(aln[0][:n_aln[0]]),) if n_aln[0] else None,
                               end_aln_index=(tuple(aln[1][:n_aln[1]]),) if n_aln[1] else None)
        elif name == 'svlen':
            value = b.ro_end[q] - b.ro_pos[q]
        elif name == 'id':
            value = '{}-{}-INV-{}'.format(b.names[0][b.ro_seq[q]], b.ro_pos[q] + 1, b.ro_end[q] - b.ro_pos[q])
        elif name == 'region_flag':
            value = b.region_flags[self._i]
        elif name in ('n_near_tie', 'n_unresolved'):                                  # near-tie guard (pav_amd.h)
            value = int(b.sub[q][name])
        elif name in ('_df', 'native_table'):
            value = b.table_of(self._i)[0 if name == '_df' else 1]
        else:
            raise AttributeError(name)
        setattr(self, name, value)
        return value


class _Interval:
    __slots__ = ('begin', 'end', 'data')

    def __init__(self, begin, end, data):
        self.begin, self.end, self.data = begin, end, data


class SrsTree:
    """Minimal interval lookup with the slice of the intervaltree interface scan_for_inv uses: ``tree[x]`` returns
    the set of intervals containing ``x``; each has ``.data`` (pavlib/inv.py:259)."""

    def __init__(self):
        self.intervals = []

    def add(self, begin, end, data):
        self.intervals.append(_Interval(begin, end, data))

    def __getitem__(self, point):
        return {iv for iv in self.intervals if iv.begin <= point < iv.end}


class IntervalSet:
    """The slice of ``intervaltree.IntervalTree`` the N-gap trees of the rules use (call_lg.snakefile:76-81,
    pavlib/inv.py:214-219): ``tree[a:b] = data`` adds an interval, ``tree[a:b]`` returns the intervals overlapping
    ``[a, b)``, ``tree[x]`` those containing ``x``."""

    def __init__(self):
        self.intervals = []

    def __setitem__(self, key, data):
        self.intervals.append(_Interval(key.start, key.stop, data))

    def __getitem__(self, key):
        if isinstance(key, slice):
            return {iv for iv in self.intervals if iv.begin < key.stop and iv.end > key.start}
        return {iv for iv in self.intervals if iv.begin <= key < iv.end}

    def __len__(self):
        return len(self.intervals)


def get_srs_tree(srs_tuple_list):
    """``[(region size limit, state-run-smooth factor), ...]`` -> lookup of the factor by region size, as
    ``pavlib.inv.get_srs_tree`` (pavlib/inv.py:564-620): the factor given with a limit applies from that limit up to the next
    one; below the first limit it is 20 (or the limit if smaller); no list = 20 everywhere.  Same checks and messages, in
    the order the reference meets them (a malformed element: RuntimeError when it is a str, else the TypeError of the
    reference's message concatenation)."""
    tree = SrsTree()
    if srs_tuple_list is None or len(srs_tuple_list) == 0:
        tree.add(0, np.inf, DEFAULT_STATE_RUN_SMOOTH)
        return tree
    for item in srs_tuple_list:
        if len(item) != 2:
            # the message is built by concatenation as in inv.py:573: a str element raises this RuntimeError, anything else
            # the TypeError of the concatenation itself
            raise RuntimeError('Element in "state run smooth" tuple list that is not length 2: ' + item)
    lower = factor = None
    for limit, value in sorted(srs_tuple_list):
        limit, value = int(limit), int(value)
        if lower is None:
            if limit < 0:
                raise RuntimeError('State run inversion size limits must be 0 or greater: {}'.format(limit))
            if value < 4:
                raise RuntimeError('Not tested with "state run smooth" factor less than 4: {}'.format(value))
            if limit > 0:
                tree.add(0, limit, np.min([limit, 20]))
        else:
            if value < 20:
                raise RuntimeError('Not tested with "state run smooth" factor less than 20: {}'.format(value))
            if limit == lower:
                raise RuntimeError('Duplicate limit in state run limits: {}'.format(limit))
            tree.add(lower, limit, factor)
        lower, factor = limit, value
    tree.add(lower, np.inf, factor)
    return tree


def _write_log(message, log):
    if log is None:
        return
    log.write(message)
    log.write('\n')
    log.flush()


class _Scan:
    """One flagged region's scan as a state machine: ``next_job()`` gives the density job of the next iteration (or
    None when the scan ended), ``feed()`` consumes the device result.  Control flow = pavlib/inv.py:203-454."""

    def __init__(self, region_flag, ref_fa_name, tig_fa_name, align_lift, k_util, n_tree, max_region_size, log, srs_tree,
                 min_exp_count, ref_index, tig_index):
        self.region_flag = region_flag
        self.ref_fa_name, self.tig_fa_name = ref_fa_name, tig_fa_name
        self.align_lift, self.k_util, self.log = align_lift, k_util, log
        self.min_exp_count = DEFAULT_MIN_EXP_COUNT if min_exp_count is None else min_exp_count
        self.max_region_size = MAX_REGION_SIZE if max_region_size is None else max_region_size
        self.ref_index, self.tig_index = ref_index, tig_index
        self.done = False
        self.result = None
        self.error = None
        _write_log('Scanning for inversions in flagged region: {} (flagged region record id = {})'.format(
            region_flag, region_flag.region_id()), log)
        self.df_fai = read_fai(ref_fa_name + '.fai')
        self.region_ref = region_flag.copy()
        self.region_ref.expand(INITIAL_EXPAND, min_pos=0, max_end=self.df_fai, shift=True)
        self.expansion_count = 0
        self.n_tree_chrom = n_tree[self.region_ref.chrom] if n_tree is not None and self.region_ref.chrom in n_tree.keys() else None
        self.srs_tree = get_srs_tree(None) if srs_tree is None else srs_tree
        if not hasattr(self.srs_tree, '__getitem__'):
            raise NotImplementedError('Custom state-run-smooth parameters are not currently implemented')
        self.region_tig = None

    def _finish(self, result=None):
        self.done = True
        self.result = result
        return None

    def next_job(self):
        """Top of the ``while True`` loop (inv.py:223-260) up to the density call."""
        region_ref = self.region_ref
        if 0 < self.max_region_size < len(region_ref):
            _write_log('Region size exceeds max: {} ({} > {})'.format(region_ref, len(region_ref), self.max_region_size), self.log)
            return self._finish()
        if self.n_tree_chrom is not None:
            if len(self.n_tree_chrom[region_ref.pos:region_ref.end]) > 0:
                _write_log('Region overlaps N bases: {}'.format(region_ref), self.log)
        try:
            region_tig = self.align_lift.lift_region_to_qry(region_ref)
        except RuntimeError as ex:
            self.error = ex
            return self._finish()
        if region_tig is None:
            _write_log('Could not lift reference region onto contigs: {}'.format(region_ref), self.log)
            return self._finish()
        self.region_tig = region_tig
        self.expansion_count += 1
        _write_log('Scanning region: {}'.format(region_ref), self.log)
        srs = int(list(self.srs_tree[len(region_tig)])[0].data)
        if region_ref.chrom not in self.ref_index or region_tig.chrom not in self.tig_index:
            self.error = RuntimeError('Sequence {} / {} is not loaded on the device'.format(region_ref.chrom, region_tig.chrom))
            return self._finish()
        return _lib.DenJob(self.ref_index[region_ref.chrom], self.tig_index[region_tig.chrom], region_ref.pos, region_ref.end,
                           region_tig.pos, region_tig.end, 1 if region_tig.is_rev else 0, srs)

    def feed(self, res, state_rl):
        """After the density call (inv.py:268-351).  Returns True when the region is flanked by reference-oriented
        k-mers and must be characterised (table needed)."""
        region_ref, log = self.region_ref, self.log
        if res.status == _lib.DEN_FAIL:
            if res.fail_kind == 1:
                stderr = ''                                            # message goes to stdout in density.py:511
            else:
                stderr = 'K-mer count exceeds max: {} > {} ({}): {}\n'.format(
                    res.max_count, density.MAX_REF_KMER_COUNT, self.k_util.to_string(res.max_kmer), region_ref)
            _write_log('Received return code {} from scripts/density.py for region {}:\n{}'.format(
                ERR_INV_FAIL, str(region_ref), stderr), log)
            self._finish()
            return False
        if res.n_rows == 0:
            _write_log('No informative reference k-mers in forward or reverse orientation in region', log)
            self._finish()
            return False
        self.state_rl = state_rl
        condensed_states = [record[0] for record in state_rl]
        if len(state_rl) == 1 and state_rl[0][0] in {0, -1} and self.expansion_count >= self.min_exp_count:
            _write_log('Found no inverted k-mer states after {} expansion(s)'.format(self.expansion_count), log)
            self._finish()
            return False
        if len(condensed_states) > 2 and condensed_states[0] == 0 and condensed_states[-1] == 0:
            return True
        # Expand (inv.py:309-342)
        last_len = len(region_ref)
        expand_bp = np.int32(len(region_ref) * EXPAND_FACTOR)
        if len(condensed_states) > 2:
            if condensed_states[0] == 0:
                balance = 0.25
            elif condensed_states[-1] == 0:
                balance = 0.75
            else:
                balance = 0.5
        else:
            balance = 0.5
        region_ref.expand(expand_bp, min_pos=0, max_end=self.df_fai, shift=True, balance=balance)
        if len(region_ref) == last_len:
            _write_log('Reached reference limits, cannot expand', log)
            self._finish()
        return False

    def characterise(self, ctx, job):
        """inv.py:353-454 with the flanked table resident on the device as ``job`` of the last batch."""
        state_rl, log, k_util = self.state_rl, self.log, self.k_util
        region_ref, region_tig = self.region_ref, self.region_tig
        if not np.any([record[0] == 2 for record in state_rl]):
            _write_log('No inverted states found', log)
            return self._finish()
        max_inv_run = np.max([record[1] for record in state_rl if record[0] == 2])
        if max_inv_run < MIN_INV_KMER_RUN:
            _write_log('Longest run of strictly inverted k-mers ({}) does not meet the minimum threshold ({})'.format(
                max_inv_run, MIN_INV_KMER_RUN), log)
            return self._finish()
        if state_rl[0][0] != 0 or state_rl[-1][0] != 0:
            self.error = RuntimeError('Found INV region not flanked by reference sequence (program bug): {}'.format(region_ref))
            return self._finish()
        state_rl_inv = [record for record in state_rl if record[0] == 2]
        region_tig_outer = seq.Region(region_tig.chrom, state_rl[1][2] + region_tig.pos,
                                      state_rl[-2][3] + region_tig.pos + k_util.k_size, is_rev=region_tig.is_rev)
        region_tig_inner = seq.Region(region_tig.chrom, state_rl_inv[0][2] + region_tig.pos,
                                      state_rl_inv[-1][3] + region_tig.pos + k_util.k_size, is_rev=region_tig.is_rev)
        try:
            region_ref_outer = self.align_lift.lift_region_to_sub(region_tig_outer)
            if region_ref_outer is None:
                _write_log('Failed lifting outer INV region to reference: {}'.format(region_tig_outer), log)
                return self._finish()
            region_ref_inner = self.align_lift.lift_region_to_sub(region_tig_inner, gap=True)
        except RuntimeError as ex:
            self.error = ex
            return self._finish()
        if region_ref_inner is None:
            region_ref_inner = region_ref_outer
        print('INV Found: outer={}, inner={} (ref outer={}, inner={})'.format(
            region_tig_outer, region_tig_inner, region_ref_outer, region_ref_inner))
        if len(region_ref_outer) < len(region_tig_outer) * MIN_QRY_REF_PROP:
            _write_log('Reference region too short: Reference region length ({:,d}) is not within {:.2f}% of the contig region length ({:,d})'.format(
                len(region_ref_outer), MIN_QRY_REF_PROP * 100, len(region_tig_outer)), log)
            return self._finish()
        if len(region_tig_outer) < len(region_ref_outer) * MIN_QRY_REF_PROP:
            _write_log('Contig region too short: Contig region length ({:,d}) is not within {:.2f}% of the reference region length ({:,d})'.format(
                len(region_tig_outer), MIN_QRY_REF_PROP * 100, len(region_ref_outer)), log)
            return self._finish()
        # density table + INV-DUP annotation (inv.py:440-442, 457-561)
        res_rows = self.n_rows
        cols = ctx.density_table(job, res_rows)
        codes = _flank_match_codes(res_rows, region_ref_outer, region_ref_inner, region_tig_outer, region_tig_inner,
                                   region_ref, ctx, job, self.ref_index)

        def df():
            return density.table_frame(cols, finalised=True, extra=_flank_match_text(*codes))
        inv_call = InvCall(region_ref_outer, region_ref_inner, region_tig_outer, region_tig_inner, region_ref, region_tig,
                           self.region_flag, df)
        _write_log('Found inversion: {}'.format(inv_call), log)
        return self._finish(inv_call)


def _flank_match_text(flank, match):
    m = _MATCH_TEXT[match]
    m[match == 3] = np.nan                                             # 'NA' -> NaN (inv.py:555)
    return {'FLANK': _FLANK_TEXT[flank], 'MATCH': m}


def _flank_match_columns(*args):
    return _flank_match_text(*_flank_match_codes(*args))


def _flank_match_codes(n_rows, region_ref_outer, region_ref_inner, region_tig_outer, region_tig_inner,
                       region_tig_discovery, ctx, job, ref_index):
    """FLANK / MATCH codes of pavlib/inv.py:480-555 for the table resident on the device as ``job``."""
    region_dup_ref_up = seq.Region(region_ref_outer.chrom, region_ref_outer.pos, region_ref_inner.pos)
    region_dup_ref_dn = seq.Region(region_ref_outer.chrom, region_ref_inner.end, region_ref_outer.end)
    region_dup_tig_up = seq.Region(region_tig_outer.chrom, region_tig_outer.pos, region_tig_inner.pos)
    region_dup_tig_dn = seq.Region(region_tig_outer.chrom, region_tig_inner.end, region_tig_outer.end)
    flank, match = ctx.density_annotate(
        job, n_rows, ref_index[region_ref_outer.chrom],
        (region_dup_ref_up.pos, region_dup_ref_up.end), (region_dup_ref_dn.pos, region_dup_ref_dn.end),
        int(region_tig_discovery.pos),
        (region_dup_tig_up.pos, region_dup_tig_up.end), (region_dup_tig_dn.pos, region_dup_tig_dn.end))
    return flank, match


def annotate_inv_dup_mers(df, region_ref_outer, region_ref_inner, region_tig_outer, region_tig_inner,
                          region_tig_discovery, ref_fa, k_util, ctx=None, job=None, ref_index=None):
    """
    Annotate inverted duplications flanking an inversion: FLANK (UP / DN by contig index) and MATCH (SAME / OTHER /
    NaN) columns (pavlib/inv.py:457-561).  Note the reference passes the *reference* discovery region as
    ``region_tig_discovery`` (inv.py:440-442) and tests the raw KMER against canonical k-mer sets; both are kept.
    Runs on the table resident on the device (``ctx``, ``job``).
    """
    if ctx is None or job is None:
        raise _lib.PavDeviceError('annotate_inv_dup_mers needs the device context holding the density table')
    extra = _flank_match_columns(df.shape[0], region_ref_outer, region_ref_inner, region_tig_outer, region_tig_inner,
                                 region_tig_discovery, ctx, job, ref_index)
    df['FLANK'] = extra['FLANK']
    df['MATCH'] = extra['MATCH']
    return df


# ---------------------------------------------------------------------------------------------------------
# Drivers
# ---------------------------------------------------------------------------------------------------------

def _seq_index(ctx):
    """name -> record number for both stores (rebuilt only when a store was loaded again)."""
    r, t = ctx.seq_names(_lib.PAV_ROLE_REF), ctx.seq_names(_lib.PAV_ROLE_TIG)
    cache = getattr(ctx, '_seq_index_cache', None)
    if cache is None or cache[0] is not r or cache[1] is not t:
        cache = (r, t, {n: i for i, n in enumerate(r)}, {n: i for i, n in enumerate(t)})
        ctx._seq_index_cache = cache
    return cache[2], cache[3]


def ensure_sequences(ctx, ref_fa_name, tig_fa_name):
    """Upload both FASTA files once per context (records are addressed by name afterwards)."""
    key = (str(ref_fa_name), str(tig_fa_name))
    if getattr(ctx, '_inv_loaded', None) != key:
        ref_fa, tig_fa = open_fasta(ref_fa_name), open_fasta(tig_fa_name)
        ctx.seq_load_fasta(_lib.PAV_ROLE_REF, ref_fa.native, ref_fa.record_numbers(ref_fa.names))
        ctx.seq_load_fasta(_lib.PAV_ROLE_TIG, tig_fa.native, tig_fa.record_numbers(tig_fa.names))
        ctx._inv_loaded = key


def _drive(ctx, scans, params, max_batch_bp=64_000_000):
    """Run scan state machines in lock-step: every round is one batched device call over the live regions."""
    live = list(scans)
    while live:
        jobs, owners = [], []
        budget = 0
        rest = []
        for sc in live:
            if budget > max_batch_bp and jobs:
                rest.append(sc)
                continue
            job = sc.next_job()
            if job is None:
                continue
            jobs.append(job)
            owners.append(sc)
            budget += (job.ref_end - job.ref_pos) + (job.tig_end - job.tig_pos)
        if not jobs:
            live = rest
            continue
        results = ctx.density_batch(jobs, params)
        nxt = []
        for j, (sc, res) in enumerate(zip(owners, results)):
            runs = ctx.density_runs(j, res.n_runs) if res.status != _lib.DEN_FAIL and res.n_rows else []
            sc.n_rows = res.n_rows
            if sc.feed(res, runs):
                sc.characterise(ctx, j)          # table of job j is still resident
            elif not sc.done:
                nxt.append(sc)
        live = nxt + rest


def _check_k_size(k_util):
    """k-mers are 2-bit packed into one 64-bit word on the device (k <= 32, ``PAV_E_LIMIT`` in include/pav_amd.h; 32-mers use the
    HBM-table kernels: an LDS slot has no room for the two orientation bits beside 64 bits of k-mer); kanapy's Python integers
    have no such limit, so ``inv_k_size >= 33`` in config.json is refused here with the reason instead of a generic device
    error - before any device work."""
    k = int(k_util.k_size)
    if not 1 <= k <= 32:
        raise RuntimeError('k-mer size {} is not supported by pav_amd (1..32): k-mers are packed 2 bits per base into one '
                           '64-bit word on the device; set inv_k_size <= 32'.format(k))


def scan_for_inv(region_flag, ref_fa_name, tig_fa_name, align_lift, k_util, n_tree=None, max_region_size=None, threads=1,
                 log=None, srs_tree=None, min_exp_count=DEFAULT_MIN_EXP_COUNT, ctx=None, device_id=0):
    """
    Scan a flagged region for an inversion, expanding as necessary (same contract as pavlib/inv.py:149-185).

    :return: An ``InvCall`` describing the inversion found or ``None``.  ``threads`` is accepted for compatibility.
    """
    _check_k_size(k_util)
    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        ensure_sequences(ctx, ref_fa_name, tig_fa_name)
        ref_index, tig_index = _seq_index(ctx)
        sc = _Scan(region_flag, ref_fa_name, tig_fa_name, align_lift, k_util, n_tree, max_region_size, log, srs_tree,
                   min_exp_count, ref_index, tig_index)
        _drive(ctx, [sc], density.den_params(k=k_util.k_size))
    finally:
        if own:
            ctx.close()
    if sc.error is not None:
        raise sc.error
    return sc.result


_DEFAULT_SRS = []


def _native_srs(srs_tree):
    """The state-run-smooth lookup as the array of ``pav_srs`` the native driver takes; built once per tree (a scan per pass of
    a haplotype asks with the same tree every time)."""
    if srs_tree is None:
        if not _DEFAULT_SRS:
            _DEFAULT_SRS.append(get_srs_tree(None))
        srs_tree = _DEFAULT_SRS[0]
    ivs = list(getattr(srs_tree, 'intervals', None) or srs_tree)      # SrsTree or a real intervaltree.IntervalTree
    key = tuple((float(iv.begin), float(iv.end), int(iv.data)) for iv in ivs)
    cached = getattr(srs_tree, '_pav_native_srs', None)
    if cached is None or cached[0] != key:
        arr = (_lib.Srs * max(1, len(ivs)))(*[_lib.Srs(b, e, d, 0) for b, e, d in key])
        cached = (key, arr, len(ivs))
        try:
            srs_tree._pav_native_srs = cached
        except AttributeError:                                        # an object without a __dict__: built every time
            pass
    return cached[1], cached[2]


def pack_lift_table(align_lift, ref_index, tig_index):
    """The trimmed alignment table of an :class:`pav_amd.align.AlignLift` as the arrays ``pav_inv_load_alignments`` takes: row
    records (record numbers of the two stores, spans, strand, INDEX), the CIGAR strings as one text block, their offsets."""
    df = align_lift.df
    n = df.shape[0]
    aln = np.zeros(n, dtype=_lib.INV_ALN_DTYPE)
    if n:
        aln['ref_id'] = [ref_index[str(c)] for c in df['#CHROM']]
        aln['tig_id'] = [tig_index[str(c)] for c in df['QRY_ID']]
        for col, name in (('pos', 'POS'), ('end', 'END'), ('qry_pos', 'QRY_POS'), ('qry_end', 'QRY_END'), ('index', 'INDEX')):
            aln[col] = df[name].to_numpy(dtype=np.int64)
        aln['rev'] = [1 if bool(v) else 0 for v in df['REV']]
    cig = [str(c).encode() for c in df['CIGAR']] if n else []
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(c) for c in cig], dtype=np.uint64)
    text = np.frombuffer(b''.join(cig), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    return aln, text, off


def _native_scan(ctx, region_flags, align_lift, k_util, max_region_size, logs, srs_tree, min_exp_count, ref_index, tig_index,
                 eager_tables=True, log=None, found_out=None):
    """All regions through the library's native driver (pav_inv_scan_batch, csrc/invscan.cpp)."""
    import ctypes
    import time
    _t = [time.perf_counter()]
    _timing = bool(os.environ.get('PAV_TIMING'))

    def _lap(what):
        if _timing:
            t = time.perf_counter()
            print('[pav timing] _native_scan %-18s %.1f ms' % (what, (t - _t[0]) * 1e3), file=__import__('sys').stderr)
            _t[0] = t
    if getattr(align_lift, '_native_loaded', None) is not ctx:
        aln, text, off = pack_lift_table(align_lift, ref_index, tig_index)
        _lap('marshal table')
        ctx.inv_load_alignments(aln, text, off)
        align_lift._native_loaded = ctx
        _lap('load_alignments')
    srs, n_srs = _native_srs(srs_tree)
    params = _lib.InvParams(int(MAX_REGION_SIZE if max_region_size is None else max_region_size),
                            int(DEFAULT_MIN_EXP_COUNT if min_exp_count is None else min_exp_count), n_srs,
                            ctypes.cast(srs, ctypes.POINTER(_lib.Srs)), density.den_params(k=k_util.k_size),
                            0 if eager_tables else 1, 0)
    regions = np.zeros(len(region_flags), dtype=_lib.INV_REGION_DTYPE)
    if isinstance(region_flags, RegionColumns):
        regions['ref_id'], regions['pos'], regions['end'] = region_flags.ref_id, region_flags.pos, region_flags.end
    elif len(region_flags):
        regions['ref_id'] = [ref_index[rf.chrom] for rf in region_flags]
        regions['pos'] = [rf.pos for rf in region_flags]
        regions['end'] = [rf.end for rf in region_flags]
    _lap('marshal regions')
    res = ctx.inv_scan_batch(regions, params)
    _lap('inv_scan_batch')
    ref_names, tig_names = ctx.seq_names(_lib.PAV_ROLE_REF), ctx.seq_names(_lib.PAV_ROLE_TIG)

    n_rgn = len(region_flags)
    resv = np.frombuffer(res, dtype=_lib.INV_RESULT_DTYPE, count=n_rgn) if n_rgn else np.zeros(0, dtype=_lib.INV_RESULT_DTYPE)
    outcome, found = resv['outcome'], resv['found']
    if eager_tables:
        all_cols, all_flank, all_match, row_off = ctx.inv_tables(np.where(outcome == _lib.INV_CALL, resv['n_rows'], 0))
    generation = ctx._inv_generation
    _lap('inv_tables')
    if log is not None and n_rgn:                                      # one sink for the batch: the texts in region order
        log.write(''.join(ctx.inv_texts(0, n_rgn, int(resv['log_bytes'].sum(dtype=np.int64)), joined=True)))
        log.flush()
    if logs is not None and n_rgn:
        texts = ctx.inv_texts(0, n_rgn, int(resv['log_bytes'].sum(dtype=np.int64)))
        for lg, text in zip(logs, texts):
            if lg is not None and text:
                lg.write(text)
                lg.flush()
    errors = ctx.inv_texts(1, n_rgn, int(resv['error_bytes'].sum(dtype=np.int64))) if (outcome == _lib.INV_ERROR).any() else None
    out = [None] * n_rgn
    hit = np.flatnonzero((found != 0) | (outcome != _lib.INV_NONE))
    sub = resv[hit]                                                   # the few regions with something to report
    names = (ref_names, tig_names)

    if eager_tables:
        def table_of(i):
            sl = slice(int(row_off[i]), int(row_off[i + 1]))
            cols = {name: arr[sl] for name, arr in all_cols.items()}
            flank, match = all_flank[sl], all_match[sl]
            return (lambda: density.table_frame(cols, finalised=True, extra=_flank_match_text(flank, match))), (ctx, i, generation)
    else:
        def table_of(i):                        # views of the library's pinned host copy; valid until the next scan
            def df():
                cols, flank, match = ctx.inv_table_view(i, generation)
                return density.table_frame(cols, finalised=True, extra=_flank_match_text(flank.copy(), match.copy()))
            return df, (ctx, i, generation)     # native_table: the library's host copy, Context.inv_write_tables writes it as text
    batch = _NativeBatch(sub, hit, names, region_flags, table_of)
    sub_outcome = sub['outcome'].tolist()
    for q, i in enumerate(hit.tolist()):
        if sub_outcome[q] == _lib.INV_ERROR:
            out[i] = RuntimeError(errors[i])
        elif sub_outcome[q] == _lib.INV_CALL:
            out[i] = _NativeInvCall(batch, q, i)
    if found.any():                                                    # inv.py:408, one line per region in region order
        print(ctx.inv_texts(2, n_rgn, joined=True)[0], end='', file=found_out)
    _lap('results')
    return out


class RegionColumns:
    """Flagged regions as three columns - record number in the context's reference store, POS, END - for batches that never
    existed as Python objects (the loci ``pav_cigar_flag`` has just produced).  Behaves like a list of Regions."""

    def __init__(self, ref_id, pos, end, names):
        self.ref_id = np.ascontiguousarray(ref_id, dtype=np.uint32)
        self.pos = np.ascontiguousarray(pos, dtype=np.int64)
        self.end = np.ascontiguousarray(end, dtype=np.int64)
        self.names = names

    def __len__(self):
        return self.ref_id.shape[0]

    def __getitem__(self, i):
        return seq.Region(self.names[int(self.ref_id[i])], int(self.pos[i]), int(self.end[i]))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def loci_regions(ctx, loci, only_try_inv=True):
    """``pav_flag_locus`` records (``Context.cigar_flag`` / ``flag_merge_loci``; their ``chrom`` is the rank of the
    chromosome name in str order, as the flag tables sort) -> :class:`RegionColumns` of the loci rule call_inv_batch would
    hand to ``scan_for_inv``: those with TRY_INV (rules/call_inv.snakefile:145-146, 185-196)."""
    names = ctx.seq_names(_lib.PAV_ROLE_REF)
    cached = getattr(ctx, '_ref_by_rank', None)                      # (the order of the record names: the same for every pass of a context)
    if cached is None or cached[0] is not names:
        cached = ctx._ref_by_rank = (names, np.array(sorted(range(len(names)), key=lambda i: names[i]), dtype=np.uint32))
    by_rank = cached[1]
    sel = loci[loci['try_inv'] != 0] if only_try_inv else loci
    return RegionColumns(by_rank[sel['chrom']] if sel.shape[0] else np.zeros(0, np.uint32), sel['pos'], sel['end'], names)


def scan_for_inv_batch(region_flags, ref_fa_name, tig_fa_name, align_lift, k_util, n_tree=None, max_region_size=None,
                       logs=None, srs_tree=None, min_exp_count=DEFAULT_MIN_EXP_COUNT, ctx=None, device_id=0, native=None,
                       eager_tables=True, log=None, found_out=None):
    """Scan many flagged regions; returns a list of ``InvCall`` / ``None`` / ``RuntimeError`` (one per region, the
    error object where ``scan_for_inv`` would have raised).  ``logs``: one file-like object per region or None.
    ``log``: one file-like object for the whole batch - what rule call_inv_batch passes to every ``scan_for_inv`` call
    (rules/call_inv.snakefile:172,191-196); it receives the log text of all regions in region order.

    ``native``: run the whole scan loop inside the library (csrc/invscan.cpp).  Default: yes when ``align_lift`` is a
    :class:`pav_amd.align.AlignLift` and no N-tree is given; the Python state machine below is the same algorithm and
    is used otherwise (e.g. for a ``pavlib.align.AlignLift`` object).

    ``eager_tables`` (native driver): copy every call's density table into numpy arrays before returning (default).  With
    ``False`` ``InvCall.df`` is assembled on first access from the library's host copy, which lives until the next scan
    on the same context (reading it later raises); the rule mirror and bench.py use that.  The tables then stay packed in
    HBM during the scan and cross PCIe when the first of them is read.

    ``found_out``: where the 'INV Found: ...' lines the reference prints (pavlib/inv.py:408) go (native driver; default
    ``sys.stdout``) - callers that scan from several threads give each its own sink."""
    _check_k_size(k_util)
    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        ensure_sequences(ctx, ref_fa_name, tig_fa_name)
        ref_index, tig_index = _seq_index(ctx)
        from .align import AlignLift as _OurLift
        if native is None:
            native = isinstance(align_lift, _OurLift) and n_tree is None and \
                os.environ.get('PAV_INV_DRIVER', '').lower() != 'python'
        if native:
            # a context that is closed on return cannot serve lazy tables: they are copied out before it goes
            return _native_scan(ctx, region_flags, align_lift, k_util, max_region_size, logs, srs_tree, min_exp_count,
                                ref_index, tig_index, eager_tables=eager_tables or own, log=log, found_out=found_out)
        # lock-step scans write into private buffers; the text goes to the caller's sinks afterwards, in region order
        import io
        want_text = log is not None or logs is not None
        private = [io.StringIO() if want_text else None for _ in region_flags]
        scans = [_Scan(rf, ref_fa_name, tig_fa_name, align_lift, k_util, n_tree, max_region_size, private[i], srs_tree,
                       min_exp_count, ref_index, tig_index) for i, rf in enumerate(region_flags)]
        try:
            _drive(ctx, scans, density.den_params(k=k_util.k_size))
        finally:
            # also when a scan raised: what the scans wrote so far reaches the caller's sinks, as with the reference's
            # streaming log (pavlib/inv.py:623-640)
            if logs is not None:
                for sink, buf in zip(logs, private):
                    if sink is not None and buf.getvalue():
                        sink.write(buf.getvalue())
                        sink.flush()
            if log is not None:
                log.write(''.join(buf.getvalue() for buf in private))
                log.flush()
    finally:
        if own:
            ctx.close()
    return [sc.error if sc.error is not None else sc.result for sc in scans]
