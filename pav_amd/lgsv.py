"""
Large alignment-truncating variants - the interface of ``pavlib.lgsv.scan_for_events`` (pavlib/lgsv.py:31-642; rule
``call_lg_discover``, rules/call_lg.snakefile:40-104) for this package.

How it is organised here
  * the alignment table is split once into per-(chromosome, contig) column arrays (``_Group``); the partner search of every
    record walks those arrays instead of re-indexing the DataFrame per step;
  * a candidate pair is described by a ``_Gap`` (the unaligned interval between two same-strand records on both axes) which
    classifies itself as DEL / INS / INV-candidate / nothing;
  * INS / DEL rows are collected column-wise by ``_Events`` together with four breakpoint-homology queries each, and all
    queries of the table go to the GPU in ONE ``pav_homology`` call after the walk - possible because the event list does
    not depend on the homology values (``match_bp`` below);
  * inversion candidates go through ``pav_amd.inv.scan_for_inv`` (k-mer density scan on the GPU); sequences are resident
    once per context instead of the reference's whole-chromosome ``SeqCache`` (pavlib/lgsv.py:645-695).

What must stay as the reference has it (tests/golden/lgsv_hap is the reference's own output): the order in which partners are
tried, including the step of two after a same-strand pair that yields nothing (pavlib/lgsv.py:434-437 and :556-557 both
advance), the distance-proportion filter (:160-170), the three RuntimeError texts, the log lines, the columns.
"""

import os
import sys

import numpy as np
import pandas as pd

from . import _lib, inv, seq
from .align import AlignLift
from .kmer import KmerUtil

MAX_QRY_DIST_PROP = 1       # lgsv.py:18
MAX_REF_DIST_PROP = 3       # lgsv.py:19
DIST_PROP_LEN_MAPQ = (20000, 40)   # lgsv.py:21
MIN_SV = 50                 # both axes: a gap of at least this many bases is an event (lgsv.py:173, 261, 351)
CALL_SOURCE = 'ALNTRUNC'
CALL_SOURCE_INV_DENSITY = 'ALNTRUNC-DEN'
CALL_SOURCE_INV_NO_DENSITY = 'ALNTRUNC-NODEN'

INSDEL_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI', 'ALIGN_INDEX', 'LEFT_SHIFT',
                  'HOM_REF', 'HOM_TIG', 'CALL_SOURCE', 'FILTER', 'SEQ']
INV_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI', 'RGN_REF_INNER', 'RGN_QRY_INNER',
               'RGN_REF_DISC', 'RGN_QRY_DISC', 'FLAG_ID', 'FLAG_TYPE', 'ALIGN_INDEX', 'CALL_SOURCE', 'FILTER', 'SEQ']
_HOM_COL, _HOM_TIG_COL = INSDEL_COLUMNS.index('HOM_REF'), INSDEL_COLUMNS.index('HOM_TIG')
_REF, _TIG = _lib.PAV_ROLE_REF, _lib.PAV_ROLE_TIG
_LEFT, _RIGHT = 0, 1        # pav_hom_query.dir


def match_bp(record, right_end):
    """``pavlib.align.match_bp`` (align.py:325-361) compares the operation *characters* of ``cigar_str_to_tuples`` with BAM
    integer codes, so no operation ever counts as a match: the first one breaks the loop and the result is 0 for every
    record with a non-empty CIGAR.  The left shift of alignment-truncating INS / DEL is therefore always 0 in PAV 2.4.6;
    kept as is."""
    return 0


class _Group:
    """The records of one (chromosome, contig) pair in table order, as column arrays."""

    def __init__(self, df, rows, chrom, tig_id):
        self.chrom, self.tig_id, self.n = chrom, tig_id, len(rows)
        take = lambda col: df[col].to_numpy()[rows]            # noqa: E731
        self.pos, self.end = take('POS'), take('END')
        self.qpos, self.qend = take('QRY_POS'), take('QRY_END')
        self.rev, self.mapq, self.index = take('REV'), take('MAPQ'), take('INDEX')
        self.span = self.end - self.pos                        # what the reference keeps in its QRY_LEN column (lgsv.py:70)

    def label(self, i):
        return f'Index {self.index[i]} ({self.tig_id}:{self.qpos[i]}-{self.qend[i]})'

    def index_list(self, *members):
        return ','.join(str(self.index[i]) for i in members)


def _groups(df):
    """(chromosome, contig) pairs with more than one record, in order of first appearance (lgsv.py:89-90)."""
    if df.shape[0] == 0:
        return
    by_pair = df.groupby(['#CHROM', 'QRY_ID'], sort=False).indices
    for (chrom, tig_id), rows in by_pair.items():
        if len(rows) > 1:
            yield _Group(df, np.sort(rows), chrom, tig_id)


class _Gap:
    """What lies between record ``a`` and a later record ``b`` of the same strand: ``[q0, q1)`` on the contig (whichever of
    the two comes first there), ``dist_ref`` bases on the reference."""

    def __init__(self, g, a, b):
        first, second = (a, b) if g.qpos[a] < g.qpos[b] else (b, a)
        if g.qpos[second] < g.qend[first]:
            raise RuntimeError('Contig ranges overlap for two alignment records (should not occur after alignment trimming): '
                               f'{g.label(a)} and {g.label(b)}')
        self.q0, self.q1 = g.qend[first], g.qpos[second]
        self.dist_tig = self.q1 - self.q0
        self.dist_ref = g.pos[b] - g.end[a]
        if self.dist_tig < 0:
            raise RuntimeError('Contig query positions are out of order (program bug): Contig distance is negative '
                               f'({self.dist_tig}): {g.label(a)} and {g.label(b)}')
        self.shortest = np.min([g.span[a], g.span[b]])
        self.weakest = np.min([g.mapq[a], g.mapq[b]])

    def too_far(self, max_tig_prop, max_ref_prop):
        """Short or poorly mapped flanks do not support a gap that is large relative to them (lgsv.py:160-170)."""
        if self.shortest >= DIST_PROP_LEN_MAPQ[0] and self.weakest >= DIST_PROP_LEN_MAPQ[1]:
            return False
        return bool(np.abs(self.dist_tig) / self.shortest > max_tig_prop or np.abs(self.dist_ref) / self.shortest > max_ref_prop)

    def kind(self):
        big_ref, big_tig = bool(self.dist_ref >= MIN_SV), bool(self.dist_tig >= MIN_SV)
        return {(True, False): 'DEL', (False, True): 'INS', (True, True): 'INV'}.get((big_ref, big_tig))


class _Events:
    """INS / DEL rows (their two homology columns are filled in by ``resolve``) and INV rows of one table."""

    def __init__(self, ctx, hap, log, ref_fa_name, tig_fa_name, density_out_dir):
        self.ctx, self.hap, self.log = ctx, hap, log
        self.ref_fa_name, self.tig_fa_name, self.density_out_dir = ref_fa_name, tig_fa_name, density_out_dir
        self.ref_id, self.tig_id = inv._seq_index(ctx)
        self.rows = {'INS': [], 'DEL': []}
        self.order = []                                        # (svtype, row) per event, the order of their homology queries
        self.queries = []
        self.inv_rows = []
        self.inv_seen = set()                                  # the same inversion can be reached from several gaps

    def say(self, what, item):
        self.log.write(f'{what}: {item}\n')
        self.log.flush()

    def _query(self, where, pos, sv, direction):
        """Scan ``where`` = (role, record, oriented reverse?) from ``pos`` against the SV sequence ``sv`` = (role, record,
        oriented reverse?, start on the oriented record, length)."""
        ids = (self.ref_id, self.tig_id)
        self.queries.append((where[0], ids[where[0]][where[1]], int(bool(where[2])), 0, int(pos),
                             sv[0], ids[sv[0]][sv[1]], int(bool(sv[2])), 0, int(sv[3]), int(sv[4]), direction))

    def _indel(self, svtype, g, a, b, fields, ref_scan, tig_scan, sv):
        self.say(svtype, fields[3])
        on_ref, on_tig = (_REF, g.chrom, False), (_TIG, g.tig_id, g.rev[a])
        self._query(on_ref, ref_scan[0], sv, _LEFT)
        self._query(on_ref, ref_scan[1], sv, _RIGHT)
        self._query(on_tig, tig_scan[0], sv, _LEFT)            # contig coordinates as given, on the oriented contig
        self._query(on_tig, tig_scan[1], sv, _RIGHT)           # (pavlib/lgsv.py:237-238, 309-310)
        self.order.append(svtype)
        self.rows[svtype].append(fields)

    def deletion(self, g, a, b, gap):
        start, stop, at = g.end[a], g.pos[b], gap.q0
        shift = np.min([match_bp(None, True), 0])              # always 0: nothing is moved (see match_bp)
        bases = seq.region_seq_fasta(seq.Region(g.chrom, start, stop), self.ref_fa_name)
        fields = [g.chrom, start, stop, f'{g.chrom}-{start}-DEL-{gap.dist_ref}', 'DEL', gap.dist_ref, self.hap,
                  f'{g.tig_id}:{at + 1}-{at + 1}', '-' if g.rev[a] else '+', gap.dist_tig, g.index_list(a, b), shift, None, None,
                  CALL_SOURCE, 'PASS', bases]
        self._indel('DEL', g, a, b, fields, (start - 1, stop), (at - 1, at), (_REF, g.chrom, False, start, gap.dist_ref))

    def insertion(self, g, a, b, gap, tig_len):
        at = g.end[a]
        rev = g.rev[a]
        where = seq.Region(g.tig_id, gap.q0, gap.q1, is_rev=rev)
        shift = np.min([match_bp(None, True), 0])
        bases = seq.region_seq_fasta(where, self.tig_fa_name, rev_compl=rev)
        fields = [g.chrom, at, at + 1, f'{g.chrom}-{at}-INS-{gap.dist_tig}', 'INS', gap.dist_tig, self.hap, where.to_base1_string(),
                  '-' if rev else '+', gap.dist_ref, g.index_list(a, b), shift, None, None, CALL_SOURCE, 'PASS', bases]
        # the inserted bases in alignment orientation: contig[q0:q1], read from the other end when the record is reverse
        sv_start = tig_len - int(gap.q1) if rev else int(gap.q0)
        self._indel('INS', g, a, b, fields, (at - 1, at), (gap.q0 - 1, gap.q1), (_TIG, g.tig_id, bool(rev), sv_start, gap.dist_tig))

    def inversion(self, label, call, g, members, source):
        """Record ``call`` unless it was seen before.  Returns True when it was recorded."""
        if call is None or call.id in self.inv_seen:
            return False
        self.inv_seen.add(call.id)
        self.say(label, call)
        rev = g.rev[members[0]]
        outer = call.region_tig_outer
        self.inv_rows.append([
            call.region_ref_outer.chrom, call.region_ref_outer.pos, call.region_ref_outer.end, call.id, 'INV', call.svlen, self.hap,
            outer.to_base1_string(), '-' if rev else '+', 0, call.region_ref_inner.to_base1_string(),
            call.region_tig_inner.to_base1_string(), call.region_ref_discovery.to_base1_string(),
            call.region_tig_discovery.to_base1_string(), call.region_flag.region_id(), 'ALNTRUNC', g.index_list(*members), source,
            'PASS', seq.region_seq_fasta(outer, self.tig_fa_name, rev_compl=rev)])
        if self.density_out_dir is not None and call.df is not None:
            call.df.to_csv(os.path.join(self.density_out_dir, f'density_{call.id}_{self.hap}.tsv.gz'), sep='\t', index=False,
                           compression='gzip')
        return True

    def resolve(self):
        """One device call for the breakpoint homology of every INS / DEL, then the three tables."""
        if self.queries:
            q = np.array(self.queries, dtype=_lib.HOM_QUERY_DTYPE)
            hom = self.ctx.homology(q).reshape(-1, 4)
            cursor = {'INS': 0, 'DEL': 0}
            for svtype, (ref_l, ref_r, tig_l, tig_r) in zip(self.order, hom):
                row = self.rows[svtype][cursor[svtype]]
                cursor[svtype] += 1
                row[_HOM_COL], row[_HOM_TIG_COL] = f'{int(ref_l)},{int(ref_r)}', f'{int(tig_l)},{int(tig_r)}'
        return _table(self.rows['INS'], INSDEL_COLUMNS), _table(self.rows['DEL'], INSDEL_COLUMNS), _table(self.inv_rows, INV_COLUMNS)


def _table(rows, columns):
    """All-object frame in the reference's sort order (pd.concat(...).T of Series gives object columns, lgsv.py:561-636)."""
    if not rows:
        return pd.DataFrame([], columns=columns)
    out = pd.DataFrame({c: pd.Series([r[i] for r in rows], dtype=object) for i, c in enumerate(columns)}, columns=columns)
    out.sort_values(['#CHROM', 'POS', 'END', 'ID'], inplace=True)
    return out


def scan_for_events(df, df_tig_fai, hap, ref_fa_name, tig_fa_name, k_size, n_tree=None, threads=1, log=sys.stdout,
                    density_out_dir=None, max_tig_dist_prop=None, max_ref_dist_prop=None, srs_tree=None, max_region_size=None,
                    version_id=False, ctx=None, device_id=0):
    """Same arguments and return value as ``pavlib.lgsv.scan_for_events``: ``(df_ins, df_del, df_inv)``.  ``version_id``
    defaults to ``False`` here (``True`` in the reference, pavlib/lgsv.py:31-36, whose only caller passes ``False``,
    call_lg.snakefile:98): ``True`` raises, see below and INTEGRATION.md section 5."""
    if version_id:
        raise NotImplementedError('version_id=True needs svpoplib.variant.version_id (un-vendored submodule); '
                                  'rule call_lg_discover passes version_id=False (call_lg.snakefile:98)')
    if max_tig_dist_prop is None:
        max_tig_dist_prop = MAX_QRY_DIST_PROP
    if max_ref_dist_prop is None:
        max_ref_dist_prop = MAX_REF_DIST_PROP

    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        inv.ensure_sequences(ctx, ref_fa_name, tig_fa_name)
        lift_table = df.assign(QRY_LEN=df['END'] - df['POS'])      # the table AlignLift sees in the reference (lgsv.py:70-73)
        align_lift = AlignLift(lift_table, df_tig_fai)
        k_util = KmerUtil(k_size)
        events = _Events(ctx, hap, log, ref_fa_name, tig_fa_name, density_out_dir)

        def try_inversion(g, left, right):
            flagged = seq.Region(g.chrom, g.end[left], g.pos[right], is_rev=g.rev[left])
            return inv.scan_for_inv(flagged, ref_fa_name, tig_fa_name, align_lift, k_util, max_region_size=max_region_size,
                                    threads=threads, n_tree=n_tree, srs_tree=srs_tree, log=log, min_exp_count=1, ctx=ctx)

        def same_strand(g, a, b):
            """Returns how far the partner index moves on; 0 = record ``a`` is done."""
            gap = _Gap(g, a, b)
            if gap.too_far(max_tig_dist_prop, max_ref_dist_prop):
                return 1
            kind = gap.kind()
            if kind == 'DEL':
                events.deletion(g, a, b, gap)
                return 0
            if kind == 'INS':
                events.insertion(g, a, b, gap, int(df_tig_fai[g.tig_id]))
                return 0
            if kind == 'INV' and events.inversion('INV (2-tig)', try_inversion(g, a, b), g, (a, b), CALL_SOURCE_INV_DENSITY):
                return 0
            return 2                                               # nothing called: the reference advances twice here

        def flanked(g, a, b, c):
            """``b`` is on the other strand: an inversion if ``c`` continues ``a`` and ``b`` lies between them on the contig."""
            if g.rev[c] != g.rev[a]:
                return 1
            mid = (g.qpos[b] + g.qend[b]) // 2
            between = (g.qpos[c] < mid < g.qend[a]) if g.rev[a] else (g.qend[a] < mid < g.qpos[c])
            if not between:
                return 1
            call, source = try_inversion(g, a, c), CALL_SOURCE_INV_DENSITY
            if call is None and (b, c) == (a + 1, a + 2):
                # three consecutive records and no density support: the middle record itself is the call (lgsv.py:478-497)
                on_ref = seq.Region(g.chrom, g.pos[b], g.end[b])
                on_tig = seq.Region(g.tig_id, g.qpos[b], g.qend[b])
                call, source = inv.InvCall(on_ref, on_ref, on_tig, on_tig, on_ref, on_tig, on_ref, None), CALL_SOURCE_INV_NO_DENSITY
            return 0 if events.inversion('INV (3-tig)', call, g, (a, b, c), source) else 1

        for g in _groups(df):
            for a in range(g.n - 1):
                b = a + 1
                while b < g.n:
                    if g.rev[b] == g.rev[a]:
                        step = same_strand(g, a, b)
                    elif b + 1 < g.n:
                        step = flanked(g, a, b, b + 1)
                    else:
                        step = 1
                    if not step:
                        break
                    b += step
        return events.resolve()
    finally:
        if own:
            ctx.close()
