"""
Alignment lift-over between reference and contig coordinates - host mirror of ``pavlib/align/lift.py``.

The reference builds two interval trees per alignment record (pavlib/align/lift.py:380-476) and queries them with
point lookups.  The CIGAR operations tile both coordinate axes, so the same answers come from sorted operation
boundaries + binary search; the tables are built with numpy from a vectorised tokenizer.  All quirks are kept:
intervals whose *target* span is 1 (length-1 match operations, insertions, deletions) lift to ``data[1]``
(lift.py:163-181, 257-258), a query exactly at an alignment end matches the last interval (lift.py:122-138),
reversed records translate through ``QRY_LEN - pos`` (lift.py:116, 260).
"""

import bisect
import collections

import numpy as np

from .. import seq as pavseq
from .cigar import tokenize

_OP_TO_CODE = np.full(256, 15, dtype=np.int64)
for _i, _c in enumerate(b'MIDNSHP=X'):
    _OP_TO_CODE[_c] = _i


_MATCH_CODES = np.zeros(16, dtype=bool)
_MATCH_CODES[[0, 7, 8]] = True            # M, =, X


class _OpTable:
    """One alignment record's operations as parallel arrays (BAM codes, lengths, start on both axes) with the point
    lookup the reference performs on its per-record interval trees.  ``axis`` 0 = subject, 1 = query."""

    def __init__(self, ops, sub_begin, qry_begin):
        self.code = (ops & 15).astype(np.int64)
        self.len = (ops >> 4).astype(np.int64)
        self.begin = (sub_begin.astype(np.int64), qry_begin.astype(np.int64))

    def view(self, axis):
        return _AxisView(self, axis)


class _AxisView:
    """Intervals of one axis: subject axis = M/=/X and D operations, query axis = M/=/X and I operations
    (pavlib/align/lift.py:437-461).  Payload (d0, d1) = span on the other axis, or (x, x + 1) for I / D."""

    def __init__(self, table, axis):
        self.t, self.axis = table, axis
        self.gap_code = 2 if axis == 0 else 1

    def at(self, pos):
        """Index of the operation whose interval contains ``pos`` or -1."""
        t = self.t
        b = t.begin[self.axis]
        k = int(np.searchsorted(b, pos, side='right')) - 1
        if k < 0:
            return -1
        c = t.code[k]
        if not (_MATCH_CODES[c] or c == self.gap_code):
            return -1
        return k if pos < b[k] + t.len[k] else -1

    def interval(self, k):
        """(begin, end, d0, d1) of operation k."""
        t = self.t
        begin = int(t.begin[self.axis][k])
        ln = int(t.len[k])
        o = int(t.begin[1 - self.axis][k])
        return begin, begin + ln, o, (o + ln if _MATCH_CODES[t.code[k]] else o + 1)


class AlignLift:
    """Lift coordinates through alignment records (same constructor and methods as pavlib.align.AlignLift)."""

    def __init__(self, df, df_fai, cache_align=10, ctx=None):
        """``ctx``: optional :class:`pav_amd._lib.Context`; the CIGARs of all rows are then tokenised and prefix-scanned
        on the GPU in one call (``pav_align_index``) instead of per record on first use."""
        self.df = df
        self.df_fai = df_fai
        self.cache_align = cache_align
        self._dev = None
        self._dev_ctx = ctx if df.shape[0] else None          # the index is built when the first lift through this object needs it:
                                                                # the native scan driver keeps its own and never asks
        if len(set(df.index)) != df.shape[0]:
            raise RuntimeError('Cannot create AlignLift object with duplicate index values')
        # per subject / per query: list of (begin, end, index); lookups need "exactly one record contains pos"
        self.ref_tree = collections.defaultdict(list)
        self.tig_tree = collections.defaultdict(list)
        self._rows = {}
        cols = ['#CHROM', 'POS', 'END', 'INDEX', 'QRY_ID', 'QRY_POS', 'QRY_END', 'REV']
        for index, vals in zip(df.index, zip(*[df[c].tolist() for c in cols])):
            row = dict(zip(cols, vals))
            self._rows[index] = row
            if row['END'] > row['POS']:
                self.ref_tree[row['#CHROM']].append((row['POS'], row['END'], index))
            if row['QRY_END'] > row['QRY_POS']:
                self.tig_tree[row['QRY_ID']].append((row['QRY_POS'], row['QRY_END'], index))
        for tree in (self.ref_tree, self.tig_tree):
            for key in tree:
                tree[key].sort(key=lambda t: (t[0], t[1]))
        self._starts = {id(tree): {k: [t[0] for t in v] for k, v in tree.items()} for tree in (self.ref_tree, self.tig_tree)}
        self._maxlen = {id(tree): {k: max(t[1] - t[0] for t in v) for k, v in tree.items()} for tree in (self.ref_tree, self.tig_tree)}
        self._tables = collections.OrderedDict()           # index -> (subject-axis view, query-axis view), least recent first
        self.ref_cache = _AxisCache(self._tables, 0)
        self.tig_cache = _AxisCache(self._tables, 1)

    def _containing(self, tree, key, pos):
        """Indexes of the records of ``key`` whose [begin, end) contains ``pos`` (sorted starts + bisect)."""
        records = tree.get(key)
        if not records:
            return []
        starts = self._starts[id(tree)][key]
        hi = bisect.bisect_right(starts, pos)
        lo = bisect.bisect_left(starts, pos - self._maxlen[id(tree)][key])
        return [records[i][2] for i in range(lo, hi) if records[i][1] > pos]

    # ---- point lifts ------------------------------------------------------------------------------------------
    # A lifted point is (sequence, position, record is reverse, lowest, highest position, record indexes).
    @staticmethod
    def _each(coord, lift_one):
        if isinstance(coord, (list, tuple)):
            return [lift_one(p) for p in coord]
        return lift_one(coord)

    @staticmethod
    def _through(view, k, at):
        """Position ``at`` inside operation ``k`` of ``view`` carried to the other axis: operations that cover one
        position there (insertions, deletions, one-base matches) give that payload's end (lift.py:163-181, 257-258)."""
        begin, _, d0, d1 = view.interval(k)
        return d0 + (int(at) - begin) if d1 - d0 > 1 else d1

    def lift_to_sub(self, query_id, coord, gap=False):
        """Contig position(s) -> subject (pavlib/align/lift.py:51-185): the one record containing the position carries it
        over; no record and ``gap``: interpolated between the flanking records; otherwise None."""
        def one(pos):
            owners = self._containing(self.tig_tree, query_id, pos)
            if len(owners) != 1:
                return self._get_subject_gap(query_id, pos) if (gap and not owners) else None
            index = owners[0]
            self._add_align(index)
            view, rec = self.tig_cache[index], self._rows[index]
            at = self.df_fai[query_id] - pos if rec['REV'] else pos      # reverse records count from the other contig end
            k = view.at(at)
            if k < 0:                                                    # exactly at the alignment end (lift.py:122-138)
                k = view.at(at - 1)
                if k < 0 or view.interval(k)[1] != at:
                    raise RuntimeError((
                        'Found no matches in a lift-tree for a record within a '
                        'global to-subject tree: {}:{} (index={}, gap={})'
                    ).format(query_id, pos, index, gap))
            there = self._through(view, k, at)
            return rec['#CHROM'], there, rec['REV'], there, there, (rec['INDEX'],)
        return self._each(coord, one)

    def lift_to_qry(self, subject_id, coord):
        """Subject position(s) -> contig (pavlib/align/lift.py:187-272); None unless exactly one record contains it."""
        def one(pos):
            owners = self._containing(self.ref_tree, subject_id, pos)
            if len(owners) != 1:
                return None
            index = owners[0]
            self._add_align(index)
            view, rec = self.ref_cache[index], self._rows[index]
            k = view.at(pos)
            if k < 0:
                raise RuntimeError((
                    'Program bug: Found no matches in a lift-tree for a record withing a '
                    'global to-query tree: {}:{} (index={})'
                ).format(subject_id, pos, index))
            there = self._through(view, k, pos)
            if rec['REV']:
                there = self.df_fai[rec['QRY_ID']] - there
            return rec['QRY_ID'], there, rec['REV'], there, there, (rec['INDEX'],)
        return self._each(coord, one)

    @staticmethod
    def _span(first, last, is_rev):
        """Region between two lifted points ((name, pos, rev, lo, hi, records) each) on the same sequence."""
        if first is None or last is None or first[0] != last[0]:
            return None
        (name, pos, _, pos_lo, pos_hi, pos_rec), (_, end, _, end_lo, end_hi, end_rec) = first, last
        return pavseq.Region(name, pos, end, is_rev=is_rev, pos_min=pos_lo, pos_max=pos_hi, end_min=end_lo, end_max=end_hi,
                             pos_aln_index=(pos_rec,), end_aln_index=(end_rec,))

    def lift_region_to_sub(self, region, gap=False):
        """Query region -> subject region (never marked reverse), or None when an end does not lift, the ends land on
        different subjects, or both orientations are known and disagree (pavlib/align/lift.py:274-302)."""
        first, last = self.lift_to_sub(region.chrom, (region.pos, region.end), gap)
        if first is not None and last is not None and None not in (first[2], last[2]) and first[2] != last[2]:
            return None
        return self._span(first, last, False)

    def lift_region_to_qry(self, region):
        """Subject region -> query region carrying the record's orientation, or None when an end does not lift or the ends
        land on different contigs / orientations (pavlib/align/lift.py:304-331)."""
        first, last = self.lift_to_qry(region.chrom, (region.pos, region.end))
        if first is None or last is None or first[2] != last[2]:
            return None
        return self._span(first, last, first[2])

    def _get_subject_gap(self, query_id, pos):
        """A contig position no record covers, placed between the record that ends last before it and the one that starts
        first after it - if both exist and sit on the same subject (pavlib/align/lift.py:333-378).  The position reported
        is the midpoint of the contig gap; orientation only when both neighbours agree."""
        if pos is None:
            return None
        members = np.flatnonzero(self.df['QRY_ID'].to_numpy() == query_id)
        ends, starts = self.df['QRY_END'].to_numpy()[members], self.df['QRY_POS'].to_numpy()[members]
        before, after = np.flatnonzero(ends < pos), np.flatnonzero(starts > pos)
        if before.size == 0 or after.size == 0:
            return None
        # the same (unstable) sort the reference's Series.sort_values() uses decides between equal coordinates
        left = self.df.iloc[members[before[np.argsort(ends[before], kind='quicksort')[-1]]]]
        right = self.df.iloc[members[after[np.argsort(starts[after], kind='quicksort')[0]]]]
        if left['#CHROM'] != right['#CHROM']:
            return None
        gap_from, gap_to = left['QRY_END'], right['QRY_POS']
        return (left['#CHROM'], int((gap_from + gap_to) / 2), left['REV'] if left['REV'] == right['REV'] else None,
                gap_from, gap_to, (left['INDEX'], right['INDEX']))

    def _add_align(self, index):
        """Make the operation tables of one record available (the job of pavlib/align/lift.py:380-476); most recently used
        records stay at the end of ``_tables``."""
        if index in self._tables:
            self._tables.move_to_end(index)
            return
        self._check_and_clear()
        if self._dev is None and self._dev_ctx is not None:
            df, ctx, self._dev_ctx = self.df, self._dev_ctx, None
            cig = [str(c).encode() for c in df['CIGAR']]
            off = np.zeros(len(cig) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(c) for c in cig], dtype=np.uint64)
            if getattr(ctx, 'handle', None):
                # a device or library failure here (out of memory, PAV_E_LIMIT, a HIP fault) is raised, not papered over
                ops, op_off, sub_b, qry_b = ctx.align_index(df['POS'].to_numpy(dtype=np.uint32),
                                                             np.frombuffer(b''.join(cig), dtype=np.uint8), off)
                self._dev = (ops, op_off.astype(np.int64), sub_b, qry_b, {ix: i for i, ix in enumerate(df.index)})
            # (a context that has been closed since this object was made: the host tokenizer below gives the same tables,
            #  record by record)
        if self._dev is not None:
            ops_all, op_off, sub_all, qry_all, where = self._dev
            r = where[index]
            sl = slice(int(op_off[r]), int(op_off[r + 1]))
            ops, sub_b, qry_b = ops_all[sl], sub_all[sl], qry_all[sl]
        else:                                           # host tokenizer (no GPU context given)
            row = self.df.loc[index]
            lens, opc = tokenize(row['CIGAR'])
            code = _OP_TO_CODE[opc]
            ops = (lens.astype(np.uint32) << 4) | code.astype(np.uint32)
            match = _MATCH_CODES[code]
            sub_adv = np.where(match | (code == 2), lens, 0)
            qry_adv = np.where(match | (code == 1) | (code == 4) | (code == 5), lens, 0)
            sub_b = int(row['POS']) + np.concatenate(([0], np.cumsum(sub_adv)[:-1]))
            qry_b = np.concatenate(([0], np.cumsum(qry_adv)[:-1]))
        table = _OpTable(ops, sub_b, qry_b)
        bad = (table.code == 3) | (table.code == 6)
        if np.any(bad):
            row = self._rows[index]
            raise RuntimeError('Unhandled CIGAR operation: {}: Alignment {}:{} ({})'.format(
                'NP'[int(table.code[np.flatnonzero(bad)[0]] == 6)], row['#CHROM'], row['POS'], row['QRY_ID']))
        if np.any((table.len == 0) & (_MATCH_CODES[table.code] | (table.code == 1) | (table.code == 2))):
            raise ValueError('IntervalTree: Null Interval objects not allowed in IntervalTree: zero-length CIGAR operation')
        self._tables[index] = (table.view(0), table.view(1))

    def _check_and_clear(self):
        """Drop the least recently used tables.  The reference keeps ``cache_align`` = 10 records; a table here is three
        small arrays and eviction has no observable effect, so at least 4096 stay."""
        keep = max(self.cache_align, 4096)
        while len(self._tables) >= keep:
            self._tables.popitem(last=False)

    @property
    def cache_queue(self):
        """Cached record indexes, most recently used first (the reference's deque, for introspection)."""
        return collections.deque(reversed(self._tables))


class _AxisCache:
    """``ref_cache`` / ``tig_cache`` of the reference as read-only views of the table cache (one axis each)."""

    def __init__(self, tables, axis):
        self._tables, self._axis = tables, axis

    def __contains__(self, index):
        return index in self._tables

    def __getitem__(self, index):
        return self._tables[index][self._axis]

    def __len__(self):
        return len(self._tables)

    def keys(self):
        return self._tables.keys()
