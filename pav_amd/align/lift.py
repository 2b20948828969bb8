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
        if ctx is not None and df.shape[0]:
            cig = [str(c).encode() for c in df['CIGAR']]
            off = np.zeros(len(cig) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(c) for c in cig], dtype=np.uint64)
            ops, op_off, sub_b, qry_b = ctx.align_index(df['POS'].to_numpy(dtype=np.uint32),
                                                         np.frombuffer(b''.join(cig), dtype=np.uint8), off)
            self._dev = (ops, op_off.astype(np.int64), sub_b, qry_b, {ix: i for i, ix in enumerate(df.index)})
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
        self.cache_queue = collections.deque()
        self.ref_cache = dict()
        self.tig_cache = dict()

    def _containing(self, tree, key, pos):
        """Indexes of the records of ``key`` whose [begin, end) contains ``pos`` (sorted starts + bisect)."""
        records = tree.get(key)
        if not records:
            return []
        starts = self._starts[id(tree)][key]
        hi = bisect.bisect_right(starts, pos)
        lo = bisect.bisect_left(starts, pos - self._maxlen[id(tree)][key])
        return [records[i][2] for i in range(lo, hi) if records[i][1] > pos]

    # ---- query -> subject (lift.py:51-185) -------------------------------------------------------------
    def lift_to_sub(self, query_id, coord, gap=False):
        ret_list = issubclass(coord.__class__, (list, tuple))
        if not ret_list:
            coord = (coord,)
        out = []
        for pos in coord:
            pos_org = pos
            hits = self._containing(self.tig_tree, query_id, pos)
            if len(hits) == 1:
                index = hits[0]
            elif len(hits) == 0 and gap:
                out.append(self._get_subject_gap(query_id, pos))
                continue
            else:
                out.append(None)
                continue
            if index not in self.tig_cache:
                self._add_align(index)
            tree = self.tig_cache[index]
            row = self._rows[index]
            if row['REV']:
                pos = self.df_fai[query_id] - pos
            i = tree.at(pos)
            if i < 0:
                i = tree.at(pos - 1)                    # a query exactly at the alignment end (lift.py:122-138)
                if i < 0 or tree.interval(i)[1] != pos:
                    raise RuntimeError((
                        'Found no matches in a lift-tree for a record within a '
                        'global to-subject tree: {}:{} (index={}, gap={})'
                    ).format(query_id, pos_org, index, gap))
            begin, _, d0, d1 = tree.interval(i)
            lift_pos = d0 + (int(pos) - begin) if d1 - d0 > 1 else d1
            out.append((row['#CHROM'], lift_pos, row['REV'], lift_pos, lift_pos, (row['INDEX'],)))
        return out if ret_list else out[0]

    # ---- subject -> query (lift.py:187-272) ------------------------------------------------------------
    def lift_to_qry(self, subject_id, coord):
        ret_list = issubclass(coord.__class__, (list, tuple))
        if not ret_list:
            coord = (coord,)
        out = []
        for pos in coord:
            hits = self._containing(self.ref_tree, subject_id, pos)
            if len(hits) != 1:
                out.append(None)
                continue
            index = hits[0]
            if index not in self.ref_cache:
                self._add_align(index)
            tree = self.ref_cache[index]
            row = self._rows[index]
            i = tree.at(pos)
            if i < 0:
                raise RuntimeError((
                    'Program bug: Found no matches in a lift-tree for a record withing a '
                    'global to-query tree: {}:{} (index={})'
                ).format(subject_id, pos, index))
            begin, _, d0, d1 = tree.interval(i)
            qry_pos = d0 + (int(pos) - begin) if d1 - d0 > 1 else d1
            if row['REV']:
                qry_pos = self.df_fai[row['QRY_ID']] - qry_pos
            out.append((row['QRY_ID'], qry_pos, row['REV'], qry_pos, qry_pos, (row['INDEX'],)))
        return out if ret_list else out[0]

    def lift_region_to_sub(self, region, gap=False):
        """Query region -> subject region, or None (lift.py:274-302)."""
        sub_pos, sub_end = self.lift_to_sub(region.chrom, (region.pos, region.end), gap)
        if sub_pos is None or sub_end is None:
            return None
        if sub_pos[0] != sub_end[0] or (sub_pos[2] is not None and sub_end[2] is not None and sub_pos[2] != sub_end[2]):
            return None
        return pavseq.Region(sub_pos[0], sub_pos[1], sub_end[1], is_rev=False, pos_min=sub_pos[3], pos_max=sub_pos[4],
                             end_min=sub_end[3], end_max=sub_end[4], pos_aln_index=(sub_pos[5],), end_aln_index=(sub_end[5],))

    def lift_region_to_qry(self, region):
        """Subject region -> query region, or None (lift.py:304-331)."""
        query_pos, query_end = self.lift_to_qry(region.chrom, (region.pos, region.end))
        if query_pos is None or query_end is None:
            return None
        if query_pos[0] != query_end[0] or query_pos[2] != query_end[2]:
            return None
        return pavseq.Region(query_pos[0], query_pos[1], query_end[1], is_rev=query_pos[2], pos_min=query_pos[3],
                             pos_max=query_pos[4], end_min=query_end[3], end_max=query_end[4],
                             pos_aln_index=(query_pos[5],), end_aln_index=(query_end[5],))

    def _get_subject_gap(self, query_id, pos):
        """Interpolate into the gap between two records of one contig (lift.py:333-378)."""
        if pos is None:
            return None
        subdf = self.df.loc[self.df['QRY_ID'] == query_id]
        if not np.any(subdf['QRY_END'] < pos) or not np.any(subdf['QRY_POS'] > pos):
            return None
        row_l = subdf.loc[subdf.loc[subdf['QRY_END'] < pos, 'QRY_END'].sort_values().index[-1]]
        row_r = subdf.loc[subdf.loc[subdf['QRY_POS'] > pos, 'QRY_POS'].sort_values().index[0]]
        if row_l['#CHROM'] != row_r['#CHROM']:
            return None
        return (row_l['#CHROM'], int((row_l['QRY_END'] + row_r['QRY_POS']) / 2),
                row_l['REV'] if row_l['REV'] == row_r['REV'] else None, row_l['QRY_END'], row_r['QRY_POS'],
                (row_l['INDEX'], row_r['INDEX']))

    def _add_align(self, index):
        """Build and cache the two operation tables of one record (lift.py:380-476)."""
        if index in self.ref_cache:
            while index in self.cache_queue:
                self.cache_queue.remove(index)
            self.cache_queue.appendleft(index)
            return
        self._check_and_clear()
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
        self.ref_cache[index] = table.view(0)
        self.tig_cache[index] = table.view(1)
        self.cache_queue.appendleft(index)

    def _check_and_clear(self):
        # The reference keeps 10 records (lift.py:20-49); eviction has no observable effect, so keep many more
        while len(self.cache_queue) >= max(self.cache_align, 4096):
            index = self.cache_queue.pop()
            del self.ref_cache[index]
            del self.tig_cache[index]
