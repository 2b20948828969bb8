"""
Regions and FASTA region access on the host - the interface of the parts of ``pavlib/seq.py`` the inversion path calls
(``Region`` :20-258, ``region_from_string`` :260-285, ``region_from_id`` :288-302, ``region_seq_fasta`` :328-360), written
for this package: coordinates are plain Python integers, expansion is interval arithmetic on a (lo, hi) pair, sequences are
byte arrays from the native FASTA reader.

Behaviour that callers (and the golden vectors, tests/golden/region_kat.json) depend on and that is therefore kept:
  * a region given with pos > end is stored swapped, marked reverse, and its uncertainty bounds end up crossed
    (pavlib/seq.py:54-66);
  * ``expand`` only honours ``max_end`` when it is a pandas Series holding the chromosome (pavlib/seq.py:157-162);
  * ``region_id()`` prints the 0-based start while ``region_from_id()`` reads the number as 1-based (:110, :302);
  * ``to_bed_string()`` prints ``pos + 1`` (:92-96).
"""

import re

import numpy as np
import pandas as pd

from .fasta import open_fasta

# IUPAC complement table for byte arrays (Bio.Seq.reverse_complement semantics: case kept, U -> A, others unchanged)
_COMP = np.arange(256, dtype=np.uint8)
_COMP[list(b'ACGTRYSWKMBDHVNUacgtryswkmbdhvnu')] = list(b'TGCAYRSWMKVHDBNAtgcayrswmkvhdbna')

_REGION_TEXT = re.compile(r'(?P<chrom>[^:]+):(?P<first>\d+)-(?P<last>\d+)$')
_ITEM_KEYS = ('chrom', 'pos', 'pos1', 'end')


def _bound(value, default):
    return default if value is None else int(value)


class Region:
    """Half-open interval ``[pos, end)`` on ``chrom`` (BED coordinates) with an orientation flag, optional
    uncertainty bounds for both ends, and the alignment-record indexes the ends were lifted through."""

    def __init__(self, chrom, pos, end, is_rev=None, pos_min=None, pos_max=None, end_min=None, end_max=None,
                 pos_aln_index=None, end_aln_index=None):
        lo, hi = int(pos), int(end)
        flipped = lo > hi
        if flipped:
            # Stored in ascending order.  The bounds follow the reference's bookkeeping: each end's bounds come from the
            # arguments of the *other* end and default to the other end's coordinate.
            lo, hi = hi, lo
            pos_min, pos_max, end_min, end_max = end_min, end_max, pos_min, pos_max
            pos_aln_index, end_aln_index = end_aln_index, pos_aln_index
        self.chrom, self.pos, self.end = str(chrom), lo, hi
        near, far = (hi, lo) if flipped else (lo, hi)
        self.pos_min, self.pos_max = _bound(pos_min, near), _bound(pos_max, near)
        self.end_min, self.end_max = _bound(end_min, far), _bound(end_max, far)
        self.pos_aln_index, self.end_aln_index = pos_aln_index, end_aln_index
        self.is_rev = flipped if is_rev is None else is_rev

    # ---- text forms -----------------------------------------------------------------------------------------
    def to_base1_string(self):
        """``chrom:first-last``, 1-based closed (samtools / browser notation)."""
        return f'{self.chrom}:{self.pos + 1}-{self.end}'

    __repr__ = to_base1_string

    def to_bed_string(self):
        return '\t'.join((self.chrom, str(self.pos + 1), str(self.end)))

    def region_id(self):
        return f'{self.chrom}-{self.pos}-RGN-{len(self)}'

    # ---- container protocol ---------------------------------------------------------------------------------
    def __len__(self):
        return self.end - self.pos

    def __getitem__(self, key):
        if key not in _ITEM_KEYS:
            raise IndexError('No key in Region: {}'.format(key))
        return self.pos + 1 if key == 'pos1' else getattr(self, key)

    def _key(self):
        return self.chrom, self.pos, self.end

    def __eq__(self, other):
        return self._key() == (other.chrom, other.pos, other.end)

    def __lt__(self, other):
        return self._key() < (other.chrom, other.pos, other.end)

    __hash__ = None

    def copy(self):
        return Region(*self._key(), self.is_rev, self.pos_min, self.pos_max, self.end_min, self.end_max)

    # ---- expansion --------------------------------------------------------------------------------------------
    def expand(self, expand_bp, min_pos=0, max_end=None, shift=True, balance=0.5):
        """Grow the interval by ``expand_bp`` bases, ``int(expand_bp * balance)`` of them on the left.

        The result never leaves ``[min_pos, max_end[chrom]]``; with ``shift`` the part that would have crossed one limit
        is given to the other end (which is then clipped as well).  An interval that would turn inside out collapses to
        its midpoint.  Uncertainty bounds are reset to the new ends."""
        balance = 0.5 if balance is None else balance
        try:
            in_range = 0 <= balance <= 1
        except ValueError as not_a_number:
            raise RuntimeError('balance is not numeric: {}'.format(balance)) from not_a_number
        if not in_range:
            raise RuntimeError('balance must be in range [0, 1]: {}'.format(balance))
        left = int(expand_bp * balance)
        right = max(0, int(expand_bp - left))
        lo, hi = self.pos - left, self.end + right

        ceiling = None                                   # only a per-chromosome table is honoured
        if isinstance(max_end, pd.Series) and self.chrom in max_end.index:
            ceiling = max_end[self.chrom]

        if min_pos is not None and lo < min_pos:
            hi += (min_pos - lo) if shift else 0
            lo = min_pos
        if ceiling is not None and hi > ceiling:
            if shift:
                lo -= hi - ceiling
                if min_pos is not None:
                    lo = max(lo, min_pos)
            hi = ceiling
        if hi < lo:
            lo = hi = (lo + hi) // 2

        self.pos = self.pos_min = self.pos_max = int(lo)
        self.end = self.end_min = self.end_max = int(hi)


def region_from_string(rgn_str, is_rev=None, base0half=False):
    """Parse ``chrom:pos-end`` (thousands separators allowed).  The numbers are 1-based closed unless ``base0half``."""
    found = _REGION_TEXT.match(rgn_str.replace(',', ''))
    if found is None:
        raise RuntimeError('Region is not in expected format (chrom:pos-end): {}'.format(rgn_str))
    first, last = int(found['first']), int(found['last'])
    return Region(found['chrom'], first if base0half else first - 1, last, is_rev=is_rev)


def region_from_id(region_id):
    """``CHROM-POS-SVTYPE-LEN`` (e.g. an "RGN" ID) to a Region; POS is read as 1-based."""
    fields = region_id.split('-')
    if len(fields) != 4:
        raise RuntimeError('Unrecognized region ID: {}'.format(region_id))
    chrom, start1, _, length = fields
    start = int(start1) - 1
    return Region(chrom, start, start + int(length))


def region_seq_resident(ctx, role, region, rev_compl=None):
    """:func:`region_seq_fasta` for a record that is resident on ``ctx`` (``pav_seq_fetch``): the bases come from HBM, no FASTA
    file is parsed on the host for them."""
    names = ctx.seq_names(role)
    if type(region) is str:
        rec, flip = names.index(region), bool(rev_compl)
        bases = ctx.seq_fetch(role, rec, 0, ctx.seq_lengths(role)[rec])
    elif type(region) is Region:
        rec = names.index(region.chrom)
        bases = ctx.seq_fetch(role, rec, region.pos, region.end)
        flip = bool(region.is_rev if rev_compl is None else rev_compl)
    else:
        raise RuntimeError('Unrecognized region type: {}: Expected Region (pavlib.seq) or str'.format(type(region).__name__))
    if flip:
        bases = _COMP[bases[::-1]]
    return bases.tobytes().decode()


def region_seq_fasta(region, fa_file_name, rev_compl=None):
    """Bases of a Region - or of the whole record named by a str - as text.  Reverse-complemented when ``rev_compl`` is
    true, or, when it is None, for a Region with ``is_rev`` set."""
    records = open_fasta(fa_file_name)
    if type(region) is str:
        bases, flip = records[region], bool(rev_compl)
    elif type(region) is Region:
        bases = records[region.chrom][region.pos:region.end]
        flip = bool(region.is_rev if rev_compl is None else rev_compl)
    else:
        raise RuntimeError('Unrecognized region type: {}: Expected Region (pavlib.seq) or str'.format(type(region).__name__))
    if flip:
        bases = _COMP[bases[::-1]]
    return bases.tobytes().decode()
