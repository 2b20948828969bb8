"""
Regions and FASTA region access - host mirror of the parts of ``pavlib/seq.py`` the inversion path calls
(``Region`` :20-258, ``region_from_string`` :260-285, ``region_from_id`` :288-302, ``region_seq_fasta`` :328-360).
Same coordinates, same string forms, same quirks (``region_id()`` is 0-based while ``region_from_id()`` assumes a
1-based ID: pavlib/seq.py:110,302).
"""

import re

import numpy as np
import pandas as pd

from .fasta import open_fasta

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b'ACGTRYSWKMBDHVNUacgtryswkmbdhvnu', b'TGCAYRSWMKVHDBNAtgcayrswmkvhdbna'):
    _COMP[_a] = _b


class Region:
    """0-based half-open region (BED) with orientation flag and optional alignment-record indexes."""

    def __init__(self, chrom, pos, end, is_rev=None, pos_min=None, pos_max=None, end_min=None, end_max=None,
                 pos_aln_index=None, end_aln_index=None):
        self.chrom = str(chrom)
        self.pos = int(pos)
        self.end = int(end)
        self.pos_min = self.pos if pos_min is None else int(pos_min)
        self.pos_max = self.pos if pos_max is None else int(pos_max)
        self.end_min = self.end if end_min is None else int(end_min)
        self.end_max = self.end if end_max is None else int(end_max)
        self.pos_aln_index = pos_aln_index
        self.end_aln_index = end_aln_index
        if self.pos > self.end:                       # reversed coordinates: swap and mark reverse (seq.py:54-70)
            self.pos, self.end = self.end, self.pos
            self.end_min = self.pos if pos_min is None else int(pos_min)
            self.end_max = self.pos if pos_max is None else int(pos_max)
            self.pos_min = self.end if end_min is None else int(end_min)
            self.pos_max = self.end if end_max is None else int(end_max)
            self.pos_aln_index, self.end_aln_index = self.end_aln_index, self.pos_aln_index
            if is_rev is None:
                is_rev = True
        self.is_rev = False if is_rev is None else is_rev

    def __repr__(self):
        return self.to_base1_string()

    def to_base1_string(self):
        return '{}:{}-{}'.format(self.chrom, self.pos + 1, self.end)

    def to_bed_string(self):
        return '{}\t{}\t{}'.format(self.chrom, self.pos + 1, self.end)     # sic (seq.py:92-96)

    def __len__(self):
        return self.end - self.pos

    def region_id(self):
        return '{}-{}-RGN-{}'.format(self.chrom, self.pos, self.end - self.pos)

    def expand(self, expand_bp, min_pos=0, max_end=None, shift=True, balance=0.5):
        """Grow by ``expand_bp`` split ``balance`` / ``1 - balance`` between the two ends, clipped to
        ``[min_pos, max_end[chrom]]`` and shifted to keep the size when ``shift`` (seq.py:112-188)."""
        if balance is None:
            balance = 0.5
        try:
            if not (0 <= balance <= 1):
                raise RuntimeError('balance must be in range [0, 1]: {}'.format(balance))
        except ValueError:
            raise RuntimeError('balance is not numeric: {}'.format(balance))
        expand_pos = int(expand_bp * balance)
        expand_end = np.max([0, expand_bp - expand_pos])
        new_pos = int(self.pos - expand_pos)
        new_end = int(self.end + expand_end)
        if min_pos is not None and new_pos < min_pos:
            if shift:
                new_end += min_pos - new_pos
            new_pos = min_pos
        if max_end is not None:
            if max_end.__class__ == pd.core.series.Series and self.chrom in max_end.index:
                max_end = max_end[self.chrom]
            else:
                max_end = None
        if max_end is not None and new_end > max_end:
            if shift:
                new_pos -= new_end - max_end
                if new_pos < min_pos:
                    new_pos = min_pos
            new_end = max_end
        if new_end < new_pos:
            new_end = new_pos = (new_end + new_pos) // 2
        self.pos = new_pos
        self.end = new_end
        self.pos_min = self.pos_max = self.pos
        self.end_min = self.end_max = self.end

    def __getitem__(self, key):
        if key not in {'chrom', 'pos', 'pos1', 'end'}:
            raise IndexError('No key in Region: {}'.format(key))
        return self.pos + 1 if key == 'pos1' else self.__dict__[key]

    def __eq__(self, other):
        return self.chrom == other.chrom and self.pos == other.pos and self.end == other.end

    def __lt__(self, other):
        return (self.chrom, self.pos, self.end) < (other.chrom, other.pos, other.end)

    def copy(self):
        return Region(self.chrom, self.pos, self.end, self.is_rev, self.pos_min, self.pos_max, self.end_min, self.end_max)


def region_from_string(rgn_str, is_rev=None, base0half=False):
    """"chrom:pos-end" (1-based closed unless ``base0half``) -> Region (seq.py:260-285)."""
    match_obj = re.match(r'^([^:]+):(\d+)-(\d+)$', rgn_str.replace(',', ''))
    if match_obj is None:
        raise RuntimeError('Region is not in expected format (chrom:pos-end): {}'.format(rgn_str))
    pos = int(match_obj[2])
    end = int(match_obj[3])
    if not base0half:
        pos -= 1
    return Region(match_obj[1], pos, end, is_rev=is_rev)


def region_from_id(region_id):
    """CHROM-POS-SVTYPE-LEN -> Region, POS taken as 1-based (seq.py:288-302)."""
    tok = region_id.split('-')
    if len(tok) != 4:
        raise RuntimeError('Unrecognized region ID: {}'.format(region_id))
    return Region(tok[0], int(tok[1]) - 1, int(tok[1]) - 1 + int(tok[3]))


def region_seq_fasta(region, fa_file_name, rev_compl=None):
    """Sequence of a Region (or of a whole record when ``region`` is a str); reverse-complemented when
    ``rev_compl`` or, if that is None, when ``region.is_rev`` (seq.py:328-360)."""
    fa = open_fasta(fa_file_name)
    if region.__class__ == str:
        arr, is_region = fa[region], False
    elif region.__class__ == Region:
        arr, is_region = fa[region.chrom][region.pos:region.end], True
    else:
        raise RuntimeError('Unrecognized region type: {}: Expected Region (pavlib.seq) or str'.format(str(region.__class__.__name__)))
    do_rc = (is_region and region.is_rev) if rev_compl is None else bool(rev_compl)
    if do_rc:
        arr = _COMP[arr[::-1]]
    return arr.tobytes().decode()
