"""
CIGAR text on the host: a vectorised tokenizer for the lift-over tables and the reference's iterator interface.

``cigar_str_to_tuples`` keeps the contract of pavlib/align/align.py:286-322 (yields ``(int length, op char)``,
same RuntimeError texts); the variant caller itself tokenises on the GPU (csrc/cigar.hip).
"""

import numpy as np
import pandas as pd

_OPS = b'MIDNSHP=X'
_OP_CODE = np.full(256, -1, dtype=np.int8)
for _i, _c in enumerate(_OPS):
    _OP_CODE[_c] = _i


def tokenize(cigar):
    """-> (lengths int64[n], op bytes uint8[n]).  Raises like the reference for malformed text."""
    b = np.frombuffer(cigar.encode() if isinstance(cigar, str) else bytes(cigar), dtype=np.uint8)
    if b.size == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.uint8)
    is_digit = (b >= 48) & (b <= 57)
    op_pos = np.flatnonzero(~is_digit)
    if is_digit[-1]:
        raise IndexError('string index out of range')                  # cigar[len_pos] past the end (align.py:307)
    starts = np.concatenate(([0], op_pos[:-1] + 1))
    n_digits = op_pos - starts
    bad = np.flatnonzero(n_digits == 0)
    unk = np.flatnonzero(_OP_CODE[b[op_pos]] < 0)
    first_bad = bad[0] if bad.size else None
    first_unk = unk[0] if unk.size else None
    if first_bad is not None and (first_unk is None or first_bad <= first_unk):
        raise _TokError('missing', int(starts[first_bad]))
    if first_unk is not None:
        raise _TokError('unknown', int(starts[first_unk]))
    # value of each token: digits weighted by powers of ten (lengths < 2^53 stay exact in float64)
    d_idx = np.flatnonzero(is_digit)
    tok = np.searchsorted(op_pos, d_idx)
    power = op_pos[tok] - d_idx - 1
    vals = np.bincount(tok, weights=(b[d_idx] - 48) * np.power(10.0, power), minlength=op_pos.size)
    return vals.astype(np.int64), b[op_pos]


class _TokError(Exception):
    def __init__(self, kind, pos):
        super().__init__(kind)
        self.kind, self.pos = kind, pos


def cigar_str_to_tuples(record):
    """Iterator of ``(cigar-len, cigar-op)`` tuples for an alignment record or a CIGAR string."""
    cigar = record['CIGAR'] if type(record) == pd.Series else record
    try:
        lens, ops = tokenize(cigar)
    except _TokError as ex:
        if ex.kind == 'missing':
            raise RuntimeError('Missing length in CIGAR string for contig {} alignment starting at {}:{}: CIGAR index {}'.format(
                record['QRY_ID'], record['#CHROM'], record['POS'], ex.pos))
        raise RuntimeError('Unknown CIGAR operation for contig {} alignment starting at {}:{}: CIGAR operation {}'.format(
            record['QRY_ID'], record['#CHROM'], record['POS'], cigar[ex.pos]))
    for length, op in zip(lens.tolist(), ops.tobytes().decode()):
        yield (length, op)
