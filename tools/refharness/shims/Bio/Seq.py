"""IUPAC-aware, case-preserving reverse complement (what pavlib/cigarcall.py:70 and pavlib/seq.py:355-358 use)."""
_COMP = str.maketrans('ACGTRYSWKMBDHVNUacgtryswkmbdhvnu', 'TGCAYRSWMKVHDBNAtgcayrswmkvhdbna')


class Seq:
    def __init__(self, data):
        self._data = str(data)

    def reverse_complement(self):
        return Seq(self._data.translate(_COMP)[::-1])

    def __str__(self):
        return self._data

    def __len__(self):
        return len(self._data)
