"""Oracle-harness stand-in for the absent `pysam` module (this container only).

Only what the reference's hot path touches: ``pysam.FastaFile(path).fetch(name, start, end)``
as a context manager (pavlib/cigarcall.py:59-66, pavlib/seq.py:339-351).  Written for this
repo's golden-vector generator; it is never shipped with, nor imported by, the product.
"""
import gzip


class FastaFile:
    _cache = {}

    def __init__(self, path):
        self.path = str(path)
        if self.path not in FastaFile._cache:
            FastaFile._cache[self.path] = self._read(self.path)
        self._seqs = FastaFile._cache[self.path]

    @staticmethod
    def _read(path):
        opener = gzip.open if path.endswith('.gz') else open
        seqs, name, chunks = {}, None, []
        with opener(path, 'rt') as fh:
            for line in fh:
                line = line.rstrip('\n')
                if line.startswith('>'):
                    if name is not None:
                        seqs[name] = ''.join(chunks)
                    name, chunks = line[1:].split()[0], []
                elif line:
                    chunks.append(line)
        if name is not None:
            seqs[name] = ''.join(chunks)
        return seqs

    @property
    def references(self):
        return list(self._seqs)

    def fetch(self, reference=None, start=None, end=None):
        seq = self._seqs[str(reference)]
        if start is None and end is None:
            return seq
        return seq[(0 if start is None else int(start)):(len(seq) if end is None else int(end))]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class AlignmentFile:  # imported by pavlib.align at module load; never used on the hot path
    def __init__(self, *a, **k):
        raise NotImplementedError('pysam.AlignmentFile is not available in the oracle harness')
