"""K-mer helpers with the published kanapy behaviour assumed by every call site in the reference.

A=0 C=1 G=2 T=3, two bits per base, first base most significant; reverse complement = reversed
order with each base XOR 3; canonical = numeric min(kmer, rc); ``stream`` restarts after any
non-ACGT character (either case) and, with ``index=True``, yields the 0-based offset of the
k-mer's first base (pavlib/inv.py:380-389 adds k to run ends).  Parity of the numeric KMER column
and of MATCH is therefore *unpinned* against real kanapy (its source is absent from the snapshot).
"""
_B2I = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'a': 0, 'c': 1, 'g': 2, 't': 3}
_I2B = 'ACGT'


class KmerUtil:
    def __init__(self, k_size):
        self.k_size = int(k_size)
        self.k_bit_size = 2 * self.k_size
        self.k_mask = (1 << self.k_bit_size) - 1

    def append(self, kmer, base):
        return ((kmer << 2) | _B2I[base]) & self.k_mask

    def rev_complement(self, kmer):
        rc = 0
        for _ in range(self.k_size):
            rc = (rc << 2) | ((kmer & 3) ^ 3)
            kmer >>= 2
        return rc

    def canonical_complement(self, kmer):
        rc = self.rev_complement(kmer)
        return kmer if kmer <= rc else rc

    def to_string(self, kmer):
        return ''.join(_I2B[(kmer >> (2 * (self.k_size - 1 - i))) & 3] for i in range(self.k_size))

    def to_kmer(self, s):
        kmer = 0
        for ch in s:
            kmer = (kmer << 2) | _B2I[ch]
        return kmer


def stream(seq, k_util, index=False):
    k, mask = k_util.k_size, k_util.k_mask
    kmer, load = 0, 0
    for i, ch in enumerate(seq):
        code = _B2I.get(ch)
        if code is None:
            kmer, load = 0, 0
            continue
        kmer = ((kmer << 2) | code) & mask
        load += 1
        if load >= k:
            yield (kmer, i - k + 1) if index else kmer
