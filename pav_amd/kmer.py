"""
K-mer utility with the interface the hot path uses from ``kanapy.util.kmer.KmerUtil`` (k_size, rev_complement,
canonical_complement, to_string, to_kmer): rules/call_inv.snakefile:172, pavlib/inv.py:256,382,508-516,
scripts/density.py:173,524.

kanapy is not vendored in the reference snapshot; the integer encoding assumed here and
in the device kernels is A=0 C=1 G=2 T=3, two bits per base, first base most significant, reverse complement =
reversed order with every base XOR 3, canonical = numeric minimum.  A real kanapy ``KmerUtil`` can be passed to
``pav_amd.inv.scan_for_inv`` instead: only ``k_size`` and ``to_string`` are read from it.
"""

_I2B = 'ACGT'
_B2I = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'a': 0, 'c': 1, 'g': 2, 't': 3}


class KmerUtil:
    def __init__(self, k_size):
        self.k_size = int(k_size)
        self.k_bit_size = 2 * self.k_size
        self.k_mask = (1 << self.k_bit_size) - 1

    def rev_complement(self, kmer):
        kmer = int(kmer)
        rc = 0
        for _ in range(self.k_size):
            rc = (rc << 2) | ((kmer & 3) ^ 3)
            kmer >>= 2
        return rc

    def canonical_complement(self, kmer):
        rc = self.rev_complement(kmer)
        return int(kmer) if int(kmer) <= rc else rc

    def to_string(self, kmer):
        kmer = int(kmer)
        return ''.join(_I2B[(kmer >> (2 * (self.k_size - 1 - i))) & 3] for i in range(self.k_size))

    def to_kmer(self, s):
        kmer = 0
        for ch in s:
            kmer = (kmer << 2) | _B2I[ch]
        return kmer
