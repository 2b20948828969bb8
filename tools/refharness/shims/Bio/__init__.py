"""Oracle-harness stand-in for Biopython (only Bio.Seq.Seq.reverse_complement is functional)."""
from . import Seq, SeqIO, bgzf  # noqa: F401
