from . import kmer  # noqa: F401
