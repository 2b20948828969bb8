"""Oracle-harness stand-in for the un-vendored svpoplib (dep/svpop is an empty submodule in the reference snapshot)."""
from . import ref, variant, seq, svmerge, vcf  # noqa: F401
