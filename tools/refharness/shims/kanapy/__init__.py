"""Oracle-harness stand-in for the un-vendored kanapy k-mer utilities (semantics: SURVEY.md section 8(c))."""
from . import util, plot  # noqa: F401
