"""Alignment helpers on the host side of the hot path (mirror of the used parts of ``pavlib/align``)."""
from .lift import AlignLift  # noqa: F401
from .cigar import cigar_str_to_tuples, tokenize  # noqa: F401
from .trim import trim_alignments, trim_alignment_record  # noqa: F401
from .ingest import get_align_bed  # noqa: F401
