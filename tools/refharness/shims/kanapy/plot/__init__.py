from . import dotplot  # noqa: F401
