"""Make the read-only reference importable in THIS container (never on the GPU box).

Usage: ``import refenv; pavlib = refenv.import_pavlib()``.  Puts, in order, on sys.path and on
PYTHONPATH (pavlib/inv.py:249-266 spawns ``python3 scripts/density.py``, which inherits it):
  1. tools/refharness/shims      - stand-ins for pysam / Bio / svpoplib / kanapy (absent here)
  2. a scratch dir of symlinks to the pure-Python intervaltree 3.1.0 found under /opt/conda
  3. /root/reference             - the reference itself, imported unmodified
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SHIMS = os.path.join(HERE, 'shims')
REFERENCE = os.environ.get('PAV_REFERENCE', '/root/reference')
_CONDA_SP = '/opt/conda/lib/python3.9/site-packages'


def _third_party_dir():
    d = os.path.join(tempfile.gettempdir(), 'pav_refharness_3p')
    os.makedirs(d, exist_ok=True)
    link = os.path.join(d, 'intervaltree')
    if not os.path.exists(link):
        os.symlink(os.path.join(_CONDA_SP, 'intervaltree'), link)
    return d


def setup():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError(f'reference not present at {REFERENCE}: golden vectors can only be regenerated '
                           'in the build container')
    paths = [SHIMS, _third_party_dir(), REFERENCE]
    for p in reversed(paths):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['PYTHONPATH'] = os.pathsep.join(paths + [os.environ.get('PYTHONPATH', '')]).rstrip(os.pathsep)
    return paths


def import_pavlib():
    setup()
    import pavlib  # noqa: E402
    return pavlib
