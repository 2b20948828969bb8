"""The C-ABI library loads and exports every symbol include/pav_amd.h declares (no compute calls: CPU box)."""
import os
import re

from pav_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    with open(os.path.join(ROOT, 'include', 'pav_amd.h')) as fh:
        text = fh.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pav_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(built):
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} is declared in include/pav_amd.h but not exported'
    assert sorted(_lib.SYMBOLS) == declared, 'pav_amd/_lib.py prototypes out of sync with include/pav_amd.h'
    assert lib.pav_abi_version() == 2


def test_struct_sizes_match_header():
    assert _lib.SNV_DTYPE.itemsize == 16
    assert _lib.INDEL_DTYPE.itemsize == 64
    assert _lib.ALN_DTYPE.itemsize == 16


def test_no_cpu_fallback(built):
    """Without a GPU pav_create must fail loudly (PavDeviceError), never return a CPU context."""
    import pytest
    lib = _lib.load()
    if lib.pav_device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(_lib.PavDeviceError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """The product package must not import, link or execute anything under oracle/ (parity would be void)."""
    pat = re.compile(r'(^|\s)(import|from)\s+oracle\b|oracle/|libpavoracle|pav_oracle', re.M)
    pkg = os.path.join(ROOT, 'pav_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.c', '.cpp')):
                with open(os.path.join(dirpath, f)) as fh:
                    assert not pat.search(fh.read()), f'{os.path.join(dirpath, f)} reaches into oracle/'
