"""
pav_amd - MI355X (gfx950) native variant-calling core for PAV's hot path.

Host-side mirror of the two reference entry points:

* ``pav_amd.cigarcall.make_insdel_snv_calls``  <->  ``pavlib.cigarcall.make_insdel_snv_calls``
* ``pav_amd.inv.scan_for_inv``                 <->  ``pavlib.inv.scan_for_inv``

Both run on the GPU through ``libpav_amd.so`` (C ABI: include/pav_amd.h).  There is no CPU fallback.
"""

__version__ = '0.1.0'
