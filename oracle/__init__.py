"""CPU oracle -- TEST INFRASTRUCTURE, not product code.

Importable only from tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of bench.py.  The product package (``docker-montage-gan_amd``) never imports
this; on a GPU tensor its ops call libmontage_hip.so or raise.
Parity status: pinned by tests/golden (vectors produced by the reference's own
CPU implementations, generator committed as tests/golden/make_golden.py).
"""
