"""CPU oracle of the EPSM manifold-gradient hot path -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``epsm_mitsuba3_amd``) never does.
"""
