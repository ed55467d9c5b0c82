"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the DASpeech hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; the product (``daspeech_amd``) never does.
"""
