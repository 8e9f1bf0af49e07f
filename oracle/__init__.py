"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``ziragroundingdino_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
