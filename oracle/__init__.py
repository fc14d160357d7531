"""ORACLE package — CPU restatements of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under ``edgedict_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and there
only as the checker / the timed CPU baseline, never as the thing shipped.
"""
