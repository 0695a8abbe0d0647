"""CPU oracle for the CFDBench FNO hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cfdbench_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do.  The shipped path is the HIP extension and fails loudly
when it is missing.

Parity pinning: the reference (luo-yining/CFDBench) ships no tests and no
golden vectors, so this oracle is pinned against outputs
of the reference's own Python modules imported in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``, checked by
``tests/test_oracle_golden.py``).
"""
