"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the CFUN volumetric hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the *checker*.  The product (``cfun_amd``) never imports
this package and fails loudly when its HIP library is missing.

Parity status: **pinned** -- every function in ``cfun_oracle`` is checked by
``tests/test_oracle_golden.py`` against vectors in ``tests/golden/*.npz`` that
were produced in the build container by importing the reference's own Python
(``tests/golden/gen_golden.py``; torch 2.10.0 CPU).  The reference ships no
tests/golden vectors of its own, so the pin is "outputs of
the reference itself run here".
"""
