"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the KZG segment-prover hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import,
link or execute it, and only as the checker / reported baseline.

PARITY STATUS: **parity unpinned** against the real reference prover.  The
arithmetic of the reference lives in the external Rust binary
``apollozkp/fourier`` (reference ``requirements.txt:3``, un-pinned git
dependency, not present under /root/reference, no Rust toolchain here), and the
reference's own tests assert no commitment / proof constant (SRS regenerated per
session, reference ``tests/conftest.py:50-65``).  What *is* pinned:

* the Fr wire encoding and ``eval`` -- by the one known-answer vector the
  reference holds (``tests/test_miner.py:33-55``: TEST_POLY / TEST_POINT /
  TEST_EVAL), checked in ``tests/test_oracle.py``;
* public constants of BLS12-381 (generator, ZCash 48-byte compression, 2-adic
  root of unity derived from 7).

The oracle restates the *published* algorithms (BLS12-381, KZG10, Pianist
per-worker basis); a group element has one canonical encoding, so any correct
implementation agrees bit for bit on the same (SRS, scalars, alpha).
"""
