"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference hot path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  Nothing under ultrare_amd/ imports it; the
product path has no CPU fallback and fails loudly when the HIP library is absent.

Parity status: PINNED.  Every function here is checked in tests/test_oracle_golden.py
against golden vectors produced by the real reference run in the build container
(tests/golden/make_golden.py).  One exception, stated where it applies
(cpu_ref.ot_cluster): the reference's LP solver is POT 0.9.0 `ot.emd`, which is not
in /root/reference nor installable offline; the goldens used an exact HiGHS LP in its
place, so the OT labels are pinned to "the exact LP optimum", not to POT's code.
"""
