"""CPU oracle for the OpenESS hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy for integer/byte/event arithmetic, plain
PyTorch fp32 on CPU for the floating-point network pieces, plus a scalar C port of the
voxelizers in ``voxel_oracle.c``) of the reference algorithms on the path named by
BASELINE.json ``north_star``.  Every function cites the reference file:line it follows.

It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  Nothing under ``openess_amd/`` imports, links or
executes anything from here; the product path fails loudly when the HIP library is absent.

Pinning: the reference ships no golden vectors or tests for this path,
so the oracle is pinned against outputs of the reference itself, imported from
``/root/reference`` in the build container by ``tests/golden/gen_golden.py`` and committed
as ``tests/golden/*.npz`` (inputs + expected outputs only).  ``tests/test_oracle_golden.py``
checks every oracle function against those fixtures.
"""
