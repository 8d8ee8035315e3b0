"""CPU oracle for the batched GN/LM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a CPU restatement (torch-CPU tensors in fp32/fp64, LAPACK potrf via
``torch.linalg.cholesky`` exactly like the reference) of the algorithm the reference runs for
``DenseLinearization`` + ``CholeskyDenseSolver`` + ``LevenbergMarquardt`` on SE3 pose graphs.
Every function cites the reference file:line it follows (paths relative to /root/reference).

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
  * nothing under ``theseus_amd/`` imports it, ever -- the product path is HIP-only and fails
    loudly when ``libtheseus_hip.so`` is missing.

Pinning: ``oracle/gen_golden.py`` imports the real reference from /root/reference (this container
only) and writes small fixtures to ``tests/golden/``; ``tests/test_oracle_golden.py`` checks the
restatement against them on every CPU run (the fixtures travel to the GPU box, the reference does not).
"""
