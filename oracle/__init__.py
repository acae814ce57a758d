"""CPU oracle for the TGTC-Style render hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain CPU (numpy / PyTorch-CPU, fp32+fp64) restatement of the
reference algorithm for the path named in BASELINE.json: ray generation, coarse
and inverse-CDF fine sampling, positional encoding, the NeRF / style MLPs, alpha
compositing and the forward pass of the 2-D style transformer / CNN decoder / VGG
encoder.  Every function cites the reference file:line it follows.

Parity pinning: the oracle is checked against golden vectors produced by
importing the *reference itself* in the build container
(tests/golden/gen_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / reported baseline.  The product
(tgtc-style_amd/) never imports it and fails loudly if the HIP library is missing.
"""
