"""TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT.

CPU restatement (plain PyTorch CPU ops, fp32 or fp64) of the reference's training hot path
(/root/reference model.py:370-552 and the arch/ modules it instantiates).  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only as the
checker / CPU baseline.  The product (semi-supervised-segmentation-cyclegan_amd/) never imports it and
has no CPU path of its own.

Parity status: PINNED.  tests/golden/gen_golden.py imports the real reference in the build container, loads
the same keyed weights into the reference's own modules, asserts this restatement reproduces them, and
writes the golden vectors in tests/golden/ (tests/test_oracle_golden.py re-checks the restatement
against those vectors everywhere, including on the GPU box where /root/reference does not exist).
"""
