"""ORACLE — test infrastructure only.

CPU restatements of the reference's algorithms for the hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (ieee_amd/) never does and fails loudly without its
HIP library.
"""
