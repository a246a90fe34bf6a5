"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU checker for the NLTGV2-L1 hot path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under flame_amd/ (the product) imports it.  PARITY UNPINNED -- see nltgv2_oracle.c header.
"""
