"""oracle/ — TEST INFRASTRUCTURE, not product.

CPU restatement of the reference's algorithm for the DiST training hot path
(dist_oracle.py) plus the script that pins it against the real reference
(make_golden.py, runs only where /root/reference exists).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import from here.
"""
