"""oracle/ -- TEST INFRASTRUCTURE (CPU parity checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (calm_amd/) never does.
"""
