"""CPU oracle for the attention-forward hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker -- never as the thing measured or shipped.
"""
