"""CPU oracle of the DiT hot path: TEST INFRASTRUCTURE, not part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / reported baseline.
"""
