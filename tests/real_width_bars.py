"""Per-block parity bars at the real Flux.1 width (relative L2 per output stream, one block on
the oracle's input), shared by tests/test_real_width_gpu.py (HIP path against the oracle) and
tests/test_oracle_sensitivity.py (two faithful evaluations of the oracle against each other:
the floor these bars are read against).  Keys: (block kind "d" double / "s" single, W8A8?)."""
BAR = {("d", False): 2e-3, ("s", False): 1e-3, ("d", True): 3e-3, ("s", True): 2e-3}
