"""Per-block parity bars at the real Flux.1 width (relative L2 per output stream, one block on
the oracle's input), shared by tests/test_real_width_gpu.py (HIP path against the oracle) and
tests/test_oracle_sensitivity.py (two faithful evaluations of the oracle against each other:
the floor these bars are read against).  Keys: (block kind "d" double / "s" single, W8A8?)."""
# measured on MI355X: double bf16 1.67e-3, single bf16 3.5e-4, double W8A8 2.2e-3, single W8A8 1.2e-3;
# floor (oracle re-evaluated with another GEMM order and half-precision probabilities): 1.63e-3, 3.1e-4,
# 1.65e-3, 4.5e-4; the reference's own flash kernels' bf16 probabilities: 1.97e-3, 5.2e-4, 2.7e-3, 9.6e-4
BAR = {("d", False): 2e-3, ("s", False): 6e-4, ("d", True): 2.7e-3, ("s", True): 1.5e-3}
