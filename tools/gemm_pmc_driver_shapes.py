"""The twelve GEMM shapes that take the most time in a B = 64 training step (profiles/r03/gemm_profile_e2.json), shared by
tools/gemm_pmc_driver.py and tools/e2_bench.py.  (kind, H, Cin, Cout, k): fwd = forward / stride-1 data gradient (the same GEMM
class), wgrad = weight gradient."""
SHAPES = [('fwd', 16, 256, 256, 3), ('fwd', 32, 128, 128, 3), ('fwd', 64, 128, 128, 3), ('fwd', 64, 64, 64, 3),
          ('wgrad', 16, 256, 256, 3), ('wgrad', 32, 128, 128, 3), ('fwd', 8, 512, 512, 3), ('fwd', 64, 64, 256, 1),
          ('wgrad', 64, 64, 64, 3), ('fwd', 16, 1024, 256, 1), ('fwd', 16, 256, 1024, 1), ('fwd', 32, 128, 512, 1)]
