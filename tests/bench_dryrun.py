"""Launch target of tests/test_bench_cpu.py: `bench.py --device cpu` under torchrun with the CPU test double behind the
operators (test infrastructure; bench.py itself never installs it outside its cpu_baseline leg)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from minsu3d_amd import backend  # noqa: E402
from oracle.oracle_backend import OracleBackend  # noqa: E402

backend.set_backend(OracleBackend())
import bench  # noqa: E402

bench.main(sys.argv[1:] + ["--device", "cpu"])
