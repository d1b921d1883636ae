"""Regenerate every exact-arithmetic golden file (tests/golden/*_lattice.npz) from its own script, into `out_dir`
(default: in place).  Each generator is seeded and uses Python integers / fractions only — neither the oracle nor the
library — so the regenerated arrays must equal the committed ones (tests/test_golden_regeneration.py).

    python tests/golden/regenerate_all.py [out_dir]
"""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
GENERATORS = ("ops", "lines", "join", "contains")


def main(out_dir: str = HERE) -> None:
    os.makedirs(out_dir, exist_ok=True)
    for name in GENERATORS:
        importlib.import_module(f"tests.golden.make_{name}_golden").main(out_dir)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
