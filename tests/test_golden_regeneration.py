"""The exact-arithmetic golden files are reproducible: tests/golden/regenerate_all.py (seeded generators on Python integers /
fractions, neither the oracle nor the library) writes arrays equal to the committed tests/golden/*_lattice.npz."""
import os

import numpy as np

from tests.golden import regenerate_all

HERE = os.path.dirname(os.path.abspath(__file__))


def test_regenerated_golden_files_equal_the_committed_ones(tmp_path):
    regenerate_all.main(str(tmp_path))
    for name in regenerate_all.GENERATORS:
        a = np.load(os.path.join(HERE, "golden", f"{name}_lattice.npz"))
        b = np.load(os.path.join(str(tmp_path), f"{name}_lattice.npz"))
        assert set(a.files) == set(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (name, k)
