"""The decision rule of the chain kernels for "test" sub-cells (gpk_join.hip: pip_tile_chain / pip_tile_route; CPU mirror: tools/proto_local_chain.py): winding number of a
point of a padded sub-cell = a per-sub-cell constant + the contributions of the LOCAL CHAIN of ring edges.  Checked against
the full ring walk on random and on-edge points; CPU only, a few seconds."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["c2", "small", "c4"])
@pytest.mark.parametrize("mode", [[], ["--as-kernel"]])  # the rule itself / the cyclic single-arc form chain_aux_kernel builds
def test_local_chain_rule_agrees_with_the_full_ring_walk(workload, mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "proto_local_chain.py"), workload] + mode, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "disagreements: 0" in r.stdout
