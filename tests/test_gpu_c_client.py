"""The plain-C client of include/geopolars_hip.h (tests/c_abi_client.c) on the GPU box: compiled as strict C99, linked against the library,
it uploads a polygon, computes its area, sends the column out through gpk_geoarray_to_arrow and back in through gpk_geoarray_from_arrow
(the two halves of the reference's FFI seam, py-geopolars/src/ffi.rs:12-52) and releases the structs — no Python between the C caller
and the library."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_host_cpu import build_and_run_c_client  # noqa: E402

pytestmark = pytest.mark.gpu


def test_plain_c_client_computes_and_round_trips_on_the_device(gpk, tmp_path):
    r = build_and_run_c_client(tmp_path)
    assert "area: rc=0 value=0.5" in r.stdout and "area after the round trip: rc=0 value=0.5" in r.stdout, r.stdout
