"""GPU vs oracle / golden on the reference's real datasets (committed as tests/golden/*.npz; the GPU box
has no /root/reference): WKB -> GeoArrow -> HBM -> operators through the C ABI."""
import os

import numpy as np
import pytest

from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import join_pairs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def series(name) -> tuple[GeoSeries, np.lib.npyio.NpzFile]:
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    return GeoSeries(GeoArrowArray.from_wkb(z["wkb_values"], z["wkb_offsets"])), z


def rel_close(got, exp, rtol=1e-9):
    exp = np.asarray(exp)
    assert np.all(np.abs(got - exp) <= rtol * np.maximum(np.abs(exp), 1e-300))


def test_config1_cities_centroid_and_bounds(gpk):
    """BASELINE.json configs[0] / py-geopolars/example.py:1-11 on the GPU path."""
    s, z = series("cities")
    assert len(s) == 202
    assert np.array_equal(s.centroid().array.xy, s.array.xy)  # centroid(point) == point, bit exact
    assert np.array_equal(s.bounds(), np.concatenate([s.array.xy, s.array.xy], axis=1))
    assert np.array_equal(s.envelope().array.xy, s.array.xy)
    assert np.array_equal(s.translate(10.0, 10.0).array.xy, s.array.xy + 10.0)  # benches/affine.rs:25


@pytest.mark.parametrize("name", ["naturalearth_lowres", "nybb"])
def test_polygon_measures_match_golden(gpk, name):
    s, z = series(name)
    rel_close(s.area(), z["oracle_area"])
    rel_close(s.euclidean_length(), z["oracle_length"])
    rel_close(s.centroid().array.xy, z["oracle_centroid"])
    assert np.array_equal(s.bounds(), z["oracle_bounds"])
    env = s.envelope().array
    assert env.n_coords == 5 * len(s) and np.array_equal(env.xy[0::5], z["oracle_bounds"][:, :2])
    if name == "nybb":
        assert np.allclose(s.area(), z["Shape_Area"], rtol=5e-6)
        from tests.test_host_cpu import all_rings_as_multilinestring

        # Shape_Leng (the fixture's own column) is the length of ALL rings of a borough: the HIP length of that MultiLineString
        assert np.allclose(GeoSeries(all_rings_as_multilinestring(s.array)).euclidean_length(), z["Shape_Leng"], rtol=5e-5)


def test_countries_contain_cities_join(gpk, oracle):
    countries, _ = series("naturalearth_lowres")
    cities, _ = series("naturalearth_cities")
    exp_pairs, exp_counts, _ = oracle.spatial_join(cities.array, countries.array, "intersects", mode=0)
    got_pairs, got_counts = join_pairs(cities, countries, "intersects")
    assert len(exp_pairs) > 150
    assert np.array_equal(got_pairs, exp_pairs) and np.array_equal(got_counts, exp_counts)
    d = cities.distance(countries, exp_pairs[:, 1][: len(cities)] if len(exp_pairs) >= len(cities) else np.zeros(len(cities), np.uint32))
    assert d.shape == (len(cities),)


def test_spatial_join_table_shapes(gpk):
    """spatial_join_test / _with_suffixes (spatial_index.rs:432-556): result shapes (2, 4) and (9, 4),
    suffixed column names."""
    import struct

    import pyarrow as pa

    from geopolars_amd.spatial_index import SpatialJoinArgs, spatial_join

    pts = [(0.0, 10.0), (1.0, 1.0), (10.0, 1.0), (1.0, -1.0), (0.0, -10.0), (-1.0, -1.0), (-10.0, 0.0), (-1.0, 1.0), (0.0, 10.0)]
    wkb_pts = [struct.pack("<BIdd", 1, 1, *p) for p in pts]
    ring = [(0.0, 0.0), (20.0, 0.0), (20.0, 20.0), (0.0, 20.0), (0.0, 0.0)]
    wkb_poly = struct.pack("<BIII", 1, 3, 1, len(ring)) + b"".join(struct.pack("<dd", *c) for c in ring)
    point_df = pa.table({"geometry": pa.array(wkb_pts, pa.binary()), "point_values": pa.array([1.0, 2, 3, 4, 5, 6, 7, 8, 9])})
    polygon_df = pa.table({"geometry": pa.array([wkb_poly], pa.binary()), "string_col": pa.array(["test"])})
    inner = spatial_join(point_df, polygon_df, SpatialJoinArgs(join_type="inner", l_suffix="_left!", r_suffix="_right!"))
    left = spatial_join(point_df, polygon_df, SpatialJoinArgs(join_type="left"))
    assert (inner.num_rows, inner.num_columns) == (2, 4)
    assert (left.num_rows, left.num_columns) == (9, 4)
    assert inner.column_names == ["geometry_left!", "point_values_left!", "geometry_right!", "string_col_right!"]
    assert inner.column("point_values_left!").to_pylist() == [2.0, 3.0]
    assert left.column("string_col_right").null_count == 7
