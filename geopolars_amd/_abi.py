"""ctypes binding of include/geopolars_hip.h — the only door into libgeopolars_hip.so.

There is no CPU fallback behind this module: if the shared library is missing or a compute entry
point reports GPK_ERR_DEVICE, a GeopolarsHipError is raised.  `import torch` happens BEFORE the
library is opened so that libgeopolars_hip.so binds to the HIP runtime PyTorch already loaded
(same soname `libamdhip64.so.7`): device pointers from torch tensors and torch streams are then
valid inside the library, and torch.cuda.Event sees the library's launches.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPK_LIB_PATH") or os.path.join(HERE, "libgeopolars_hip.so")  # override: A/B builds

# ---- constants (mirror of the header) ----------------------------------------------------------
GPK_OK = 0
GPK_ERR_MISMATCHED_GEOMETRY = 1
GPK_ERR_INVALID_OFFSETS = 2
GPK_ERR_NULL_UNSUPPORTED = 3
GPK_ERR_DEVICE = 4
GPK_ERR_OOM = 5
GPK_ERR_INVALID_ARGUMENT = 6
GPK_ERR_CAPACITY = 7

GEOM_POINT = 0
GEOM_LINESTRING = 1
GEOM_POLYGON = 3
GEOM_MULTIPOINT = 4
GEOM_MULTILINESTRING = 5
GEOM_MULTIPOLYGON = 6
ARROW_WKB, ARROW_INTERLEAVED, ARROW_STRUCT = 0, 1, 2  # gpk_geoarray_to_arrow layouts

MEM_HOST = 0
MEM_DEVICE = 1

PRED_INTERSECTS = 0
PRED_CONTAINS = 1
PRED_WITHIN = 2
INDEX_BBOX_GRID = 1
INDEX_PIP = 2
INDEX_PIP_LIGHT = 4
INDEX_PIP_FULL = 8
QUERY_CONTAINED, QUERY_INTERSECTING = 0, 1  # gpk_index_query_envelope modes
PREDICATES = {"intersects": PRED_INTERSECTS, "contains": PRED_CONTAINS, "within": PRED_WITHIN}


class GeopolarsHipError(RuntimeError):
    """Mirror of GeopolarsError (geopolars/geopolars-geo/src/error.rs:9-28) on the Python side."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[gpk status {code}] {message}")
        self.code = code


class MismatchedGeometry(GeopolarsHipError):
    """GeopolarsError::MismatchedGeometry (error.rs:12-16)."""


class GeoArrowDesc(C.Structure):
    _fields_ = [
        ("geom_type", C.c_int32),
        ("mem_space", C.c_int32),
        ("n_geoms", C.c_int64),
        ("n_coords", C.c_int64),
        ("xy", C.c_void_p),
        ("geom_offsets", C.c_void_p),
        ("part_offsets", C.c_void_p),
        ("ring_offsets", C.c_void_p),
        ("n_parts", C.c_int64),
        ("n_rings", C.c_int64),
        ("validity", C.c_void_p),
        ("x", C.c_void_p),  # separated coordinates (xy NULL): Struct<x, y> GeoArrow arrays
        ("y", C.c_void_p),
    ]


class ArrowSchema(C.Structure):
    """Arrow C Data Interface (what `pyarrow.Array._export_to_c` fills: py-geopolars/src/ffi.rs:12-32)"""


ArrowSchema._fields_ = [
    ("format", C.c_char_p),
    ("name", C.c_char_p),
    ("metadata", C.c_void_p),
    ("flags", C.c_int64),
    ("n_children", C.c_int64),
    ("children", C.POINTER(C.POINTER(ArrowSchema))),
    ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.CFUNCTYPE(None, C.POINTER(ArrowSchema))),
    ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64),
    ("null_count", C.c_int64),
    ("offset", C.c_int64),
    ("n_buffers", C.c_int64),
    ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.CFUNCTYPE(None, C.POINTER(ArrowArray))),
    ("private_data", C.c_void_p),
]


# every symbol the header declares: name -> (restype, argtypes)
_VP = C.c_void_p
_PROTOS = {
    "gpk_version": (C.c_char_p, []),
    "gpk_last_error": (C.c_int32, [C.c_char_p, C.c_size_t]),
    "gpk_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "gpk_device_info": (C.c_int32, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]),
    "gpk_geoarray_upload": (C.c_int32, [C.POINTER(GeoArrowDesc), _VP, C.POINTER(_VP)]),
    "gpk_geoarray_from_arrow": (C.c_int32, [_VP, _VP, C.c_int32, _VP, C.POINTER(_VP), C.POINTER(C.c_int32)]),
    "gpk_geoarray_to_arrow": (C.c_int32, [_VP, C.c_int32, _VP, _VP, _VP]),
    "gpk_geoarray_free": (C.c_int32, [_VP]),
    "gpk_geoarray_invalidate": (C.c_int32, [_VP]),
    "gpk_geoarray_nbytes": (C.c_int32, [_VP, C.POINTER(C.c_int64)]),
    "gpk_wkb_decode": (C.c_int32, [_VP, _VP, C.c_int64, _VP, C.POINTER(C.c_int64), _VP, _VP, _VP, _VP]),
    "gpk_geoarray_from_wkb": (C.c_int32, [_VP, _VP, C.c_int64, _VP, C.c_int32, _VP, C.POINTER(_VP), C.POINTER(C.c_int32)]),
    "gpk_geoarray_download": (C.c_int32, [_VP, C.POINTER(C.c_int64), _VP, _VP, _VP, _VP, _VP]),
    "gpk_area": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_signed_area": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_centroid": (C.c_int32, [_VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_bounds": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_euclidean_length": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_affine_transform": (C.c_int32, [_VP, C.POINTER(C.c_double), _VP, C.c_int32, _VP]),
    "gpk_affine_transform_rows": (C.c_int32, [_VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_affine_about_origin": (C.c_int32, [_VP, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double, _VP, C.c_int32, _VP]),
    "gpk_envelope": (C.c_int32, [_VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_exterior": (C.c_int32, [_VP, _VP, _VP, C.POINTER(C.c_int64), C.c_int32, _VP]),
    "gpk_explode": (C.c_int32, [_VP, _VP, C.c_int32, _VP, C.POINTER(_VP)]),
    "gpk_geoarray_validity": (C.c_int32, [_VP, _VP, C.POINTER(C.c_int32), _VP]),
    "gpk_geoarray_len": (C.c_int32, [_VP, C.POINTER(C.c_int64)]),
    "gpk_geom_type": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_is_empty": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_is_ring": (C.c_int32, [_VP, _VP, C.c_int32, _VP]),
    "gpk_point_xy": (C.c_int32, [_VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_geodesic_length": (C.c_int32, [_VP, C.c_int32, _VP, C.c_int32, _VP]),
    "gpk_simplify": (C.c_int32, [_VP, C.c_double, _VP, _VP, C.POINTER(C.c_int64), C.c_int32, _VP]),
    "gpk_convex_hull": (C.c_int32, [_VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_distance_rowwise": (C.c_int32, [_VP, _VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_rowmap_build": (C.c_int32, [_VP, _VP, C.c_int64, C.c_int32, _VP, C.POINTER(_VP)]),
    "gpk_rowmap_free": (C.c_int32, [_VP]),
    "gpk_rowmap_nbytes": (C.c_int32, [_VP, C.POINTER(C.c_int64)]),
    "gpk_distance_rowmap": (C.c_int32, [_VP, _VP, _VP, _VP, C.c_int32, _VP]),
    "gpk_predicate_rowwise": (C.c_int32, [_VP, _VP, _VP, C.c_int32, _VP, C.c_int32, _VP]),
    "gpk_index_build": (C.c_int32, [_VP, _VP, C.POINTER(_VP)]),
    "gpk_index_build_ex": (C.c_int32, [_VP, C.c_int32, _VP, _VP, C.POINTER(_VP)]),
    "gpk_index_free": (C.c_int32, [_VP]),
    "gpk_index_nbytes": (C.c_int32, [_VP, C.POINTER(C.c_int64)]),
    "gpk_index_describe": (C.c_int32, [_VP, C.POINTER(C.c_int64)]),
    "gpk_index_query_envelope": (C.c_int32, [_VP, _VP, C.c_int64, C.c_int32, _VP, _VP, C.c_int64, C.POINTER(C.c_int64), C.c_int32, _VP]),
    "gpk_spatial_join": (
        C.c_int32,
        [_VP, _VP, _VP, C.c_int32, C.c_uint32, _VP, _VP, C.c_int64, C.POINTER(C.c_int64), C.c_int32, _VP],
    ),
    "gpk_spatial_join_async": (C.c_int32, [_VP, _VP, _VP, C.c_int32, C.c_uint32, _VP, _VP, C.c_int64, _VP, _VP]),
    "gpk_wkb_encode": (C.c_int32, [_VP, _VP, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "gpk_geoarray_to_wkb": (C.c_int32, [_VP, _VP, _VP, C.c_int64, C.POINTER(C.c_int64), C.c_int32, _VP]),
    "gpk_join_indices": (C.c_int32, [_VP, _VP, C.c_int64, C.c_int64, C.c_uint32, C.c_int32, _VP, _VP, C.c_int64, C.POINTER(C.c_int64), C.c_int32, _VP]),
    "gpk_take_fixed": (C.c_int32, [_VP, C.c_int32, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, C.c_int32, _VP]),
    "gpk_take_binary": (C.c_int32, [_VP, _VP, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, C.c_int64, C.POINTER(C.c_int64), _VP, C.c_int32, _VP]),
    "gpk_comm_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "gpk_comm_init": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.POINTER(_VP)]),
    "gpk_comm_free": (C.c_int32, [_VP]),
    "gpk_comm_mock_world": (C.c_int32, [C.c_int32, C.POINTER(_VP)]),
    "gpk_comm_init_mock": (C.c_int32, [C.c_int32, _VP, C.POINTER(_VP)]),
    "gpk_comm_mock_world_free": (C.c_int32, [_VP]),
    "gpk_comm_info": (C.c_int32, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gpk_allgatherv_geoarray": (C.c_int32, [_VP, _VP, _VP, C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gpk_geoarray_concat": (C.c_int32, [C.POINTER(_VP), C.c_int32, _VP, C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gpk_allgatherv_rows_f64": (C.c_int32, [_VP, _VP, C.c_int64, C.c_int32, _VP, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _VP]),
    "gpk_device_cache_release": (C.c_int32, []),
    "gpk_join_stats_enable": (C.c_int32, [C.c_int32]),
    "gpk_join_stats": (C.c_int32, [C.POINTER(C.c_int64), C.c_int32]),
    "gpk_profile_enable": (C.c_int32, [C.c_int32]),
    "gpk_profile_filter": (C.c_int32, [C.c_char_p]),
    "gpk_profile_reset": (C.c_int32, []),
    "gpk_profile_query": (C.c_int32, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def lib() -> C.CDLL:
    """Open libgeopolars_hip.so (once).  Fails loudly when the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GeopolarsHipError(
            GPK_ERR_DEVICE,
            f"{LIB_PATH} is missing: build it with `python -m geopolars_amd.build` "
            "(hipcc, gfx950).  geopolars_amd has no CPU fallback.",
        )
    try:  # make the process-wide HIP runtime the one torch ships, before our DT_NEEDED is resolved
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, the C ABI works without it
        pass
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(handle, name)  # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().gpk_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(status: int) -> None:
    if status == GPK_OK:
        return
    msg = last_error()
    if status == GPK_ERR_MISMATCHED_GEOMETRY:
        raise MismatchedGeometry(status, msg)
    raise GeopolarsHipError(status, msg)


def device_count() -> int:
    n = C.c_int32(0)
    rc = lib().gpk_device_count(C.byref(n))
    return int(n.value) if rc == GPK_OK else 0


def device_info() -> tuple[str, int]:
    buf = C.create_string_buffer(256)
    cus = C.c_int32(0)
    check(lib().gpk_device_info(buf, 256, C.byref(cus)))
    return buf.value.decode(), int(cus.value)
