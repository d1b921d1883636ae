/*
 * geopolars_hip.h — flat C ABI of libgeopolars_hip.so, the MI355X (gfx950) geometry-kernel
 * backend that sits under GeoPolars' `GeoSeries` operator surface.
 *
 * Every entry point replaces one `todo!()` body (or one dead-code call site) in the reference:
 *   trait GeoSeries / impl GeoSeries for Series   geopolars/geopolars-geo/src/geoseries.rs:10-181,183-279
 *   spatial_join + SpatialIndex                    geopolars/src/spatial_index.rs:37-204,314-350
 *   row codec (WKB <-> geometry)                   geopolars/geopolars-geo/src/util.rs:11-37
 * The only FFI convention the reference has is the Arrow C Data Interface
 * (py-geopolars/src/ffi.rs:12-52): single-chunk arrays, inputs BORROWED for the call, outputs owned
 * by the producer until released.  This ABI keeps that ownership model but speaks raw GeoArrow
 * buffers (coords FixedSizeList<f64,2> interleaved + i32 List offsets) so a Rust shim can pass
 * `array.values().as_ptr()` straight through (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - All functions return int32_t status (GPK_OK == 0).  Nothing throws or aborts across the ABI;
 *     `gpk_last_error` returns a thread-local message.  Status codes map onto
 *     `GeopolarsError` (geopolars/geopolars-geo/src/error.rs:9-28) in the shim.
 *   - Pointers carry a memory space tag (GPK_MEM_HOST / GPK_MEM_DEVICE).  Host buffers are copied to
 *     HBM once by `gpk_geoarray_upload`; device buffers are borrowed (zero copy) — that is how a
 *     caller that already holds data in HBM (another kernel, a torch tensor's data_ptr) plugs in.
 *   - `stream` is a hipStream_t passed as void* (NULL = HIP's legacy default stream, which orders
 *     against every other blocking stream of the device; pass your own stream for concurrency).
 *     Calls with host outputs block until the result is host-visible; calls with device outputs are
 *     stream-ordered and do not synchronise.
 *   - Geometry handles are immutable after upload and may be shared between threads
 *     (mirrors `Arc<SpatialIndex>`, spatial_index.rs:20-21).
 *   - Null rows: validity bitmaps are Arrow LSB-first; null in -> null out (never a panic, unlike
 *     util.rs:32).
 *   - There is NO CPU fallback in this library.  Without a gfx950 device every compute entry point
 *     returns GPK_ERR_DEVICE.
 */
#ifndef GEOPOLARS_HIP_H
#define GEOPOLARS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define GPK_OK                        0
#define GPK_ERR_MISMATCHED_GEOMETRY   1 /* -> GeopolarsError::MismatchedGeometry (error.rs:12-16) */
#define GPK_ERR_INVALID_OFFSETS       2 /* -> PolarsError::ComputeError */
#define GPK_ERR_NULL_UNSUPPORTED      3
#define GPK_ERR_DEVICE                4 /* HIP error / no device / extension not usable */
#define GPK_ERR_OOM                   5
#define GPK_ERR_INVALID_ARGUMENT      6
#define GPK_ERR_CAPACITY              7 /* caller-provided pair buffer too small; n_pairs is set */

/* ---- geometry type ids: identical to GeoSeries::geom_type (geoseries.rs:60-73) --------------- */
#define GPK_GEOM_POINT             0
#define GPK_GEOM_LINESTRING        1
#define GPK_GEOM_POLYGON           3
#define GPK_GEOM_MULTIPOINT        4
#define GPK_GEOM_MULTILINESTRING   5
#define GPK_GEOM_MULTIPOLYGON      6

#define GPK_MEM_HOST    0
#define GPK_MEM_DEVICE  1

/* ---- predicates: `Predicate` of spatial_index.rs:13,28 -------------------------------------- */
#define GPK_PRED_INTERSECTS  0 /* default, spatial_index.rs:28 */
#define GPK_PRED_CONTAINS    1
#define GPK_PRED_WITHIN      2 /* within(a,b) == contains(b,a) */

/*
 * One single-chunk GeoArrow array.  Nesting by type:
 *   POINT                         coords
 *   LINESTRING / MULTIPOINT       geom_offsets[n+1] -> coords
 *   POLYGON / MULTILINESTRING     geom_offsets[n+1] -> rings ; ring_offsets[n_rings+1] -> coords
 *   MULTIPOLYGON                  geom_offsets[n+1] -> parts ; part_offsets[n_parts+1] -> rings ;
 *                                 ring_offsets[n_rings+1] -> coords
 * coords are interleaved xy (FixedSizeList<f64,2>) — or SEPARATED: `xy` NULL and `x`, `y` two arrays of n_coords doubles, the
 * Struct<x: f64, y: f64> coordinate arrays the reference's Python layer builds with `pyarrow.StructArray.from_arrays([x, y])`
 * (py-geopolars/python/geopolars/internals/geoseries.py:86-113).  Kernels read interleaved coordinates (one 16-byte request per
 * vertex); a separated descriptor is interleaved ON THE DEVICE while it is uploaded — one pass, no host-side copy — so the
 * handle owns its coordinates even when the descriptor's buffers are device memory (offsets and validity are still borrowed).
 * Rings are stored closed (first == last).  Unused offset pointers are NULL.  All buffers of one descriptor live in `mem_space`.
 */
typedef struct gpk_geoarrow_desc {
    int32_t        geom_type;     /* GPK_GEOM_* */
    int32_t        mem_space;     /* GPK_MEM_HOST or GPK_MEM_DEVICE */
    int64_t        n_geoms;
    int64_t        n_coords;
    const double*  xy;            /* 2*n_coords doubles */
    const int32_t* geom_offsets;  /* n_geoms+1 or NULL (POINT) */
    const int32_t* part_offsets;  /* n_parts+1, MULTIPOLYGON only */
    const int32_t* ring_offsets;  /* n_rings+1, POLYGON / MULTILINESTRING / MULTIPOLYGON */
    int64_t        n_parts;       /* MULTIPOLYGON only */
    int64_t        n_rings;       /* POLYGON / MULTILINESTRING / MULTIPOLYGON */
    const uint8_t* validity;      /* Arrow bitmap, NULL = all valid */
    const double*  x;             /* separated coordinates (xy == NULL): n_coords doubles each, else NULL */
    const double*  y;
} gpk_geoarrow_desc;

typedef struct gpk_geoarray gpk_geoarray; /* device-resident SoA copy (or borrowed view) */
typedef struct gpk_index    gpk_index;    /* device-resident spatial index over one geoarray */

/* ---- library ------------------------------------------------------------------------------ */
const char* gpk_version(void);
int32_t gpk_last_error(char* buf, size_t cap);
int32_t gpk_device_count(int32_t* out_n);
/* name + CU count of the current HIP device; fails with GPK_ERR_DEVICE when it is not gfx950 */
int32_t gpk_device_info(char* name_buf, size_t cap, int32_t* out_cus);
/* Index tables and build temporaries are recycled through a cache of device blocks inside the library (a freed index's tables serve
 * the next build: SpatialIndex values come and go with the dataframe pipeline, spatial_index.rs:37-71, and hipMalloc / hipFree of
 * gigabytes cost up to hundreds of milliseconds).  The cache holds at most GPK_DEVICE_CACHE_MB (default: a sixteenth of the device's
 * memory, 16 GB at most; 0 = no cache); this call hands every idle block back to the driver now. */
int32_t gpk_device_cache_release(void);

/* ---- "copied once to HBM as SoA" ---------------------------------------------------------- */
/* replaces the per-op row decode of util.rs:27-37 (iter_geom).  A DEVICE view whose offsets do not start at 0 (a sliced Arrow list
 * array: unrebased offsets next to the slice of the child buffer they index) is accepted: the level gets an owned, rebased copy, so
 * every entry point sees children indexed from 0 (one 4-byte read-back per offsets level and upload of a device view). */
int32_t gpk_geoarray_upload(const gpk_geoarrow_desc* desc, void* stream, gpk_geoarray** out);
int32_t gpk_geoarray_free(gpk_geoarray* a);
/* A handle derives tables from its OFFSETS on first use and keeps them (the size classes of the streaming reductions, the strip table and
 * ring records of their one-pass form: csrc/gpk_unary.hip, csrc/gpk_ringstream.hip) — handles are immutable (above).  A handle over
 * BORROWED device buffers (GPK_MEM_DEVICE descriptors) is only as immutable as its owner keeps those buffers: rewriting COORDINATES in
 * place is harmless to the tables, rewriting OFFSETS is not — call this (or make a new handle: no copy either way) before the next
 * call on the handle.  Waits for the device.  (The index gpk_spatial_join keeps on a handle is kept only on handles that own their
 * buffers: nothing to drop there.) */
int32_t gpk_geoarray_invalidate(gpk_geoarray* a);
/* HBM bytes held by the handle (owned + borrowed), for roofline accounting */
int32_t gpk_geoarray_nbytes(const gpk_geoarray* a, int64_t* out_bytes);

/* WKB BinaryArray<i32> (util.rs:27-37 input format) -> GeoArrow buffers, host side (both byte orders; Z / M ordinates are read past).
 * Pass 1 (out == NULL): validates and fills `counts` = {geom_type, n_geoms, n_parts, n_rings, n_coords}.
 * Pass 2: fills caller-allocated buffers of exactly those sizes.  Mixed Polygon/MultiPolygon input is
 * promoted to MULTIPOLYGON, mixed LineString/MultiLineString to MULTILINESTRING. */
int32_t gpk_wkb_decode(const uint8_t* wkb_values, const int32_t* wkb_offsets, int64_t n_rows,
                       const uint8_t* validity, int64_t counts[5], double* xy,
                       int32_t* geom_offsets, int32_t* part_offsets, int32_t* ring_offsets);

/* The same decode on the GPU: the raw WKB column (values + offsets, in `mem_space`) is copied to HBM once and
 * decoded there into a device-resident handle (scan -> prefix sums -> fill); the GeoArrow SoA never exists on
 * the host.  Little-endian ISO WKB / EWKB+SRID, 2D, types 1-6, same promotion rules as gpk_wkb_decode.  A HOST column that also
 * holds big-endian records or Z / M ordinates (EWKB flags, ISO 1000-codes) is parsed by the host decoder instead — both byte orders,
 * Z / M dropped like geozero's to_geo drops them for the reference — and uploaded: same handle, without the GPU's parse rate; in a
 * DEVICE column such rows are reported (GPK_ERR_MISMATCHED_GEOMETRY).  Type 7 (GeometryCollection, geoseries.rs:60-73) has no GeoArrow
 * nesting and is reported either way. */
int32_t gpk_geoarray_from_wkb(const uint8_t* wkb_values, const int32_t* wkb_offsets, int64_t n_rows,
                              const uint8_t* validity, int32_t mem_space, void* stream,
                              gpk_geoarray** out, int32_t* out_geom_type);
/* GeoArrow -> WKB BinaryArray<i32>: the column format geometry-valued results leave the reference in
 * (from_geom_vec, util.rs:11-24).  Little-endian ISO WKB, 2D, one WKB type per column (the array's), zero-length
 * records for null rows (the validity bitmap travels separately).
 *   out_offsets[n_geoms + 1]   Arrow offsets (may be NULL)
 *   out_values[capacity]       WKB bytes (NULL + capacity 0 = size query)
 *   *n_bytes                   total bytes, always set; GPK_ERR_CAPACITY when > capacity, or when the column
 *                              cannot fit i32 offsets (encode row slices)
 * gpk_wkb_encode: host buffers in, host buffers out, no device.  gpk_geoarray_to_wkb: encodes a device-resident
 * handle ON the GPU (sizes -> scan -> headers -> bodies); outputs in `out_space`. */
int32_t gpk_wkb_encode(const gpk_geoarrow_desc* desc, int32_t* out_offsets, uint8_t* out_values,
                       int64_t capacity, int64_t* n_bytes);
int32_t gpk_geoarray_to_wkb(const gpk_geoarray* a, int32_t* out_offsets, uint8_t* out_values,
                            int64_t capacity, int64_t* n_bytes, int32_t out_space, void* stream);
/* The reference's FFI seam — the Arrow C Data Interface (py-geopolars/src/ffi.rs:12-32: a pyarrow array exported with `_export_to_c`,
 * rechunked to ONE array first, :56) — as an entry point: `array` / `schema` are the two exported structs of a geometry column in
 * host memory, BORROWED for the call (the caller releases them as it would after any import).  Accepted columns
 *   Binary / LargeBinary ("z" / "Z")                     WKB rows: decoded on the GPU like gpk_geoarray_from_wkb
 *   [List<]* Struct<x: f64, y: f64>                       native GeoArrow with SEPARATED coordinates, as the reference's Python layer
 *                                                         builds them (internals/geoseries.py:86-113); 0 - 3 list levels, "+l" or "+L"
 *   [List<]* FixedSizeList<f64, 2>                        native GeoArrow, interleaved
 * Sliced arrays (offset != 0 at any level), 64-bit list offsets (narrowed after a range check) and validity bitmaps with a bit
 * offset are handled.  One and two list levels are two geometry types each: `geom_type_hint` (GPK_GEOM_*, or -1) or the schema's
 * ARROW:extension:name (geoarrow.multipoint / geoarrow.multilinestring) picks MULTIPOINT / MULTILINESTRING, else LINESTRING / POLYGON.
 * *out_geom_type (may be NULL) = the handle's type.  Anything else: GPK_ERR_MISMATCHED_GEOMETRY. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char* format;
    const char* name;
    const char* metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema** children;
    struct ArrowSchema* dictionary;
    void (*release)(struct ArrowSchema*);
    void* private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    struct ArrowArray** children;
    struct ArrowArray* dictionary;
    void (*release)(struct ArrowArray*);
    void* private_data;
};
#endif
int32_t gpk_geoarray_from_arrow(const struct ArrowArray* array, const struct ArrowSchema* schema, int32_t geom_type_hint,
                                void* stream, gpk_geoarray** out, int32_t* out_geom_type);
/* ... and the way back (py-geopolars/src/ffi.rs:35-52: `to_py_array` hands every result to Python as an ArrowArray / ArrowSchema pair
 * that the importer releases): the handle's column in HOST memory behind the two caller-provided structs, every buffer owned by the
 * library until the importer calls the structs' `release` callbacks (which free the buffers, the children and the private data; a
 * struct whose release is NULL has been released).  `layout`:
 *   GPK_ARROW_WKB          Binary ("z", i32 offsets) of ISO WKB, encoded on the GPU (gpk_geoarray_to_wkb) — how the reference holds
 *                          geometry columns (util.rs:11-24); ARROW:extension:name geoarrow.wkb
 *   GPK_ARROW_STRUCT       native GeoArrow, 0 - 3 "+l" levels by geometry type over Struct<x: f64, y: f64> — what the reference's Python
 *                          layer builds (internals/geoseries.py:86-113)
 *   GPK_ARROW_INTERLEAVED  the same nesting over FixedSizeList<f64, 2>
 * Native layouts carry ARROW:extension:name geoarrow.point / linestring / polygon / multipoint / multilinestring / multipolygon, so
 * that gpk_geoarray_from_arrow(to_arrow(x)) == x for every type.  Null rows keep their (empty) slots; the outermost array carries the
 * validity bitmap and null_count. */
#define GPK_ARROW_WKB 0
#define GPK_ARROW_INTERLEAVED 1
#define GPK_ARROW_STRUCT 2
int32_t gpk_geoarray_to_arrow(const gpk_geoarray* a, int32_t layout, void* stream, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
/* Device -> host copy of a handle's GeoArrow buffers.  sizes[4] = {n_coords, n_parts, n_rings, n_geoms} is always
 * filled; NULL buffers are skipped (call once with NULLs to size the buffers). */
int32_t gpk_geoarray_download(const gpk_geoarray* a, int64_t sizes[4], double* xy, int32_t* geom_offsets,
                              int32_t* part_offsets, int32_t* ring_offsets, void* stream);

/* The Arrow validity bitmap of a handle, device -> host: *out_has_validity = 0 when every row is valid (nothing is
 * written); out_bitmap[(n_geoms + 7) / 8] may be NULL to ask only that. */
int32_t gpk_geoarray_validity(const gpk_geoarray* a, uint8_t* out_bitmap, int32_t* out_has_validity, void* stream);

/* ---- unary operators: GeoSeries::{area, centroid, envelope/bounds, affine_transform, ...} --- */
/* out arrays live in `out_space`; sizes are in elements.                                      */
/* area: geoseries.rs:14-16,188-190.  out[n_geoms] */
int32_t gpk_area(const gpk_geoarray* a, double* out, int32_t out_space, void* stream);
/* signed area (exterior orientation sign), same layout */
int32_t gpk_signed_area(const gpk_geoarray* a, double* out, int32_t out_space, void* stream);
/* centroid: geoseries.rs:18-21,192-194.  out_xy[2*n_geoms]; out_valid[n_geoms] bytes (0 = empty geometry -> null) */
int32_t gpk_centroid(const gpk_geoarray* a, double* out_xy, uint8_t* out_valid, int32_t out_space,
                     void* stream);
/* bounds (north-star) / envelope (geoseries.rs:28-33,200-202): out[4*n_geoms] = minx,miny,maxx,maxy;
 * empty geometry -> NaN x4 */
int32_t gpk_bounds(const gpk_geoarray* a, double* out4, int32_t out_space, void* stream);
/* euclidean_length: geoseries.rs:35-41.  out[n_geoms] */
int32_t gpk_euclidean_length(const gpk_geoarray* a, double* out, int32_t out_space, void* stream);
/* affine_transform: geoseries.rs:11-12,184-186.  m = [a, b, xoff, d, e, yoff] (upstream
 * `AffineTransform::from([f64;6])` order, which py-geopolars/src/geo.rs:10-16 passes through; the
 * Python docstring georust/geoseries.py:33 says [a,b,d,e,xoff,yoff] — that docstring is wrong).
 * out_xy[2*n_coords]; offsets are unchanged and shared with the input. */
int32_t gpk_affine_transform(const gpk_geoarray* a, const double m[6], double* out_xy,
                             int32_t out_space, void* stream);
/* rotate / scale / skew about a per-geometry origin (geoseries.rs:85-139,163-174; TransformOrigin of
 * py-geopolars/src/utils.rs:5-27 = centroid | bbox center | point) reduce to one affine matrix PER ROW:
 * matrices[6*n_geoms] in the same [a, b, xoff, d, e, yoff] order, living in `out_space`. */
int32_t gpk_affine_transform_rows(const gpk_geoarray* a, const double* matrices, double* out_xy,
                                  int32_t out_space, void* stream);
/* rotate / scale / skew about a per-geometry origin in ONE call (geoseries.rs:85-93,95-107,118-139): the origins
 * (TransformOrigin of py-geopolars/src/utils.rs:5-27: centroid | centre of the bounding box | a point) and the per-row
 * matrices are computed on the device, then applied like gpk_affine_transform_rows.
 *   kind    GPK_AFFINE_ROTATE  p0 = angle in degrees, counter-clockwise                      [c, -s, ox - c ox + s oy, s, c, oy - s ox - c oy]
 *           GPK_AFFINE_SCALE   p0 = xfact, p1 = yfact                                        [xf, 0, ox (1 - xf), 0, yf, oy (1 - yf)]
 *           GPK_AFFINE_SKEW    p0 = xs, p1 = ys in degrees (matrix of geoseries.rs:129-138)   [1, tan xs, -oy tan xs, tan ys, 1, -ox tan ys]
 *   origin  GPK_ORIGIN_CENTROID | GPK_ORIGIN_CENTER | GPK_ORIGIN_POINT (ox, oy)
 * out_xy[2*n_coords]; offsets are unchanged and shared with the input. */
#define GPK_AFFINE_ROTATE 0
#define GPK_AFFINE_SCALE  1
#define GPK_AFFINE_SKEW   2
#define GPK_ORIGIN_CENTROID 0
#define GPK_ORIGIN_CENTER   1
#define GPK_ORIGIN_POINT    2
int32_t gpk_affine_about_origin(const gpk_geoarray* a, int32_t kind, double p0, double p1, int32_t origin,
                                double ox, double oy, double* out_xy, int32_t out_space, void* stream);
/* envelope as a geometry (geoseries.rs:28-33; geo BoundingRect -> Rect::to_polygon): one closed 5-coordinate rectangle
 * per row (minx miny, maxx miny, maxx maxy, minx maxy, minx miny) — a POLYGON column whose ring_offsets are 5 i.
 * out_xy[10*n_geoms]; out_valid[n_geoms] bytes (0 = null or empty row: its rectangle is NaN).  The envelope of a point is
 * the point: POINT columns are reported (GPK_ERR_MISMATCHED_GEOMETRY), pass them through unchanged. */
int32_t gpk_envelope(const gpk_geoarray* a, double* out_xy, uint8_t* out_valid, int32_t out_space, void* stream);
/* exterior (geoseries.rs:43-47): the outer ring of each polygon as a LINESTRING column.  POLYGON columns only
 * (GeopolarsError::MismatchedGeometry otherwise).  out_geom_offsets[n_geoms+1]; out_xy capacity 2*n_coords(a) doubles
 * (NULL = size query); *n_out_coords = coordinates written.  Null rows give empty linestrings. */
int32_t gpk_exterior(const gpk_geoarray* a, double* out_xy, int32_t* out_geom_offsets, int64_t* n_out_coords,
                     int32_t out_space, void* stream);
/* explode (geoseries.rs:49-50; benches/explode.rs:10-24): one row per member of a multi-part geometry.  Pure offset
 * surgery — MULTIPOINT -> POINT, MULTILINESTRING -> LINESTRING, MULTIPOLYGON -> POLYGON, single-part columns explode to
 * themselves — so *out is a VIEW of `a` (coordinates and inner offsets are shared: `a` must outlive it; free it with
 * gpk_geoarray_free).  Members of a null row are null.  out_parent (optional, n_members i32 in `parent_space`): the row
 * each member came from — the index a dataframe repeats its other columns by. */
int32_t gpk_explode(const gpk_geoarray* a, int32_t* out_parent, int32_t parent_space, void* stream, gpk_geoarray** out);
/* rows of a handle (n_geoms) — e.g. of an exploded view */
int32_t gpk_geoarray_len(const gpk_geoarray* a, int64_t* out_n);
/* geom_type (geoseries.rs:60-73): the column's pygeos type id per row, -1 for null rows.  out[n_geoms] i8 */
int32_t gpk_geom_type(const gpk_geoarray* a, int8_t* out, int32_t out_space, void* stream);
/* is_empty (geoseries.rs:75-76; geo HasDimensions::is_empty): out[n_geoms] bytes 0/1, 0 for null rows */
int32_t gpk_is_empty(const gpk_geoarray* a, uint8_t* out, int32_t out_space, void* stream);
/* is_ring (geoseries.rs:78-83; geo-types LineString::is_closed: first == last, an empty linestring counts as closed).
 * LINESTRING columns only.  out[n_geoms] bytes 0/1, 0 for null rows */
int32_t gpk_is_ring(const gpk_geoarray* a, uint8_t* out, int32_t out_space, void* stream);
/* x / y (geoseries.rs:177-180): POINT columns only; either output may be NULL; NaN for null rows */
int32_t gpk_point_xy(const gpk_geoarray* a, double* out_x, double* out_y, int32_t out_space, void* stream);
/* geodesic_length (geoseries.rs:52-58,216-218; methods of py-geopolars/src/geo.rs:61-78): metres along the ellipsoid /
 * sphere, coordinates = (lon, lat) degrees; linestrings = their segments, polygons = exterior rings only, points = 0
 * (the row rule of euclidean_length).  out[n_geoms]; null rows NaN.
 *   GPK_GEODESIC_HAVERSINE  geo 0.27 HaversineLength: great circle on the mean-radius sphere (6371008.8 m)
 *   GPK_GEODESIC_VINCENTY   geo 0.27 VincentyLength: Vincenty's inverse formula on WGS84; a row holding a segment upstream
 *                           answers with Err(FailedToConverge) (nearly antipodal end points) is NaN
 *   GPK_GEODESIC_KARNEY     "geodesic", the Python default: geo 0.27 GeodesicLength = Karney's algorithm (J. Geodesy 87, 2013,
 *                           through geographiclib-rs) on WGS84 — order-6 series, Newton's method with bisection fallback */
#define GPK_GEODESIC_KARNEY    0
#define GPK_GEODESIC_HAVERSINE 1
#define GPK_GEODESIC_VINCENTY  2
int32_t gpk_geodesic_length(const gpk_geoarray* a, int32_t method, double* out, int32_t out_space, void* stream);
/* simplify (geoseries.rs:108-116,240-242): Ramer-Douglas-Peucker as geo 0.27 runs it (algorithm/simplify.rs): per
 * coordinate sequence (linestring / ring), the farthest point from the chord decides (the LAST one among equals), a
 * range whose farthest point is within `epsilon` loses its interior unless the sequence would drop below 2 (linestrings)
 * / 4 (polygon rings) coordinates; end points are always kept; epsilon <= 0 returns the input.  The nesting above the
 * sequences does not change: the result has the input's geom / part offsets and
 *   out_seq_offsets[n_seq + 1]   the new innermost offsets (ring_offsets, or geom_offsets of a LINESTRING column)
 *   out_xy                       capacity 2 * n_coords(a) doubles (NULL = size query); *n_out_coords = coordinates kept.
 * LINESTRING / MULTILINESTRING / POLYGON / MULTIPOLYGON columns (points: GPK_ERR_MISMATCHED_GEOMETRY, pass through). */
int32_t gpk_simplify(const gpk_geoarray* a, double epsilon, double* out_xy, int32_t* out_seq_offsets,
                     int64_t* n_out_coords, int32_t out_space, void* stream);
/* convex_hull: geoseries.rs:23-26,196-198.  Output = POLYGON array, one closed CCW ring per geometry.
 * out_ring_offsets[n_geoms+1]; out_xy capacity must be >= 2*(n_coords + n_geoms) doubles. */
int32_t gpk_convex_hull(const gpk_geoarray* a, double* out_xy, int32_t* out_ring_offsets,
                        int32_t out_space, void* stream);

/* ---- row-wise binary operators ------------------------------------------------------------ */
/* distance: geoseries.rs:141-146,248-251 ("1-to-1 row-wise").  `b_rows` (optional, same space as
 * out) maps row i of `a` to row b_rows[i] of `b` — the take() a caller would otherwise materialise;
 * NULL = identity (then n_geoms must match); an entry >= n_geoms(b) behaves like a null row of b
 * (distance NaN, predicate false — the out-of-range rule of gpk_take_*).  out[n_geoms(a)].  Supported: POINT x {POINT,
 * LINESTRING, POLYGON, MULTIPOLYGON, MULTILINESTRING, MULTIPOINT} and the mirrored pairs. */
int32_t gpk_distance_rowwise(const gpk_geoarray* a, const gpk_geoarray* b, const uint32_t* b_rows,
                             double* out, int32_t out_space, void* stream);
/* A row map that is used more than once (a dataframe's foreign-key column joined against the same geometry column for
 * every batch of points) can be prepared ONCE: gpk_rowmap_build orders the left rows by target (targets by descending
 * vertex count, so that the 64 rows one wave takes walk equally long linestrings) and keeps the order in HBM;
 * gpk_distance_rowmap then runs the distance kernel alone, stream-ordered.  LINESTRING right sides (the grouped schedule);
 * gpk_distance_rowwise with b_rows builds, uses and drops such a map internally when a target has 8 or more rows on
 * average.  b_rows in `rows_space`; the build synchronises the stream. */
typedef struct gpk_rowmap gpk_rowmap;
int32_t gpk_rowmap_build(const gpk_geoarray* b, const uint32_t* b_rows, int64_t n_rows, int32_t rows_space,
                         void* stream, gpk_rowmap** out);
int32_t gpk_rowmap_free(gpk_rowmap* map);
int32_t gpk_rowmap_nbytes(const gpk_rowmap* map, int64_t* out_bytes);
/* out[n_geoms(a)] as gpk_distance_rowwise(a, b, b_rows, ...) would fill it; a POINT, b the LINESTRING array of the map */
int32_t gpk_distance_rowmap(const gpk_geoarray* a, const gpk_geoarray* b, const gpk_rowmap* map, double* out,
                            int32_t out_space, void* stream);
/* contains / within / intersects row-wise (north-star additions to the trait; semantics from the
 * dispatch table spatial_index.rs:89-137, geo 0.27 traits).  out[n] bytes 0/1.  Pairs with an answer:
 * point x polygonal (all three), polygonal x polygonal (intersects; contains / within = "the contained side
 * is not empty and a subset of the other", every member of a multipolygon counted), lineal contains point /
 * point within lineal, point x point (equality); any other pair is false. */
int32_t gpk_predicate_rowwise(const gpk_geoarray* a, const gpk_geoarray* b, const uint32_t* b_rows,
                              int32_t predicate, uint8_t* out, int32_t out_space, void* stream);

/* ---- spatial index + join: spatial_index.rs:37-204,314-350 -------------------------------- */
/* SpatialIndex::try_from(&Series) (spatial_index.rs:320-334): bbox per geometry + a uniform-grid
 * directory over the bboxes (the GPU replacement for rstar's R-tree) + per-polygon edge slabs. */
int32_t gpk_index_build(const gpk_geoarray* a, void* stream, gpk_index** out);
/* The same with a choice of tables and, optionally, precomputed leaves:
 *   parts      GPK_INDEX_BBOX_GRID (always built: what rstar holds, the candidate generator of every join arm)
 *              | GPK_INDEX_PIP (point-in-polygon raster + edge slabs; only point x polygonal joins read them — a
 *                polygon x polygon join served by an index without them costs a fraction of the build).  A point join
 *                against an index built without GPK_INDEX_PIP still answers exactly, through the slow generic walk.
 *   bbox4_dev  NULL, or n_geoms x (minx, miny, maxx, maxy) in DEVICE memory (NaN x4 = empty geometry): the leaves
 *              another rank computed with gpk_bounds for its shard and all-gathered — the bounds
 *              pass over the coordinates is skipped.
 * gpk_index_build(a, ...) == gpk_index_build_ex(a, GPK_INDEX_BBOX_GRID | GPK_INDEX_PIP, NULL, ...). */
#define GPK_INDEX_BBOX_GRID 1
#define GPK_INDEX_PIP       2
#define GPK_INDEX_PIP_LIGHT 4 /* with GPK_INDEX_PIP: no per-entry level-2 records for cells where several parts meet — about half
                                 the build time on overlapping right sides for ~10 % slower point joins: what gpk_spatial_join builds
                                 for itself when it is handed no index (an index that serves one join) */
#define GPK_INDEX_PIP_FULL  8 /* with GPK_INDEX_PIP: those records ALWAYS.  By default a column of very many small parts (more than two per
                                 cell of the 2048 x 2048 raster: 5M power-law multipolygons) gets a 4096 x 4096 raster, plain entry lists
                                 and a box per part instead: 118 ms / 1.7 GB / 2.0 ms per 6.25M-point join against 215 ms / 2.7 GB / 1.6 ms
                                 with the records — worth it for an index that serves a few hundred joins */
int32_t gpk_index_build_ex(const gpk_geoarray* a, int32_t parts, const double* bbox4_dev, void* stream,
                           gpk_index** out);
int32_t gpk_index_free(gpk_index* idx);
/* A query on the index by itself — what the reference's own index tests do (spatial_index.rs:383-393,422-429):
 * `r_tree.locate_in_envelope(&AABB::from_corners(lo, hi))` = every leaf (one per indexed geometry: its bounding box) CONTAINED in the
 * query box, `locate_in_envelope_intersecting` = every leaf that meets it; both with closed intervals (a leaf touching the query box's
 * border is contained / intersecting — rstar AABB::contains_envelope / intersects).  Batched: n_boxes query boxes
 *   boxes4[4 * n_boxes]      two corners (x0, y0, x1, y1) per query, in `space`; ordered as AABB::from_corners orders them (component-wise
 *                            min / max); a box with a NaN matches nothing
 *   out_counts[n_boxes]      u32 leaves per query (may be NULL)
 *   out_pairs[2 * capacity]  u32 (query, geometry index) interleaved, sorted by (query, index) (NULL + capacity 0: count only)
 *   *n_pairs                 total, always set (GPK_ERR_CAPACITY when > capacity and pairs were asked for)
 * Null and empty geometries have no leaf.  No geometry is read: the answer comes from the index's boxes and grid directory. */
#define GPK_QUERY_CONTAINED    0 /* rstar RTree::locate_in_envelope */
#define GPK_QUERY_INTERSECTING 1 /* rstar RTree::locate_in_envelope_intersecting */
int32_t gpk_index_query_envelope(const gpk_index* idx, const double* boxes4, int64_t n_boxes, int32_t mode,
                                 uint32_t* out_counts, uint32_t* out_pairs, int64_t pair_capacity, int64_t* n_pairs,
                                 int32_t space, void* stream);
int32_t gpk_index_nbytes(const gpk_index* idx, int64_t* out_bytes);
/* What the index holds (tests and bench.py report it; nothing in a join depends on the caller knowing):
 *   out = {raster side R (0: no point-in-polygon tables), one-part-per-cell ("lean") 0/1, local chains 0/1 (`test` sub-cells
 *          decided from one or two ring edges in the owning lane), LDS routing image 0/1 (R <= 512), entry lists dominate 0/1 (overlapping parts,
 *          very many small parts: the general tile kernel runs one point per lane), 0, 0, 0} */
int32_t gpk_index_describe(const gpk_index* idx, int64_t out[8]);

/*
 * spatial_join refine (spatial_index.rs:74-143): all (l, r) with predicate(left[l], right[r]) true,
 * emitted SORTED by (l, r) (rstar's traversal order is unspecified; the shim may compare sets).
 *   out_counts[n_left]   u32 hits per left row (may be NULL)
 *   out_pairs[2*cap]     u32 (l, r) interleaved (may be NULL when cap == 0: count-only mode)
 *   *n_pairs             total hits (always set; GPK_ERR_CAPACITY if > cap and pairs requested)
 * `right_index` may be NULL (built on the fly like spatial_index.rs:60-71).  The index such a call builds STAYS on the right-side
 * handle WHEN THE HANDLE OWNS ITS BUFFERS (host uploads, gpk_geoarray_from_wkb / _from_arrow, results of this library: nobody can
 * change those bytes, so the index cannot go stale) and is freed with it: the reference's default call shape —
 * SpatialJoinArgs::default() has r_index: None, spatial_index.rs:24-35 — repeated against the same series pays for one build.
 * A handle that BORROWS device buffers (GPK_MEM_DEVICE descriptors: the caller may rewrite them between calls) gets a fresh index
 * in every such call.  Only indexes of at most GPK_AUTO_INDEX_MAX_MB (environment, default 256) are kept; GPK_AUTO_INDEX=0 builds and
 * frees per call; gpk_geoarray_nbytes includes the kept indexes.
 * Geometry dispatch = the match of spatial_index.rs:89-137: point <-> polygon / multipolygon on either side
 * (`poly.contains(point)` whatever the predicate), polygonal x polygonal `intersects`, polygon / multipolygon x POLYGON
 * `contains` (:99-101,107-111; upstream's DE-9IM relate restated as "right is not empty and a subset of left"),
 * point <-> linestring / multilinestring on either side (`line.contains(point)`); every other combination (e.g. `contains`
 * with a multipolygon on the right, `within` for polygonal pairs) is upstream's `_ => false`: an empty result, not an error.
 * `left_row_base` is added to every emitted l (row-sharded multi-GPU runs).
 */
int32_t gpk_spatial_join(const gpk_geoarray* left, const gpk_geoarray* right,
                         const gpk_index* right_index, int32_t predicate, uint32_t left_row_base,
                         uint32_t* out_counts, uint32_t* out_pairs, int64_t pair_capacity,
                         int64_t* n_pairs, int32_t out_space, void* stream);

/*
 * Stream-ordered form of gpk_spatial_join for callers that keep everything in HBM (the idiom a pipeline of
 * kernels on one HIP stream wants; the reference's call is synchronous, spatial_index.rs:44-58): the join is
 * ENQUEUED on `stream` and the call returns without waiting.
 *   - point x polygon / multipolygon only, `right_index` required (nothing to build or free behind the stream);
 *   - out_counts / out_pairs are device buffers (either may be NULL as above);
 *   - *n_pairs_dev (device or device-mapped host memory, may be NULL) receives the total number of hits when the
 *     stream reaches that point; pairs beyond pair_capacity are dropped, so compare the two after synchronising.
 * Scratch comes from an arena owned by (calling thread, stream): calls enqueued on different streams do not share
 * buffers; calls on ONE stream reuse them in stream order.
 */
int32_t gpk_spatial_join_async(const gpk_geoarray* left, const gpk_geoarray* right,
                               const gpk_index* right_index, int32_t predicate, uint32_t left_row_base,
                               uint32_t* out_counts, uint32_t* out_pairs, int64_t pair_capacity,
                               int64_t* n_pairs_dev, void* stream);

/* ---- join assembly (spatial_index.rs:145-203) ------------------------------------------------ */
/* The reference turns the (l, r) pairs into two u64 index Series and lets polars `inner_join` / `left_join` pull
 * the attribute columns (spatial_index.rs:147-199).  Here: the row indices of the joined table, then a gather per
 * column.  All buffers of one call live in `space`. */
#define GPK_JOIN_INNER 0
#define GPK_JOIN_LEFT  1
/* (counts[n_left], sorted pairs[2*n_pairs] as produced by gpk_spatial_join with `left_row_base`) -> out_l / out_r
 * [capacity] i64 row indices, sorted by l.  Left join: a left row without hits appears once with r = -1
 * (JoinType::Left, spatial_index.rs:186-199); other join types do not exist upstream (:200-202).
 * *n_rows is always set (capacity 0 = size query; GPK_ERR_CAPACITY when it does not fit). */
int32_t gpk_join_indices(const uint32_t* counts, const uint32_t* pairs, int64_t n_left, int64_t n_pairs,
                         uint32_t left_row_base, int32_t join_type, int64_t* out_l, int64_t* out_r,
                         int64_t capacity, int64_t* n_rows, int32_t space, void* stream);
/* out[i] = values[idx[i]] for a fixed-width Arrow column (elem_bits 1 = boolean bitmap, 8, 16, 32, 64, 128);
 * idx[i] = -1 (or out of range) and null source rows give a null: out_validity (Arrow bitmap, may be NULL). */
int32_t gpk_take_fixed(const void* values, int32_t elem_bits, const uint8_t* validity, int64_t n_values,
                       const int64_t* idx, int64_t n_idx, void* out_values, uint8_t* out_validity,
                       int32_t space, void* stream);
/* The same for an Arrow Binary / Utf8 column (i32 offsets + bytes; the reference's geometry column is one).
 * out_offsets[n_idx + 1]; out_values NULL + capacity 0 = size query; *n_bytes always set. */
int32_t gpk_take_binary(const uint8_t* values, const int32_t* offsets, const uint8_t* validity, int64_t n_values,
                        const int64_t* idx, int64_t n_idx, int32_t* out_offsets, uint8_t* out_values,
                        int64_t capacity, int64_t* n_bytes, uint8_t* out_validity, int32_t space, void* stream);

/* ---- chunked columns ------------------------------------------------------------------------------------ */
/* K chunks of one column held by this process -> ONE array (a new handle, gpk_geoarray_free): Arrow's rechunk — the reference turns
 * every Series into a single chunk before it looks at it (py-geopolars/src/ffi.rs:56,73,93).  Device-resident: the chunks' buffers
 * are placed with device copies, offsets rebased (a chunk's offsets need not start at 0: a sliced Arrow array), validity bits
 * repacked across chunk boundaries that do not fall on a byte.  The placement / rebase / repack code is the all-gatherv's own
 * (below): what runs here with K chunks is what runs there with K ranks.  All chunks must have the same geometry type
 * (GPK_ERR_MISMATCHED_GEOMETRY); 1 <= n_chunks <= 64.  out_row_bases[n_chunks + 1] (host, may be NULL): first row of every chunk. */
int32_t gpk_geoarray_concat(const gpk_geoarray* const* chunks, int32_t n_chunks, void* stream, gpk_geoarray** out,
                            int64_t* out_row_bases, int64_t* out_bytes);

/* ---- multi-GPU: the one collective of the path ------------------------------------ */
/* One process per GPU; the LEFT series is sharded by rows and needs no collective (disjoint output rows, pairs carry
 * `left_row_base`).  A RIGHT side that is itself produced sharded is exchanged once — where `spatial_join` receives its right
 * side and its index, spatial_index.rs:37-76 — with an all-gatherv of its GeoArrow buffers over RCCL / xGMI, and the leaves of
 * its index (per-geometry boxes: the NodeEnvelopes of spatial_index.rs:206-312, gpk_bounds of the shard) travel the same way,
 * so that gpk_index_build_ex assembles the gathered index from them.  Device-resident end to end: lengths first (one small
 * all-gather, the only host read), then one grouped round of broadcasts per buffer with every piece landing at its final
 * offset; offsets are rebased and validity repacked on the device.  RCCL is opened at run time (GPK_RCCL_PATH, else the
 * copy the process already holds, else the system's): there is no link-time dependency.
 *   gpk_comm_unique_id   rank 0 draws the 128-byte id (ncclUniqueId) and hands it to the other ranks by any side channel
 *   gpk_comm_init        every rank, same id: a communicator on the current device (collective call)
 *   gpk_allgatherv_geoarray   shard -> the whole column in rank order (a new handle, gpk_geoarray_free); *out_row_base = first
 *                        row of this rank's shard in it; every rank passes the same geometry type (GPK_ERR_MISMATCHED_GEOMETRY)
 *   gpk_allgatherv_rows_f64   n_local rows of `width` doubles (device) -> all rows in rank order; out_dev NULL = total only
 *                        (still a collective: every rank must make the same call); out_counts[world] host, may be NULL */
typedef struct gpk_comm gpk_comm;
int32_t gpk_comm_unique_id(uint8_t out_id[128]);
int32_t gpk_comm_init(int32_t rank, int32_t world, const uint8_t id[128], gpk_comm** out);
int32_t gpk_comm_free(gpk_comm* comm);
int32_t gpk_comm_info(const gpk_comm* comm, int32_t* out_rank, int32_t* out_world);
/* An IN-PROCESS transport for tests of the exchange (no RCCL): `world` threads of one process on one device stand in for the ranks.
 * gpk_comm_mock_world makes the meeting place, every thread opens its communicator on it with gpk_comm_init_mock and calls
 * gpk_allgatherv_* as ranks would — every line of the exchange but RCCL's own runs, with ranks that are apart in time.  Free the
 * communicators first, then the world. */
int32_t gpk_comm_mock_world(int32_t world, void** out_world);
int32_t gpk_comm_init_mock(int32_t rank, void* world, gpk_comm** out);
int32_t gpk_comm_mock_world_free(void* world);
int32_t gpk_allgatherv_geoarray(gpk_comm* comm, const gpk_geoarray* shard, void* stream, gpk_geoarray** out,
                                int64_t* out_row_base, int64_t* out_bytes);
int32_t gpk_allgatherv_rows_f64(gpk_comm* comm, const double* local_dev, int64_t n_local, int32_t width, double* out_dev,
                                int64_t out_capacity_rows, int64_t* out_total_rows, int64_t* out_counts, void* stream);

/* ---- join statistics (bench.py's edge_tests/s) ---------------------------------- */
/* While enabled, the point x polygonal join kernels count what their exact phase does (a few atomics per tile:
 * leave it off in timed regions).  out = {(point, part) pairs sent to the exact winding walk, edges walked for them,
 * left rows a chain-kernel join deferred to the generic walk (list cells, sub-cells without a chain entry, orientations
 * the floating-point filter could not certify), 0}, accumulated over the joins since the last reset; gpk_join_stats waits
 * for the device. */
int32_t gpk_join_stats_enable(int32_t on);
int32_t gpk_join_stats(int64_t out[4], int32_t reset);

/* ---- profiling hooks (bench.py's roofline leg) -------------------------------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on its stream. */
int32_t gpk_profile_enable(int32_t on);
/* Restrict the bracketing to kernels whose name contains `substr` (NULL or "" = every kernel).  Every event pair
 * drains the stream around its kernel (a few microseconds), so a timed region brackets only what it reports. */
int32_t gpk_profile_filter(const char* substr);
int32_t gpk_profile_reset(void);
/* accumulated milliseconds + launch count of kernels whose name contains `substr` */
int32_t gpk_profile_query(const char* substr, double* out_ms, int64_t* out_launches);

#ifdef __cplusplus
}
#endif
#endif /* GEOPOLARS_HIP_H */
