// gpk_structural.hip — the operators of `trait GeoSeries` that only re-arrange or inspect a column:
//   envelope      geoseries.rs:28-33    bounding rectangle as a geometry (geo: BoundingRect -> Rect::to_polygon)
//   exterior      geoseries.rs:43-47    outer ring of each polygon as a linestring column
//   explode       geoseries.rs:49-50    one row per member of a multi-part geometry (benches/explode.rs:10-24)
//   geom_type     geoseries.rs:60-73    pygeos type ids, -1 for missing
//   is_empty      geoseries.rs:75-76
//   is_ring       geoseries.rs:78-83    closed linestrings (geo-types LineString::is_closed)
//   x / y         geoseries.rs:177-180
//   rotate / scale / skew about a per-geometry origin   geoseries.rs:85-139, TransformOrigin of py-geopolars/src/utils.rs:5-27
// None of them is arithmetic-heavy; they are here so that a Rust shim needs no geometry code of its own:
// every trait method has one entry point.  All are maps over rows (or offset surgery), stream-ordered for device outputs.
#include <cmath>

#include "gpk_device.h"
#include "gpk_index.h"
#include "gpk_scan.h"

namespace gpk {

// coordinate range of one row (every nesting level resolved)
__device__ __forceinline__ void row_coords(const DevGeo& a, int64_t g, int& c0, int& c1) {
    switch (a.type) {
    case GPK_GEOM_POINT:
        c0 = (int)g;
        c1 = (int)g + 1;
        break;
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT:
        c0 = a.geom_off[g];
        c1 = a.geom_off[g + 1];
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING:
        c0 = a.ring_off[a.geom_off[g]];
        c1 = a.ring_off[a.geom_off[g + 1]];
        break;
    default:
        c0 = a.ring_off[a.part_off[a.geom_off[g]]];
        c1 = a.ring_off[a.part_off[a.geom_off[g + 1]]];
    }
}

// envelope: the closed 5-coordinate rectangle (minx miny, maxx miny, maxx maxy, minx maxy, minx miny) of geo's Rect::to_polygon
__global__ void envelope_kernel(const double4* __restrict__ box, const uint8_t* __restrict__ validity, int64_t n, double2* __restrict__ out,
                                uint8_t* __restrict__ out_valid) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const double4 b = box[g];
    const bool ok = dev::valid_row(validity, g) && b.x == b.x;  // NaN box: empty geometry -> null
    double2* o = out + 5 * g;
    o[0] = make_double2(b.x, b.y);
    o[1] = make_double2(b.z, b.y);
    o[2] = make_double2(b.z, b.w);
    o[3] = make_double2(b.x, b.w);
    o[4] = make_double2(b.x, b.y);
    if (out_valid) out_valid[g] = ok ? 1 : 0;
}

// exterior: coordinates of ring geom_off[g] (none for a null row or a polygon without rings)
__global__ void exterior_sizes_kernel(DevGeo a, int32_t* __restrict__ sizes) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const int r0 = a.geom_off[g], r1 = a.geom_off[g + 1];
    sizes[g] = (dev::valid_row(a.validity, g) && r1 > r0) ? a.ring_off[r0 + 1] - a.ring_off[r0] : 0;
}
template <int G>
__global__ __launch_bounds__(256) void exterior_copy_kernel(DevGeo a, const int32_t* __restrict__ off, double2* __restrict__ out) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t g = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; g < a.n_geoms; g += groups) {
        const int o = off[g], n = off[g + 1] - o;
        if (n == 0) continue;
        const int c0 = a.ring_off[a.geom_off[g]];
        for (int i = lane; i < n; i += G) out[o + i] = a.xy[c0 + i];
    }
}

// explode: validity of the members (a member of a null row is null) and the row each member came from
__global__ void explode_members_kernel(DevGeo a, uint8_t* __restrict__ member_valid_bits, int32_t* __restrict__ parent) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const bool ok = dev::valid_row(a.validity, g);
    const int m0 = a.type == GPK_GEOM_POINT || a.type == GPK_GEOM_LINESTRING || a.type == GPK_GEOM_POLYGON ? (int)g : a.geom_off[g];
    const int m1 = a.type == GPK_GEOM_POINT || a.type == GPK_GEOM_LINESTRING || a.type == GPK_GEOM_POLYGON ? (int)g + 1 : a.geom_off[g + 1];
    for (int m = m0; m < m1; ++m) {
        if (parent) parent[m] = (int32_t)g;
        if (member_valid_bits && ok) atomicOr(reinterpret_cast<unsigned int*>(member_valid_bits) + (m >> 5), 1u << (m & 31));
    }
}

__global__ void geom_type_kernel(DevGeo a, int8_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < a.n_geoms) out[g] = dev::valid_row(a.validity, g) ? (int8_t)a.type : (int8_t)-1;
}

// geo's HasDimensions::is_empty: a point without coordinates (NaN here), a line without coordinates, a polygon whose
// exterior has none, a multi-geometry all of whose members are empty
__global__ void is_empty_kernel(DevGeo a, uint8_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    bool empty = true;
    switch (a.type) {
    case GPK_GEOM_POINT: {
        const double2 p = a.xy[g];
        empty = !(p.x == p.x) || !(p.y == p.y);
        break;
    }
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT:
        empty = a.geom_off[g + 1] == a.geom_off[g];
        break;
    case GPK_GEOM_MULTILINESTRING: {
        int c0, c1;
        row_coords(a, g, c0, c1);
        empty = c1 == c0;
        break;
    }
    default: {  // polygonal: some member polygon has a non-empty exterior
        int p0, p1;
        dev::geom_parts(a, g, p0, p1);
        for (int p = p0; p < p1 && empty; ++p) {
            int r0, r1;
            dev::part_rings(a, p, r0, r1);
            if (r1 > r0 && a.ring_off[r0 + 1] > a.ring_off[r0]) empty = false;
        }
    }
    }
    out[g] = (dev::valid_row(a.validity, g) && empty) ? 1 : 0;
}

// LineString::is_closed of geo-types 0.7: first == last; an EMPTY linestring counts as closed (the crate adopts the JTS
// LinearRing rule for every linestring — documented there)
__global__ void is_ring_kernel(DevGeo a, uint8_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const int c0 = a.geom_off[g], c1 = a.geom_off[g + 1];
    bool closed = true;
    if (c1 > c0) {
        const double2 f = a.xy[c0], l = a.xy[c1 - 1];
        closed = f.x == l.x && f.y == l.y;
    }
    out[g] = (dev::valid_row(a.validity, g) && closed) ? 1 : 0;
}

__global__ void point_xy_kernel(DevGeo a, double* __restrict__ x, double* __restrict__ y) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const bool ok = dev::valid_row(a.validity, g);
    const double2 p = a.xy[g];
    if (x) x[g] = ok ? p.x : NAN;
    if (y) y[g] = ok ? p.y : NAN;
}

// one affine matrix [a, b, xoff, d, e, yoff] per geometry for a rotation / scaling / skew about its origin.  The
// expressions are evaluated in the order the reference's own formulas have (geoseries.rs:129-138 for the skew), so that the
// result is the same IEEE value a host implementation of those formulas produces (the library is built -ffp-contract=off).
//   kind 0 rotate: k0 = cos, k1 = sin      [ c, -s, ox - c ox + s oy,  s, c, oy - s ox - c oy ]
//   kind 1 scale:  k0 = xfact, k1 = yfact  [ xf, 0, ox (1 - xf),  0, yf, oy (1 - yf) ]
//   kind 2 skew:   k0 = tan xs, k1 = tan ys [ 1, tx, -oy tx,  ty, 1, -ox ty ]
// origin: 0 = centroid (cxy), 1 = centre of the bounding box (box), 2 = the point (px, py)
__global__ void origin_matrices_kernel(int64_t n, int kind, double k0, double k1, int origin, const double2* __restrict__ cxy,
                                       const double4* __restrict__ box, double px, double py, double* __restrict__ m) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    double ox = px, oy = py;
    if (origin == 0) {
        ox = cxy[g].x;
        oy = cxy[g].y;
    } else if (origin == 1) {
        const double4 b = box[g];
        ox = (b.x + b.z) / 2.0;
        oy = (b.y + b.w) / 2.0;
    }
    double* o = m + 6 * g;
    if (kind == 0) {
        o[0] = k0;
        o[1] = -k1;
        o[2] = ox - k0 * ox + k1 * oy;
        o[3] = k1;
        o[4] = k0;
        o[5] = oy - k1 * ox - k0 * oy;
    } else if (kind == 1) {
        o[0] = k0;
        o[1] = 0.0;
        o[2] = ox * (1.0 - k0);
        o[3] = 0.0;
        o[4] = k1;
        o[5] = oy * (1.0 - k1);
    } else {
        o[0] = 1.0;
        o[1] = k0;
        o[2] = -oy * k0;
        o[3] = k1;
        o[4] = 1.0;
        o[5] = -ox * k1;
    }
}

static inline dim3 rows_grid(int64_t n) { return dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)); }

// small map with one output array: stage through the auxiliary arena for host outputs
template <typename T, typename LAUNCH>
static int32_t map_out(T* out, int64_t n, int32_t out_space, hipStream_t s, LAUNCH&& launch) {
    if (n == 0) return GPK_OK;
    T* dev = out;
    if (out_space != GPK_MEM_DEVICE) {
        GPK_TRY(workspace_aux(0).begin(sizeof(T) * (size_t)n + 256));
        dev = (T*)workspace_aux(0).take(sizeof(T) * (size_t)n);
    }
    GPK_TRY(launch(dev));
    return copy_out(out, out_space, dev, sizeof(T) * (size_t)n, s);
}

}  // namespace gpk

using namespace gpk;

extern "C" {

int32_t gpk_envelope(const gpk_geoarray* a, double* out_xy, uint8_t* out_valid, int32_t out_space, void* stream) {
    if (!a || (!out_xy && a->d.n_geoms > 0)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");  // (an empty column has no output to point at)
    GPK_TRY(require_device());
    if (a->d.type == GPK_GEOM_POINT)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "envelope: the envelope of a point is the point itself — pass a POINT column through unchanged");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    const bool host = out_space != GPK_MEM_DEVICE;
    const size_t bb = sizeof(double4) * (size_t)n, ob = sizeof(double2) * 5 * (size_t)n;
    GPK_TRY(workspace_aux(0).begin(align256(bb) + (host ? align256(ob) + align256((size_t)n) : 0) + 512));
    double4* box = (double4*)workspace_aux(0).take(bb);
    double2* out_dev = host ? (double2*)workspace_aux(0).take(ob) : (double2*)out_xy;
    uint8_t* valid_dev = out_valid ? (host ? (uint8_t*)workspace_aux(0).take((size_t)n) : out_valid) : nullptr;
    GPK_TRY(gpk_bounds(a, (double*)box, GPK_MEM_DEVICE, stream));
    GPK_LAUNCH("gpk_envelope", envelope_kernel, rows_grid(n), dim3(256), 0, s, (const double4*)box, a->d.validity, n, out_dev, valid_dev);
    if (out_valid) GPK_TRY(copy_out(out_valid, out_space, valid_dev, (size_t)n, s));
    return copy_out(out_xy, out_space, out_dev, ob, s);
}

int32_t gpk_exterior(const gpk_geoarray* a, double* out_xy, int32_t* out_geom_offsets, int64_t* n_out_coords, int32_t out_space, void* stream) {
    if (!a || !out_geom_offsets || !n_out_coords) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    if (a->d.type != GPK_GEOM_POLYGON)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "exterior: expected a POLYGON column (found type %d)", a->d.type);  // MismatchedGeometry, error.rs:12-16
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms, nc = a->d.n_coords;
    const bool host = out_space != GPK_MEM_DEVICE;
    const size_t offb = sizeof(int32_t) * (size_t)(n + 1), xyb = sizeof(double2) * (size_t)(nc > 0 ? nc : 1);
    const int64_t nb = (n + 255) / 256;
    GPK_TRY(workspace_aux(0).begin(2 * align256(offb) + align256(sizeof(unsigned long long) * (size_t)(nb + 2)) + (host ? align256(xyb) : 0) + 512));
    int32_t* sizes = (int32_t*)workspace_aux(0).take(offb);
    int32_t* off_dev = host ? (int32_t*)workspace_aux(0).take(offb) : out_geom_offsets;
    unsigned long long* btot = (unsigned long long*)workspace_aux(0).take(sizeof(unsigned long long) * (size_t)(nb + 2));
    double2* out_dev = host ? (double2*)workspace_aux(0).take(xyb) : (double2*)out_xy;
    if (n == 0) {
        GPK_HIP(hipMemsetAsync(off_dev, 0, sizeof(int32_t), s));
        *n_out_coords = 0;
        return copy_out(out_geom_offsets, out_space, off_dev, sizeof(int32_t), s);
    }
    GPK_LAUNCH("gpk_exterior_sizes", exterior_sizes_kernel, rows_grid(n), dim3(256), 0, s, a->d, sizes);
    GPK_TRY(exclusive_scan_i32(sizes, n, off_dev, nullptr, btot, s));
    if (out_xy) {
        const double mean = (double)nc / (double)n;
        int64_t blocks = (n + 15) / 16;
        const int64_t cap = (int64_t)cu_count() * 16;
        if (blocks > cap) blocks = cap;
        if (mean <= 12.0)
            GPK_LAUNCH("gpk_exterior_copy", exterior_copy_kernel<4>, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, s, a->d, (const int32_t*)off_dev, out_dev);
        else
            GPK_LAUNCH("gpk_exterior_copy", exterior_copy_kernel<16>, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, s, a->d, (const int32_t*)off_dev, out_dev);
    }
    unsigned long long total = 0;  // the call reports the size: one read-back
    GPK_HIP(hipMemcpyAsync(&total, btot + nb, sizeof total, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    *n_out_coords = (int64_t)total;
    if (host) {
        GPK_TRY(copy_out(out_geom_offsets, out_space, off_dev, offb, s));
        if (out_xy) GPK_TRY(copy_out(out_xy, out_space, out_dev, sizeof(double2) * (size_t)total, s));
    }
    return GPK_OK;
}

int32_t gpk_explode(const gpk_geoarray* a, int32_t* out_parent, int32_t parent_space, void* stream, gpk_geoarray** out) {
    if (!a || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const DevGeo& d = a->d;
    gpk_geoarray* e = new gpk_geoarray;
    memset(e, 0, sizeof *e);
    e->device = a->device;
    e->d = d;  // a view: coordinates and inner offsets are the input's (the input must outlive the result)
    e->d.validity = nullptr;
    int64_t members = d.n_geoms;
    switch (d.type) {
    case GPK_GEOM_MULTIPOINT:  // every coordinate becomes a POINT row
        e->d.type = GPK_GEOM_POINT;
        members = d.n_coords;
        e->d.geom_off = nullptr;
        break;
    case GPK_GEOM_MULTILINESTRING:  // every member linestring a LINESTRING row: the ring offsets become the row offsets
        e->d.type = GPK_GEOM_LINESTRING;
        members = d.n_rings;
        e->d.geom_off = d.ring_off;
        e->d.ring_off = nullptr;
        e->d.n_rings = 0;
        break;
    case GPK_GEOM_MULTIPOLYGON:  // every member polygon a POLYGON row: the part offsets become the row offsets
        e->d.type = GPK_GEOM_POLYGON;
        members = d.n_parts;
        e->d.geom_off = d.part_off;
        e->d.part_off = nullptr;
        break;
    default:  // single-part columns explode to themselves
        break;
    }
    e->d.n_geoms = members;
    if (is_polygonal(e->d.type)) e->d.n_parts = members;
    e->nbytes = a->nbytes;
    const bool multi = e->d.type != d.type;
    auto cleanup = [&](int32_t rc) {
        gpk_geoarray_free(e);
        return rc;
    };
    const bool need_valid = d.validity != nullptr && members > 0;
    int32_t* parent_dev = nullptr;
    const bool host_parent = out_parent && parent_space != GPK_MEM_DEVICE;
    if (out_parent && members > 0) {
        parent_dev = out_parent;
        if (host_parent) {
            const int32_t rc = workspace_aux(0).begin(sizeof(int32_t) * (size_t)members + 256);
            if (rc != GPK_OK) return cleanup(rc);
            parent_dev = (int32_t*)workspace_aux(0).take(sizeof(int32_t) * (size_t)members);
        }
    }
    if (need_valid) {  // the members' bitmap is owned by the result
        const size_t vb = (((size_t)members + 31) / 32) * 4;
        void* v = nullptr;
        if (device_malloc(&v, vb) != hipSuccess) return cleanup(fail(GPK_ERR_OOM, "explode: device_malloc(%zu) failed", vb));
        e->owned[4] = v;
        e->d.validity = (const uint8_t*)v;
        if (hipMemsetAsync(v, 0, vb, s) != hipSuccess) return cleanup(fail(GPK_ERR_DEVICE, "explode: memset failed"));
    }
    if ((need_valid || parent_dev) && d.n_geoms > 0) {
        auto run = [&]() -> int32_t {
            GPK_LAUNCH("gpk_explode_members", explode_members_kernel, rows_grid(d.n_geoms), dim3(256), 0, s, d, need_valid ? (uint8_t*)e->owned[4] : (uint8_t*)nullptr,
                       parent_dev);
            return GPK_OK;
        };
        const int32_t rc = run();
        if (rc != GPK_OK) return cleanup(rc);
    }
    (void)multi;
    if (host_parent && members > 0) {
        const int32_t rc = copy_out(out_parent, parent_space, parent_dev, sizeof(int32_t) * (size_t)members, s);
        if (rc != GPK_OK) return cleanup(rc);
    }
    *out = e;
    return GPK_OK;
}

int32_t gpk_geom_type(const gpk_geoarray* a, int8_t* out, int32_t out_space, void* stream) {
    if (!a || (!out && a->d.n_geoms > 0)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    return map_out(out, n, out_space, s, [&](int8_t* dev) -> int32_t {
        GPK_LAUNCH("gpk_geom_type", geom_type_kernel, rows_grid(n), dim3(256), 0, s, a->d, dev);
        return GPK_OK;
    });
}

int32_t gpk_is_empty(const gpk_geoarray* a, uint8_t* out, int32_t out_space, void* stream) {
    if (!a || (!out && a->d.n_geoms > 0)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    return map_out(out, n, out_space, s, [&](uint8_t* dev) -> int32_t {
        GPK_LAUNCH("gpk_is_empty", is_empty_kernel, rows_grid(n), dim3(256), 0, s, a->d, dev);
        return GPK_OK;
    });
}

int32_t gpk_is_ring(const gpk_geoarray* a, uint8_t* out, int32_t out_space, void* stream) {
    if (!a || (!out && a->d.n_geoms > 0)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    if (a->d.type != GPK_GEOM_LINESTRING)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "is_ring: expected a LINESTRING column (found type %d)", a->d.type);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    return map_out(out, n, out_space, s, [&](uint8_t* dev) -> int32_t {
        GPK_LAUNCH("gpk_is_ring", is_ring_kernel, rows_grid(n), dim3(256), 0, s, a->d, dev);
        return GPK_OK;
    });
}

int32_t gpk_point_xy(const gpk_geoarray* a, double* out_x, double* out_y, int32_t out_space, void* stream) {
    if (!a || (!out_x && !out_y)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    if (a->d.type != GPK_GEOM_POINT) return fail(GPK_ERR_MISMATCHED_GEOMETRY, "x / y: expected a POINT column (found type %d)", a->d.type);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    const bool host = out_space != GPK_MEM_DEVICE;
    const size_t b = sizeof(double) * (size_t)n;
    double *xd = out_x, *yd = out_y;
    if (host) {
        GPK_TRY(workspace_aux(0).begin(2 * align256(b) + 256));
        if (out_x) xd = (double*)workspace_aux(0).take(b);
        if (out_y) yd = (double*)workspace_aux(0).take(b);
    }
    GPK_LAUNCH("gpk_point_xy", point_xy_kernel, rows_grid(n), dim3(256), 0, s, a->d, xd, yd);
    if (out_x) GPK_TRY(copy_out(out_x, out_space, xd, b, s));
    if (out_y) GPK_TRY(copy_out(out_y, out_space, yd, b, s));
    return GPK_OK;
}

int32_t gpk_affine_about_origin(const gpk_geoarray* a, int32_t kind, double p0, double p1, int32_t origin, double ox, double oy, double* out_xy,
                                int32_t out_space, void* stream) {
    if (!a || (!out_xy && a->d.n_coords > 0)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (kind < GPK_AFFINE_ROTATE || kind > GPK_AFFINE_SKEW) return fail(GPK_ERR_INVALID_ARGUMENT, "unknown transform kind %d", kind);
    if (origin < GPK_ORIGIN_CENTROID || origin > GPK_ORIGIN_POINT) return fail(GPK_ERR_INVALID_ARGUMENT, "unknown origin %d", origin);  // "Invalid argument", utils.rs:21
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    if (n == 0 || a->d.n_coords == 0) return GPK_OK;
    // parameters of the matrix: angles arrive in degrees (geoseries.rs:85-93,118-139)
    const double rad = 3.14159265358979323846 / 180.0;
    double k0 = p0, k1 = p1;
    if (kind == GPK_AFFINE_ROTATE) {
        const double t = p0 * rad;
        k0 = cos(t);
        k1 = sin(t);
    } else if (kind == GPK_AFFINE_SKEW) {
        k0 = tan(p0 * rad);
        k1 = tan(p1 * rad);
    }
    const size_t mb = sizeof(double) * 6 * (size_t)n, ob = sizeof(double4) * (size_t)n;
    GPK_TRY(workspace_aux(1).begin(align256(mb) + align256(ob) + align256((size_t)n) + 512));
    double* mats = (double*)workspace_aux(1).take(mb);
    void* org = workspace_aux(1).take(ob);  // centroids (double2) or boxes (double4)
    uint8_t* cvalid = (uint8_t*)workspace_aux(1).take((size_t)n);
    if (origin == GPK_ORIGIN_CENTROID) GPK_TRY(gpk_centroid(a, (double*)org, cvalid, GPK_MEM_DEVICE, stream));
    if (origin == GPK_ORIGIN_CENTER) GPK_TRY(gpk_bounds(a, (double*)org, GPK_MEM_DEVICE, stream));
    GPK_LAUNCH("gpk_origin_matrices", origin_matrices_kernel, rows_grid(n), dim3(256), 0, s, n, (int)kind, k0, k1, (int)origin, (const double2*)org,
               (const double4*)org, ox, oy, mats);
    return affine_rows_impl(a, mats, GPK_MEM_DEVICE, out_xy, out_space, s);
}

}  // extern "C"
