// gpk_wkb.cpp — host-side WKB -> GeoArrow decoder (the one-time replacement for the per-op,
// per-row `Wkb(value.to_vec()).to_geo()` of geopolars/geopolars-geo/src/util.rs:27-37, which the
// reference's README.md:83 calls out as the dominant cost).  Handles ISO WKB and EWKB (SRID flag),
// both byte orders, 2D only; types 1-6.  Pure host code: no device involved.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "gpk_common.h"

namespace {

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool le = true;

    int extra = 0;  // Z / M ordinates per coordinate of the geometry being read (dropped: GeoSeries is 2D, as geozero's to_geo is)
    void skip_extra() {
        for (int i = 0; i < extra; ++i) (void)f64();
    }
    uint8_t u8() {
        if (p + 1 > end) {
            ok = false;
            return 0;
        }
        return *p++;
    }
    uint32_t u32() {
        if (p + 4 > end) {
            ok = false;
            return 0;
        }
        uint32_t v;
        if (le) {
            v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        } else {
            v = (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24);
        }
        p += 4;
        return v;
    }
    double f64() {
        if (p + 8 > end) {
            ok = false;
            return 0;
        }
        uint64_t v = 0;
        if (le)
            for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
        else
            for (int i = 0; i < 8; ++i) v = (v << 8) | p[i];
        p += 8;
        double d;
        memcpy(&d, &v, 8);
        return d;
    }
};

struct Sink {  // counts always; writes when buffers are present
    int64_t n_parts = 0, n_rings = 0, n_coords = 0;
    double* xy = nullptr;
    int32_t* part_off = nullptr;
    int32_t* ring_off = nullptr;
    void coord(double x, double y) {
        if (xy) {
            xy[2 * n_coords] = x;
            xy[2 * n_coords + 1] = y;
        }
        ++n_coords;
    }
    void end_ring() {
        ++n_rings;
        if (ring_off) ring_off[n_rings] = (int32_t)n_coords;
    }
    void end_part() {
        ++n_parts;
        if (part_off) part_off[n_parts] = (int32_t)n_rings;
    }
};

// header: returns base type 1..6 or 0 on error
int read_header(Reader& r) {
    const uint8_t bo = r.u8();
    if (!r.ok || bo > 1) return 0;
    r.le = bo == 1;
    uint32_t t = r.u32();
    if (!r.ok) return 0;
    if (t & 0x20000000u) (void)r.u32();  // EWKB SRID
    // Z / M ordinates are read past: EWKB flags 0x80000000 (Z) / 0x40000000 (M); ISO type codes 1000 + t (Z), 2000 + t (M), 3000 + t (ZM)
    int extra = ((t & 0x80000000u) ? 1 : 0) + ((t & 0x40000000u) ? 1 : 0);
    t &= 0x0FFFFFFFu;
    if (t >= 1000) {
        const uint32_t dim = t / 1000;
        if (dim > 3 || extra) return 0;
        extra = dim == 3 ? 2 : 1;
        t %= 1000;
    }
    if (t < 1 || t > 6) return 0;  // (7 = GeometryCollection: no GeoArrow nesting holds it)
    r.extra = extra;
    return (int)t;
}

enum Family { FAM_NONE = 0, FAM_POINT, FAM_LINE, FAM_POLY };
Family family_of(int t) {
    switch (t) {
    case 1:
    case 4: return FAM_POINT;
    case 2:
    case 5: return FAM_LINE;
    case 3:
    case 6: return FAM_POLY;
    }
    return FAM_NONE;
}

void read_coords(Reader& r, uint32_t n, Sink& s) {
    for (uint32_t i = 0; i < n && r.ok; ++i) {
        const double x = r.f64(), y = r.f64();
        r.skip_extra();
        if (r.ok) s.coord(x, y);
    }
}
void read_polygon_body(Reader& r, Sink& s) {
    const uint32_t nr = r.u32();
    for (uint32_t k = 0; k < nr && r.ok; ++k) {
        const uint32_t n = r.u32();
        read_coords(r, n, s);
        s.end_ring();
    }
}

}  // namespace

using namespace gpk;

extern "C" int32_t gpk_wkb_decode(const uint8_t* values, const int32_t* offsets, int64_t n_rows,
                                  const uint8_t* validity, int64_t counts[5], double* xy,
                                  int32_t* geom_offsets, int32_t* part_offsets, int32_t* ring_offsets) {
    if (!offsets || !counts || (n_rows > 0 && !values))
        return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_wkb_decode: NULL argument");
    const bool fill = xy != nullptr || geom_offsets != nullptr;

    // pass A: family + whether any multi-geometry appears
    Family fam = FAM_NONE;
    bool any_multi = false;
    for (int64_t i = 0; i < n_rows; ++i) {
        if (validity && !((validity[i >> 3] >> (i & 7)) & 1)) continue;
        Reader r{values + offsets[i], values + offsets[i + 1]};
        const int t = read_header(r);
        if (!t) return fail(GPK_ERR_MISMATCHED_GEOMETRY, "row %lld: unsupported or malformed WKB", (long long)i);
        const Family f = family_of(t);
        if (fam == FAM_NONE) fam = f;
        if (f != fam)
            return fail(GPK_ERR_MISMATCHED_GEOMETRY,
                        "row %lld: mixed geometry families in one column (expected family %d, found type %d)",
                        (long long)i, (int)fam, t);
        any_multi |= t >= 4;
    }
    if (fam == FAM_NONE) fam = FAM_POINT;
    int out_type;
    if (fam == FAM_POINT)
        out_type = any_multi ? GPK_GEOM_MULTIPOINT : GPK_GEOM_POINT;
    else if (fam == FAM_LINE)
        out_type = any_multi ? GPK_GEOM_MULTILINESTRING : GPK_GEOM_LINESTRING;
    else
        out_type = any_multi ? GPK_GEOM_MULTIPOLYGON : GPK_GEOM_POLYGON;

    Sink s;
    if (fill) {
        s.xy = xy;
        // which offset levels exist for out_type
        if (out_type == GPK_GEOM_MULTIPOLYGON) {
            s.part_off = part_offsets;
            s.ring_off = ring_offsets;
            if (!part_offsets || !ring_offsets || !geom_offsets)
                return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_wkb_decode: offset buffers missing");
            part_offsets[0] = 0;
            ring_offsets[0] = 0;
        } else if (out_type == GPK_GEOM_POLYGON || out_type == GPK_GEOM_MULTILINESTRING) {
            s.ring_off = ring_offsets;
            if (!ring_offsets || !geom_offsets)
                return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_wkb_decode: offset buffers missing");
            ring_offsets[0] = 0;
        } else if (out_type != GPK_GEOM_POINT && !geom_offsets) {
            return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_wkb_decode: offset buffers missing");
        }
        if (geom_offsets) geom_offsets[0] = 0;
    }

    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int64_t i = 0; i < n_rows; ++i) {
        const bool valid = !validity || ((validity[i >> 3] >> (i & 7)) & 1);
        if (valid) {
            Reader r{values + offsets[i], values + offsets[i + 1]};
            const int t = read_header(r);
            switch (t) {
            case 1: {
                const double x = r.f64(), y = r.f64();
                r.skip_extra();
                if (out_type == GPK_GEOM_POINT)
                    s.coord(x, y);
                else if (!(std::isnan(x) && std::isnan(y)))
                    s.coord(x, y);  // POINT EMPTY adds nothing to a MULTIPOINT row
                break;
            }
            case 2: {
                const uint32_t n = r.u32();
                read_coords(r, n, s);
                if (out_type == GPK_GEOM_MULTILINESTRING) s.end_ring();
                break;
            }
            case 3:
                read_polygon_body(r, s);
                if (out_type == GPK_GEOM_MULTIPOLYGON) s.end_part();
                break;
            case 4:
            case 5:
            case 6: {
                const uint32_t k = r.u32();
                for (uint32_t m = 0; m < k && r.ok; ++m) {
                    const bool outer_le = r.le;
                    const int ct = read_header(r);
                    if (ct != t - 3) {
                        r.ok = false;
                        break;
                    }
                    if (ct == 1) {
                        const double x = r.f64(), y = r.f64();
                        r.skip_extra();
                        if (!(std::isnan(x) && std::isnan(y))) s.coord(x, y);
                    } else if (ct == 2) {
                        const uint32_t n = r.u32();
                        read_coords(r, n, s);
                        s.end_ring();
                    } else {
                        read_polygon_body(r, s);
                        s.end_part();
                    }
                    r.le = outer_le;
                }
                break;
            }
            default:
                r.ok = false;
            }
            if (!r.ok) return fail(GPK_ERR_INVALID_OFFSETS, "row %lld: truncated or malformed WKB", (long long)i);
        } else if (out_type == GPK_GEOM_POINT) {
            s.coord(nan, nan);  // a null point row still owns one coordinate slot
        }
        if (fill && geom_offsets) {
            int64_t level1;
            switch (out_type) {
            case GPK_GEOM_MULTIPOLYGON: level1 = s.n_parts; break;
            case GPK_GEOM_POLYGON:
            case GPK_GEOM_MULTILINESTRING: level1 = s.n_rings; break;
            default: level1 = s.n_coords;
            }
            if (out_type != GPK_GEOM_POINT) geom_offsets[i + 1] = (int32_t)level1;
        }
        if (s.n_coords > INT32_MAX) return fail(GPK_ERR_INVALID_OFFSETS, "column exceeds i32 offsets");
    }
    counts[0] = out_type;
    counts[1] = n_rows;
    counts[2] = s.n_parts;
    counts[3] = s.n_rings;
    counts[4] = s.n_coords;
    return GPK_OK;
}

// ---- GeoArrow -> WKB, host side (from_geom_vec of util.rs:11-24) -------------------------------------
// Two passes over the rows: sizes -> offsets, then the bytes.  Same byte stream as the device encoder
// (gpk_wkb_encode.hip): little-endian ISO WKB, 2D, one type per column, zero-length records for null rows.
namespace {
struct Writer {
    uint8_t* p;
    void u8(uint8_t v) { *p++ = v; }
    void u32(uint32_t v) {
        memcpy(p, &v, 4);  // little-endian hosts only (x86-64 here)
        p += 4;
    }
    void header(uint32_t type, uint32_t count) {
        u8(1);
        u32(type);
        u32(count);
    }
    // (interleaved coordinates, or the descriptor's separated x / y arrays)
    void coords(const gpk_geoarrow_desc* d, int64_t c0, int64_t c1) {
        if (d->xy) {
            memcpy(p, d->xy + 2 * c0, sizeof(double) * 2 * (size_t)(c1 - c0));
            p += 16 * (c1 - c0);
            return;
        }
        for (int64_t c = c0; c < c1; ++c) {
            memcpy(p, d->x + c, 8);
            memcpy(p + 8, d->y + c, 8);
            p += 16;
        }
    }
};
inline bool row_valid(const uint8_t* v, int64_t i) { return !v || ((v[i >> 3] >> (i & 7)) & 1); }
int64_t host_row_bytes(const gpk_geoarrow_desc* d, int64_t g) {
    if (!row_valid(d->validity, g)) return 0;
    const int32_t* go = d->geom_offsets;
    switch (d->geom_type) {
    case GPK_GEOM_POINT: return 21;
    case GPK_GEOM_LINESTRING: return 9 + 16 * (int64_t)(go[g + 1] - go[g]);
    case GPK_GEOM_MULTIPOINT: return 9 + 21 * (int64_t)(go[g + 1] - go[g]);
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING:
        return 9 + (d->geom_type == GPK_GEOM_POLYGON ? 4 : 9) * (int64_t)(go[g + 1] - go[g]) +
               16 * (int64_t)(d->ring_offsets[go[g + 1]] - d->ring_offsets[go[g]]);
    default: {
        const int32_t r0 = d->part_offsets[go[g]], r1 = d->part_offsets[go[g + 1]];
        return 9 + 9 * (int64_t)(go[g + 1] - go[g]) + 4 * (int64_t)(r1 - r0) + 16 * (int64_t)(d->ring_offsets[r1] - d->ring_offsets[r0]);
    }
    }
}
}  // namespace

extern "C" int32_t gpk_wkb_encode(const gpk_geoarrow_desc* d, int32_t* out_offsets, uint8_t* out_values, int64_t capacity,
                                  int64_t* n_bytes) {
    using gpk::fail;
    if (!d || !n_bytes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->mem_space != GPK_MEM_HOST) return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_wkb_encode takes host buffers (use gpk_geoarray_to_wkb for a device handle)");
    if (capacity < 0 || (capacity > 0 && !out_values)) return fail(GPK_ERR_INVALID_ARGUMENT, "capacity without out_values");
    const int t = d->geom_type;
    if (t != GPK_GEOM_POINT && t != GPK_GEOM_LINESTRING && t != GPK_GEOM_POLYGON && t != GPK_GEOM_MULTIPOINT &&
        t != GPK_GEOM_MULTILINESTRING && t != GPK_GEOM_MULTIPOLYGON)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "unknown geometry type %d", t);
    if (d->n_geoms > 0 && ((t != GPK_GEOM_POINT && !d->geom_offsets) ||
                           ((t == GPK_GEOM_POLYGON || t == GPK_GEOM_MULTILINESTRING || t == GPK_GEOM_MULTIPOLYGON) && !d->ring_offsets) ||
                           (t == GPK_GEOM_MULTIPOLYGON && !d->part_offsets) || (d->n_coords > 0 && !d->xy && !(d->x && d->y))))
        return fail(GPK_ERR_INVALID_OFFSETS, "missing offsets / coordinates for geometry type %d", t);
    int64_t total = 0;
    if (out_offsets) out_offsets[0] = 0;
    for (int64_t g = 0; g < d->n_geoms; ++g) {
        total += host_row_bytes(d, g);
        if (total > 0x7FFFFFFFLL)
            return fail(GPK_ERR_CAPACITY, "wkb_encode: the column does not fit BinaryArray<i32> offsets; encode row slices");
        if (out_offsets) out_offsets[g + 1] = (int32_t)total;
    }
    *n_bytes = total;
    if (!out_values) return GPK_OK;
    if (total > capacity) return fail(GPK_ERR_CAPACITY, "wkb_encode: %lld bytes but capacity %lld", (long long)total, (long long)capacity);
    Writer w{out_values};
    const int32_t *go = d->geom_offsets, *po = d->part_offsets, *ro = d->ring_offsets;
    auto polygon = [&](int32_t r0, int32_t r1) {
        w.header(3u, (uint32_t)(r1 - r0));
        for (int32_t r = r0; r < r1; ++r) {
            w.u32((uint32_t)(ro[r + 1] - ro[r]));
            w.coords(d, ro[r], ro[r + 1]);
        }
    };
    for (int64_t g = 0; g < d->n_geoms; ++g) {
        if (!row_valid(d->validity, g)) continue;
        switch (t) {
        case GPK_GEOM_POINT:
            w.u8(1);
            w.u32(1u);
            w.coords(d, g, g + 1);
            break;
        case GPK_GEOM_LINESTRING:
            w.header(2u, (uint32_t)(go[g + 1] - go[g]));
            w.coords(d, go[g], go[g + 1]);
            break;
        case GPK_GEOM_MULTIPOINT:
            w.header(4u, (uint32_t)(go[g + 1] - go[g]));
            for (int32_t i = go[g]; i < go[g + 1]; ++i) {
                w.u8(1);
                w.u32(1u);
                w.coords(d, i, i + 1);
            }
            break;
        case GPK_GEOM_POLYGON: polygon(go[g], go[g + 1]); break;
        case GPK_GEOM_MULTILINESTRING:
            w.header(5u, (uint32_t)(go[g + 1] - go[g]));
            for (int32_t l = go[g]; l < go[g + 1]; ++l) {
                w.header(2u, (uint32_t)(ro[l + 1] - ro[l]));
                w.coords(d, ro[l], ro[l + 1]);
            }
            break;
        default:
            w.header(6u, (uint32_t)(go[g + 1] - go[g]));
            for (int32_t q = go[g]; q < go[g + 1]; ++q) polygon(po[q], po[q + 1]);
        }
    }
    return GPK_OK;
}
