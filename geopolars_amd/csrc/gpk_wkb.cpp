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
    if (t & 0xC0000000u) return 0;        // EWKB Z / M
    t &= 0x0FFFFFFFu;
    if (t >= 1000) return 0;  // ISO Z / M / ZM
    if (t < 1 || t > 6) return 0;
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
