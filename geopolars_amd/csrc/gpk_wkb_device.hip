// gpk_wkb_device.hip — WKB -> GeoArrow decoding on the GPU (the step right before the
// hot path; the reference pays a per-row, per-op WKB parse, geopolars/geopolars-geo/src/util.rs:27-37, and
// calls it "expensive", README.md:83).  The raw WKB column (Arrow BinaryArray<i32>: values + offsets) is
// copied to HBM once and decoded there, so the GeoArrow SoA never exists on the host:
//   scan  : one lane per row parses the headers and counts parts / rings / coordinates;
//   scans : three exclusive scans give every row its output positions;
//   fill  : one lane per row re-parses and writes coordinates and offsets.
// Handles little-endian ISO WKB and EWKB with SRID, 2D, types 1-6 with the same promotion rules as the host
// decoder (gpk_wkb.cpp); anything else (big-endian, Z/M, mixed families, truncation) is reported, never guessed.
#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

struct WkbCursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok;
    __device__ __forceinline__ uint32_t u8() {
        if (p + 1 > end) {
            ok = false;
            return 0;
        }
        return *p++;
    }
    __device__ __forceinline__ uint32_t u32() {
        if (p + 4 > end) {
            ok = false;
            return 0;
        }
        uint32_t v;
        __builtin_memcpy(&v, p, 4);  // one (unaligned) dword load: legal on gfx9+ under HSA, little-endian like the payload
        p += 4;
        return v;
    }
    __device__ __forceinline__ double f64() {
        if (p + 8 > end) {
            ok = false;
            return 0.0;
        }
        double v;
        __builtin_memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    __device__ __forceinline__ void skip(size_t n) {
        if (p + n > end)
            ok = false;
        else
            p += n;
    }
    // header -> base type 1..6, 0 on error (big-endian, Z/M, unknown)
    __device__ __forceinline__ int header() {
        const uint32_t bo = u8();
        if (!ok || bo != 1u) {
            ok = false;
            return 0;
        }
        uint32_t t = u32();
        if (t & 0x20000000u) (void)u32();
        if (!ok || (t & 0xC0000000u)) {
            ok = false;
            return 0;
        }
        t &= 0x0FFFFFFFu;
        if (t < 1u || t > 6u) {
            ok = false;
            return 0;
        }
        return (int)t;
    }
};

struct RowCount {
    int32_t type;    // 0 = null row
    int32_t k;       // sub-geometries of a multi type (points / linestrings / polygons); 1 otherwise
    int32_t rings;   // rings (polygon family) or linestrings (line family)
    int32_t coords;  // coordinates, POINT EMPTY members of a MULTIPOINT not counted
};

// `emit` < 0: count only.  Otherwise write coordinates from position `emit` and ring / part end offsets.
template <bool FILL>
__device__ inline bool parse_row(WkbCursor& r, RowCount& rc, double2* __restrict__ xy, int64_t cpos, int32_t* __restrict__ ring_off,
                                 int64_t rpos, int32_t* __restrict__ part_off, int64_t ppos, bool out_point,
                                 const uint8_t* __restrict__ values = nullptr, int32_t* __restrict__ seq_src = nullptr,
                                 int64_t seq_pos = 0) {
    const int t = r.header();
    if (!t) return false;
    rc.type = t;
    rc.k = 1;
    rc.rings = 0;
    rc.coords = 0;
    auto put = [&](double x, double y) {
        if (FILL) xy[cpos + rc.coords] = make_double2(x, y);
        ++rc.coords;
    };
    auto end_ring = [&]() {
        ++rc.rings;
        if (FILL && ring_off) ring_off[rpos + rc.rings] = (int32_t)(cpos + rc.coords);
    };
    int parts_done = 0;
    auto end_part = [&]() {
        ++parts_done;
        if (FILL && part_off) part_off[ppos + parts_done] = (int32_t)(rpos + rc.rings);
    };
    // a run of n coordinates (linestring / ring body) is never read here: the scan only needs its length, and the fill
    // records where it starts so that wkb_copy_kernel moves it with many lanes (a 100k-vertex ring is one header to
    // the lane that owns the row)
    int seqs_done = 0;
    auto coords_run = [&](uint32_t n) {
        if (FILL && seq_src) seq_src[seq_pos + seqs_done] = (int32_t)(r.p - values);
        ++seqs_done;
        r.skip(16 * (size_t)n);
        if (r.ok) rc.coords += (int32_t)n;
    };
    auto polygon_body = [&]() {
        const uint32_t nr = r.u32();
        for (uint32_t q = 0; q < nr && r.ok; ++q) {
            const uint32_t n = r.u32();
            coords_run(n);
            end_ring();
        }
    };
    switch (t) {
    case 1: {
        const double x = r.f64(), y = r.f64();
        if (out_point || !(isnan(x) && isnan(y))) put(x, y);
        break;
    }
    case 2: {
        const uint32_t n = r.u32();
        coords_run(n);
        end_ring();  // counted as one linestring; only used when the column is MULTILINESTRING
        break;
    }
    case 3:
        polygon_body();
        end_part();
        break;
    default: {
        const uint32_t k = r.u32();
        rc.k = (int32_t)k;
        for (uint32_t m = 0; m < k && r.ok; ++m) {
            const int ct = r.header();
            if (ct != t - 3) {
                r.ok = false;
                break;
            }
            if (ct == 1) {
                const double x = r.f64(), y = r.f64();
                if (!(isnan(x) && isnan(y))) put(x, y);
            } else if (ct == 2) {
                const uint32_t n = r.u32();
                coords_run(n);
                end_ring();
            } else {
                polygon_body();
                end_part();
            }
        }
    }
    }
    return r.ok;
}

__global__ void wkb_scan_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ offsets, int64_t n_rows,
                                const uint8_t* __restrict__ validity, RowCount* __restrict__ rows, uint32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    RowCount rc{0, 0, 0, 0};
    uint32_t bits = 0;
    if (i < n_rows && dev::valid_row(validity, i)) {
        WkbCursor r{values + offsets[i], values + offsets[i + 1], true};
        if (!parse_row<false>(r, rc, nullptr, 0, nullptr, 0, nullptr, 0, false)) {
            bits = 0x80000000u;  // malformed / unsupported
            rc = RowCount{0, 0, 0, 0};
        } else {
            bits = 1u << rc.type;
        }
    }
    if (i < n_rows) rows[i] = rc;
    // the column's type flags: OR-ed across the wave, and only a wave that brings a NEW bit touches the word (one atomicOr per row on
    // one address was 8M serialised atomics for a column of 8M small polygons — most of this kernel's 0.73 ms)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, o, 64);
    if ((threadIdx.x & 63) == 0 && bits && (bits & ~__atomic_load_n(flags, __ATOMIC_RELAXED)) != 0u) atomicOr(flags, bits);
}

// per-row output extents for the chosen column type
__global__ void wkb_extent_kernel(const RowCount* __restrict__ rows, int64_t n_rows, int out_type, int32_t* __restrict__ n_coords,
                                  int32_t* __restrict__ n_rings, int32_t* __restrict__ n_parts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const RowCount rc = rows[i];
    int c = rc.coords, rg = 0, pt = 0;
    if (out_type == GPK_GEOM_POINT) {
        c = 1;  // a null or empty point row still owns one (NaN) coordinate slot
    } else if (out_type == GPK_GEOM_MULTILINESTRING || out_type == GPK_GEOM_POLYGON || out_type == GPK_GEOM_MULTIPOLYGON) {
        rg = rc.rings;
        if (out_type == GPK_GEOM_MULTIPOLYGON) pt = rc.type == 0 ? 0 : (rc.type == 3 ? 1 : rc.k);
    }
    n_coords[i] = c;
    n_rings[i] = rg;
    n_parts[i] = pt;
}

__global__ void wkb_fill_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ offsets, int64_t n_rows,
                                const uint8_t* __restrict__ validity, int out_type, const int32_t* __restrict__ cpos,
                                const int32_t* __restrict__ rpos, const int32_t* __restrict__ ppos, double2* __restrict__ xy,
                                int32_t* __restrict__ geom_off, int32_t* __restrict__ part_off, int32_t* __restrict__ ring_off,
                                int32_t* __restrict__ seq_src) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const bool out_point = out_type == GPK_GEOM_POINT;
    const bool has_ring = out_type == GPK_GEOM_MULTILINESTRING || out_type == GPK_GEOM_POLYGON || out_type == GPK_GEOM_MULTIPOLYGON;
    const bool has_part = out_type == GPK_GEOM_MULTIPOLYGON;
    if (dev::valid_row(validity, i)) {
        WkbCursor r{values + offsets[i], values + offsets[i + 1], true};
        RowCount rc;
        // sequences are numbered like the rings (ring types) or like the rows (a LINESTRING column: one per row)
        (void)parse_row<true>(r, rc, xy, cpos[i], has_ring ? ring_off : nullptr, has_ring ? rpos[i] : 0, has_part ? part_off : nullptr,
                              has_part ? ppos[i] : 0, out_point, values, seq_src, has_ring ? (int64_t)rpos[i] : i);
    } else if (seq_src && !has_ring) {
        seq_src[i] = 0;  // null row of a LINESTRING column: an empty sequence
    } else if (out_point) {
        xy[cpos[i]] = make_double2(NAN, NAN);
    }
    if (i == 0) {
        if (geom_off) geom_off[0] = 0;
        if (has_ring) ring_off[0] = 0;
        if (has_part) part_off[0] = 0;
    }
    if (geom_off) {
        // level-1 offsets: end position of row i at the column's first nesting level
        const int32_t* lvl = has_part ? ppos : (has_ring ? rpos : cpos);
        geom_off[i + 1] = lvl[i + 1];
    }
}

// ---- coordinate runs: WKB bytes -> xy ---------------------------------------------------------------------------
// sequence q (a ring / member line, or the row of a LINESTRING column) starts at values + seq_src[q] and fills
// xy[seq_off[q] .. seq_off[q + 1]).  8 lanes per sequence; sequences longer than WKB_LONG are listed for
// wkb_copy_long_kernel, which spreads each of them over the whole grid.
constexpr int WKB_COPY_GS = 8, WKB_LONG = 4096;
__device__ __forceinline__ double2 load_xy_unaligned(const uint8_t* p) {
    double2 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__global__ __launch_bounds__(256) void wkb_copy_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ seq_src,
                                                        const int32_t* __restrict__ seq_off, int64_t n_seq, double2* __restrict__ xy,
                                                        int32_t* __restrict__ long_list) {
    const int lane = threadIdx.x & (WKB_COPY_GS - 1);
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WKB_COPY_GS;
    if (q >= n_seq) return;
    const int c0 = seq_off[q], n = seq_off[q + 1] - c0;
    if (n <= 0) return;
    if (n > WKB_LONG) {
        if (lane == 0) long_list[1 + atomicAdd(&long_list[0], 1)] = (int32_t)q;
        return;
    }
    const uint8_t* src = values + seq_src[q];
    for (int i = lane; i < n; i += WKB_COPY_GS) xy[c0 + i] = load_xy_unaligned(src + 16 * (size_t)i);
}
__global__ __launch_bounds__(256) void wkb_copy_long_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ seq_src,
                                                             const int32_t* __restrict__ seq_off, const int32_t* __restrict__ long_list,
                                                             double2* __restrict__ xy) {
    const int n_long = long_list[0];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int k = 0; k < n_long; ++k) {
        const int q = long_list[1 + k];
        const int c0 = seq_off[q], n = seq_off[q + 1] - c0;
        const uint8_t* src = values + seq_src[q];
        for (int64_t i = tid; i < n; i += stride) xy[c0 + i] = load_xy_unaligned(src + 16 * (size_t)i);
    }
}

}  // namespace gpk

using namespace gpk;

extern "C" int32_t gpk_geoarray_from_wkb(const uint8_t* wkb_values, const int32_t* wkb_offsets, int64_t n_rows, const uint8_t* validity,
                                         int32_t mem_space, void* stream, gpk_geoarray** out, int32_t* out_geom_type) {
    if (!out || !wkb_offsets || (n_rows > 0 && !wkb_values)) return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_geoarray_from_wkb: NULL argument");
    *out = nullptr;
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    if (n_rows > INT32_MAX) return fail(GPK_ERR_INVALID_OFFSETS, "column exceeds i32 offsets");

    // temporaries of this call (freed on every path); the decoded buffers are owned by the returned handle
    void* tmp[16] = {nullptr};
    int n_tmp = 0;
    gpk_geoarray* a = nullptr;
    auto done = [&](int32_t rc) {
        for (int i = 0; i < n_tmp; ++i) (void)hipFree(tmp[i]);
        if (rc != GPK_OK && a) gpk_geoarray_free(a);
        return rc;
    };
    // temporaries come from the thread's auxiliary arena while they fit (capped: mapping gigabytes costs more than the
    // allocations it replaces); the decoded buffers are the handle's own allocations
    {
        size_t est = (size_t)n_rows * 64 + (1u << 20);
        if (est > (size_t(256) << 20)) est = size_t(256) << 20;
        (void)workspace_aux(1).begin(est);
    }
    auto dalloc = [&](void** p, size_t bytes, bool temporary) -> int32_t {
        if (temporary) {
            *p = workspace_aux(1).take(bytes ? bytes : 8);
            if (*p) return GPK_OK;
        }
        hipError_t e = device_malloc(p, bytes ? bytes : 8);
        if (e != hipSuccess) return fail(GPK_ERR_OOM, "gpk_geoarray_from_wkb: device_malloc(%zu): %s", bytes, hipGetErrorString(e));
        if (temporary) tmp[n_tmp++] = *p;
        return GPK_OK;
    };
#define W_TRY(x)                              \
    do {                                      \
        int32_t _rc = (x);                    \
        if (_rc != GPK_OK) return done(_rc);  \
    } while (0)
#define W_HIP(x)                                                                                        \
    do {                                                                                                \
        hipError_t _e = (x);                                                                            \
        if (_e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "%s: %s", #x, hipGetErrorString(_e)));   \
    } while (0)

    const uint8_t* values_dev = wkb_values;
    const int32_t* offsets_dev = wkb_offsets;
    const uint8_t* validity_dev = validity;
    if (mem_space == GPK_MEM_HOST) {
        const size_t vbytes = n_rows > 0 ? (size_t)wkb_offsets[n_rows] : 0;
        void *v = nullptr, *o = nullptr, *vd = nullptr;
        W_TRY(dalloc(&v, vbytes, true));
        W_TRY(dalloc(&o, sizeof(int32_t) * (size_t)(n_rows + 1), true));
        W_HIP(hipMemcpyAsync(v, wkb_values, vbytes, hipMemcpyHostToDevice, s));
        W_HIP(hipMemcpyAsync(o, wkb_offsets, sizeof(int32_t) * (size_t)(n_rows + 1), hipMemcpyHostToDevice, s));
        values_dev = (const uint8_t*)v;
        offsets_dev = (const int32_t*)o;
        if (validity) {
            W_TRY(dalloc(&vd, (size_t)((n_rows + 7) / 8), true));
            W_HIP(hipMemcpyAsync(vd, validity, (size_t)((n_rows + 7) / 8), hipMemcpyHostToDevice, s));
            validity_dev = (const uint8_t*)vd;
        }
    }
    RowCount* rows = nullptr;
    uint32_t* flags = nullptr;
    int32_t *nc = nullptr, *nr = nullptr, *np = nullptr, *cpos = nullptr, *rpos = nullptr, *ppos = nullptr;
    unsigned long long* btot = nullptr;
    W_TRY(dalloc((void**)&rows, sizeof(RowCount) * (size_t)n_rows, true));
    W_TRY(dalloc((void**)&flags, 64, true));
    const size_t ib = sizeof(int32_t) * (size_t)(n_rows + 1);
    W_TRY(dalloc((void**)&nc, ib, true));
    W_TRY(dalloc((void**)&nr, ib, true));
    W_TRY(dalloc((void**)&np, ib, true));
    W_TRY(dalloc((void**)&cpos, ib, true));
    W_TRY(dalloc((void**)&rpos, ib, true));
    W_TRY(dalloc((void**)&ppos, ib, true));
    W_TRY(dalloc((void**)&btot, sizeof(unsigned long long) * (size_t)((n_rows + 255) / 256 + 4), true));
    W_HIP(hipMemsetAsync(flags, 0, 64, s));
    const dim3 grid((unsigned)((n_rows + 255) / 256 > 0 ? (n_rows + 255) / 256 : 1)), block(256);
    auto launch1 = [&]() -> int32_t {
        if (n_rows > 0)
            GPK_LAUNCH("gpk_wkb_scan", wkb_scan_kernel, grid, block, 0, s, values_dev, offsets_dev, n_rows, validity_dev, rows, flags);
        return GPK_OK;
    };
    W_TRY(launch1());
    uint32_t hflags = 0;
    W_HIP(hipMemcpyAsync(&hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    W_HIP(hipStreamSynchronize(s));
    if (hflags & 0x80000000u)
        return done(fail(GPK_ERR_MISMATCHED_GEOMETRY, "gpk_geoarray_from_wkb: malformed, big-endian or Z/M WKB in the column (use gpk_wkb_decode on the host)"));
    const bool fp = hflags & ((1u << 1) | (1u << 4)), fl = hflags & ((1u << 2) | (1u << 5)), fg = hflags & ((1u << 3) | (1u << 6));
    if ((int)fp + (int)fl + (int)fg > 1)
        return done(fail(GPK_ERR_MISMATCHED_GEOMETRY, "gpk_geoarray_from_wkb: mixed geometry families in one column"));
    const bool multi = hflags & ((1u << 4) | (1u << 5) | (1u << 6));
    int out_type = GPK_GEOM_POINT;
    if (fl) out_type = multi ? GPK_GEOM_MULTILINESTRING : GPK_GEOM_LINESTRING;
    if (fg) out_type = multi ? GPK_GEOM_MULTIPOLYGON : GPK_GEOM_POLYGON;
    if (fp) out_type = multi ? GPK_GEOM_MULTIPOINT : GPK_GEOM_POINT;

    int32_t tot_c = 0, tot_r = 0, tot_p = 0;
    auto launch2 = [&]() -> int32_t {
        if (n_rows > 0) {
            GPK_LAUNCH("gpk_wkb_extent", wkb_extent_kernel, grid, block, 0, s, (const RowCount*)rows, n_rows, out_type, nc, nr, np);
            GPK_TRY(exclusive_scan_i32(nc, n_rows, cpos, nullptr, btot, s));
            GPK_TRY(exclusive_scan_i32(nr, n_rows, rpos, nullptr, btot, s));
            GPK_TRY(exclusive_scan_i32(np, n_rows, ppos, nullptr, btot, s));
            GPK_HIP(hipMemcpyAsync(&tot_c, cpos + n_rows, 4, hipMemcpyDeviceToHost, s));
            GPK_HIP(hipMemcpyAsync(&tot_r, rpos + n_rows, 4, hipMemcpyDeviceToHost, s));
            GPK_HIP(hipMemcpyAsync(&tot_p, ppos + n_rows, 4, hipMemcpyDeviceToHost, s));
            GPK_HIP(hipStreamSynchronize(s));
        }
        return GPK_OK;
    };
    W_TRY(launch2());

    a = new gpk_geoarray;
    memset(a, 0, sizeof *a);
    W_HIP(hipGetDevice(&a->device));
    const bool has_ring = out_type == GPK_GEOM_MULTILINESTRING || out_type == GPK_GEOM_POLYGON || out_type == GPK_GEOM_MULTIPOLYGON;
    const bool has_part = out_type == GPK_GEOM_MULTIPOLYGON;
    a->d.type = out_type;
    a->d.n_geoms = n_rows;
    a->d.n_coords = tot_c;
    a->d.n_rings = has_ring ? tot_r : 0;
    a->d.n_parts = has_part ? tot_p : (is_polygonal(out_type) ? n_rows : 0);
    double2* xy = nullptr;
    int32_t *go = nullptr, *po = nullptr, *ro = nullptr;
    W_TRY(dalloc((void**)&xy, sizeof(double2) * (size_t)tot_c, false));
    a->owned[0] = xy;
    if (out_type != GPK_GEOM_POINT) {
        W_TRY(dalloc((void**)&go, sizeof(int32_t) * (size_t)(n_rows + 1), false));
        a->owned[1] = go;
    }
    if (has_part) {
        W_TRY(dalloc((void**)&po, sizeof(int32_t) * (size_t)(tot_p + 1), false));
        a->owned[2] = po;
    }
    if (has_ring) {
        W_TRY(dalloc((void**)&ro, sizeof(int32_t) * (size_t)(tot_r + 1), false));
        a->owned[3] = ro;
    }
    uint8_t* vcopy = nullptr;
    if (validity_dev) {
        W_TRY(dalloc((void**)&vcopy, (size_t)((n_rows + 7) / 8), false));
        a->owned[4] = vcopy;
        W_HIP(hipMemcpyAsync(vcopy, validity_dev, (size_t)((n_rows + 7) / 8), hipMemcpyDeviceToDevice, s));
    }
    // coordinate runs of line / polygon columns are moved by wkb_copy_kernel from the positions the fill records
    const bool has_seq = out_type == GPK_GEOM_LINESTRING || has_ring;
    const int64_t n_seq = has_ring ? (int64_t)tot_r : n_rows;
    int32_t *seq_src = nullptr, *long_list = nullptr;
    if (has_seq && n_rows > 0) {
        W_TRY(dalloc((void**)&seq_src, sizeof(int32_t) * (size_t)(n_seq + 1), true));
        W_TRY(dalloc((void**)&long_list, sizeof(int32_t) * (size_t)(n_seq + 2), true));
        W_HIP(hipMemsetAsync(long_list, 0, sizeof(int32_t), s));
    }
    auto launch3 = [&]() -> int32_t {
        if (n_rows > 0) {
            GPK_LAUNCH("gpk_wkb_fill", wkb_fill_kernel, grid, block, 0, s, values_dev, offsets_dev, n_rows, validity_dev, out_type,
                       (const int32_t*)cpos, (const int32_t*)rpos, (const int32_t*)ppos, xy, go, po, ro, seq_src);
            if (has_seq && n_seq > 0 && tot_c > 0) {
                const int32_t* seq_off = has_ring ? (const int32_t*)ro : (const int32_t*)go;
                GPK_LAUNCH("gpk_wkb_copy", wkb_copy_kernel, dim3((unsigned)((n_seq * WKB_COPY_GS + 255) / 256)), dim3(256), 0, s, values_dev,
                           (const int32_t*)seq_src, seq_off, n_seq, xy, long_list);
                GPK_LAUNCH("gpk_wkb_copy_long", wkb_copy_long_kernel, dim3((unsigned)(cu_count() * 8)), dim3(256), 0, s, values_dev,
                           (const int32_t*)seq_src, seq_off, (const int32_t*)long_list, xy);
            }
        } else if (go) {
            GPK_HIP(hipMemsetAsync(go, 0, sizeof(int32_t), s));
        }
        return GPK_OK;
    };
    W_TRY(launch3());
    W_HIP(hipStreamSynchronize(s));
#undef W_TRY
#undef W_HIP
    a->d.xy = xy;
    a->d.geom_off = go;
    a->d.part_off = po;
    a->d.ring_off = ro;
    a->d.validity = vcopy;
    a->nbytes = (int64_t)(sizeof(double2) * (size_t)tot_c + (go ? sizeof(int32_t) * (size_t)(n_rows + 1) : 0) +
                          (po ? sizeof(int32_t) * (size_t)(tot_p + 1) : 0) + (ro ? sizeof(int32_t) * (size_t)(tot_r + 1) : 0));
    if (out_geom_type) *out_geom_type = out_type;
    *out = a;
    return done(GPK_OK);
}

// Copy a device-resident array back as host GeoArrow buffers (tests, and callers that want the decoded column).
// sizes[4] = {n_coords, n_parts, n_rings, n_geoms}; pass NULL buffers to query sizes only.
extern "C" int32_t gpk_geoarray_download(const gpk_geoarray* a, int64_t sizes[4], double* xy, int32_t* geom_offsets, int32_t* part_offsets,
                                         int32_t* ring_offsets, void* stream) {
    if (!a || !sizes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    sizes[0] = a->d.n_coords;
    sizes[1] = a->d.type == GPK_GEOM_MULTIPOLYGON ? a->d.n_parts : 0;
    sizes[2] = a->d.n_rings;
    sizes[3] = a->d.n_geoms;
    if (xy && a->d.n_coords) GPK_HIP(hipMemcpyAsync(xy, a->d.xy, sizeof(double2) * (size_t)a->d.n_coords, hipMemcpyDeviceToHost, s));
    if (geom_offsets && a->d.geom_off)
        GPK_HIP(hipMemcpyAsync(geom_offsets, a->d.geom_off, sizeof(int32_t) * (size_t)(a->d.n_geoms + 1), hipMemcpyDeviceToHost, s));
    if (part_offsets && a->d.part_off)
        GPK_HIP(hipMemcpyAsync(part_offsets, a->d.part_off, sizeof(int32_t) * (size_t)(a->d.n_parts + 1), hipMemcpyDeviceToHost, s));
    if (ring_offsets && a->d.ring_off)
        GPK_HIP(hipMemcpyAsync(ring_offsets, a->d.ring_off, sizeof(int32_t) * (size_t)(a->d.n_rings + 1), hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}
