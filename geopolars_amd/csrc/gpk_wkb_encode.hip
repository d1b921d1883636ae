// gpk_wkb_encode.hip — GeoArrow -> WKB on the GPU (second half: geometry-valued results —
// centroid, convex_hull, envelope, affine_transform — leave the reference as WKB `binary` series through
// from_geom_vec, geopolars/geopolars-geo/src/util.rs:11-24; this writes that column straight from the HBM-resident
// GeoArrow buffers).  Little-endian ISO WKB, 2D, the single type of the array for every row; null rows are
// zero-length (validity travels separately).
//   sizes   : one lane per row -> bytes of its record (closed form from the offsets);
//   scan    : exclusive scan -> Arrow i32 offsets;
//   headers : one lane per row writes the geometry header (and the member headers of multi types);
//   bodies  : 8 lanes per ring / linestring write `count` + coordinates (16-byte unaligned stores).
// A record's bytes depend only on the row's own offsets, so every stage is a map; HBM-bound
// (16 B read + 16 B written per coordinate).
#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

constexpr int WKB_GS = 8;  // lanes per ring / linestring in the body kernel

__device__ __forceinline__ void put_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void put_xy(uint8_t* p, double2 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ void put_header(uint8_t* p, uint32_t type, uint32_t count) {  // byte order, type, count
    p[0] = 1;
    put_u32(p + 1, type);
    put_u32(p + 5, count);
}

__device__ __forceinline__ int64_t wkb_row_bytes(const DevGeo& a, int64_t g) {
    if (!dev::valid_row(a.validity, g)) return 0;
    switch (a.type) {
    case GPK_GEOM_POINT: return 21;
    case GPK_GEOM_LINESTRING: return 9 + 16 * (int64_t)(a.geom_off[g + 1] - a.geom_off[g]);
    case GPK_GEOM_MULTIPOINT: return 9 + 21 * (int64_t)(a.geom_off[g + 1] - a.geom_off[g]);
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING: {
        const int r0 = a.geom_off[g], r1 = a.geom_off[g + 1];
        const int64_t per = a.type == GPK_GEOM_POLYGON ? 4 : 9;
        return 9 + per * (r1 - r0) + 16 * (int64_t)(a.ring_off[r1] - a.ring_off[r0]);
    }
    default: {  // MULTIPOLYGON
        const int p0 = a.geom_off[g], p1 = a.geom_off[g + 1];
        const int r0 = a.part_off[p0], r1 = a.part_off[p1];
        return 9 + 9 * (int64_t)(p1 - p0) + 4 * (int64_t)(r1 - r0) + 16 * (int64_t)(a.ring_off[r1] - a.ring_off[r0]);
    }
    }
}

__global__ __launch_bounds__(256) void wkb_sizes_kernel(DevGeo a, int32_t* __restrict__ sizes) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const int64_t b = wkb_row_bytes(a, g);
    sizes[g] = (int32_t)(b < 0x7FFFFFFF ? b : 0x7FFFFFFF);  // the host has bounded the column below 2 GiB
}

// geometry headers; POINT rows are complete here.  For the ring types the lane also walks the row's rings / member lines:
// it writes their headers (count, or a full LineString header) and records where each one's coordinates start
// (seq_dst[r], -1 for a null row), so the body kernel needs no search for the owner of a ring.
__global__ __launch_bounds__(256) void wkb_headers_kernel(DevGeo a, const int32_t* __restrict__ off, uint8_t* __restrict__ out,
                                                           int32_t* __restrict__ seq_dst) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const bool valid = dev::valid_row(a.validity, g);
    const int32_t o = off[g];
    uint8_t* p = out + o;
    switch (a.type) {
    case GPK_GEOM_POINT:
        if (!valid) return;
        p[0] = 1;
        put_u32(p + 1, 1u);
        put_xy(p + 5, a.xy[g]);  // an empty point is NaN NaN in both encodings
        break;
    case GPK_GEOM_LINESTRING:
        seq_dst[g] = valid ? o + 9 : -1;
        if (valid) put_header(p, 2u, (uint32_t)(a.geom_off[g + 1] - a.geom_off[g]));
        break;
    case GPK_GEOM_MULTIPOINT:
        if (valid) put_header(p, 4u, (uint32_t)(a.geom_off[g + 1] - a.geom_off[g]));
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING: {
        const int r0 = a.geom_off[g], r1 = a.geom_off[g + 1];
        const bool poly = a.type == GPK_GEOM_POLYGON;
        if (valid) put_header(p, poly ? 3u : 5u, (uint32_t)(r1 - r0));
        int32_t at = o + 9;
        for (int r = r0; r < r1; ++r) {
            const int n = a.ring_off[r + 1] - a.ring_off[r];
            if (!valid) {
                seq_dst[r] = -1;
                continue;
            }
            if (poly)
                put_u32(out + at, (uint32_t)n);
            else
                put_header(out + at, 2u, (uint32_t)n);
            at += poly ? 4 : 9;
            seq_dst[r] = at;
            at += 16 * n;
        }
        break;
    }
    default: {
        const int p0 = a.geom_off[g], p1 = a.geom_off[g + 1];
        if (valid) put_header(p, 6u, (uint32_t)(p1 - p0));
        int32_t at = o + 9;
        for (int q = p0; q < p1; ++q) {
            const int r0 = a.part_off[q], r1 = a.part_off[q + 1];
            if (valid) {
                put_header(out + at, 3u, (uint32_t)(r1 - r0));
                at += 9;
            }
            for (int r = r0; r < r1; ++r) {
                const int n = a.ring_off[r + 1] - a.ring_off[r];
                if (!valid) {
                    seq_dst[r] = -1;
                    continue;
                }
                put_u32(out + at, (uint32_t)n);
                at += 4;
                seq_dst[r] = at;
                at += 16 * n;
            }
        }
    }
    }
}

// largest index i in [0, n) with off[i] <= v  (off ascending, off[0] <= v < off[n])
__device__ __forceinline__ int owner_of(const int32_t* __restrict__ off, int64_t n, int v) {
    int64_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= v)
            lo = mid;
        else
            hi = mid;
    }
    return (int)lo;
}

// bodies: WKB_GS lanes per sequence (a ring / member line, or the row of a LINESTRING column) copy its coordinates to
// out + seq_dst[q]; sequences longer than WKB_ENC_LONG are listed for wkb_bodies_long_kernel, which spreads each of them
// over the whole grid (a 100k-vertex ring would otherwise be 8 lanes' job)
constexpr int WKB_ENC_LONG = 4096;
__global__ __launch_bounds__(256) void wkb_bodies_kernel(DevGeo a, const int32_t* __restrict__ seq_dst, uint8_t* __restrict__ out,
                                                          int32_t* __restrict__ long_list) {
    const int lane = threadIdx.x & (WKB_GS - 1);
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WKB_GS;
    const bool rows = a.type == GPK_GEOM_LINESTRING;
    const int64_t n_seq = rows ? a.n_geoms : a.n_rings;
    if (q >= n_seq) return;
    const int32_t dst = seq_dst[q];
    if (dst < 0) return;
    const int32_t* so = rows ? a.geom_off : a.ring_off;
    const int c0 = so[q], c1 = so[q + 1];
    if (c1 - c0 > WKB_ENC_LONG) {
        if (lane == 0) long_list[1 + atomicAdd(&long_list[0], 1)] = (int32_t)q;
        return;
    }
    uint8_t* p = out + dst;
    for (int i = c0 + lane; i < c1; i += WKB_GS) put_xy(p + 16 * (int64_t)(i - c0), a.xy[i]);
}
__global__ __launch_bounds__(256) void wkb_bodies_long_kernel(DevGeo a, const int32_t* __restrict__ seq_dst, uint8_t* __restrict__ out,
                                                               const int32_t* __restrict__ long_list) {
    const int n_long = long_list[0];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const int32_t* so = a.type == GPK_GEOM_LINESTRING ? a.geom_off : a.ring_off;
    for (int k = 0; k < n_long; ++k) {
        const int q = long_list[1 + k];
        const int c0 = so[q], c1 = so[q + 1];
        uint8_t* p = out + seq_dst[q];
        for (int64_t i = c0 + tid; i < c1; i += stride) put_xy(p + 16 * (i - c0), a.xy[i]);
    }
}

// The same bodies cut by INPUT coordinate (the decoder's wkb_copy_kernel in reverse): a work-group owns WKB_EB_TILE consecutive
// coordinates of the column whatever sequences they belong to — a power-law column keeps half of its coordinates in rings of at most
// 16 and a few in rings of 100k, and eight lanes per ring serve neither.  The offsets of the sequences that meet the tile and their
// destinations are staged in LDS; a lane finds the sequence of each of its coordinates there (a null row's sequences have no
// destination: their coordinates are skipped).
constexpr int WKB_EB_BLOCK = 256, WKB_EB_PER = 8, WKB_EB_TILE = WKB_EB_BLOCK * WKB_EB_PER, WKB_EB_SEQS = 2560;
__global__ __launch_bounds__(WKB_EB_BLOCK) void wkb_bodies_flat_kernel(DevGeo a, const int32_t* __restrict__ seq_dst, uint8_t* __restrict__ out) {
    __shared__ int32_t s_off[WKB_EB_SEQS + 1], s_dst[WKB_EB_SEQS];
    __shared__ int32_t s_q[2];
    const bool rows = a.type == GPK_GEOM_LINESTRING;
    const int64_t n_seq = rows ? a.n_geoms : a.n_rings;
    const int32_t* __restrict__ so = rows ? a.geom_off : a.ring_off;
    const int64_t c_lo = (int64_t)blockIdx.x * WKB_EB_TILE, c_hi = c_lo + WKB_EB_TILE < a.n_coords ? c_lo + WKB_EB_TILE : a.n_coords;
    auto seq_of = [&](int64_t c) {  // last sequence that starts at or before coordinate c
        int64_t lo = 0, hi = n_seq;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)so[mid] <= c)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    };
    if (threadIdx.x < 2) s_q[threadIdx.x] = (int32_t)seq_of(threadIdx.x == 0 ? c_lo : c_hi - 1);
    __syncthreads();
    const int q0 = s_q[0], q1 = s_q[1], nq = q1 - q0 + 1;
    const bool staged = nq <= WKB_EB_SEQS;
    if (staged) {
        for (int t = threadIdx.x; t <= nq; t += WKB_EB_BLOCK) s_off[t] = so[q0 + t];
        for (int t = threadIdx.x; t < nq; t += WKB_EB_BLOCK) s_dst[t] = seq_dst[q0 + t];
    }
    __syncthreads();
    double2 v[WKB_EB_PER];
#pragma unroll
    for (int j = 0; j < WKB_EB_PER; ++j) {
        const int64_t c = c_lo + j * WKB_EB_BLOCK + threadIdx.x;
        v[j] = c < c_hi ? a.xy[c] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int j = 0; j < WKB_EB_PER; ++j) {
        const int64_t c = c_lo + j * WKB_EB_BLOCK + threadIdx.x;
        if (c >= c_hi) continue;
        int64_t dst, first;
        if (staged) {
            int lo = 0, hi = nq;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((int64_t)s_off[mid] <= c)
                    lo = mid;
                else
                    hi = mid;
            }
            dst = s_dst[lo];
            first = s_off[lo];
        } else {
            const int64_t q = seq_of(c);
            dst = seq_dst[q];
            first = so[q];
        }
        if (dst >= 0) put_xy(out + dst + 16 * (c - first), v[j]);
    }
}

// MULTIPOINT members: one lane per coordinate, 21 bytes each
__global__ __launch_bounds__(256) void wkb_multipoint_kernel(DevGeo a, const int32_t* __restrict__ off, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_coords) return;
    const int g = owner_of(a.geom_off, a.n_geoms, (int)i);
    if (!dev::valid_row(a.validity, g)) return;
    uint8_t* p = out + off[g] + 9 + 21 * (int64_t)(i - a.geom_off[g]);
    p[0] = 1;
    put_u32(p + 1, 1u);
    put_xy(p + 5, a.xy[i]);
}

}  // namespace gpk

using namespace gpk;

extern "C" int32_t gpk_geoarray_to_wkb(const gpk_geoarray* a, int32_t* out_offsets, uint8_t* out_values, int64_t capacity,
                                       int64_t* n_bytes, int32_t out_space, void* stream) {
    if (!a || !n_bytes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (capacity < 0 || (capacity > 0 && !out_values)) return fail(GPK_ERR_INVALID_ARGUMENT, "capacity without out_values");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const DevGeo& d = a->d;
    const int64_t n = d.n_geoms;
    *n_bytes = 0;
    // upper bound of the column (every row valid): it must fit Arrow's i32 offsets
    const bool pointish = d.type == GPK_GEOM_POINT || d.type == GPK_GEOM_MULTIPOINT;
    const int64_t bound = (pointish ? 21 : 16) * d.n_coords + 9 * (n + d.n_parts + d.n_rings);
    if (bound > 0x7FFFFFFFLL)
        return fail(GPK_ERR_CAPACITY, "to_wkb: up to %lld bytes do not fit BinaryArray<i32> offsets; encode row slices", (long long)bound);
    const bool host_out = out_space != GPK_MEM_DEVICE;
    const int64_t nb = (n + 255) / 256;
    const int64_t n_seq_enc = d.type == GPK_GEOM_LINESTRING ? n : d.n_rings;
    size_t need = 2 * align256(sizeof(int32_t) * (size_t)(n + 1)) + align256(sizeof(unsigned long long) * (size_t)(nb + 2)) +
                  2 * align256(sizeof(int32_t) * (size_t)(n_seq_enc + 2)) + 1024;
    if (host_out && out_values) need += align256((size_t)capacity);
    GPK_TRY(workspace().begin(need));
    int32_t* sizes = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* off_dev = (host_out || !out_offsets) ? (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n + 1)) : out_offsets;
    unsigned long long* btot = (unsigned long long*)workspace().take(sizeof(unsigned long long) * (size_t)(nb + 2));
    uint8_t* val_dev = out_values ? (host_out ? (uint8_t*)workspace().take((size_t)capacity) : out_values) : nullptr;
    int32_t* long_list = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_seq_enc + 2));
    int32_t* seq_dst = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_seq_enc + 2));

    int32_t total = 0;
    if (n > 0) {
        GPK_LAUNCH("gpk_wkb_sizes", wkb_sizes_kernel, dim3((unsigned)nb), dim3(256), 0, s, d, sizes);
        GPK_TRY(exclusive_scan_i32(sizes, n, off_dev, nullptr, btot, s));
        GPK_HIP(hipMemcpyAsync(&total, off_dev + n, sizeof total, hipMemcpyDeviceToHost, s));
        GPK_HIP(hipStreamSynchronize(s));
    } else {
        GPK_HIP(hipMemsetAsync(off_dev, 0, sizeof(int32_t), s));
    }
    *n_bytes = (int64_t)total;
    if (out_offsets && host_out) GPK_TRY(copy_out(out_offsets, out_space, off_dev, sizeof(int32_t) * (size_t)(n + 1), s));
    if (!out_values) return GPK_OK;  // size query
    if ((int64_t)total > capacity)
        return fail(GPK_ERR_CAPACITY, "to_wkb: %lld bytes but capacity %lld", (long long)total, (long long)capacity);
    if (n > 0 && total > 0) {
        GPK_LAUNCH("gpk_wkb_headers", wkb_headers_kernel, dim3((unsigned)nb), dim3(256), 0, s, d, (const int32_t*)off_dev, val_dev, seq_dst);
        if (d.type == GPK_GEOM_MULTIPOINT) {
            if (d.n_coords > 0)
                GPK_LAUNCH("gpk_wkb_multipoint", wkb_multipoint_kernel, dim3((unsigned)((d.n_coords + 255) / 256)), dim3(256), 0, s, d,
                           (const int32_t*)off_dev, val_dev);
        } else if (d.type != GPK_GEOM_POINT) {
            const int64_t n_seq = n_seq_enc;
            // the bodies are cut by coordinate (2M x 64-vertex polygons 1.41 -> 0.98 ms, 1M power-law multipolygons 0.67 -> 0.42 ms, 8M x 8
            // 1.25 -> 1.28 ms); GPK_WKB_ENC_GROUPS=1: the round-3 form, eight lanes per sequence (A/B runs)
            static const bool force_groups = getenv("GPK_WKB_ENC_GROUPS") != nullptr;
            const bool flat = !force_groups && n_seq > 0 && d.n_coords > 0;
            if (flat) {
                GPK_LAUNCH("gpk_wkb_bodies", wkb_bodies_flat_kernel, dim3((unsigned)((d.n_coords + WKB_EB_TILE - 1) / WKB_EB_TILE)), dim3(WKB_EB_BLOCK), 0, s, d,
                           (const int32_t*)seq_dst, val_dev);
            } else if (n_seq > 0) {
                GPK_HIP(hipMemsetAsync(long_list, 0, sizeof(int32_t), s));
                GPK_LAUNCH("gpk_wkb_bodies", wkb_bodies_kernel, dim3((unsigned)((n_seq * WKB_GS + 255) / 256)), dim3(256), 0, s, d,
                           (const int32_t*)seq_dst, val_dev, long_list);
                GPK_LAUNCH("gpk_wkb_bodies_long", wkb_bodies_long_kernel, dim3((unsigned)(cu_count() * 8)), dim3(256), 0, s, d,
                           (const int32_t*)seq_dst, val_dev, (const int32_t*)long_list);
            }
        }
    }
    if (host_out) GPK_TRY(copy_out(out_values, out_space, val_dev, (size_t)total, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}
