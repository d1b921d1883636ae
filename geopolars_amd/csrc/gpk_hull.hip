// gpk_hull.hip — convex_hull (geoseries.rs:23-26,196-198; geo 0.27 convex_hull/qhull.rs semantics:
// closed counter-clockwise exterior, collinear vertices dropped).
//
// The hull of a point set is unique, so any exact algorithm returns the same ring up to its start
// vertex; this kernel emits it starting at the lexicographically smallest vertex.  One lane per
// geometry for the chain, 16 lanes per geometry for the sort (bitonic network in LDS; a work-group per geometry beyond 128
// points), then a monotone chain driven by the exact
// orientation kernel.  Irregular per-row output sizes go through size -> scan -> compact.
// This is the lowest-traffic operator of the surface; the sort's working set sits in LDS.
#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

__device__ __forceinline__ bool xy_less(double2 a, double2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

// monotone chain over n SORTED points read through ld(0..n) (duplicates allowed: they are skipped on the fly, in both
// directions).  The stack holds point INDICES, behind put(k, i) / get(k) (the lane's column of an LDS tile, or global
// scratch for geometries too large for it): a stack of coordinates in global memory turned every push into a partial
// cache-line write from half a million concurrent streams.  The two topmost entries also live in registers as
// coordinates, so a step reads one point, and one more per pop.  The closed hull is written to h; returns its size.
template <typename LD, typename PUT, typename GET>
__device__ __forceinline__ int chain_of(LD ld, int n, PUT put, GET get, double2* __restrict__ h) {
    const double2 p0 = ld(0), pl = ld(n - 1);
    int k = 0, m = 0;          // stack depth, distinct points seen
    double2 t1 = p0, t2 = p0;  // points of stack entries k-1, k-2
    auto push = [&](int i, double2 v) {
        put(k++, i);
        t2 = t1;
        t1 = v;
    };
    auto pop = [&]() {
        --k;
        t1 = t2;
        if (k >= 2) t2 = ld(get(k - 2));
    };
    double2 prev = make_double2(0, 0);
    double2 nxt = p0;  // the next point is loaded one step ahead of its use
    for (int i = 0; i < n; ++i) {
        const double2 v = nxt;
        if (i + 1 < n) nxt = ld(i + 1);
        if (i > 0 && v.x == prev.x && v.y == prev.y) continue;
        prev = v;
        ++m;
        while (k >= 2 && dev::orient2d(t2.x, t2.y, t1.x, t1.y, v.x, v.y) <= 0) pop();
        push(i, v);
    }
    if (m < 3) {  // one or two distinct points
        h[0] = p0;
        if (m == 2) h[1] = pl;
        h[m] = p0;
        return m + 1;
    }
    const int lo = k + 1;
    prev = pl;  // the last distinct point is already on the stack: the way back starts below it
    nxt = n >= 2 ? ld(n - 2) : p0;
    for (int i = n - 2; i >= 0; --i) {
        const double2 v = nxt;
        if (i > 0) nxt = ld(i - 1);
        if (v.x == prev.x && v.y == prev.y) continue;
        prev = v;
        while (k >= lo && dev::orient2d(t2.x, t2.y, t1.x, t1.y, v.x, v.y) <= 0) pop();
        push(i, v);
    }
    --k;
    if (k < 3) {  // all collinear: the two extremes
        h[0] = p0;
        h[1] = pl;
        h[2] = p0;
        return 3;
    }
    for (int j = 0; j < k; ++j) h[j] = ld(get(j));
    h[k] = h[0];  // close the ring
    return k + 1;
}

__device__ __forceinline__ void geom_coord_range(const DevGeo& a, int64_t g, int& c0, int& c1) {
    switch (a.type) {
    case GPK_GEOM_POINT: {
        const double2 p = a.xy[g];
        c0 = (int)g;
        c1 = (isnan(p.x) || isnan(p.y)) ? (int)g : (int)g + 1;
        break;
    }
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

// scratch layout per geometry g with coordinate range [c0, c1): sorted copy at sorted[c0..c0 + n_pts[g]), chain stack at
// stack[2*c0 + 2*g .. 2*c1 + 2*g + 2).  Three stages:
//   hull_sort_kernel      HULL_GS lanes per geometry of at most HULL_CAP points: a bitonic network over the geometry's
//                         points in LDS (coalesced loads and stores, no data-dependent addressing, 16 geometries per
//                         work-group) — the lane-per-geometry heap sort made ~800 scattered accesses per geometry;
//   hull_sort_big_kernel  one work-group per geometry beyond HULL_CAP: bitonic network in LDS / in the global scratch;
//   hull_chain_kernel     one lane per geometry: monotone chain over its sorted points (sequential reads).
// The closing duplicate of a ring is dropped before sorting (the chain skips duplicates anyway).
constexpr int HULL_CAP = 128, HULL_GS = 16;
__device__ __forceinline__ int hull_points(const DevGeo& a, int64_t g, int& c0) {  // -1: null / empty row
    int c1;
    geom_coord_range(a, g, c0, c1);
    if (!dev::valid_row(a.validity, g) || c1 == c0) return -1;
    int n = c1 - c0;
    if (n >= 2) {
        const double2 f = a.xy[c0], l = a.xy[c1 - 1];
        if (f.x == l.x && f.y == l.y) --n;
    }
    return n;
}
__global__ __launch_bounds__(256) void hull_sort_kernel(DevGeo a, double2* __restrict__ sorted, int32_t* __restrict__ n_pts,
                                                        int32_t* __restrict__ big_list, int32_t* __restrict__ big_count) {
    __shared__ double2 lds[(256 / HULL_GS) * HULL_CAP];
    const int lane = threadIdx.x & (HULL_GS - 1);
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / HULL_GS;
    if (g >= a.n_geoms) return;
    int c0;
    const int n = hull_points(a, g, c0);
    if (lane == 0) {
        n_pts[g] = n;
        if (n > HULL_CAP) big_list[atomicAdd(big_count, 1)] = (int32_t)g;  // order is irrelevant: one work-group each
    }
    if (n <= 0 || n > HULL_CAP) return;  // empty, or left to hull_sort_big_kernel
    double2* __restrict__ v = lds + (threadIdx.x / HULL_GS) * HULL_CAP;
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = lane; i < P; i += HULL_GS) v[i] = i < n ? a.xy[c0 + i] : make_double2(INFINITY, INFINITY);  // sentinels sort last
    auto sync = [] {  // the lanes of a group sit in one wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            sync();
            for (int t = lane; t < P / 2; t += HULL_GS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;  // 2j * (t / j) + t % j for a power of two j
                const bool up = (i & k) == 0;
                const double2 x = v[i], y = v[l];
                if (xy_less(y, x) == up) {
                    v[i] = y;
                    v[l] = x;
                }
            }
        }
    sync();
    for (int i = lane; i < n; i += HULL_GS) sorted[c0 + i] = v[i];
}
// One work-group per geometry beyond HULL_CAP points (their ids were appended to big_list by hull_sort_kernel): the
// normalised bitonic network — every comparator ascending, so a partner index beyond n is simply skipped, which is the
// same as padding with +inf — over the geometry's points in LDS (up to HULL_BIG_LDS of them) or in place in the global
// scratch.  A 100k-point ring takes 153 steps of 256 lanes instead of 3.4M dependent accesses of one lane.
constexpr int HULL_BIG_LDS = 4096;
__global__ __launch_bounds__(256) void hull_sort_big_kernel(DevGeo a, double2* __restrict__ sorted, const int32_t* __restrict__ n_pts,
                                                            const int32_t* __restrict__ big_list, const int32_t* __restrict__ big_count) {
    __shared__ double2 lds[HULL_BIG_LDS];
    const int n_big = *big_count;
    for (int b = blockIdx.x; b < n_big; b += gridDim.x) {
        const int64_t g = big_list[b];
        const int n = n_pts[g];
        int c0, c1;
        geom_coord_range(a, g, c0, c1);
        const bool in_lds = n <= HULL_BIG_LDS;
        double2* __restrict__ v = in_lds ? lds : sorted + c0;
        for (int i = threadIdx.x; i < n; i += 256) v[i] = a.xy[c0 + i];
        int P = 2;
        while (P < n) P <<= 1;
        auto step = [&](int mask) {  // partner of i: i ^ mask; the lower index of a pair owns the exchange
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += 256) {
                const int l = i ^ mask;
                if (l > i && l < n) {
                    const double2 x = v[i], y = v[l];
                    if (xy_less(y, x)) {
                        v[i] = y;
                        v[l] = x;
                    }
                }
            }
        };
        for (int k = 2; k <= P; k <<= 1) {
            step(k - 1);  // flip: i <-> its mirror inside the block of k
            for (int j = k >> 2; j > 0; j >>= 1) step(j);
        }
        __syncthreads();
        if (in_lds)
            for (int i = threadIdx.x; i < n; i += 256) sorted[c0 + i] = v[i];
        __syncthreads();  // the LDS tile is reused by the next geometry of this work-group
    }
}
// one wave per work-group: when every geometry of the wave has at most HULL_STACK points the index stacks sit in an LDS
// tile, interleaved by lane (entry k of lane l at [k * 64 + l]); otherwise the wave keeps them in global scratch
constexpr int HULL_STACK = 129;  // stack depth: up to n + 1 entries for n points
__global__ __launch_bounds__(64) void hull_chain_kernel(DevGeo a, const double2* __restrict__ sorted, const int32_t* __restrict__ n_pts,
                                                         double2* __restrict__ stack, int32_t* __restrict__ idx_scratch,
                                                         int32_t* __restrict__ sizes) {
    __shared__ uint16_t s_idx[(HULL_STACK + 1) * 64];
    const int lane = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * 64 + lane;
    const int n = g < a.n_geoms ? n_pts[g] : 0;
    const bool all_fit = __all(n <= HULL_STACK - 1);  // wave-uniform
    if (n <= 0) {
        if (g < a.n_geoms) sizes[g] = 0;
        return;
    }
    int c0, c1;
    geom_coord_range(a, g, c0, c1);
    const double2* __restrict__ p = sorted + c0;
    double2* __restrict__ h = stack + 2 * (int64_t)c0 + 2 * g;
    auto ld = [&](int i) { return p[i]; };
    if (all_fit) {
        sizes[g] = chain_of(ld, n, [&](int k, int i) { s_idx[k * 64 + lane] = (uint16_t)i; }, [&](int k) { return (int)s_idx[k * 64 + lane]; }, h);
    } else {
        int32_t* __restrict__ st = idx_scratch + 2 * (int64_t)c0 + 2 * g;
        sizes[g] = chain_of(ld, n, [&](int k, int i) { st[k] = i; }, [&](int k) { return (int)st[k]; }, h);
    }
}

__global__ void hull_compact_kernel(DevGeo a, const double2* __restrict__ stack, const int32_t* __restrict__ off,
                                    double2* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    int c0, c1;
    geom_coord_range(a, g, c0, c1);
    const double2* h = stack + 2 * (int64_t)c0 + 2 * g;
    const int o = off[g], n = off[g + 1] - o;
    for (int i = 0; i < n; ++i) out[o + i] = h[i];
}

}  // namespace gpk

using namespace gpk;

extern "C" int32_t gpk_convex_hull(const gpk_geoarray* a, double* out_xy, int32_t* out_ring_offsets, int32_t out_space,
                                   void* stream) {
    if (!a || !out_xy || !out_ring_offsets) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms, nc = a->d.n_coords;
    const size_t cap_coords = (size_t)(nc + n);
    const size_t off_bytes = sizeof(int32_t) * (size_t)(n + 1);
    const int64_t nb = (n + 255) / 256;
    const bool host_out = out_space != GPK_MEM_DEVICE;
    size_t need = align256(sizeof(double2) * (size_t)(nc + 1)) + align256(sizeof(double2) * (2 * (size_t)nc + 2 * (size_t)n + 2)) +
                  2 * align256(off_bytes) + align256(sizeof(unsigned long long) * (size_t)(nb + 2)) +
                  align256(sizeof(int32_t) * (2 * (size_t)nc + 2 * (size_t)n + 2)) + align256(off_bytes) + 1024;
    if (host_out) need += align256(sizeof(double2) * cap_coords) + align256(off_bytes);
    GPK_TRY(workspace().begin(need));
    double2* sorted = (double2*)workspace().take(sizeof(double2) * (size_t)(nc + 1));
    double2* stack = (double2*)workspace().take(sizeof(double2) * (2 * (size_t)nc + 2 * (size_t)n + 2));
    int32_t* sizes = (int32_t*)workspace().take(off_bytes);
    int32_t* n_pts = (int32_t*)workspace().take(off_bytes);
    int32_t* idx_scratch = (int32_t*)workspace().take(sizeof(int32_t) * (2 * (size_t)nc + 2 * (size_t)n + 2));
    unsigned long long* btot = (unsigned long long*)workspace().take(sizeof(unsigned long long) * (size_t)(nb + 2));
    int32_t* big_list = (int32_t*)workspace().take(off_bytes);  // [0, n): ids of the geometries beyond HULL_CAP points, [n]: their number
    double2* out_dev = host_out ? (double2*)workspace().take(sizeof(double2) * cap_coords) : (double2*)out_xy;
    int32_t* off_dev = host_out ? (int32_t*)workspace().take(off_bytes) : out_ring_offsets;
    if (n == 0) {
        GPK_HIP(hipMemsetAsync(off_dev, 0, sizeof(int32_t), s));
        return copy_out(out_ring_offsets, out_space, off_dev, sizeof(int32_t), s);
    }
    const dim3 grid((unsigned)nb), block(256);
    GPK_HIP(hipMemsetAsync(big_list + n, 0, sizeof(int32_t), s));
    GPK_LAUNCH("gpk_hull_sort", hull_sort_kernel, dim3((unsigned)((n * HULL_GS + 255) / 256)), dim3(256), 0, s, a->d, sorted, n_pts, big_list,
               big_list + n);
    {
        const int64_t big_blocks = n < (int64_t)cu_count() * 8 ? n : (int64_t)cu_count() * 8;
        GPK_LAUNCH("gpk_hull_sort_big", hull_sort_big_kernel, dim3((unsigned)big_blocks), block, 0, s, a->d, sorted, (const int32_t*)n_pts,
                   (const int32_t*)big_list, (const int32_t*)(big_list + n));
    }
    GPK_LAUNCH("gpk_hull_chain", hull_chain_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a->d, (const double2*)sorted, (const int32_t*)n_pts, stack,
               idx_scratch, sizes);
    GPK_TRY(exclusive_scan_i32(sizes, n, off_dev, nullptr, btot, s));
    GPK_LAUNCH("gpk_hull_compact", hull_compact_kernel, grid, block, 0, s, a->d, stack, off_dev, out_dev);
    if (host_out) {
        GPK_TRY(copy_out(out_ring_offsets, out_space, off_dev, off_bytes, s));
        const int32_t total = out_ring_offsets[n];
        GPK_TRY(copy_out(out_xy, out_space, out_dev, sizeof(double2) * (size_t)total, s));
    }
    return GPK_OK;
}
