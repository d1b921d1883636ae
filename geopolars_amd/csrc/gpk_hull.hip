// gpk_hull.hip — convex_hull (geoseries.rs:23-26,196-198; geo 0.27 convex_hull/qhull.rs semantics:
// closed counter-clockwise exterior, collinear vertices dropped).
//
// The hull of a point set is unique, so any exact algorithm returns the same ring up to its start
// vertex; this kernel emits it starting at the lexicographically smallest vertex.  Geometries of up to 128 points — the
// common case — are solved whole by 16 cooperating lanes in LDS (bitonic network, then parallel left-turn filtering of
// the candidate cycle to its fixed point: hull_small_kernel); larger ones get a work-group each for the sort and one lane
// for a monotone chain.  Every orientation is the exact kernel's.  Irregular per-row output sizes go through
// size -> scan -> compact.
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
// hull_small_kernel: the whole hull of a geometry of at most HULL_CAP points by HULL_GS cooperating lanes, in LDS:
//   1. bitonic network over the points (coalesced loads, no data-dependent addressing);
//   2. the candidate cycle L, points strictly below the chord L -> R ascending, R, points strictly above it descending
//      (L / R = lexicographic extremes; duplicates and points on the chord never enter);
//   3. rounds of parallel filtering: every candidate whose triple (predecessor, itself, successor) in the current cycle is
//      not a strict left turn leaves, the survivors are compacted in order (group ballot + prefix popcount), until a round
//      removes nothing.  A removed point lies on or above (below) a chord of points of its own half, so it is no hull
//      vertex whatever else leaves in the same round; a cycle of strict left turns through L and R with every other point
//      proven inside is THE hull — the ring the monotone chain (chain_of) writes, vertex for vertex, because both start at
//      the smallest point, run counter-clockwise and drop collinear points.  Orientations are exact (dev::orient2d).
// The hull goes to the geometry's slice of `stack` (coalesced), its size to sizes[g]; rows beyond HULL_CAP points are
// listed for hull_sort_big_kernel + hull_chain_big_kernel.  The lane-per-geometry chain this replaces walked 2 x 64 points
// with dependent LDS stack accesses while 63 of 64 memory lanes idled: 4.8 of the 7.4 ms of 2M x 64-vertex polygons.
// CAP = 64 first (half the LDS per work-group: eight work-groups per CU instead of four — the kernel is bound by the
// latency of its barrier-free but strictly sequential LDS steps, so occupancy is what it buys), rows of 65 .. 128 points are
// listed and taken by the CAP = 128 instantiation (LISTED: the work-groups stride over the list).
template <int CAP, bool LISTED>
__global__ __launch_bounds__(256) void hull_small_kernel(DevGeo a, int32_t* __restrict__ n_pts, const int32_t* __restrict__ in_list,
                                                         const int32_t* __restrict__ in_count, int32_t* __restrict__ mid_list,
                                                         int32_t* __restrict__ mid_count, int32_t* __restrict__ big_list,
                                                         int32_t* __restrict__ big_count, double2* __restrict__ stack,
                                                         int32_t* __restrict__ sizes) {
    __shared__ double2 lds[(256 / HULL_GS) * CAP];
    __shared__ uint8_t s_idx[256 / HULL_GS][2][CAP + 8];
    const int lane = threadIdx.x & (HULL_GS - 1), grp = threadIdx.x / HULL_GS;
    const int gshift = (threadIdx.x & 63) & ~(HULL_GS - 1);  // bit position of this group's lanes in a wave ballot
    const int64_t n_items = LISTED ? (int64_t)*in_count : a.n_geoms;
    for (int64_t item = (int64_t)blockIdx.x * (256 / HULL_GS) + grp; item < n_items; item += (int64_t)gridDim.x * (256 / HULL_GS)) {
    const int64_t g = LISTED ? (int64_t)in_list[item] : item;
    int c0;
    const int n = hull_points(a, g, c0);
    if (!LISTED && lane == 0) {
        n_pts[g] = n;
        if (n > HULL_CAP) big_list[atomicAdd(big_count, 1)] = (int32_t)g;  // order is irrelevant: one work-group each
        else if (n > CAP) mid_list[atomicAdd(mid_count, 1)] = (int32_t)g;
        if (n <= 0) sizes[g] = 0;
    }
    if (n <= 0 || n > CAP) continue;
    double2* __restrict__ v = lds + grp * CAP;
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
    double2* __restrict__ h = stack + 2 * (int64_t)c0 + 2 * g;
    const double2 L = v[0], R = v[n - 1];
    if (L.x == R.x && L.y == R.y) {  // one distinct point
        if (lane == 0) {
            h[0] = L;
            h[1] = L;
            sizes[g] = 2;
        }
        continue;
    }
    auto gballot = [&](bool c) -> uint32_t { return (uint32_t)(__ballot(c) >> gshift) & ((1u << HULL_GS) - 1u); };
    const uint32_t below = (1u << lane) - 1u;
    uint8_t* cur = s_idx[grp][0];
    uint8_t* nxt = s_idx[grp][1];
    // candidate cycle: interior points i = 1 .. n-2 in chunks of HULL_GS, lane = point; side[] keeps the chord test
    constexpr int CH = (CAP + HULL_GS - 1) / HULL_GS;
    int side[CH];
    int cnt = 1;
    if (lane == 0) cur[0] = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = 1 + c * HULL_GS + lane;
        side[c] = 0;
        if (i < n - 1) {
            const double2 p = v[i], q = v[i - 1];
            if (!(p.x == q.x && p.y == q.y)) side[c] = dev::orient2d(L.x, L.y, R.x, R.y, p.x, p.y);
        }
        const uint32_t m = gballot(side[c] < 0);
        if (side[c] < 0) cur[cnt + __popc(m & below)] = (uint8_t)i;
        cnt += __popc(m);
    }
    if (lane == 0) cur[cnt] = (uint8_t)(n - 1);
    ++cnt;
#pragma unroll
    for (int c = CH - 1; c >= 0; --c) {
        const int i = 1 + c * HULL_GS + lane;
        const uint32_t m = gballot(side[c] > 0);
        if (side[c] > 0) cur[cnt + __popc(m >> (lane + 1))] = (uint8_t)i;  // descending within the chunk
        cnt += __popc(m);
    }
    for (;;) {
        sync();
        int kept = 0;
        for (int b = 0; b < cnt; b += HULL_GS) {
            const int t = b + lane;
            bool keep = false;
            int i = 0;
            if (t < cnt) {
                i = cur[t];
                if (i == 0 || i == n - 1) {
                    keep = true;
                } else {
                    const double2 pa = v[cur[t - 1]], pb = v[i], pc = v[cur[t + 1 == cnt ? 0 : t + 1]];
                    keep = dev::orient2d(pa.x, pa.y, pb.x, pb.y, pc.x, pc.y) > 0;
                }
            }
            const uint32_t m = gballot(keep);
            if (keep) nxt[kept + __popc(m & below)] = (uint8_t)i;
            kept += __popc(m);
        }
        uint8_t* t = cur;
        cur = nxt;
        nxt = t;
        const bool done = kept == cnt;  // uniform within the group
        cnt = kept;
        if (done) break;
    }
    sync();
    for (int j = lane; j < cnt; j += HULL_GS) h[j] = v[cur[j]];
    if (lane == 0) {
        h[cnt] = L;  // close the ring
        sizes[g] = cnt + 1;
    }
    sync();  // the group's LDS slices are rewritten by its next item
    }
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
// geometries beyond HULL_CAP points (sorted by hull_sort_big_kernel): one lane per listed geometry runs the monotone
// chain with its index stack in the global scratch
__global__ __launch_bounds__(64) void hull_chain_big_kernel(DevGeo a, const double2* __restrict__ sorted, const int32_t* __restrict__ n_pts,
                                                             const int32_t* __restrict__ big_list, const int32_t* __restrict__ big_count,
                                                             double2* __restrict__ stack, int32_t* __restrict__ idx_scratch,
                                                             int32_t* __restrict__ sizes) {
    const int n_big = *big_count;
    for (int b = blockIdx.x * 64 + threadIdx.x; b < n_big; b += gridDim.x * 64) {
        const int64_t g = big_list[b];
        const int n = n_pts[g];
        int c0, c1;
        geom_coord_range(a, g, c0, c1);
        const double2* __restrict__ p = sorted + c0;
        double2* __restrict__ h = stack + 2 * (int64_t)c0 + 2 * g;
        int32_t* __restrict__ st = idx_scratch + 2 * (int64_t)c0 + 2 * g;
        sizes[g] = chain_of([&](int i) { return p[i]; }, n, [&](int k, int i) { st[k] = i; }, [&](int k) { return (int)st[k]; }, h);
    }
}

// HULL_GS lanes per geometry copy its hull from the scratch slice to its place in the output (coalesced both ways)
__global__ __launch_bounds__(256) void hull_compact_kernel(DevGeo a, const double2* __restrict__ stack, const int32_t* __restrict__ off,
                                                           double2* __restrict__ out) {
    const int lane = threadIdx.x & (HULL_GS - 1);
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / HULL_GS;
    if (g >= a.n_geoms) return;
    int c0, c1;
    geom_coord_range(a, g, c0, c1);
    const double2* __restrict__ h = stack + 2 * (int64_t)c0 + 2 * g;
    const int o = off[g], n = off[g + 1] - o;
    for (int i = lane; i < n; i += HULL_GS) out[o + i] = h[i];
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
                  align256(sizeof(int32_t) * (2 * (size_t)nc + 2 * (size_t)n + 2)) + 2 * align256(off_bytes) + 1024;
    if (host_out) need += align256(sizeof(double2) * cap_coords) + align256(off_bytes);
    GPK_TRY(workspace().begin(need));
    double2* sorted = (double2*)workspace().take(sizeof(double2) * (size_t)(nc + 1));
    double2* stack = (double2*)workspace().take(sizeof(double2) * (2 * (size_t)nc + 2 * (size_t)n + 2));
    int32_t* sizes = (int32_t*)workspace().take(off_bytes);
    int32_t* n_pts = (int32_t*)workspace().take(off_bytes);
    int32_t* idx_scratch = (int32_t*)workspace().take(sizeof(int32_t) * (2 * (size_t)nc + 2 * (size_t)n + 2));
    unsigned long long* btot = (unsigned long long*)workspace().take(sizeof(unsigned long long) * (size_t)(nb + 2));
    int32_t* big_list = (int32_t*)workspace().take(off_bytes);  // [0, n): ids of the geometries beyond HULL_CAP points, [n]: their number
    int32_t* mid_list = (int32_t*)workspace().take(off_bytes);  // the same for 65 .. HULL_CAP points
    double2* out_dev = host_out ? (double2*)workspace().take(sizeof(double2) * cap_coords) : (double2*)out_xy;
    int32_t* off_dev = host_out ? (int32_t*)workspace().take(off_bytes) : out_ring_offsets;
    if (n == 0) {
        GPK_HIP(hipMemsetAsync(off_dev, 0, sizeof(int32_t), s));
        return copy_out(out_ring_offsets, out_space, off_dev, sizeof(int32_t), s);
    }
    const dim3 grid((unsigned)nb), block(256);
    GPK_HIP(hipMemsetAsync(big_list + n, 0, sizeof(int32_t), s));
    GPK_HIP(hipMemsetAsync(mid_list + n, 0, sizeof(int32_t), s));
    const dim3 ggrid((unsigned)((n * HULL_GS + 255) / 256));
    GPK_LAUNCH("gpk_hull_small", (hull_small_kernel<64, false>), ggrid, block, 0, s, a->d, n_pts, (const int32_t*)nullptr, (const int32_t*)nullptr, mid_list,
               mid_list + n, big_list, big_list + n, stack, sizes);
    {  // rows of 65 .. 128 points (listed by the launch above)
        const int64_t mid_blocks = (n + 15) / 16 < (int64_t)cu_count() * 8 ? (n + 15) / 16 : (int64_t)cu_count() * 8;
        GPK_LAUNCH("gpk_hull_small_mid", (hull_small_kernel<HULL_CAP, true>), dim3((unsigned)mid_blocks), block, 0, s, a->d, n_pts, (const int32_t*)mid_list,
                   (const int32_t*)(mid_list + n), (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, stack, sizes);
    }
    {
        const int64_t big_blocks = n < (int64_t)cu_count() * 8 ? n : (int64_t)cu_count() * 8;
        GPK_LAUNCH("gpk_hull_sort_big", hull_sort_big_kernel, dim3((unsigned)big_blocks), block, 0, s, a->d, sorted, (const int32_t*)n_pts,
                   (const int32_t*)big_list, (const int32_t*)(big_list + n));
        const int64_t chain_blocks = (n + 63) / 64 < (int64_t)cu_count() * 16 ? (n + 63) / 64 : (int64_t)cu_count() * 16;
        GPK_LAUNCH("gpk_hull_chain_big", hull_chain_big_kernel, dim3((unsigned)chain_blocks), dim3(64), 0, s, a->d, (const double2*)sorted,
                   (const int32_t*)n_pts, (const int32_t*)big_list, (const int32_t*)(big_list + n), stack, idx_scratch, sizes);
    }
    GPK_TRY(exclusive_scan_i32(sizes, n, off_dev, nullptr, btot, s));
    GPK_LAUNCH("gpk_hull_compact", hull_compact_kernel, ggrid, block, 0, s, a->d, stack, off_dev, out_dev);
    if (host_out) {
        GPK_TRY(copy_out(out_ring_offsets, out_space, off_dev, off_bytes, s));
        const int32_t total = out_ring_offsets[n];
        GPK_TRY(copy_out(out_xy, out_space, out_dev, sizeof(double2) * (size_t)total, s));
    }
    return GPK_OK;
}
