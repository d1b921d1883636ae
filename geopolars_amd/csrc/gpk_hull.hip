// gpk_hull.hip — convex_hull (geoseries.rs:23-26,196-198; geo 0.27 convex_hull/qhull.rs semantics:
// closed counter-clockwise exterior, collinear vertices dropped).
//
// The hull of a point set is unique, so any exact algorithm returns the same ring up to its start
// vertex; this kernel emits it starting at the lexicographically smallest vertex.  One lane per
// geometry: in-place heap sort of a copy (LDS, or global scratch for large geometries), then a monotone chain driven by the exact
// orientation kernel.  Irregular per-row output sizes go through size -> scan -> compact.
// This is the lowest-traffic operator of the surface; the sort's working set sits in LDS.
#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

__device__ __forceinline__ bool xy_less(double2 a, double2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

// in-place heap sort of n points reached through ld(i) / st(i, v) (global scratch or the lane's LDS column)
template <typename LD, typename ST>
__device__ __forceinline__ void heap_sort(LD ld, ST st, int n) {
    auto sift = [&](int root, int end) {
        double2 rv = ld(root);
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            double2 cv = ld(child);
            if (child + 1 < end) {
                const double2 c2 = ld(child + 1);
                if (xy_less(cv, c2)) {
                    cv = c2;
                    ++child;
                }
            }
            if (!xy_less(rv, cv)) break;
            st(root, cv);
            root = child;
        }
        st(root, rv);
    };
    for (int start = n / 2 - 1; start >= 0; --start) sift(start, n);
    for (int end = n - 1; end > 0; --end) {
        const double2 t = ld(0);
        st(0, ld(end));
        st(end, t);
        sift(0, end);
    }
}

// sort + dedup + monotone chain over the points ld(0..n); the hull is written to h (global), closed; returns its size.
// The two topmost stack entries live in registers, so the chain touches memory once per push and once per pop.
template <typename LD, typename ST>
__device__ __forceinline__ int hull_of(LD ld, ST st, int n, double2* __restrict__ h) {
    heap_sort(ld, st, n);
    int m = 0;
    {
        double2 prev = make_double2(0, 0);
        for (int i = 0; i < n; ++i) {
            const double2 v = ld(i);
            if (m == 0 || v.x != prev.x || v.y != prev.y) {
                st(m++, v);
                prev = v;
            }
        }
    }
    const double2 p0 = ld(0);
    int k = 0;
    if (m < 3) {
        for (int i = 0; i < m; ++i) h[k++] = ld(i);
    } else {
        double2 t1 = p0, t2 = p0;  // h[k-1], h[k-2]
        auto push = [&](double2 v) {
            h[k++] = v;
            t2 = t1;
            t1 = v;
        };
        auto pop = [&]() {
            --k;
            t1 = t2;
            if (k >= 2) t2 = h[k - 2];
        };
        for (int i = 0; i < m; ++i) {
            const double2 v = ld(i);
            while (k >= 2 && dev::orient2d(t2.x, t2.y, t1.x, t1.y, v.x, v.y) <= 0) pop();
            push(v);
        }
        const int lo = k + 1;
        for (int i = m - 2; i >= 0; --i) {
            const double2 v = ld(i);
            while (k >= lo && dev::orient2d(t2.x, t2.y, t1.x, t1.y, v.x, v.y) <= 0) pop();
            push(v);
        }
        --k;
        if (k < 3) {  // all collinear: the two extremes
            k = 2;
            h[0] = p0;
            h[1] = ld(m - 1);
        }
    }
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

// scratch layout per geometry g with coordinate range [c0, c1): sorted copy at sorted[c0..c1) (global path only),
// chain stack at stack[2*c0 + 2*g .. 2*c1 + 2*g + 2).
// One lane per geometry, one wave per work-group.  When every geometry of the wave has at most HULL_CAP points (the
// closing duplicate of a ring does not count) the lanes keep their points in LDS, interleaved (element i of lane l at
// [i * 64 + l]: conflict-free for equal i, and the sort's data-dependent indices never leave the lane's column); the heap
// sort then runs out of LDS instead of making ~800 scattered global accesses per geometry (2M x 64-vertex polygons:
// 70 ms -> see DESIGN.md).  Larger geometries sort in global scratch as before.
constexpr int HULL_CAP = 64;
__global__ __launch_bounds__(64) void hull_kernel(DevGeo a, double2* __restrict__ sorted, double2* __restrict__ stack,
                                                  int32_t* __restrict__ sizes) {
    __shared__ double2 lds[HULL_CAP * 64];
    const int lane = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * 64 + lane;
    int c0 = 0, c1 = 0;
    bool act = g < a.n_geoms;
    if (act) {
        geom_coord_range(a, g, c0, c1);
        if (!dev::valid_row(a.validity, g) || c1 == c0) {
            sizes[g] = 0;
            act = false;
        }
    }
    int n = c1 - c0;
    if (act && n >= 2) {  // a closing duplicate would only be removed by the dedup pass: drop it before sorting
        const double2 f = a.xy[c0], l = a.xy[c1 - 1];
        if (f.x == l.x && f.y == l.y) --n;
    }
    const bool fits = !act || n <= HULL_CAP;
    const bool all_fit = __all(fits);  // wave-uniform
    if (!act) return;
    double2* h = stack + 2 * (int64_t)c0 + 2 * g;
    if (all_fit) {
        for (int i = 0; i < n; ++i) lds[i * 64 + lane] = a.xy[c0 + i];
        sizes[g] = hull_of([&](int i) { return lds[i * 64 + lane]; }, [&](int i, double2 v) { lds[i * 64 + lane] = v; }, n, h);
    } else {
        double2* p = sorted + c0;
        for (int i = 0; i < n; ++i) p[i] = a.xy[c0 + i];
        sizes[g] = hull_of([&](int i) { return p[i]; }, [&](int i, double2 v) { p[i] = v; }, n, h);
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
                  align256(off_bytes) + align256(sizeof(unsigned long long) * (size_t)(nb + 2)) + 1024;
    if (host_out) need += align256(sizeof(double2) * cap_coords) + align256(off_bytes);
    GPK_TRY(workspace().begin(need));
    double2* sorted = (double2*)workspace().take(sizeof(double2) * (size_t)(nc + 1));
    double2* stack = (double2*)workspace().take(sizeof(double2) * (2 * (size_t)nc + 2 * (size_t)n + 2));
    int32_t* sizes = (int32_t*)workspace().take(off_bytes);
    unsigned long long* btot = (unsigned long long*)workspace().take(sizeof(unsigned long long) * (size_t)(nb + 2));
    double2* out_dev = host_out ? (double2*)workspace().take(sizeof(double2) * cap_coords) : (double2*)out_xy;
    int32_t* off_dev = host_out ? (int32_t*)workspace().take(off_bytes) : out_ring_offsets;
    if (n == 0) {
        GPK_HIP(hipMemsetAsync(off_dev, 0, sizeof(int32_t), s));
        return copy_out(out_ring_offsets, out_space, off_dev, sizeof(int32_t), s);
    }
    const dim3 grid((unsigned)nb), block(256);
    GPK_LAUNCH("gpk_hull", hull_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a->d, sorted, stack, sizes);
    GPK_TRY(exclusive_scan_i32(sizes, n, off_dev, nullptr, btot, s));
    GPK_LAUNCH("gpk_hull_compact", hull_compact_kernel, grid, block, 0, s, a->d, stack, off_dev, out_dev);
    if (host_out) {
        GPK_TRY(copy_out(out_ring_offsets, out_space, off_dev, off_bytes, s));
        const int32_t total = out_ring_offsets[n];
        GPK_TRY(copy_out(out_xy, out_space, out_dev, sizeof(double2) * (size_t)total, s));
    }
    return GPK_OK;
}
