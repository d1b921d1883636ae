// gpk_hull.hip — convex_hull (geoseries.rs:23-26,196-198; geo 0.27 convex_hull/qhull.rs semantics:
// closed counter-clockwise exterior, collinear vertices dropped).
//
// The hull of a point set is unique, so any exact algorithm returns the same ring up to its start
// vertex; this kernel emits it starting at the lexicographically smallest vertex.  One lane per
// geometry: in-place heap sort of a scratch copy, then a monotone chain driven by the exact
// orientation kernel.  Irregular per-row output sizes go through size -> scan -> compact.
// This is the lowest-traffic operator of the surface and is not tuned.
#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

__device__ __forceinline__ bool xy_less(double2 a, double2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

__device__ inline void heap_sort(double2* v, int n) {
    for (int start = n / 2 - 1; start >= 0; --start) {
        int root = start;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= n) break;
            if (child + 1 < n && xy_less(v[child], v[child + 1])) ++child;
            if (!xy_less(v[root], v[child])) break;
            const double2 t = v[root];
            v[root] = v[child];
            v[child] = t;
            root = child;
        }
    }
    for (int end = n - 1; end > 0; --end) {
        const double2 t = v[0];
        v[0] = v[end];
        v[end] = t;
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && xy_less(v[child], v[child + 1])) ++child;
            if (!xy_less(v[root], v[child])) break;
            const double2 u = v[root];
            v[root] = v[child];
            v[child] = u;
            root = child;
        }
    }
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

// scratch layout per geometry g with coordinate range [c0, c1): sorted copy at sorted[c0..c1),
// chain stack at stack[2*c0 + 2*g .. 2*c1 + 2*g + 2)
__global__ void hull_kernel(DevGeo a, double2* __restrict__ sorted, double2* __restrict__ stack,
                            int32_t* __restrict__ sizes) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    int c0, c1;
    geom_coord_range(a, g, c0, c1);
    const int n = c1 - c0;
    if (!dev::valid_row(a.validity, g) || n == 0) {
        sizes[g] = 0;
        return;
    }
    double2* p = sorted + c0;
    double2* h = stack + 2 * (int64_t)c0 + 2 * g;
    for (int i = 0; i < n; ++i) p[i] = a.xy[c0 + i];
    heap_sort(p, n);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (m == 0 || p[i].x != p[m - 1].x || p[i].y != p[m - 1].y) p[m++] = p[i];
    int k = 0;
    if (m < 3) {
        for (int i = 0; i < m; ++i) h[k++] = p[i];
    } else {
        for (int i = 0; i < m; ++i) {
            while (k >= 2 && dev::orient2d(h[k - 2].x, h[k - 2].y, h[k - 1].x, h[k - 1].y, p[i].x, p[i].y) <= 0) --k;
            h[k++] = p[i];
        }
        const int lo = k + 1;
        for (int i = m - 2; i >= 0; --i) {
            while (k >= lo && dev::orient2d(h[k - 2].x, h[k - 2].y, h[k - 1].x, h[k - 1].y, p[i].x, p[i].y) <= 0) --k;
            h[k++] = p[i];
        }
        --k;
        if (k < 3) {  // all collinear: the two extremes
            k = 2;
            h[0] = p[0];
            h[1] = p[m - 1];
        }
    }
    h[k] = h[0];  // close the ring
    sizes[g] = k + 1;
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
    GPK_LAUNCH("gpk_hull", hull_kernel, grid, block, 0, s, a->d, sorted, stack, sizes);
    GPK_TRY(exclusive_scan_i32(sizes, n, off_dev, nullptr, btot, s));
    GPK_LAUNCH("gpk_hull_compact", hull_compact_kernel, grid, block, 0, s, a->d, stack, off_dev, out_dev);
    if (host_out) {
        GPK_TRY(copy_out(out_ring_offsets, out_space, off_dev, off_bytes, s));
        const int32_t total = out_ring_offsets[n];
        GPK_TRY(copy_out(out_xy, out_space, out_dev, sizeof(double2) * (size_t)total, s));
    }
    return GPK_OK;
}
