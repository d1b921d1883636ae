// gpk_lineal_ops.hip — the two remaining operators of `trait GeoSeries` that walk coordinate sequences:
//   geodesic_length   geoseries.rs:52-58,216-218   (methods of py-geopolars/src/geo.rs:61-78: geodesic (Karney, gpk_karney.h) |
//                                                   haversine | vincenty)
//   simplify          geoseries.rs:108-116,240-242  Ramer-Douglas-Peucker, geo 0.27 algorithm/simplify.rs
// Both follow the upstream crate's published behaviour (the bodies in the reference are todo!()); what each restates is said
// at the function.  Coordinates are (lon, lat) degrees for the geodesic lengths, like geo's HaversineLength / VincentyLength.
#include <cfloat>
#include <cmath>

#include "gpk_device.h"
#include "gpk_karney.h"
#include "gpk_index.h"
#include "gpk_scan.h"

namespace gpk {

// innermost coordinate sequences of an array (rings / linestrings): offsets and count
static void seq_level(const DevGeo& a, const int32_t** off, int64_t* n) {
    switch (a.type) {
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT:
        *off = a.geom_off;
        *n = a.n_geoms;
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING:
    case GPK_GEOM_MULTIPOLYGON:
        *off = a.ring_off;
        *n = a.n_rings;
        break;
    default:
        *off = nullptr;
        *n = 0;
    }
}

// ---- geodesic lengths -----------------------------------------------------------------------------------------------
// geo 0.27 haversine_distance.rs: mean earth radius 6371008.8 m (IUGG), the half-angle formula
__device__ __forceinline__ double haversine_m(double lon1, double lat1, double lon2, double lat2) {
    const double rad = 0.017453292519943295;  // pi / 180 (f64::to_radians multiplies by this constant)
    const double t1 = lat1 * rad, t2 = lat2 * rad;
    const double dt = (lat2 - lat1) * rad, dl = (lon2 - lon1) * rad;
    const double sh = sin(dt / 2.0), sl = sin(dl / 2.0);
    const double a = sh * sh + cos(t1) * cos(t2) * (sl * sl);
    return 6371008.8 * (2.0 * asin(sqrt(a)));
}
// geo 0.27 vincenty_distance.rs (Vincenty's inverse formula on WGS84, at most 100 iterations, |d lambda| <= 1e-12):
// NaN where upstream returns Err(FailedToConvergeError) (nearly antipodal points)
__device__ inline double vincenty_m(double lon1, double lat1, double lon2, double lat2) {
    const double rad = 0.017453292519943295;
    const double a = 6378137.0, b = 6356752.314245, f = 1.0 / 298.257223563;
    const double L = (lon2 - lon1) * rad;
    const double U1 = atan((1.0 - f) * tan(lat1 * rad)), U2 = atan((1.0 - f) * tan(lat2 * rad));
    const double sU1 = sin(U1), cU1 = cos(U1), sU2 = sin(U2), cU2 = cos(U2);
    double lam = L, lam_p, sS = 0, cS = 0, sig = 0, c2A = 0, c2SM = 0;
    int it = 100;
    for (;;) {
        const double sl = sin(lam), cl = cos(lam);
        const double t0 = cU2 * sl, t1 = cU1 * sU2 - sU1 * cU2 * cl;
        sS = sqrt(t0 * t0 + t1 * t1);
        if (sS == 0.0) return (lon1 == lon2 && lat1 == lat2) ? 0.0 : NAN;  // coincident points: 0
        cS = sU1 * sU2 + cU1 * cU2 * cl;
        sig = atan2(sS, cS);
        const double sA = cU1 * cU2 * sl / sS;
        c2A = 1.0 - sA * sA;
        c2SM = c2A == 0.0 ? 0.0 : cS - 2.0 * sU1 * sU2 / c2A;  // equatorial line: cos^2 alpha = 0
        const double C = f / 16.0 * c2A * (4.0 + f * (4.0 - 3.0 * c2A));
        lam_p = lam;
        lam = L + (1.0 - C) * f * sA * (sig + C * sS * (c2SM + C * cS * (-1.0 + 2.0 * c2SM * c2SM)));
        if (fabs(lam - lam_p) <= 1e-12) break;
        if (--it == 0) return NAN;
    }
    const double uSq = c2A * (a * a - b * b) / (b * b);
    const double A = 1.0 + uSq / 16384.0 * (4096.0 + uSq * (-768.0 + uSq * (320.0 - 175.0 * uSq)));
    const double B = uSq / 1024.0 * (256.0 + uSq * (-128.0 + uSq * (74.0 - 47.0 * uSq)));
    const double dS = B * sS * (c2SM + B / 4.0 * (cS * (-1.0 + 2.0 * c2SM * c2SM) - B / 6.0 * c2SM * (-3.0 + 4.0 * sS * sS) * (-3.0 + 4.0 * c2SM * c2SM)));
    return b * A * (sig - dS);
}

// G lanes per sequence: lane k takes segments k, k + G, ...; the partial sums fold with DPP row operations
template <int G, int METHOD>
__global__ __launch_bounds__(256) void geodesic_seq_kernel(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                                           double* __restrict__ seq_len) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t s = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; s < n_seq; s += groups) {
        const int c0 = seq_off[s], c1 = seq_off[s + 1];
        double v = 0.0;
        for (int i = c0 + lane; i + 1 < c1; i += G) {
            const double2 p = xy[i], q = xy[i + 1];
            v += METHOD == GPK_GEODESIC_HAVERSINE ? haversine_m(p.x, p.y, q.x, q.y)
                 : METHOD == GPK_GEODESIC_VINCENTY ? vincenty_m(p.x, p.y, q.x, q.y)
                                                   : karney::k_geodesic_m(p.x, p.y, q.x, q.y);
        }
        v = dev::group_sum<G>(v);
        if (lane == 0) seq_len[s] = v;
    }
}
// rows from sequences: linestrings = their sequences, polygons = exterior rings only (the rule of euclidean_length,
// geoseries.rs:35-41,52-58), points = 0; null rows NaN
__global__ void geodesic_combine_kernel(DevGeo a, const double* __restrict__ seq_len, double* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    if (!dev::valid_row(a.validity, g)) {
        out[g] = NAN;
        return;
    }
    double v = 0.0;
    if (is_polygonal(a.type)) {
        int p0, p1;
        dev::geom_parts(a, g, p0, p1);
        for (int p = p0; p < p1; ++p) {
            int r0, r1;
            dev::part_rings(a, p, r0, r1);
            if (r1 > r0) v += seq_len[r0];
        }
    } else if (a.type == GPK_GEOM_LINESTRING) {
        v = seq_len[g];
    } else if (a.type == GPK_GEOM_MULTILINESTRING) {
        for (int s = a.geom_off[g]; s < a.geom_off[g + 1]; ++s) v += seq_len[s];
    }
    out[g] = v;
}

// ---- simplify (Ramer-Douglas-Peucker) -------------------------------------------------------------------------------
// geo-types private_utils::line_segment_distance, as geo's rdp uses it (distance from a point to the SEGMENT first-last),
// written with a plain sqrt of the sum of squares so that the CPU oracle evaluates the very same IEEE operations (upstream calls
// hypot(): at most an ulp away, which matters only for exact ties with `epsilon` or between two candidates)
__device__ __forceinline__ double seg_dist(double2 p, double2 s, double2 e) {
    const double dx = e.x - s.x, dy = e.y - s.y;
    if (s.x == e.x && s.y == e.y) return sqrt((p.x - s.x) * (p.x - s.x) + (p.y - s.y) * (p.y - s.y));
    const double d2 = dx * dx + dy * dy;
    const double r = ((p.x - s.x) * dx + (p.y - s.y) * dy) / d2;
    if (r <= 0.0) return sqrt((p.x - s.x) * (p.x - s.x) + (p.y - s.y) * (p.y - s.y));
    if (r >= 1.0) return sqrt((p.x - e.x) * (p.x - e.x) + (p.y - e.y) * (p.y - e.y));
    const double t = ((s.y - p.y) * dx - (s.x - p.x) * dy) / d2;
    return fabs(t) * sqrt(d2);
}

// geo 0.27 simplify.rs compute_rdp, iteratively, G lanes per sequence.  A range [i, j] of the sequence is settled like this:
//   the farthest interior point from the segment i-j (the LAST one among equals: upstream folds with `>=`); farther than
//   epsilon -> split there, left part first (the order matters: `simplified_len` is shared state); otherwise the interior is
//   culled — unless that would leave the whole sequence with fewer than `min_pts` points (INITIAL_MIN: 2 for linestrings, 4 for
//   polygon rings), in which case the range keeps all its points.
// keep[] marks surviving coordinates; stack[] (one (i, j) pair per coordinate of the sequence at most) lives in global scratch.
template <int G>
__global__ __launch_bounds__(256) void rdp_kernel(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq, double eps,
                                                  int min_pts, uint8_t* __restrict__ keep, int2* __restrict__ stack, int32_t* __restrict__ sizes) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t s = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; s < n_seq; s += groups) {
        const int c0 = seq_off[s], n = seq_off[s + 1] - c0;
        const double2* __restrict__ v = xy + c0;
        uint8_t* __restrict__ kp = keep + c0;
        int2* __restrict__ st = stack + c0;
        for (int i = lane; i < n; i += G) kp[i] = 1;
        // the group's lanes overwrite each other's marks below (lane (k - ri - 1) % G culls what lane k % G marked): order them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int len = n, top = 0;  // simplified_len; stack depth (identical on every lane of the group)
        int ri = 0, rj = n - 1;
        bool have = n >= 3 && eps > 0.0;  // fewer than three points, or a non-positive epsilon: unchanged (geo's rdp wrapper)
        while (have) {
            if (rj - ri >= 2) {
                const double2 a = v[ri], b = v[rj];
                double best = 0.0;
                int at = 0;
                for (int k = ri + 1 + lane; k < rj; k += G) {
                    const double d = seg_dist(v[k], a, b);
                    if (d >= best) {  // ascending k within the lane: the last maximum stays
                        best = d;
                        at = k;
                    }
                }
#pragma unroll
                for (int o = G / 2; o > 0; o >>= 1) {  // the farthest, the larger index among equals
                    const double ob = __shfl_xor(best, o, 64);
                    const int oa = __shfl_xor(at, o, 64);
                    if (ob > best || (ob == best && oa > at)) {
                        best = ob;
                        at = oa;
                    }
                }
                if (best > eps) {  // split: the right part waits on the stack, the left part is next
                    if (lane == 0) st[top] = make_int2(at, rj);
                    ++top;
                    rj = at;
                    continue;
                }
                const int culled = rj - ri - 1;
                if (len - culled >= min_pts) {
                    len -= culled;
                    for (int k = ri + 1 + lane; k < rj; k += G) kp[k] = 0;
                }
            }
            if (top == 0) break;
            --top;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // lane 0's stack writes are visible to the group
            const int2 r = st[top];
            ri = r.x;
            rj = r.y;
        }
        if (lane == 0) sizes[s] = n < 3 ? n : len;
    }
}
template <int G>
__global__ __launch_bounds__(256) void rdp_compact_kernel(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                                          const uint8_t* __restrict__ keep, const int32_t* __restrict__ out_off, double2* __restrict__ out) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t s = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; s < n_seq; s += groups) {
        const int c0 = seq_off[s], n = seq_off[s + 1] - c0;
        int o = out_off[s];
        for (int base = 0; base < n; base += G) {  // ballot-free ordered compaction: prefix of the keep flags inside the group
            const int k = base + lane;
            const int flag = k < n ? (int)keep[c0 + k] : 0;
            int pre = flag;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) {
                const int t = __shfl_up(pre, d, G);
                if (lane >= d) pre += t;
            }
            if (flag) out[o + pre - 1] = xy[c0 + k];
            o += __shfl(pre, G - 1, G);
        }
    }
}

static inline dim3 group_grid(int64_t n_items, int G) {
    const int64_t per_block = 256 / G;
    int64_t blocks = (n_items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)cu_count() * 32;
    if (blocks > cap) blocks = cap;
    return dim3((unsigned)(blocks > 0 ? blocks : 1));
}

}  // namespace gpk

using namespace gpk;

extern "C" {

int32_t gpk_geodesic_length(const gpk_geoarray* a, int32_t method, double* out, int32_t out_space, void* stream) {
    if (!a || (!out && a->d.n_geoms > 0)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (method != GPK_GEODESIC_KARNEY && method != GPK_GEODESIC_HAVERSINE && method != GPK_GEODESIC_VINCENTY)
        return fail(GPK_ERR_INVALID_ARGUMENT, "Geodesic calculation method not valid. Use one of geodesic, haversine or vincenty");  // geo.rs:68-71
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    const int32_t* seq_off;
    int64_t n_seq;
    seq_level(a->d, &seq_off, &n_seq);
    if (a->d.type == GPK_GEOM_MULTIPOINT) n_seq = 0;
    const bool host = out_space != GPK_MEM_DEVICE;
    const size_t ob = sizeof(double) * (size_t)n, sb = sizeof(double) * (size_t)(n_seq > 0 ? n_seq : 1);
    GPK_TRY(workspace_aux(0).begin(align256(sb) + (host ? align256(ob) : 0) + 512));
    double* seq_len = (double*)workspace_aux(0).take(sb);
    double* out_dev = host ? (double*)workspace_aux(0).take(ob) : out;
    if (n_seq > 0) {
        const double mean = (double)a->d.n_coords / (double)n_seq;
        // (Karney's method is a few hundred f64 operations per segment with a data-dependent Newton loop: always the wide groups)
        if (method == GPK_GEODESIC_KARNEY)
            GPK_LAUNCH("gpk_geodesic_seq", (geodesic_seq_kernel<16, GPK_GEODESIC_KARNEY>), group_grid(n_seq, 16), dim3(256), 0, s, a->d.xy, seq_off, n_seq, seq_len);
        else if (mean <= 24.0) {
            if (method == GPK_GEODESIC_HAVERSINE)
                GPK_LAUNCH("gpk_geodesic_seq", (geodesic_seq_kernel<4, GPK_GEODESIC_HAVERSINE>), group_grid(n_seq, 4), dim3(256), 0, s, a->d.xy, seq_off, n_seq, seq_len);
            else
                GPK_LAUNCH("gpk_geodesic_seq", (geodesic_seq_kernel<4, GPK_GEODESIC_VINCENTY>), group_grid(n_seq, 4), dim3(256), 0, s, a->d.xy, seq_off, n_seq, seq_len);
        } else {
            if (method == GPK_GEODESIC_HAVERSINE)
                GPK_LAUNCH("gpk_geodesic_seq", (geodesic_seq_kernel<16, GPK_GEODESIC_HAVERSINE>), group_grid(n_seq, 16), dim3(256), 0, s, a->d.xy, seq_off, n_seq, seq_len);
            else
                GPK_LAUNCH("gpk_geodesic_seq", (geodesic_seq_kernel<16, GPK_GEODESIC_VINCENTY>), group_grid(n_seq, 16), dim3(256), 0, s, a->d.xy, seq_off, n_seq, seq_len);
        }
    }
    GPK_LAUNCH("gpk_geodesic_combine", geodesic_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a->d, (const double*)seq_len, out_dev);
    return copy_out(out, out_space, out_dev, ob, s);
}

int32_t gpk_simplify(const gpk_geoarray* a, double epsilon, double* out_xy, int32_t* out_seq_offsets, int64_t* n_out_coords, int32_t out_space,
                     void* stream) {
    if (!a || !out_seq_offsets || !n_out_coords) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    if (a->d.type == GPK_GEOM_POINT || a->d.type == GPK_GEOM_MULTIPOINT)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "simplify: points have nothing to simplify — pass the column through unchanged");
    hipStream_t s = (hipStream_t)stream;
    const int32_t* seq_off;
    int64_t n_seq;
    seq_level(a->d, &seq_off, &n_seq);
    const int64_t nc = a->d.n_coords;
    const bool host = out_space != GPK_MEM_DEVICE;
    const size_t offb = sizeof(int32_t) * (size_t)(n_seq + 1), xyb = sizeof(double2) * (size_t)(nc > 0 ? nc : 1);
    const int64_t nb = (n_seq + 255) / 256;
    GPK_TRY(workspace_aux(0).begin(2 * align256(offb) + align256(sizeof(unsigned long long) * (size_t)(nb + 2)) + align256((size_t)nc + 8) +
                                   align256(sizeof(int2) * (size_t)(nc + 1)) + (host ? align256(xyb) : 0) + 1024));
    int32_t* sizes = (int32_t*)workspace_aux(0).take(offb);
    int32_t* off_dev = host ? (int32_t*)workspace_aux(0).take(offb) : out_seq_offsets;
    unsigned long long* btot = (unsigned long long*)workspace_aux(0).take(sizeof(unsigned long long) * (size_t)(nb + 2));
    uint8_t* keep = (uint8_t*)workspace_aux(0).take((size_t)nc + 8);
    int2* stack = (int2*)workspace_aux(0).take(sizeof(int2) * (size_t)(nc + 1));
    double2* out_dev = host ? (double2*)workspace_aux(0).take(xyb) : (double2*)out_xy;
    if (n_seq == 0) {
        GPK_HIP(hipMemsetAsync(off_dev, 0, sizeof(int32_t), s));
        *n_out_coords = 0;
        return copy_out(out_seq_offsets, out_space, off_dev, sizeof(int32_t), s);
    }
    const int min_pts = is_polygonal(a->d.type) ? 4 : 2;  // INITIAL_MIN of geo's rdp: rings keep at least 4 coordinates
    const double mean = (double)nc / (double)n_seq;
    if (mean <= 48.0) {
        GPK_LAUNCH("gpk_rdp", rdp_kernel<8>, group_grid(n_seq, 8), dim3(256), 0, s, a->d.xy, seq_off, n_seq, epsilon, min_pts, keep, stack, sizes);
    } else {
        GPK_LAUNCH("gpk_rdp", rdp_kernel<64>, group_grid(n_seq, 64), dim3(256), 0, s, a->d.xy, seq_off, n_seq, epsilon, min_pts, keep, stack, sizes);
    }
    GPK_TRY(exclusive_scan_i32(sizes, n_seq, off_dev, nullptr, btot, s));
    if (out_xy) {
        if (mean <= 48.0)
            GPK_LAUNCH("gpk_rdp_compact", rdp_compact_kernel<8>, group_grid(n_seq, 8), dim3(256), 0, s, a->d.xy, seq_off, n_seq, (const uint8_t*)keep,
                       (const int32_t*)off_dev, out_dev);
        else
            GPK_LAUNCH("gpk_rdp_compact", rdp_compact_kernel<64>, group_grid(n_seq, 64), dim3(256), 0, s, a->d.xy, seq_off, n_seq, (const uint8_t*)keep,
                       (const int32_t*)off_dev, out_dev);
    }
    unsigned long long total = 0;
    GPK_HIP(hipMemcpyAsync(&total, btot + nb, sizeof total, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    *n_out_coords = (int64_t)total;
    if (host) {
        GPK_TRY(copy_out(out_seq_offsets, out_space, off_dev, offb, s));
        if (out_xy) GPK_TRY(copy_out(out_xy, out_space, out_dev, sizeof(double2) * (size_t)total, s));
    }
    return GPK_OK;
}

}  // extern "C"
