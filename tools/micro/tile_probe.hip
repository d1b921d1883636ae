// Microbenchmark: how far is a minimal "points in -> one raster-word gather -> code + count out" kernel from
// the level-1 part of gpk_pip_tile?  Adds, step by step, what the real kernel has around the gather.
// hipcc --offload-arch=gfx950 -O3 tile_probe.hip -o tile_probe && ./tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2f64 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ double2 ld_nt(const double2* p) { const v2f64 v = __builtin_nontemporal_load(reinterpret_cast<const v2f64*>(p)); return make_double2(v.x, v.y); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: gather, one output          1: two outputs (code + count)      2: + nt loads / nt stores
// MODE 3: + LDS round trip (s_cnt/s_hit written by the owner lane, barrier, read back)
// MODE 4: + 32-byte second-level gather for 30% of the points
// MODE 5: MODE 3 with 4 points per thread
template <int MODE, int PPT>
__global__ __launch_bounds__(256) void probe(const double2* __restrict__ pts, long n, const unsigned* __restrict__ table, int R,
                                             double inv, unsigned* __restrict__ out, unsigned* __restrict__ out2,
                                             const uint4* __restrict__ sub) {
    __shared__ unsigned s_cnt[256 * PPT], s_hit[256 * PPT * 2];
    const long base = (long)blockIdx.x * (256 * PPT);
    double2 p[PPT];
    unsigned w[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        p[k] = i < n ? (MODE >= 2 ? ld_nt(pts + i) : pts[i]) : make_double2(0, 0);
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int cx = (int)fmin(fmax(p[k].x * inv, 0.0), (double)(R - 1)), cy = (int)fmin(fmax(p[k].y * inv, 0.0), (double)(R - 1));
        w[k] = table[(unsigned)cy * (unsigned)R + (unsigned)cx];
    }
    if (MODE >= 4) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if ((w[k] & 0xFF) < 77) {  // ~30% of the points
                const uint4 a = sub[(w[k] >> 8) * 2], b = sub[(w[k] >> 8) * 2 + 1];
                w[k] ^= a.x + a.w + b.y + b.z;
            }
        }
    }
    if (MODE >= 3) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int li = k * 256 + threadIdx.x;
            s_cnt[li] = w[k] & 1;
            s_hit[li * 2] = w[k] >> 1;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int li = k * 256 + threadIdx.x;
            w[k] = s_cnt[li] ? s_hit[li * 2] : 0xFFFFFFFFu;
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (i >= n) continue;
        if (MODE >= 2) {
            __builtin_nontemporal_store(w[k], out + i);
            __builtin_nontemporal_store(w[k] != 0xFFFFFFFFu ? 1u : 0u, out2 + i);
        } else {
            out[i] = w[k];
            if (MODE >= 1) out2[i] = w[k] != 0xFFFFFFFFu ? 1u : 0u;
        }
    }
}

template <int MODE, int PPT>
static float run(const double2* d, long n, const unsigned* t, int R, double inv, unsigned* o, unsigned* o2, const uint4* sub) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipEventRecord(a));
        const unsigned grid = (unsigned)((n + 256 * PPT - 1) / (256 * PPT));
        hipLaunchKernelGGL((probe<MODE, PPT>), dim3(grid), dim3(256), 0, 0, d, n, t, R, inv, o, o2, sub);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const long n = 10000000;
    std::vector<double2> h(n);
    srand(1);
    for (long i = 0; i < n; ++i) h[i] = make_double2(1000.0 * rand() / RAND_MAX, 1000.0 * rand() / RAND_MAX);
    double2* d; unsigned *t, *o, *o2; uint4* sub;
    const int R = 512;
    std::vector<unsigned> ht((size_t)R * R);
    for (auto& x : ht) x = ((unsigned)rand() % 100000u) << 8 | ((unsigned)rand() & 0xFF);
    CK(hipMalloc(&d, n * 16)); CK(hipMalloc(&o, n * 4)); CK(hipMalloc(&o2, n * 4)); CK(hipMalloc(&t, (size_t)R * R * 4));
    CK(hipMalloc(&sub, 100000 * 32)); CK(hipMemset(sub, 1, 100000 * 32));
    CK(hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(t, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    const double inv = R / 1000.0;
    printf("mode0 gather, 1 output:            %.1f us\n", run<0, 2>(d, n, t, R, inv, o, o2, sub));
    printf("mode1 + count output:              %.1f us\n", run<1, 2>(d, n, t, R, inv, o, o2, sub));
    printf("mode2 + nt loads/stores:           %.1f us\n", run<2, 2>(d, n, t, R, inv, o, o2, sub));
    printf("mode3 + LDS round trip + barrier:  %.1f us\n", run<3, 2>(d, n, t, R, inv, o, o2, sub));
    printf("mode4 + 32B second gather (30%%):   %.1f us\n", run<4, 2>(d, n, t, R, inv, o, o2, sub));
    printf("mode4, 4 points/thread:            %.1f us\n", run<4, 4>(d, n, t, R, inv, o, o2, sub));
    printf("mode4, 1 point/thread:             %.1f us\n", run<4, 1>(d, n, t, R, inv, o, o2, sub));
    return 0;
}
