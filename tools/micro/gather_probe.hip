// Microbenchmark: what does one random 4-byte gather per point cost on MI355X when the table is
// 1 MB / 4 MB / 16 MB, next to streaming 16 B per point in and 4 B per point out?
// hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o gather_probe && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const double2* __restrict__ pts, long n, const unsigned* __restrict__ table, int R,
                                             double inv, unsigned* __restrict__ out) {
    long i = (long)blockIdx.x * 512 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k, i += 256) {
        if (i >= n) return;
        const double2 p = pts[i];
        int cx = (int)floor(p.x * inv), cy = (int)floor(p.y * inv);
        cx = cx < 0 ? 0 : (cx >= R ? R - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= R ? R - 1 : cy);
        unsigned w;
        if (MODE == 0) w = (unsigned)(cx + cy);                 // no gather
        if (MODE == 1) w = table[(long)cy * R + cx];            // random gather
        if (MODE == 2) w = table[((long)cy * R + cx) & 1023];   // gather from a 4 KB window (L1 hits)
        if (MODE == 3) w = __builtin_nontemporal_load(table + (long)cy * R + cx);
        out[i] = w;
    }
}

int main() {
    const long n = 10000000;
    std::vector<double2> h(n);
    srand(1);
    for (long i = 0; i < n; ++i) h[i] = make_double2(1000.0 * rand() / RAND_MAX, 1000.0 * rand() / RAND_MAX);
    double2* d; unsigned *t, *o;
    CK(hipMalloc(&d, n * 16)); CK(hipMalloc(&o, n * 4));
    CK(hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int R : {512, 1024, 2048}) {
        CK(hipMalloc(&t, (size_t)R * R * 4)); CK(hipMemset(t, 1, (size_t)R * R * 4));
        const double inv = R / 1000.0;
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(a));
                const unsigned grid = (unsigned)((n + 511) / 512);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, d, n, t, R, inv, o);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, d, n, t, R, inv, o);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, d, n, t, R, inv, o);
                if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, d, n, t, R, inv, o);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("R=%d table=%.1f MB mode=%d (%s): %.1f us\n", R, R * (double)R * 4 / 1e6, mode,
                   mode == 0 ? "no gather" : mode == 1 ? "random gather" : mode == 2 ? "4KB-window gather" : "nt gather", best * 1e3);
        }
        CK(hipFree(t));
    }
    return 0;
}
