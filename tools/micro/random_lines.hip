// Microbenchmark: how many RANDOM 64-byte lines per second does one MI355X deliver, as a function of the footprint?
// Every lane issues 8 independent 16-byte loads at hashed line addresses (no dependence between them: latency is hidden,
// what remains is the memory system's throughput for scattered lines — address translation included).
// hipcc --offload-arch=gfx950 -O3 random_lines.hip -o random_lines && ./random_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ table, unsigned long long n_lines, unsigned long long salt, unsigned* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = table[(mix(i * 8 + k + salt) % n_lines) * 4];  // 4 x uint4 = one 64-byte line
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].w;
    if (acc == 0x12345678u) out[0] = acc;  // (keeps the loads)
}
int main() {
    unsigned* out; CK(hipMalloc(&out, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned long long loads = 1ull << 25;  // 33.5M lines per launch
    for (unsigned long long mb : {16ull, 64ull, 256ull, 1024ull, 3072ull, 6144ull}) {
        uint4* t; const size_t bytes = (size_t)mb << 20;
        CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 1, bytes));
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(probe, dim3((unsigned)(loads / 8 / 256)), dim3(256), 0, 0, t, (unsigned long long)(bytes / 64), (unsigned long long)rep * 977, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 1 && ms < best) best = ms;
        }
        printf("footprint %5llu MB: %.3f ms for %.1fM random lines = %.1f G lines/s = %.2f TB/s of 64-byte lines\n", mb, best, loads / 1e6, loads / best / 1e6, loads * 64.0 / best / 1e9);
        CK(hipFree(t));
    }
    return 0;
}
