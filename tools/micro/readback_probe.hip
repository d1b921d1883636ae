// what a 4-byte read-back costs on this runtime: hipMemcpyAsync into pageable / pinned host memory + stream sync, and a kernel that
// writes the word into pinned MAPPED host memory + stream sync (no copy).   hipcc --offload-arch=gfx950 -O2 readback_probe.hip -o readback_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void bump(int* p) { *p += 1; }
__global__ void publish(const int* p, int* host) { *host = *p; }
int main() {
    int* d;
    hipMalloc(&d, 4);
    hipMemset(d, 0, 4);
    hipStream_t s;
    hipStreamCreate(&s);
    int pageable = 0, *pinned, *mapped, *mapped_dev;
    hipHostMalloc(&pinned, 4, hipHostMallocDefault);
    hipHostMalloc(&mapped, 4, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&mapped_dev, mapped, 0);
    const int N = 200;
    auto run = [&](const char* name, auto body) {
        for (int i = 0; i < 10; ++i) body();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) body();
        auto t1 = std::chrono::steady_clock::now();
        printf("%-46s %7.1f us\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    };
    run("kernel + sync (no read-back)", [&] { hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipStreamSynchronize(s); });
    run("kernel + memcpyAsync to PAGEABLE + sync", [&] { hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipMemcpyAsync(&pageable, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
    run("kernel + memcpyAsync to PINNED + sync", [&] { hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
    run("kernel + publish kernel to MAPPED + sync", [&] { hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipLaunchKernelGGL(publish, 1, 1, 0, s, d, mapped_dev); hipStreamSynchronize(s); });
    run("two kernels back to back + sync", [&] { hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipStreamSynchronize(s); });
    run("ten kernels back to back + sync", [&] { for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(bump, 1, 1, 0, s, d); hipStreamSynchronize(s); });
    printf("(values: %d %d %d)\n", pageable, *pinned, *mapped);
    return 0;
}
