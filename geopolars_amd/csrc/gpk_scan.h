// gpk_scan.h — block-level scan / reduce building blocks (wave64 shuffles + one LDS hop) and the
// tiny single-block exclusive scan used for per-block totals.  Deterministic: fixed tree order.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gpk_common.h"

namespace gpk {
namespace dev {

// inclusive scan across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T w = __shfl_up(v, o, 64);
        if (lane >= o) v += w;
    }
    return v;
}

// Exclusive scan of one value per thread across a block of BLOCK threads (BLOCK % 64 == 0,
// BLOCK <= 1024).  Returns the exclusive prefix; *total receives the block sum (all threads).
// `lds` must hold BLOCK/64 + 1 elements of T.
template <typename T, int BLOCK>
__device__ __forceinline__ T block_exclusive_scan(T v, T* lds, T* total) {
    constexpr int NW = BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T incl = wave_inclusive_scan(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const T t = lds[w];
            lds[w] = run;
            run += t;
        }
        lds[NW] = run;
    }
    __syncthreads();
    const T out = lds[wave] + incl - v;
    *total = lds[NW];
    __syncthreads();  // lds reusable by the caller afterwards
    return out;
}

}  // namespace dev

// Single-block exclusive scan of n 64-bit totals, in place; writes the grand total to *grand (and to
// *grand_host when given: a pinned, device-mapped word the host reads after the stream sync, which
// saves a D2H copy per call).  n is small (one entry per work-group of a preceding kernel).  Chunks of
// 8192 values are staged in LDS with coalesced loads; every thread then scans 8 consecutive LDS values,
// one block scan stitches the threads, and the chunk is written back coalesced.
static __global__ __launch_bounds__(1024) void scan_block_totals_kernel(unsigned long long* __restrict__ v,
                                                                         int64_t n,
                                                                         unsigned long long* __restrict__ grand,
                                                                         unsigned long long* __restrict__ grand_host = nullptr) {
    constexpr int PER = 8, CHUNK = 1024 * PER;
    __shared__ unsigned long long buf[CHUNK + CHUNK / 32];  // padded against bank conflicts of the stride-8 walk
    __shared__ unsigned long long lds[17];
    unsigned long long carry = 0;
    for (int64_t base = 0; base < n; base += CHUNK) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = k * 1024 + threadIdx.x;
            buf[j + (j >> 5)] = base + j < n ? v[base + j] : 0ull;
        }
        __syncthreads();
        unsigned long long x[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = threadIdx.x * PER + k;
            x[k] = buf[j + (j >> 5)];
            sum += x[k];
        }
        unsigned long long tot;
        unsigned long long run = carry + dev::block_exclusive_scan<unsigned long long, 1024>(sum, lds, &tot);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = threadIdx.x * PER + k;
            buf[j + (j >> 5)] = run;
            run += x[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = k * 1024 + threadIdx.x;
            if (base + j < n) v[base + j] = buf[j + (j >> 5)];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *grand = carry;
        if (grand_host) *grand_host = carry;
    }
}

// (round 6: eight values a thread — 2048 a work-group, two 16-byte loads each when the arrays are 16-byte aligned — instead of one: an
// index build over a 4096 x 4096 raster runs ~30 scans of up to 16.7 M values, and at 256 values a work-group each was 65 k work-groups
// twice over plus a 65 k-entry single-block scan of their totals: 147 us a scan, 4.4 ms of a 34 ms build)
constexpr int SCAN_PER = 8, SCAN_BLOCK = 256 * SCAN_PER;
__device__ __forceinline__ void scan_load8(const int32_t* __restrict__ in, int64_t n, int64_t i0, bool vec, int32_t (&v)[SCAN_PER]) {
    if (vec && i0 + SCAN_PER <= n) {
        const int4 a = *reinterpret_cast<const int4*>(in + i0), b = *reinterpret_cast<const int4*>(in + i0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    }
}
static __global__ __launch_bounds__(256) void scan_i32_partial_kernel(const int32_t* __restrict__ in, int64_t n,
                                                               unsigned long long* __restrict__ block_tot, int vec) {
    __shared__ unsigned long long lds[5];
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    int32_t x[SCAN_PER];
    scan_load8(in, n, i0, vec != 0, x);
    unsigned long long v = 0, tot;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) v += (unsigned long long)x[k];
    (void)dev::block_exclusive_scan<unsigned long long, 256>(v, lds, &tot);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(256) void scan_i32_final_kernel(const int32_t* __restrict__ in, int64_t n,
                                                             const unsigned long long* __restrict__ block_off,
                                                             int32_t* __restrict__ out, int32_t* __restrict__ out2, int vec) {
    __shared__ unsigned long long lds[5];
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    int32_t x[SCAN_PER];
    scan_load8(in, n, i0, vec != 0, x);
    unsigned long long v = 0, tot;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) v += (unsigned long long)x[k];
    unsigned long long o = block_off[blockIdx.x] + dev::block_exclusive_scan<unsigned long long, 256>(v, lds, &tot);
    int32_t y[SCAN_PER];
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        y[k] = (int32_t)o;
        o += (unsigned long long)x[k];
    }
    // (in-place scans: every value of this thread was read above, and no other thread reads them)
    if (vec && i0 + SCAN_PER <= n) {
        *reinterpret_cast<int4*>(out + i0) = make_int4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<int4*>(out + i0 + 4) = make_int4(y[4], y[5], y[6], y[7]);
        if (out2) {
            *reinterpret_cast<int4*>(out2 + i0) = make_int4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<int4*>(out2 + i0 + 4) = make_int4(y[4], y[5], y[6], y[7]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k)
            if (i0 + k < n) {
                out[i0 + k] = y[k];
                if (out2) out2[i0 + k] = y[k];
            }
    }
    if (i0 <= n - 1 && n - 1 < i0 + SCAN_PER) out[n] = (int32_t)o;  // (o = the prefix past this thread's last value: the values past n are 0)
}


// Exclusive scan of n int32 counts into out[0..n] (out[n] = total); out2 (optional) receives a second
// copy of out[0..n-1] (fill cursors).  block_tot needs (n+255)/256 + 1 entries (callers size it so; (n + 2047) / 2048 + 1 are used);
// the grand total is left in block_tot[(n+255)/256] — the place callers read it from — as well.
static inline int32_t exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* out2,
                                         unsigned long long* block_tot, hipStream_t s) {
    if (n <= 0) return GPK_OK;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK, nb_old = (n + 255) / 256;
    const int vec = ((uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0 && (!out2 || (uintptr_t)out2 % 16 == 0)) ? 1 : 0;
    GPK_LAUNCH("gpk_scan_partial", scan_i32_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, n, block_tot, vec);
    GPK_LAUNCH("gpk_scan_totals", scan_block_totals_kernel, dim3(1), dim3(1024), 0, s, block_tot, nb, block_tot + nb_old,
               (unsigned long long*)nullptr);
    GPK_LAUNCH("gpk_scan_final", scan_i32_final_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, n, block_tot, out, out2, vec);
    return GPK_OK;
}

}  // namespace gpk
