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

static __global__ __launch_bounds__(256) void scan_i32_partial_kernel(const int32_t* __restrict__ in, int64_t n,
                                                               unsigned long long* __restrict__ block_tot) {
    __shared__ unsigned long long lds[5];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = i < n ? (unsigned long long)in[i] : 0ull, tot;
    (void)dev::block_exclusive_scan<unsigned long long, 256>(v, lds, &tot);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(256) void scan_i32_final_kernel(const int32_t* __restrict__ in, int64_t n,
                                                             const unsigned long long* __restrict__ block_off,
                                                             int32_t* __restrict__ out, int32_t* __restrict__ out2) {
    __shared__ unsigned long long lds[5];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = i < n ? (unsigned long long)in[i] : 0ull, tot;
    const unsigned long long ex = dev::block_exclusive_scan<unsigned long long, 256>(v, lds, &tot);
    const unsigned long long o = block_off[blockIdx.x] + ex;
    if (i < n) {
        out[i] = (int32_t)o;
        if (out2) out2[i] = (int32_t)o;
    }
    if (i == n - 1) {
        out[n] = (int32_t)(o + v);
    }
}


// Exclusive scan of n int32 counts into out[0..n] (out[n] = total); out2 (optional) receives a second
// copy of out[0..n-1] (fill cursors).  block_tot needs (n+255)/256 + 1 entries; the grand total is
// left in block_tot[(n+255)/256].
static inline int32_t exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* out2,
                                         unsigned long long* block_tot, hipStream_t s) {
    const int64_t nb = (n + 255) / 256;
    if (n <= 0) return GPK_OK;
    GPK_LAUNCH("gpk_scan_partial", scan_i32_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, n, block_tot);
    GPK_LAUNCH("gpk_scan_totals", scan_block_totals_kernel, dim3(1), dim3(1024), 0, s, block_tot, nb, block_tot + nb,
               (unsigned long long*)nullptr);
    GPK_LAUNCH("gpk_scan_final", scan_i32_final_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, n, block_tot, out, out2);
    return GPK_OK;
}

}  // namespace gpk
