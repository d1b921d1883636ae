// gpk_take.hip — join assembly on the GPU: what the reference does with two u64 index Series
// and polars `inner_join` / `left_join` (geopolars/src/spatial_index.rs:145-203) is, once the (l, r) pairs exist, a
// pure gather of attribute columns by row index.  Three entry points:
//   gpk_join_indices  : sorted (l, r) pairs (+ per-left-row counts) -> i64 row indices of the inner / left join
//                       (left join: unmatched left rows appear once with r = -1, rows stay sorted by l);
//   gpk_take_fixed    : fixed-width column (1/2/4/8/16-byte values) + validity bitmap gathered by i64 indices;
//   gpk_take_binary   : Arrow Binary/Utf8 column (i32 offsets + bytes): sizes -> scan -> byte copy.
// Index -1 (and any index outside [0, n)) yields a null.  HBM-bound: a gather of 4-8 byte values moves a full cache
// line per row when the indices are random; the join's l side is sorted, so its lines are shared by neighbours.
#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

// ---- staging: inputs / outputs in the caller's memory space ------------------------------------------------
struct Stage {  // device views of caller buffers; host buffers are copied through the thread's workspace
    hipStream_t s;
    int32_t space;
    template <typename T>
    int32_t in(const T* p, size_t count, const T** out) {
        *out = nullptr;
        if (!p || count == 0) return GPK_OK;
        if (space == GPK_MEM_DEVICE) {
            *out = p;
            return GPK_OK;
        }
        T* d = (T*)workspace().take(sizeof(T) * count);
        GPK_HIP(hipMemcpyAsync(d, p, sizeof(T) * count, hipMemcpyHostToDevice, s));
        *out = d;
        return GPK_OK;
    }
    template <typename T>
    T* out(T* p, size_t count) {
        if (!p) return nullptr;
        return space == GPK_MEM_DEVICE ? p : (T*)workspace().take(sizeof(T) * (count ? count : 1));
    }
    template <typename T>
    int32_t back(T* user, const T* dev, size_t count) {
        if (!user || space == GPK_MEM_DEVICE || count == 0) return GPK_OK;
        return copy_out(user, space, dev, sizeof(T) * count, s);
    }
};
static inline size_t staged(int32_t space, size_t bytes) { return space == GPK_MEM_DEVICE ? 0 : align256(bytes); }

// ---- join indices ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void left_rows_kernel(const uint32_t* __restrict__ counts, int64_t n, int32_t* __restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = counts[i] ? (int32_t)counts[i] : 1;  // an unmatched left row still produces one output row
}
__global__ __launch_bounds__(256) void inner_indices_kernel(const uint2* __restrict__ pairs, int64_t n_pairs, uint32_t left_base,
                                                             long long* __restrict__ out_l, long long* __restrict__ out_r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const uint2 p = pairs[i];
    out_l[i] = (long long)p.x - (long long)left_base;
    out_r[i] = (long long)p.y;
}
// one lane per left row: its pairs start at (row offset - unmatched rows before it) in the sorted pair list
__global__ __launch_bounds__(256) void left_indices_kernel(const uint32_t* __restrict__ counts, const int32_t* __restrict__ row_off,
                                                            const int32_t* __restrict__ pair_off, const uint2* __restrict__ pairs, int64_t n,
                                                            long long* __restrict__ out_l, long long* __restrict__ out_r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = row_off[i];
    const uint32_t c = counts[i];
    if (c == 0) {
        out_l[o] = i;
        out_r[o] = -1;
        return;
    }
    const int64_t p0 = pair_off[i];
    for (uint32_t k = 0; k < c; ++k) {
        out_l[o + k] = i;
        out_r[o + k] = (long long)pairs[p0 + k].y;
    }
}
__global__ __launch_bounds__(256) void u32_to_i32_kernel(const uint32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

// ---- gathers -----------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void take_fixed_kernel(const T* __restrict__ values, const uint8_t* __restrict__ validity, int64_t n_values,
                                                          const long long* __restrict__ idx, int64_t n_idx, T* __restrict__ out,
                                                          uint8_t* __restrict__ out_valid_bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_idx) return;
    const long long j = idx[i];
    const bool ok = j >= 0 && j < n_values && dev::valid_row(validity, j);
    T v;
    memset(&v, 0, sizeof v);
    if (ok && values) v = values[j];
    if (out) out[i] = v;
    if (out_valid_bytes) out_valid_bytes[i] = ok ? 1 : 0;
}
// one bit per row: Arrow booleans, and the data bitmap of any column
__global__ __launch_bounds__(256) void take_bits_kernel(const uint8_t* __restrict__ bits, const uint8_t* __restrict__ validity, int64_t n_values,
                                                         const long long* __restrict__ idx, int64_t n_idx, uint8_t* __restrict__ out_bytes,
                                                         uint8_t* __restrict__ out_valid_bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_idx) return;
    const long long j = idx[i];
    const bool ok = j >= 0 && j < n_values && dev::valid_row(validity, j);
    if (out_bytes) out_bytes[i] = (ok && bits) ? ((bits[j >> 3] >> (j & 7)) & 1) : 0;
    if (out_valid_bytes) out_valid_bytes[i] = ok ? 1 : 0;
}
// byte-per-row flags -> Arrow bitmap (LSB first): one lane per output byte
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ flags, int64_t n, uint8_t* __restrict__ bitmap) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (n + 7) / 8) return;
    uint32_t v = 0;
    for (int k = 0; k < 8; ++k) {
        const int64_t i = 8 * b + k;
        if (i < n && flags[i]) v |= 1u << k;
    }
    bitmap[b] = (uint8_t)v;
}
__global__ __launch_bounds__(256) void take_binary_sizes_kernel(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ validity,
                                                                 int64_t n_values, const long long* __restrict__ idx, int64_t n_idx,
                                                                 int32_t* __restrict__ sizes, uint8_t* __restrict__ out_valid_bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_idx) return;
    const long long j = idx[i];
    const bool ok = j >= 0 && j < n_values && dev::valid_row(validity, j);
    sizes[i] = ok ? offsets[j + 1] - offsets[j] : 0;
    if (out_valid_bytes) out_valid_bytes[i] = ok ? 1 : 0;
}
// TAKE_GS lanes per output row copy its bytes (rows are short strings / WKB records)
constexpr int TAKE_GS = 8;
__global__ __launch_bounds__(256) void take_binary_copy_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ offsets,
                                                                const long long* __restrict__ idx, int64_t n_idx,
                                                                const int32_t* __restrict__ out_off, uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & (TAKE_GS - 1);
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / TAKE_GS;
    if (i >= n_idx) return;
    const int32_t o0 = out_off[i], len = out_off[i + 1] - o0;
    if (len <= 0) return;
    const uint8_t* src = values + offsets[idx[i]];
    uint8_t* dst = out + o0;
    // 4-byte words where both sides allow it (unaligned dword accesses are legal on gfx9+ under HSA), bytes for the tail
    const int words = len >> 2;
    for (int w = lane; w < words; w += TAKE_GS) {
        uint32_t v;
        __builtin_memcpy(&v, src + 4 * w, 4);
        __builtin_memcpy(dst + 4 * w, &v, 4);
    }
    for (int b = 4 * words + lane; b < len; b += TAKE_GS) dst[b] = src[b];
}

static inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)); }

}  // namespace gpk

using namespace gpk;

extern "C" {

int32_t gpk_join_indices(const uint32_t* counts, const uint32_t* pairs, int64_t n_left, int64_t n_pairs, uint32_t left_row_base,
                         int32_t join_type, int64_t* out_l, int64_t* out_r, int64_t capacity, int64_t* n_rows, int32_t space,
                         void* stream) {
    if (!n_rows || (n_pairs > 0 && !pairs) || n_left < 0 || n_pairs < 0) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL / negative argument");
    if (join_type != GPK_JOIN_INNER && join_type != GPK_JOIN_LEFT)
        return fail(GPK_ERR_INVALID_ARGUMENT, "join type %d: only inner and left joins exist (spatial_index.rs:200-202)", join_type);
    if (join_type == GPK_JOIN_LEFT && n_left > 0 && !counts) return fail(GPK_ERR_INVALID_ARGUMENT, "a left join needs the per-row hit counts");
    if (capacity < 0 || (capacity > 0 && (!out_l || !out_r))) return fail(GPK_ERR_INVALID_ARGUMENT, "capacity without output buffers");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    Stage st{s, space};
    *n_rows = 0;
    if (join_type == GPK_JOIN_INNER) {
        *n_rows = n_pairs;
        if (capacity == 0 || n_pairs == 0) return GPK_OK;
        if (n_pairs > capacity) return fail(GPK_ERR_CAPACITY, "join_indices: %lld rows but capacity %lld", (long long)n_pairs, (long long)capacity);
        GPK_TRY(workspace().begin(staged(space, 8 * (size_t)n_pairs) * 3 + 1024));
        const uint2* p;
        GPK_TRY(st.in((const uint2*)pairs, (size_t)n_pairs, &p));
        long long *dl = st.out((long long*)out_l, (size_t)n_pairs), *dr = st.out((long long*)out_r, (size_t)n_pairs);
        GPK_LAUNCH("gpk_join_inner_indices", inner_indices_kernel, grid1(n_pairs), dim3(256), 0, s, p, n_pairs, left_row_base, dl, dr);
        GPK_TRY(st.back((long long*)out_l, dl, (size_t)n_pairs));
        GPK_TRY(st.back((long long*)out_r, dr, (size_t)n_pairs));
        GPK_HIP(hipStreamSynchronize(s));
        return GPK_OK;
    }
    if (n_left == 0) return GPK_OK;
    if (n_left >= 0x7FFFFFFFLL || n_pairs + n_left >= 0x7FFFFFFFLL)
        return fail(GPK_ERR_CAPACITY, "join_indices: more than 2^31 rows; assemble per row shard");
    const int64_t nb = (n_left + 255) / 256;
    const size_t i32n = align256(sizeof(int32_t) * (size_t)(n_left + 1));
    GPK_TRY(workspace().begin(4 * i32n + align256(8 * (size_t)(nb + 2)) + staged(space, 4 * (size_t)n_left) + staged(space, 8 * (size_t)n_pairs) +
                              2 * staged(space, 8 * (size_t)capacity) + 1024));
    const uint32_t* c;
    const uint2* p;
    GPK_TRY(st.in(counts, (size_t)n_left, &c));
    GPK_TRY(st.in((const uint2*)pairs, (size_t)n_pairs, &p));
    int32_t* rows = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_left + 1));
    int32_t* row_off = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_left + 1));
    int32_t* ci32 = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_left + 1));
    int32_t* pair_off = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_left + 1));
    unsigned long long* btot = (unsigned long long*)workspace().take(8 * (size_t)(nb + 2));
    GPK_LAUNCH("gpk_join_left_rows", left_rows_kernel, grid1(n_left), dim3(256), 0, s, c, n_left, rows);
    GPK_TRY(exclusive_scan_i32(rows, n_left, row_off, nullptr, btot, s));
    GPK_LAUNCH("gpk_join_counts_i32", u32_to_i32_kernel, grid1(n_left), dim3(256), 0, s, c, ci32, n_left);
    GPK_TRY(exclusive_scan_i32(ci32, n_left, pair_off, nullptr, btot, s));
    int32_t total = 0, total_pairs = 0;
    GPK_HIP(hipMemcpyAsync(&total, row_off + n_left, sizeof total, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipMemcpyAsync(&total_pairs, pair_off + n_left, sizeof total_pairs, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    if ((int64_t)total_pairs != n_pairs)
        return fail(GPK_ERR_INVALID_ARGUMENT, "join_indices: the counts sum to %d but %lld pairs were given", total_pairs, (long long)n_pairs);
    *n_rows = total;
    if (capacity == 0) return GPK_OK;
    if ((int64_t)total > capacity) return fail(GPK_ERR_CAPACITY, "join_indices: %d rows but capacity %lld", total, (long long)capacity);
    long long *dl = st.out((long long*)out_l, (size_t)total), *dr = st.out((long long*)out_r, (size_t)total);
    GPK_LAUNCH("gpk_join_left_indices", left_indices_kernel, grid1(n_left), dim3(256), 0, s, c, (const int32_t*)row_off,
               (const int32_t*)pair_off, p, n_left, dl, dr);
    GPK_TRY(st.back((long long*)out_l, dl, (size_t)total));
    GPK_TRY(st.back((long long*)out_r, dr, (size_t)total));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

int32_t gpk_take_fixed(const void* values, int32_t elem_bits, const uint8_t* validity, int64_t n_values, const int64_t* idx, int64_t n_idx,
                       void* out_values, uint8_t* out_validity, int32_t space, void* stream) {
    if (n_idx < 0 || n_values < 0 || (n_idx > 0 && !idx)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL / negative argument");
    if (elem_bits != 1 && elem_bits != 8 && elem_bits != 16 && elem_bits != 32 && elem_bits != 64 && elem_bits != 128)
        return fail(GPK_ERR_INVALID_ARGUMENT, "take_fixed: %d-bit values (1, 8, 16, 32, 64 or 128 supported)", elem_bits);
    GPK_TRY(require_device());
    if (n_idx == 0) return GPK_OK;
    hipStream_t s = (hipStream_t)stream;
    Stage st{s, space};
    const size_t in_bytes = elem_bits == 1 ? (size_t)(n_values + 7) / 8 : (size_t)n_values * (size_t)(elem_bits / 8);
    const size_t out_bytes = elem_bits == 1 ? (size_t)(n_idx + 7) / 8 : (size_t)n_idx * (size_t)(elem_bits / 8);
    const size_t vbytes = (size_t)(n_values + 7) / 8, obytes = (size_t)(n_idx + 7) / 8;
    GPK_TRY(workspace().begin(staged(space, in_bytes) + staged(space, vbytes) + staged(space, 8 * (size_t)n_idx) + staged(space, out_bytes) +
                              staged(space, obytes) + 3 * align256((size_t)n_idx) + 2 * align256(obytes) + 1024));
    const uint8_t *v, *val;
    const long long* ix;
    GPK_TRY(st.in((const uint8_t*)values, in_bytes, &v));
    GPK_TRY(st.in(validity, vbytes, &val));
    GPK_TRY(st.in((const long long*)idx, (size_t)n_idx, &ix));
    uint8_t* ov = st.out((uint8_t*)out_values, out_bytes);
    uint8_t* ob = st.out(out_validity, obytes);
    uint8_t* flags = out_validity ? (uint8_t*)workspace().take((size_t)n_idx) : nullptr;
    const dim3 g = grid1(n_idx);
    switch (elem_bits) {
    case 1: {
        uint8_t* bytes = out_values ? (uint8_t*)workspace().take((size_t)n_idx) : nullptr;
        GPK_LAUNCH("gpk_take_bits", take_bits_kernel, g, dim3(256), 0, s, v, val, n_values, ix, n_idx, bytes, flags);
        if (bytes) GPK_LAUNCH("gpk_take_pack", pack_bits_kernel, grid1((n_idx + 7) / 8), dim3(256), 0, s, (const uint8_t*)bytes, n_idx, ov);
        break;
    }
    case 8: GPK_LAUNCH("gpk_take_fixed8", take_fixed_kernel<uint8_t>, g, dim3(256), 0, s, v, val, n_values, ix, n_idx, ov, flags); break;
    case 16:
        GPK_LAUNCH("gpk_take_fixed16", take_fixed_kernel<uint16_t>, g, dim3(256), 0, s, (const uint16_t*)v, val, n_values, ix, n_idx, (uint16_t*)ov, flags);
        break;
    case 32:
        GPK_LAUNCH("gpk_take_fixed32", take_fixed_kernel<uint32_t>, g, dim3(256), 0, s, (const uint32_t*)v, val, n_values, ix, n_idx, (uint32_t*)ov, flags);
        break;
    case 64:
        GPK_LAUNCH("gpk_take_fixed64", take_fixed_kernel<unsigned long long>, g, dim3(256), 0, s, (const unsigned long long*)v, val, n_values, ix,
                   n_idx, (unsigned long long*)ov, flags);
        break;
    default:
        GPK_LAUNCH("gpk_take_fixed128", take_fixed_kernel<uint4>, g, dim3(256), 0, s, (const uint4*)v, val, n_values, ix, n_idx, (uint4*)ov, flags);
    }
    if (flags) GPK_LAUNCH("gpk_take_pack", pack_bits_kernel, grid1((n_idx + 7) / 8), dim3(256), 0, s, (const uint8_t*)flags, n_idx, ob);
    GPK_TRY(st.back((uint8_t*)out_values, ov, out_bytes));
    GPK_TRY(st.back(out_validity, ob, obytes));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

int32_t gpk_take_binary(const uint8_t* values, const int32_t* offsets, const uint8_t* validity, int64_t n_values, const int64_t* idx,
                        int64_t n_idx, int32_t* out_offsets, uint8_t* out_values, int64_t capacity, int64_t* n_bytes, uint8_t* out_validity,
                        int32_t space, void* stream) {
    if (!n_bytes || n_idx < 0 || n_values < 0 || (n_idx > 0 && !idx) || (n_values > 0 && !offsets))
        return fail(GPK_ERR_INVALID_ARGUMENT, "NULL / negative argument");
    if (capacity < 0 || (capacity > 0 && !out_values)) return fail(GPK_ERR_INVALID_ARGUMENT, "capacity without out_values");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    Stage st{s, space};
    *n_bytes = 0;
    if (n_idx >= 0x7FFFFFFFLL) return fail(GPK_ERR_CAPACITY, "take_binary: more than 2^31 rows");
    const size_t vbytes = (size_t)(n_values + 7) / 8, obytes = (size_t)(n_idx + 7) / 8;
    int64_t in_bytes = 0;
    if (n_values > 0) {  // the column's byte length: offsets[n_values] in the caller's space
        int32_t last = 0;
        if (space == GPK_MEM_DEVICE) {
            GPK_HIP(hipMemcpyAsync(&last, offsets + n_values, sizeof last, hipMemcpyDeviceToHost, s));
            GPK_HIP(hipStreamSynchronize(s));
        } else {
            last = offsets[n_values];
        }
        in_bytes = last;
    }
    const int64_t nb = (n_idx + 255) / 256;
    GPK_TRY(workspace().begin(staged(space, (size_t)in_bytes) + staged(space, 4 * (size_t)(n_values + 1)) + staged(space, vbytes) +
                              staged(space, 8 * (size_t)n_idx) + staged(space, (size_t)capacity) + staged(space, obytes) +
                              2 * align256(4 * (size_t)(n_idx + 1)) + align256(8 * (size_t)(nb + 2)) + align256((size_t)n_idx) + align256(obytes) + 2048));
    const uint8_t *v, *val;
    const int32_t* off;
    const long long* ix;
    GPK_TRY(st.in(values, (size_t)in_bytes, &v));
    GPK_TRY(st.in(offsets, (size_t)(n_values + 1), &off));
    GPK_TRY(st.in(validity, vbytes, &val));
    GPK_TRY(st.in((const long long*)idx, (size_t)n_idx, &ix));
    int32_t* sizes = (int32_t*)workspace().take(4 * (size_t)(n_idx + 1));
    int32_t* ooff = (space == GPK_MEM_DEVICE && out_offsets) ? out_offsets : (int32_t*)workspace().take(4 * (size_t)(n_idx + 1));
    unsigned long long* btot = (unsigned long long*)workspace().take(8 * (size_t)(nb + 2));
    uint8_t* flags = out_validity ? (uint8_t*)workspace().take((size_t)(n_idx ? n_idx : 1)) : nullptr;
    uint8_t* ob = st.out(out_validity, obytes);
    unsigned long long total = 0;
    if (n_idx > 0) {
        GPK_LAUNCH("gpk_take_binary_sizes", take_binary_sizes_kernel, grid1(n_idx), dim3(256), 0, s, off, val, n_values, ix, n_idx, sizes, flags);
        GPK_TRY(exclusive_scan_i32(sizes, n_idx, ooff, nullptr, btot, s));
        GPK_HIP(hipMemcpyAsync(&total, btot + nb, sizeof total, hipMemcpyDeviceToHost, s));  // the scan keeps a 64-bit grand total
        GPK_HIP(hipStreamSynchronize(s));
    } else {
        GPK_HIP(hipMemsetAsync(ooff, 0, sizeof(int32_t), s));
    }
    *n_bytes = (int64_t)total;
    if (total > 0x7FFFFFFFull) return fail(GPK_ERR_CAPACITY, "take_binary: %llu bytes do not fit i32 offsets; gather row slices", total);
    if (flags && n_idx > 0) GPK_LAUNCH("gpk_take_pack", pack_bits_kernel, grid1((n_idx + 7) / 8), dim3(256), 0, s, (const uint8_t*)flags, n_idx, ob);
    if (out_offsets && space != GPK_MEM_DEVICE) GPK_TRY(copy_out(out_offsets, space, ooff, 4 * (size_t)(n_idx + 1), s));
    GPK_TRY(st.back(out_validity, ob, obytes));
    if (!out_values) {
        GPK_HIP(hipStreamSynchronize(s));
        return GPK_OK;  // size query
    }
    if ((int64_t)total > capacity) return fail(GPK_ERR_CAPACITY, "take_binary: %llu bytes but capacity %lld", total, (long long)capacity);
    uint8_t* ov = st.out(out_values, (size_t)total);
    if (total > 0)
        GPK_LAUNCH("gpk_take_binary_copy", take_binary_copy_kernel, grid1(n_idx * TAKE_GS), dim3(256), 0, s, v, off, ix, n_idx, (const int32_t*)ooff, ov);
    GPK_TRY(st.back(out_values, ov, (size_t)total));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

}  // extern "C"
