// gpk_unary.hip — streaming (HBM-bound) unary operators of the GeoSeries surface:
//   area / signed_area   geoseries.rs:14-16,188-190   (geo 0.27 area.rs semantics)
//   centroid             geoseries.rs:18-21,192-194   (geo 0.27 centroid.rs semantics)
//   bounds / envelope    geoseries.rs:28-33,200-202   (geo 0.27 bounding_rect.rs semantics)
//   euclidean_length     geoseries.rs:35-41
//   affine_transform     geoseries.rs:11-12,184-186   (AffineTransform::apply, no FMA)
//
// Layout: every reduction is two stages.  Stage 1 streams the coordinate buffer once (16 B per
// vertex, double2 loads) with G lanes cooperating on one coordinate sequence (ring / linestring);
// sequences are bucketed by length once per array (G = 2 / 8 / 16 lanes, whole work-group for the
// longest) so that a wave walks sequences of similar length.  Per-sequence partials land in a
// small stats array (8 B x K per sequence).  Stage 2 is one thread per geometry folding its rings
// with the polygon / multipolygon rules.  Reduction order is fixed (DPP pairing tree), so results
// are bit-reproducible run to run.
#include <mutex>

#include "gpk_device.h"
#include "gpk_index.h"
#include "gpk_ringstream.h"
#include "gpk_scan.h"

namespace gpk {

enum : int {
    ST_AREA2 = 0,  // twice the signed ring area (shifted by the first coordinate)
    ST_ACX = 1,    // the ring's centroid x (sum (ex+sx)*cross of the shifted ring / (6 area) + the shift; a zero-area ring: unused)
    ST_ACY = 2,
    ST_LEN = 3,  // sum of segment lengths
    ST_LMX = 4,  // sum of midpoint * length
    ST_LMY = 5,
    ST_MINX = 6,
    ST_MINY = 7,
    ST_MAXX = 8,
    ST_MAXY = 9,
    ST_SUMX = 10,  // plain coordinate sums (MULTIPOINT centroid)
    ST_SUMY = 11,
    ST_COUNT = 12
};
// The four bound statistics of a sequence are ONE 32-byte record (array of structures inside the structure of arrays: rows
// ST_MINX .. ST_MAXY of the table, shifted by two doubles when n_seq is odd so that the records are 32-byte aligned — the table's
// base is 256-byte aligned, 6 * n_seq doubles precede them): one store per ring in the ring pass and one load in the per-geometry
// pass instead of four 8-byte ones n_seq apart (on a ragged column those were four partial-sector writes per ring).  The shift
// reaches two doubles into the ST_SUMX rows: no pass asks for M_BBOX and M_SUM together (seq_store).
__device__ __forceinline__ double4* bbox_records(double* stats, int64_t n_seq) {
    return reinterpret_cast<double4*>(stats + ST_MINX * n_seq + ((n_seq & 1) ? 2 : 0));
}
__device__ __forceinline__ const double4* bbox_records(const double* stats, int64_t n_seq) {
    return reinterpret_cast<const double4*>(stats + ST_MINX * n_seq + ((n_seq & 1) ? 2 : 0));
}
constexpr unsigned M_AREA = 1u << 0, M_CENT = 1u << 1, M_LEN = 1u << 2, M_BBOX = 1u << 3, M_SUM = 1u << 4;
// M_CENT = area-weighted accumulators only; M_LENC = length-weighted ones (a hypot per edge); M_DEGEN restricts a
// pass to sequences whose ring area came out zero (the only rings whose centroid needs the length partials)
constexpr unsigned M_LENC = 1u << 5, M_DEGEN = 1u << 6;

// group reductions: DPP row operations (gpk_device.h) — G <= 16 everywhere in this file
using dev::group_max;
using dev::group_min;
using dev::group_sum;

// Stage 1: per-sequence partials.  stats is [ST_COUNT][n_seq] (SoA so stage 2 reads are coalesced).
constexpr int SEQ_LONG = 512;  // sequences longer than this are reduced by a whole work-group (seq_stats_long_kernel)

// per-edge / per-vertex accumulation shared by the group kernel and the long-sequence kernel
struct SeqPartial {
    double a2 = 0, acx = 0, acy = 0, len = 0, lmx = 0, lmy = 0, sx = 0, sy = 0;
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
};
template <unsigned MASK>
__device__ __forceinline__ void seq_accumulate(SeqPartial& a, const double2* __restrict__ xy, int i, int c1, double2 first,
                                               bool closed_ring) {
    const double2 p = xy[i];
    if (MASK & M_BBOX) {
        a.mnx = p.x < a.mnx ? p.x : a.mnx;
        a.mny = p.y < a.mny ? p.y : a.mny;
        a.mxx = p.x > a.mxx ? p.x : a.mxx;
        a.mxy = p.y > a.mxy ? p.y : a.mxy;
    }
    if (MASK & M_SUM) {
        a.sx += p.x;
        a.sy += p.y;
    }
    if ((MASK & (M_AREA | M_CENT | M_LEN | M_LENC)) && i + 1 < c1) {
        const double2 q = xy[i + 1];
        if ((MASK & (M_AREA | M_CENT)) && closed_ring) {
            const double sx = p.x - first.x, sy = p.y - first.y;
            const double ex = q.x - first.x, ey = q.y - first.y;
            const double cr = sx * ey - sy * ex;
            a.a2 += cr;
            if (MASK & M_CENT) {
                a.acx += (ex + sx) * cr;
                a.acy += (ey + sy) * cr;
            }
        }
        if (MASK & (M_LEN | M_LENC)) {
            const double l = hypot(q.x - p.x, q.y - p.y);
            a.len += l;
            if (MASK & M_LENC) {
                a.lmx += (p.x + q.x) / 2.0 * l;
                a.lmy += (p.y + q.y) / 2.0 * l;
            }
        }
    }
}
// the same accumulation with the vertex and its successor already in registers
template <unsigned MASK>
__device__ __forceinline__ void seq_accumulate_pq(SeqPartial& a, double2 p, double2 q, bool has_q, double2 first, bool closed_ring) {
    if (MASK & M_BBOX) {
        a.mnx = p.x < a.mnx ? p.x : a.mnx;
        a.mny = p.y < a.mny ? p.y : a.mny;
        a.mxx = p.x > a.mxx ? p.x : a.mxx;
        a.mxy = p.y > a.mxy ? p.y : a.mxy;
    }
    if (MASK & M_SUM) {
        a.sx += p.x;
        a.sy += p.y;
    }
    if ((MASK & (M_AREA | M_CENT | M_LEN | M_LENC)) && has_q) {
        if ((MASK & (M_AREA | M_CENT)) && closed_ring) {
            const double sx = p.x - first.x, sy = p.y - first.y;
            const double ex = q.x - first.x, ey = q.y - first.y;
            const double cr = sx * ey - sy * ex;
            a.a2 += cr;
            if (MASK & M_CENT) {
                a.acx += (ex + sx) * cr;
                a.acy += (ey + sy) * cr;
            }
        }
        if (MASK & (M_LEN | M_LENC)) {
            const double l = hypot(q.x - p.x, q.y - p.y);
            a.len += l;
            if (MASK & M_LENC) {
                a.lmx += (p.x + q.x) / 2.0 * l;
                a.lmy += (p.y + q.y) / 2.0 * l;
            }
        }
    }
}
template <unsigned MASK>
__device__ __forceinline__ void seq_store(const SeqPartial& a, double* __restrict__ stats, int64_t n_seq, int64_t s) {
    if (MASK & (M_AREA | M_CENT)) stats[ST_AREA2 * n_seq + s] = a.a2;
    if (MASK & M_CENT) {
        stats[ST_ACX * n_seq + s] = a.acx;
        stats[ST_ACY * n_seq + s] = a.acy;
    }
    if (MASK & M_LENC) {
        stats[ST_LMX * n_seq + s] = a.lmx;
        stats[ST_LMY * n_seq + s] = a.lmy;
    }
    if (MASK & (M_LEN | M_LENC)) stats[ST_LEN * n_seq + s] = a.len;
    static_assert(!((MASK & M_BBOX) && (MASK & M_SUM)), "bbox_records reaches into the ST_SUMX rows");
    if (MASK & M_BBOX) bbox_records(stats, n_seq)[s] = make_double4(a.mnx, a.mny, a.mxx, a.mxy);
    if (MASK & M_SUM) {
        stats[ST_SUMX * n_seq + s] = a.sx;
        stats[ST_SUMY * n_seq + s] = a.sy;
    }
}

// Columns whose sequences ARE their geometries (a LINESTRING column; a POLYGON column of single-ring polygons, the usual
// shape of building / parcel data) skip the per-geometry combine: stage 1 writes the operator's result itself.
enum { FIN_NONE = 0, FIN_AREA, FIN_SIGNED_AREA, FIN_BOUNDS, FIN_LENGTH, FIN_CENT_POLY, FIN_CENT_DEGEN, FIN_CENT_LINE, FIN_CENT_MPOINT };
struct FinalOut {
    double* out;              // result column (bounds: 4 doubles per row, centroid: 2)
    const uint8_t* validity;  // of the geometries (= sequences)
    uint8_t* out_valid;       // centroid: "has a centroid" flags (may be nullptr)
    int32_t* degen_flag;      // centroid: set by the area-weighted pass when some ring has zero area; the degenerate-only pass
                              // leaves at once while it is 0 (nearly every column: the pass then costs a launch, not a walk)
};
// centroid of a row that contributes as a linestring (Centroid::add_line_string, centroid.rs): length-weighted
// midpoints, or the start point when every segment is degenerate; same arithmetic as wc_add_linestring + the final divide
__device__ __forceinline__ double2 line_centroid(const SeqPartial& a, double2 first, int n) {
    if (a.len > 0.0) return make_double2(a.lmx / a.len, a.lmy / a.len);
    const double k = n == 1 ? 1.0 : (double)(n - 1);
    return make_double2(first.x * k / k, first.y * k / k);
}
template <unsigned MASK, int FIN>
__device__ __forceinline__ void seq_store_any(const SeqPartial& a, double* __restrict__ stats, int64_t n_seq, int64_t s, int c0, int c1,
                                              const double2* __restrict__ xy, const FinalOut& f) {
    if ((MASK & M_CENT) && !(MASK & M_DEGEN) && f.degen_flag && c1 > c0 && a.a2 / 2.0 == 0.0) *f.degen_flag = 1;  // (benign race: every writer stores 1)
    if (FIN == FIN_NONE) {
        if ((MASK & M_CENT) && !(MASK & M_DEGEN)) {
            // the ring's centroid, not its raw moments: the per-geometry pass then needs no first coordinate of the ring (a scattered
            // 64-byte line per ring, half of centroid_combine_kernel's time on a column of multipolygons); wc_add_ring's arithmetic
            SeqPartial b = a;
            const double area = a.a2 / 2.0;
            if (c1 > c0 && area != 0.0) {
                const double2 sh = xy[c0];
                b.acx = a.acx / (6.0 * area) + sh.x;
                b.acy = a.acy / (6.0 * area) + sh.y;
            }
            seq_store<MASK>(b, stats, n_seq, s);
            return;
        }
        seq_store<MASK>(a, stats, n_seq, s);
        return;
    }
    const bool valid = dev::valid_row(f.validity, s);
    const int n = c1 - c0;
    if (FIN == FIN_AREA) f.out[s] = valid ? fabs(fabs(a.a2 / 2.0)) : NAN;  // area_combine_kernel for one ring
    if (FIN == FIN_SIGNED_AREA) {
        const double h = a.a2 / 2.0;
        f.out[s] = valid ? (h < 0.0 ? -fabs(h) : fabs(h)) : NAN;
    }
    if (FIN == FIN_LENGTH) f.out[s] = valid ? 0.0 + a.len : NAN;
    if (FIN == FIN_BOUNDS)
        reinterpret_cast<double4*>(f.out)[s] = (valid && n > 0) ? make_double4(a.mnx, a.mny, a.mxx, a.mxy) : make_double4(NAN, NAN, NAN, NAN);
    if (FIN == FIN_CENT_POLY || FIN == FIN_CENT_DEGEN || FIN == FIN_CENT_LINE || FIN == FIN_CENT_MPOINT) {
        double2* out2 = reinterpret_cast<double2*>(f.out);
        if (FIN == FIN_CENT_POLY) stats[ST_AREA2 * n_seq + s] = a.a2;  // the degenerate-only second pass filters on it
        if (!valid || n == 0) {
            if (FIN != FIN_CENT_DEGEN) {  // (the second pass leaves what the first one wrote)
                out2[s] = make_double2(NAN, NAN);
                if (f.out_valid) f.out_valid[s] = 0;
            }
            return;
        }
        double2 r;
        if (FIN == FIN_CENT_POLY) {
            const double area = a.a2 / 2.0;
            if (area == 0.0) return;  // zero-area ring: the second pass writes its linestring centroid
            const double2 sh = xy[c0];
            const double w = fabs(area);
            const double cx = a.acx / (6.0 * area) + sh.x, cy = a.acy / (6.0 * area) + sh.y;
            r = make_double2(cx * w / w, cy * w / w);  // wc_add + the final divide of centroid_combine_kernel
        } else if (FIN == FIN_CENT_MPOINT) {
            r = make_double2(a.sx / (double)n, a.sy / (double)n);
        } else {
            r = line_centroid(a, xy[c0], n);
        }
        out2[s] = r;
        if (f.out_valid) f.out_valid[s] = 1;
    }
}

// ---- size classes ---------------------------------------------------------------------------------------------
// Lanes per sequence by length: ~8 vertices per lane keeps the reduction steps per vertex low (with 64 lanes on
// 65-vertex rings the reductions outweighed the streaming work 3:1) while a group still reads G consecutive vertices
// per load.  Sequences are bucketed once per array (gpk_seq_classes) so that the lanes of a wave walk sequences of
// similar length whatever the length distribution of the column (power-law data: a few giant rings would otherwise be
// walked by one lane group while the rest of the chip idles).  Order inside a bucket is irrelevant: every sequence is
// reduced independently and deterministically.
constexpr int SEQ_LANES[3] = {2, 8, 16};
constexpr int SEQ_MAXLEN[3] = {16, 128, SEQ_LONG};  // class k takes lengths <= SEQ_MAXLEN[k] (and > SEQ_MAXLEN[k-1])
__device__ __forceinline__ int seq_class_of(int n) { return n <= SEQ_MAXLEN[0] ? 0 : (n <= SEQ_MAXLEN[1] ? 1 : (n <= SEQ_MAXLEN[2] ? 2 : 3)); }

// FILL = false: counts per class; FILL = true: ids appended at the class cursors.  A work-group classifies SEQ_CLS_ROUNDS x 256
// consecutive sequences and touches each class cursor ONCE (one atomic per wave and class put 700k atomics on four addresses for
// C5's 15M rings: 4.6 ms per pass); inside the group the ids keep their order (round, wave, lane).
constexpr int SEQ_CLS_ROUNDS = 8;
template <bool FILL>
__global__ __launch_bounds__(256) void seq_classify_kernel(const int32_t* __restrict__ seq_off, int64_t n_seq, int32_t* __restrict__ cursor,
                                                            int32_t* __restrict__ lists) {
    __shared__ int32_t s_cnt[SEQ_CLS_ROUNDS * 4][4];  // [round * 4 + wave][class]; after the prefix: the slot's first position
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * (256 * SEQ_CLS_ROUNDS);
    int cls[SEQ_CLS_ROUNDS];
#pragma unroll
    for (int r = 0; r < SEQ_CLS_ROUNDS; ++r) {
        const int64_t s = base + r * 256 + threadIdx.x;
        cls[r] = s < n_seq ? seq_class_of(seq_off[s + 1] - seq_off[s]) : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long m = __ballot(cls[r] == k);
            if (lane == 0) s_cnt[r * 4 + wave][k] = __popcll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {  // class k: exclusive prefix over the (round, wave) slots, one atomic for the group's total
        const int k = threadIdx.x;
        int total = 0;
        for (int i = 0; i < SEQ_CLS_ROUNDS * 4; ++i) {
            const int c = s_cnt[i][k];
            s_cnt[i][k] = total;
            total += c;
        }
        const int at = total ? atomicAdd(&cursor[k], total) : 0;
        if (FILL)
            for (int i = 0; i < SEQ_CLS_ROUNDS * 4; ++i) s_cnt[i][k] += at;
    }
    if (!FILL) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SEQ_CLS_ROUNDS; ++r) {
        const int c = cls[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long mk = __ballot(c == k);
            if (c == k) lists[s_cnt[r * 4 + wave][k] + __popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)(base + r * 256 + threadIdx.x);
        }
    }
}

constexpr int SEQ_CHUNK = 8192;  // coordinates of a long sequence reduced by one work-group
__global__ void long_chunk_count_kernel(const int32_t* __restrict__ seq_off, const int32_t* __restrict__ list, int64_t n_list,
                                        int32_t* __restrict__ chunks) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_list) return;
    const int64_t s = list ? (int64_t)list[k] : k;
    const int n = seq_off[s + 1] - seq_off[s];
    chunks[k] = n > 0 ? (n + SEQ_CHUNK - 1) / SEQ_CHUNK : 1;
}

// one work-group per chunk of a long sequence.  A sequence that fits one chunk is stored directly; the chunks of a
// longer one leave 12-double partials in `part` that seq_long_combine_kernel folds in chunk order (deterministic).
template <unsigned MASK, int FIN>
__device__ __forceinline__ void seq_stats_long_body(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                                    const int32_t* __restrict__ list, int64_t n_list,
                                                    const int32_t* __restrict__ chunk_begin, int64_t n_chunks, double* __restrict__ part,
                                                    double* __restrict__ stats, int block, int n_blocks, const FinalOut& fin) {
    __shared__ double red[12][4];
    for (int64_t w = block; w < n_chunks; w += n_blocks) {
        int64_t lo = 0, hi = n_list;  // largest k with chunk_begin[k] <= w (uniform across the work-group)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)chunk_begin[mid] <= w)
                lo = mid;
            else
                hi = mid;
        }
        const int64_t k = lo;
        const int64_t s = list ? (int64_t)list[k] : k;
        if ((MASK & M_DEGEN) && stats[ST_AREA2 * n_seq + s] != 0.0) continue;
        const int c0 = seq_off[s], c1 = seq_off[s + 1];
        const int n_ch = chunk_begin[k + 1] - chunk_begin[k];
        const int j = (int)(w - chunk_begin[k]);
        double2 first = make_double2(0, 0), last = make_double2(0, 0);
        if (c1 > c0) {
            first = xy[c0];
            last = xy[c1 - 1];
        }
        const bool closed_ring = c1 - c0 >= 3 && first.x == last.x && first.y == last.y;
        SeqPartial a;
        const int b0 = c0 + j * SEQ_CHUNK, b1 = b0 + SEQ_CHUNK < c1 ? b0 + SEQ_CHUNK : c1;
        for (int i = b0 + threadIdx.x; i < b1; i += 256) seq_accumulate<MASK>(a, xy, i, c1, first, closed_ring);
        double v[12] = {a.a2, a.acx, a.acy, a.len, a.lmx, a.lmy, a.sx, a.sy, a.mnx, a.mny, a.mxx, a.mxy};
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            if (q < 8)
                v[q] = dev::wave_sum(v[q]);
            else if (q < 10)
                v[q] = dev::wave_min(v[q]);
            else
                v[q] = dev::wave_max(v[q]);
        }
        __syncthreads();  // red[] reuse across iterations
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int q = 0; q < 12; ++q) red[q][threadIdx.x >> 6] = v[q];
        __syncthreads();
        if (threadIdx.x == 0) {
            double t[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                t[q] = red[q][0];
                for (int x = 1; x < 4; ++x) t[q] = q < 8 ? t[q] + red[q][x] : (q < 10 ? fmin(t[q], red[q][x]) : fmax(t[q], red[q][x]));
            }
            if (n_ch == 1) {
                SeqPartial r;
                r.a2 = t[0]; r.acx = t[1]; r.acy = t[2]; r.len = t[3]; r.lmx = t[4]; r.lmy = t[5]; r.sx = t[6]; r.sy = t[7];
                r.mnx = t[8]; r.mny = t[9]; r.mxx = t[10]; r.mxy = t[11];
                seq_store_any<MASK, FIN>(r, stats, n_seq, s, c0, c1, xy, fin);
            } else {
#pragma unroll
                for (int q = 0; q < 12; ++q) part[12 * w + q] = t[q];
            }
        }
    }
}
template <unsigned MASK, int FIN>
__global__ void seq_long_combine_kernel(const int32_t* __restrict__ list, int64_t n_list, const int32_t* __restrict__ chunk_begin,
                                        const double* __restrict__ part, int64_t n_seq, double* __restrict__ stats, FinalOut fin,
                                        const int32_t* __restrict__ seq_off, const double2* __restrict__ xy) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_list) return;
    const int w0 = chunk_begin[k], w1 = chunk_begin[k + 1];
    if (w1 - w0 <= 1) return;  // stored by its only chunk
    const int64_t s = list ? (int64_t)list[k] : k;
    if ((MASK & M_DEGEN) && stats[ST_AREA2 * n_seq + s] != 0.0) return;
    double t[12];
    for (int q = 0; q < 12; ++q) t[q] = part[12 * (int64_t)w0 + q];
    for (int w = w0 + 1; w < w1; ++w)
        for (int q = 0; q < 12; ++q) {
            const double x = part[12 * (int64_t)w + q];
            t[q] = q < 8 ? t[q] + x : (q < 10 ? fmin(t[q], x) : fmax(t[q], x));
        }
    SeqPartial r;
    r.a2 = t[0]; r.acx = t[1]; r.acy = t[2]; r.len = t[3]; r.lmx = t[4]; r.lmy = t[5]; r.sx = t[6]; r.sy = t[7];
    r.mnx = t[8]; r.mny = t[9]; r.mxx = t[10]; r.mxy = t[11];
    seq_store_any<MASK, FIN>(r, stats, n_seq, s, seq_off[s], seq_off[s + 1], xy, fin);
}

template <int G, unsigned MASK, int FIN>
__device__ __forceinline__ void seq_stats_group_body(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                                     double* __restrict__ stats, const int32_t* __restrict__ list, int64_t n_list, int block,
                                                     int n_blocks, const FinalOut& fin) {
    // one size class: the sequences list[0..n_list) (every sequence in order when list == nullptr), G lanes each
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups_per_grid = (int64_t)n_blocks * (256 / G);
    for (int64_t k = (int64_t)block * (256 / G) + threadIdx.x / G; k < n_list; k += groups_per_grid) {
        const int64_t s = list ? (int64_t)list[k] : k;
        if ((MASK & M_DEGEN) && stats[ST_AREA2 * n_seq + s] != 0.0) continue;  // group-uniform
        // Two-lane groups walk sequences of a handful of coordinates: the two offsets and the two end points were four of
        // their ~nine load instructions.  PAIR: each lane loads one offset and they swap (DPP); the first vertex is lane 0's
        // first load; whether the ring is closed is decided at the end from the last vertex a lane held (the terms are
        // accumulated as if it were and dropped when it is not — the same sum, since an open sequence added nothing before).
        constexpr bool PAIR = G == 2;
        int c0, c1;
        if (PAIR) {
            const int mine = seq_off[s + lane], other = dev::dpp_mov<0xB1>(mine);  // quad_perm [1,0,3,2]
            c0 = lane == 0 ? mine : other;
            c1 = lane == 0 ? other : mine;
        } else {
            c0 = seq_off[s];
            c1 = seq_off[s + 1];
        }
        const int n = c1 - c0;
        double a2 = 0, acx = 0, acy = 0, len = 0, lmx = 0, lmy = 0, sx_ = 0, sy_ = 0;
        double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
        double2 first = make_double2(0, 0), last = make_double2(0, 0);
        if (!PAIR && n > 0) {
            first = xy[c0];
            last = xy[c1 - 1];
        }
        // twice_signed_ring_area: < 3 coords or open -> 0 (area.rs); centroid's add_ring uses the same
        bool closed_ring = PAIR ? n >= 3 : (n >= 3 && first.x == last.x && first.y == last.y);
        // ONE load per vertex: lane l holds vertex c0 + t*G + l in round t; the edge's other end is the neighbour lane's
        // vertex (DPP row shift), and for the group's last lane the first lane's vertex of the NEXT round, which is
        // prefetched one round ahead anyway.  Trip count is uniform within the group (all its lanes stay active for DPP).
        constexpr bool EDGES = (MASK & (M_AREA | M_CENT | M_LEN | M_LENC)) != 0;
        // Two-lane groups (sequences of at most 16 coordinates: eight rounds at most) request ALL their vertices before the first round:
        // with one round of look-ahead every round still waited a full memory round trip for a few flops — the class ran at 2.3 TB/s
        // against 4.9 for the 8- and 16-lane ones.
        // (four rounds in flight, not all nine: the classes share one kernel, and nine held vertices took its register count from 52 to
        // 72 — the 8- and 16-lane classes lost a wave of occupancy and 8 % of their speed)
        constexpr int PRE = PAIR ? SEQ_MAXLEN[0] / G + 1 : 1, AHEAD = 4;
        double2 pre[PRE];
        auto vertex_of_round = [&](int t) {
            const int idx = c0 + lane + t * G;
            return idx < c1 ? xy[idx] : make_double2(0.0, 0.0);
        };
        if (PAIR) {
#pragma unroll
            for (int t = 0; t < PRE; ++t) pre[t] = t < AHEAD ? vertex_of_round(t) : make_double2(0.0, 0.0);
        }
        int i = c0 + lane;
        double2 cur = PAIR ? pre[0] : (i < c1 ? xy[i] : make_double2(0.0, 0.0));
        double2 held = cur;  // PAIR: the last vertex this lane loaded
        if (PAIR) first = make_double2(dev::dpp_mov<0xA0>(cur.x), dev::dpp_mov<0xA0>(cur.y));  // quad_perm [0,0,2,2]: lane 0's vertex
        auto round = [&](const double2 nxt) {
            const double2 p = cur;
            if (PAIR && i < c1) held = p;
            double2 q = make_double2(0.0, 0.0);
            if (EDGES) {
                const double ax = dev::dpp_mov<0x101>(cur.x), ay = dev::dpp_mov<0x101>(cur.y);                  // row_shl:1
                const double bx = dev::dpp_mov<0x110 + G - 1>(nxt.x), by = dev::dpp_mov<0x110 + G - 1>(nxt.y);  // row_shr:G-1
                q = lane == G - 1 ? make_double2(bx, by) : make_double2(ax, ay);
            }
            cur = nxt;
            if (i >= c1) return;
            if (MASK & M_BBOX) {
                mnx = p.x < mnx ? p.x : mnx;
                mny = p.y < mny ? p.y : mny;
                mxx = p.x > mxx ? p.x : mxx;
                mxy = p.y > mxy ? p.y : mxy;
            }
            if (MASK & M_SUM) {
                sx_ += p.x;
                sy_ += p.y;
            }
            if (EDGES && i + 1 < c1) {
                if ((MASK & (M_AREA | M_CENT)) && closed_ring) {
                    const double sx = p.x - first.x, sy = p.y - first.y;
                    const double ex = q.x - first.x, ey = q.y - first.y;
                    const double cr = sx * ey - sy * ex;
                    a2 += cr;
                    if (MASK & M_CENT) {
                        acx += (ex + sx) * cr;
                        acy += (ey + sy) * cr;
                    }
                }
                if (MASK & (M_LEN | M_LENC)) {
                    const double l = hypot(q.x - p.x, q.y - p.y);
                    len += l;
                    if (MASK & M_LENC) {
                        lmx += (p.x + q.x) / 2.0 * l;
                        lmy += (p.y + q.y) / 2.0 * l;
                    }
                }
            }
        };
        if constexpr (PAIR) {
#pragma unroll
            for (int t = 0; t < PRE - 1; ++t) {
                if (c0 + t * G >= c1) break;
                if (t + AHEAD < PRE) pre[t + AHEAD] = vertex_of_round(t + AHEAD);
                round(pre[t + 1]);
                i += G;
            }
        } else {  // (the loop as it was: the same body through the lambda cost the 8- and 16-lane classes 5 - 9 %)
            double2 n1 = i + G < c1 ? xy[i + G] : make_double2(0.0, 0.0);  // (two rounds ahead: one more vertex in flight per lane; a third: no gain)
            for (int base = c0; base < c1; base += G, i += G) {
                const double2 n2 = i + 2 * G < c1 ? xy[i + 2 * G] : make_double2(0.0, 0.0);
                const double2 nxt = n1;
                n1 = n2;
                const double2 p = cur;
                if (PAIR && i < c1) held = p;
                double2 q = make_double2(0.0, 0.0);
                if (EDGES) {
                    const double ax = dev::dpp_mov<0x101>(cur.x), ay = dev::dpp_mov<0x101>(cur.y);                  // row_shl:1
                    const double bx = dev::dpp_mov<0x110 + G - 1>(nxt.x), by = dev::dpp_mov<0x110 + G - 1>(nxt.y);  // row_shr:G-1
                    q = lane == G - 1 ? make_double2(bx, by) : make_double2(ax, ay);
                }
                cur = nxt;
                if (i >= c1) continue;
                if (MASK & M_BBOX) {
                    mnx = p.x < mnx ? p.x : mnx;
                    mny = p.y < mny ? p.y : mny;
                    mxx = p.x > mxx ? p.x : mxx;
                    mxy = p.y > mxy ? p.y : mxy;
                }
                if (MASK & M_SUM) {
                    sx_ += p.x;
                    sy_ += p.y;
                }
                if (EDGES && i + 1 < c1) {
                    if ((MASK & (M_AREA | M_CENT)) && closed_ring) {
                        const double sx = p.x - first.x, sy = p.y - first.y;
                        const double ex = q.x - first.x, ey = q.y - first.y;
                        const double cr = sx * ey - sy * ex;
                        a2 += cr;
                        if (MASK & M_CENT) {
                            acx += (ex + sx) * cr;
                            acy += (ey + sy) * cr;
                        }
                    }
                    if (MASK & (M_LEN | M_LENC)) {
                        const double l = hypot(q.x - p.x, q.y - p.y);
                        len += l;
                        if (MASK & M_LENC) {
                            lmx += (p.x + q.x) / 2.0 * l;
                            lmy += (p.y + q.y) / 2.0 * l;
                        }
                    }
                }
            }
        }
        if (PAIR && (MASK & (M_AREA | M_CENT))) {  // closed? the last vertex sits with lane (n - 1) & 1
            const double ox = dev::dpp_mov<0xB1>(held.x), oy = dev::dpp_mov<0xB1>(held.y);
            const double2 lastv = ((n - 1) & 1) == lane ? held : make_double2(ox, oy);
            closed_ring = closed_ring && first.x == lastv.x && first.y == lastv.y;
            if (!closed_ring) a2 = acx = acy = 0.0;
        }
        if (MASK & (M_AREA | M_CENT)) a2 = group_sum<G>(a2);
        if (MASK & M_CENT) {
            acx = group_sum<G>(acx);
            acy = group_sum<G>(acy);
        }
        if (MASK & M_LENC) {
            lmx = group_sum<G>(lmx);
            lmy = group_sum<G>(lmy);
        }
        if (MASK & (M_LEN | M_LENC)) len = group_sum<G>(len);
        if (MASK & M_BBOX) {
            mnx = group_min<G>(mnx);
            mny = group_min<G>(mny);
            mxx = group_max<G>(mxx);
            mxy = group_max<G>(mxy);
        }
        if (MASK & M_SUM) {
            sx_ = group_sum<G>(sx_);
            sy_ = group_sum<G>(sy_);
        }
        if (lane == 0) {
            SeqPartial r;
            r.a2 = a2; r.acx = acx; r.acy = acy; r.len = len; r.lmx = lmx; r.lmy = lmy; r.sx = sx_; r.sy = sy_;
            r.mnx = mnx; r.mny = mny; r.mxx = mxx; r.mxy = mxy;
            seq_store_any<MASK, FIN>(r, stats, n_seq, s, c0, c1, xy, fin);
        }
    }
}

// ---- stage 2: one thread per geometry ----------------------------------------------------------
__device__ __forceinline__ void geom_seq_range(const DevGeo& a, int64_t g, int& s0, int& s1) {
    // range of level-2 sequences (rings / member linestrings) of geometry g, for non-polygonal types
    if (a.type == GPK_GEOM_MULTILINESTRING) {
        s0 = a.geom_off[g];
        s1 = a.geom_off[g + 1];
    } else {  // LINESTRING / MULTIPOINT: the geometry is its own sequence
        s0 = (int)g;
        s1 = (int)g + 1;
    }
}

template <bool SIGNED>
__global__ void area_combine_kernel(DevGeo a, const double* __restrict__ stats, int64_t n_seq,
                                    double* __restrict__ out) {
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
            if (r1 <= r0) continue;
            double area = stats[ST_AREA2 * n_seq + r0] / 2.0;
            const bool neg = area < 0.0;
            area = fabs(area);
            for (int r = r0 + 1; r < r1; ++r) area -= fabs(stats[ST_AREA2 * n_seq + r] / 2.0);
            const double sa = neg ? -area : area;
            v += SIGNED ? sa : fabs(sa);
        }
    }
    out[g] = v;
}

__global__ void length_combine_kernel(DevGeo a, const double* __restrict__ stats, int64_t n_seq,
                                      double* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    if (!dev::valid_row(a.validity, g)) {
        out[g] = NAN;
        return;
    }
    double v = 0.0;
    if (is_polygonal(a.type)) {  // exterior rings only
        int p0, p1;
        dev::geom_parts(a, g, p0, p1);
        for (int p = p0; p < p1; ++p) {
            int r0, r1;
            dev::part_rings(a, p, r0, r1);
            if (r1 > r0) v += stats[ST_LEN * n_seq + r0];
        }
    } else if (a.type == GPK_GEOM_LINESTRING || a.type == GPK_GEOM_MULTILINESTRING) {
        int s0, s1;
        geom_seq_range(a, g, s0, s1);
        for (int s = s0; s < s1; ++s) v += stats[ST_LEN * n_seq + s];
    }
    out[g] = v;
}

__global__ void bounds_combine_kernel(DevGeo a, const double* __restrict__ stats, int64_t n_seq,
                                      double* __restrict__ out4) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    bool have = false;
    if (dev::valid_row(a.validity, g)) {
        if (is_polygonal(a.type)) {  // Polygon::bounding_rect scans the exterior only
            int p0, p1;
            dev::geom_parts(a, g, p0, p1);
            for (int p = p0; p < p1; ++p) {
                int r0, r1;
                dev::part_rings(a, p, r0, r1);
                if (r1 <= r0) continue;
                const double4 b = bbox_records(stats, n_seq)[r0];
                // (an empty exterior ring has no box: its record is the reductions' identity — as is that of a ring of NaN coordinates, which
                // does count; only such a record sends the thread to the ring's offsets, two more dependent reads a polygon otherwise)
                if (!(b.x <= b.z) && a.ring_off[r0 + 1] == a.ring_off[r0]) continue;
                have = true;
                mnx = fmin(mnx, b.x);
                mny = fmin(mny, b.y);
                mxx = fmax(mxx, b.z);
                mxy = fmax(mxy, b.w);
            }
        } else {
            int s0, s1;
            geom_seq_range(a, g, s0, s1);
            const int32_t* so = a.type == GPK_GEOM_MULTILINESTRING ? a.ring_off : a.geom_off;
            for (int s = s0; s < s1; ++s) {
                const double4 b = bbox_records(stats, n_seq)[s];
                if (!(b.x <= b.z) && so[s + 1] == so[s]) continue;
                have = true;
                mnx = fmin(mnx, b.x);
                mny = fmin(mny, b.y);
                mxx = fmax(mxx, b.z);
                mxy = fmax(mxy, b.w);
            }
        }
    }
    double4 r;
    if (have)
        r = make_double4(mnx, mny, mxx, mxy);
    else
        r = make_double4(NAN, NAN, NAN, NAN);
    reinterpret_cast<double4*>(out4)[g] = r;
}

// geo 0.27 centroid.rs WeightedCentroid: highest dimension wins.
struct WC {
    int dim;
    double w, ax, ay;
};
__device__ __forceinline__ void wc_add(WC& c, int dim, double cx, double cy, double w) {
    if (dim > c.dim) {
        c.dim = dim;
        c.w = w;
        c.ax = cx * w;
        c.ay = cy * w;
    } else if (dim == c.dim) {
        c.w += w;
        c.ax += cx * w;
        c.ay += cy * w;
    }
}
__device__ __forceinline__ void wc_add_raw(WC& c, int dim, double ax, double ay, double w) {
    if (dim > c.dim) {
        c.dim = dim;
        c.w = w;
        c.ax = ax;
        c.ay = ay;
    } else if (dim == c.dim) {
        c.w += w;
        c.ax += ax;
        c.ay += ay;
    }
}
// add_line_string from the per-sequence length partials
__device__ __forceinline__ void wc_add_linestring(WC& c, const double* stats, int64_t n_seq, int s,
                                                  const double2* xy, int c0, int n) {
    if (n == 0) return;
    const double len = stats[ST_LEN * n_seq + s];
    if (len > 0.0) {
        wc_add_raw(c, 1, stats[ST_LMX * n_seq + s], stats[ST_LMY * n_seq + s], len);
    } else {
        // every segment is zero-length: add_coord(start) per segment (or the single coord)
        const double2 p = xy[c0];
        const double k = n == 1 ? 1.0 : (double)(n - 1);
        wc_add_raw(c, 0, p.x * k, p.y * k, k);
    }
}
// add_line_string walked by one thread (only for the rare polygon whose holes cancel its exterior area
// exactly: the exterior's length partials were not computed by the degenerate-only pass)
__device__ inline void wc_add_linestring_direct(WC& c, const double2* xy, int c0, int n) {
    if (n == 0) return;
    double len = 0.0, lmx = 0.0, lmy = 0.0;
    for (int i = c0; i + 1 < c0 + n; ++i) {
        const double2 p = xy[i], q = xy[i + 1];
        const double l = hypot(q.x - p.x, q.y - p.y);
        len += l;
        lmx += (p.x + q.x) / 2.0 * l;
        lmy += (p.y + q.y) / 2.0 * l;
    }
    if (len > 0.0) {
        wc_add_raw(c, 1, lmx, lmy, len);
    } else {
        const double2 p = xy[c0];
        const double k = n == 1 ? 1.0 : (double)(n - 1);
        wc_add_raw(c, 0, p.x * k, p.y * k, k);
    }
}
__device__ __forceinline__ void wc_add_ring(WC& c, const double* stats, int64_t n_seq, int r, const DevGeo& a) {
    const double area = stats[ST_AREA2 * n_seq + r] / 2.0;
    if (area == 0.0) {  // (rare: only here are the ring's offsets and first coordinate read)
        const int c0 = a.ring_off[r];
        wc_add_linestring(c, stats, n_seq, r, a.xy, c0, a.ring_off[r + 1] - c0);
        return;
    }
    wc_add(c, 2, stats[ST_ACX * n_seq + r], stats[ST_ACY * n_seq + r], fabs(area));  // (the ring pass stored the ring's centroid: seq_store_any)
}

__global__ void centroid_combine_kernel(DevGeo a, const double* __restrict__ stats, int64_t n_seq,
                                        double* __restrict__ out_xy, uint8_t* __restrict__ out_valid) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    WC c{-1, 0, 0, 0};
    if (dev::valid_row(a.validity, g)) {
        if (is_polygonal(a.type)) {
            int p0, p1;
            dev::geom_parts(a, g, p0, p1);
            for (int p = p0; p < p1; ++p) {
                int r0, r1;
                dev::part_rings(a, p, r0, r1);
                if (r1 <= r0) continue;
                WC ext{-1, 0, 0, 0}, in{-1, 0, 0, 0};
                wc_add_ring(ext, stats, n_seq, r0, a);
                for (int r = r0 + 1; r < r1; ++r) wc_add_ring(in, stats, n_seq, r, a);
                if (ext.dim < 0) continue;
                if (in.dim >= 0 && in.dim == ext.dim) {
                    ext.w -= in.w;
                    ext.ax -= in.ax;
                    ext.ay -= in.ay;
                    if (ext.w == 0.0) {
                        const int e0 = a.ring_off[r0];
                        wc_add_linestring_direct(c, a.xy, e0, a.ring_off[r0 + 1] - e0);
                        continue;
                    }
                }
                wc_add_raw(c, ext.dim, ext.ax, ext.ay, ext.w);
            }
        } else if (a.type == GPK_GEOM_MULTIPOINT) {
            const int c0 = a.geom_off[g], n = a.geom_off[g + 1] - c0;
            if (n > 0) wc_add_raw(c, 0, stats[ST_SUMX * n_seq + g], stats[ST_SUMY * n_seq + g], (double)n);
        } else {
            int s0, s1;
            geom_seq_range(a, g, s0, s1);
            const int32_t* so = a.type == GPK_GEOM_MULTILINESTRING ? a.ring_off : a.geom_off;
            for (int s = s0; s < s1; ++s) wc_add_linestring(c, stats, n_seq, s, a.xy, so[s], so[s + 1] - so[s]);
        }
    }
    double2 r;
    if (c.dim < 0)
        r = make_double2(NAN, NAN);
    else
        r = make_double2(c.ax / c.w, c.ay / c.w);
    reinterpret_cast<double2*>(out_xy)[g] = r;
    if (out_valid) out_valid[g] = c.dim >= 0;
}

// ---- POINT arrays -------------------------------------------------------------------------------
__global__ void point_unary_kernel(DevGeo a, int op, double* __restrict__ out, uint8_t* __restrict__ out_valid) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    const double2 p = a.xy[g];
    const bool ok = dev::valid_row(a.validity, g);
    const bool empty = !ok || isnan(p.x) || isnan(p.y);
    if (op == 0) {  // area / length
        out[g] = ok ? 0.0 : NAN;
    } else if (op == 1) {  // centroid
        reinterpret_cast<double2*>(out)[g] = empty ? make_double2(NAN, NAN) : p;
        if (out_valid) out_valid[g] = !empty;
    } else {  // bounds
        reinterpret_cast<double4*>(out)[g] =
            empty ? make_double4(NAN, NAN, NAN, NAN) : make_double4(p.x, p.y, p.x, p.y);
    }
}

// ---- affine: pure elementwise stream, 16 B in + 16 B out per coordinate ------------------------------
__global__ __launch_bounds__(256) void affine_kernel(const double2* __restrict__ xy, int64_t n,
                                                     double m0, double m1, double m2, double m3,
                                                     double m4, double m5, double2* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 p = xy[i];
        double2 q;
        q.x = (m0 * p.x + m1 * p.y) + m2;  // contraction is off: matches AffineTransform::apply bit for bit
        q.y = (m3 * p.x + m4 * p.y) + m5;
        out[i] = q;
    }
}

// ---- host drivers ------------------------------------------------------------------------------
// per-row matrices: G lanes walk the contiguous coordinate range of one geometry
__device__ __forceinline__ void geom_coords(const DevGeo& a, int64_t g, int& c0, int& c1) {
    switch (a.type) {
    case GPK_GEOM_POINT: c0 = (int)g; c1 = (int)g + 1; break;
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT: c0 = a.geom_off[g]; c1 = a.geom_off[g + 1]; break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING: c0 = a.ring_off[a.geom_off[g]]; c1 = a.ring_off[a.geom_off[g + 1]]; break;
    default: c0 = a.ring_off[a.part_off[a.geom_off[g]]]; c1 = a.ring_off[a.part_off[a.geom_off[g + 1]]];
    }
}
template <int G>
__global__ __launch_bounds__(256) void affine_rows_kernel(DevGeo a, const double* __restrict__ mats, double2* __restrict__ out) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t g = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; g < a.n_geoms; g += groups) {
        int c0, c1;
        geom_coords(a, g, c0, c1);
        const double m0 = mats[6 * g], m1 = mats[6 * g + 1], m2 = mats[6 * g + 2], m3 = mats[6 * g + 3], m4 = mats[6 * g + 4],
                     m5 = mats[6 * g + 5];
        for (int i = c0 + lane; i < c1; i += G) {
            const double2 p = a.xy[i];
            out[i] = make_double2((m0 * p.x + m1 * p.y) + m2, (m3 * p.x + m4 * p.y) + m5);
        }
    }
}

// ---- class 0 (sequences of at most SEQ_MAXLEN[0] coordinates: building footprints, parcels) -------------------------
// Two lanes per sequence means a wave works on 32 sequences at a time; loading them straight from global memory costs
// one partially used load instruction per round and sequence pair (the texture-address units, not HBM, set the pace:
// 3.0 TB/s on 9-coordinate rings).  Instead the wave copies the whole coordinate span of its 32 sequences into LDS with
// full-width coalesced loads and the lanes read their vertices from there.  Spans wider than TINY_SPAN (sequences of
// other classes in between) fall back to direct loads for that round.
constexpr int TINY_SPAN = 512;  // coordinates staged per wave and round (8 KB)
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
template <unsigned MASK, int FIN>
__device__ __forceinline__ void seq_stats_tiny_body(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                                    double* __restrict__ stats, const int32_t* __restrict__ list, int64_t n_list, int block,
                                                    int n_blocks, double2* __restrict__ lds, const FinalOut& fin) {
    constexpr int G = 2;
    const int lane64 = threadIdx.x & 63, lane = threadIdx.x & (G - 1), wave = threadIdx.x >> 6;
    double2* __restrict__ buf = lds + wave * TINY_SPAN;
    const int64_t per_round = (int64_t)n_blocks * (256 / G);
    for (int64_t k0 = (int64_t)block * (256 / G) + wave * (64 / G); k0 < n_list; k0 += per_round) {  // wave-uniform trip count
        const int64_t k = k0 + lane64 / G;
        bool act = k < n_list;
        int64_t s = 0;
        int c0 = 0, c1 = 0;
        if (act) {
            s = list ? (int64_t)list[k] : k;
            c0 = seq_off[s];
            c1 = seq_off[s + 1];
            if ((MASK & M_DEGEN) && stats[ST_AREA2 * n_seq + s] != 0.0) act = false;
        }
        const bool has = act && c1 > c0;
        const int lo = wave_min_i32(has ? c0 : 0x7FFFFFFF), hi = wave_max_i32(has ? c1 : (int)0x80000000);
        const bool staged = hi > lo && hi - lo <= TINY_SPAN;  // wave-uniform
        if (staged) {
            __builtin_amdgcn_wave_barrier();  // the previous round's reads are done
            for (int t = lane64; t < hi - lo; t += 64) buf[t] = xy[lo + t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (act) {
            const int n = c1 - c0;
            SeqPartial a;
            double2 first = make_double2(0, 0), last = make_double2(0, 0);
            if (n > 0) {
                first = staged ? buf[c0 - lo] : xy[c0];
                last = staged ? buf[c1 - 1 - lo] : xy[c1 - 1];
            }
            const bool closed_ring = n >= 3 && first.x == last.x && first.y == last.y;
            for (int i = c0 + lane; i < c1; i += G) {
                const bool has_q = i + 1 < c1;
                double2 p, q = make_double2(0, 0);
                if (staged) {
                    p = buf[i - lo];
                    if (has_q) q = buf[i + 1 - lo];
                } else {
                    p = xy[i];
                    if (has_q) q = xy[i + 1];
                }
                seq_accumulate_pq<MASK>(a, p, q, has_q, first, closed_ring);
            }
            if (MASK & (M_AREA | M_CENT)) a.a2 = group_sum<G>(a.a2);
            if (MASK & M_CENT) {
                a.acx = group_sum<G>(a.acx);
                a.acy = group_sum<G>(a.acy);
            }
            if (MASK & M_LENC) {
                a.lmx = group_sum<G>(a.lmx);
                a.lmy = group_sum<G>(a.lmy);
            }
            if (MASK & (M_LEN | M_LENC)) a.len = group_sum<G>(a.len);
            if (MASK & M_BBOX) {
                a.mnx = group_min<G>(a.mnx);
                a.mny = group_min<G>(a.mny);
                a.mxx = group_max<G>(a.mxx);
                a.mxy = group_max<G>(a.mxy);
            }
            if (MASK & M_SUM) {
                a.sx = group_sum<G>(a.sx);
                a.sy = group_sum<G>(a.sy);
            }
            if (lane == 0) seq_store_any<MASK, FIN>(a, stats, n_seq, s, c0, c1, xy, fin);
        }
    }
}

// ONE launch for all size classes: consecutive ranges of work-groups take the long chunks, then the 2-, 8- and 16-lane
// classes (the long chunks first: they are the longest-running work-groups).  Separate launches would each pay a
// launch gap and a tail on columns where a class holds only a few sequences.
struct SeqPlan {
    const int32_t* list[4];  // ids per class (nullptr = all sequences in order)
    int64_t count[4];
    int blocks[4];           // work-groups given to: class 0, 1, 2, long
    const int32_t* chunk_begin;
    int64_t n_chunks;
    double* long_part;
};
template <unsigned MASK, int FIN>
__global__ __launch_bounds__(256) void seq_stats_kernel(const double2* __restrict__ xy, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                                        double* __restrict__ stats, SeqPlan p, FinalOut fin) {
    if ((MASK & M_DEGEN) && fin.degen_flag && *fin.degen_flag == 0) return;  // no zero-area ring anywhere: nothing to redo
    int b = blockIdx.x;
    if (b < p.blocks[3]) {
        seq_stats_long_body<MASK, FIN>(xy, seq_off, n_seq, p.list[3], p.count[3], p.chunk_begin, p.n_chunks, p.long_part, stats, b, p.blocks[3], fin);
        return;
    }
    b -= p.blocks[3];
    if (b < p.blocks[0]) {
        // a column of short sequences only: consecutive sequences are adjacent in memory, stage them through LDS.  In a
        // mixed column the 32 sequences of a round are interleaved with longer ones (half of the rounds of the power-law
        // test column span more than TINY_SPAN coordinates and the rest load 50 % foreign bytes): lane groups then.
        extern __shared__ double2 tiny_lds[];  // 4 waves x TINY_SPAN coordinates, allocated only for the staged form
        if (p.list[0] == nullptr)
            seq_stats_tiny_body<MASK, FIN>(xy, seq_off, n_seq, stats, p.list[0], p.count[0], b, p.blocks[0], tiny_lds, fin);
        else
            seq_stats_group_body<2, MASK, FIN>(xy, seq_off, n_seq, stats, p.list[0], p.count[0], b, p.blocks[0], fin);
        return;
    }
    b -= p.blocks[0];
    if (b < p.blocks[1]) {
        seq_stats_group_body<8, MASK, FIN>(xy, seq_off, n_seq, stats, p.list[1], p.count[1], b, p.blocks[1], fin);
        return;
    }
    b -= p.blocks[1];
    seq_stats_group_body<16, MASK, FIN>(xy, seq_off, n_seq, stats, p.list[2], p.count[2], b, p.blocks[2], fin);
}

// lanes per geometry for the per-row affine kernel: ~8 vertices per lane
static int pick_group(int64_t n_coords, int64_t n_seq) {
    const double mean = n_seq > 0 ? (double)n_coords / (double)n_seq : 1.0;
    int g = 4;
    while (g < 64 && g * 2 * 8 <= mean) g <<= 1;
    return g;
}

static void seq_view(const DevGeo& a, const int32_t** seq_off, int64_t* n_seq) {
    if (a.type == GPK_GEOM_LINESTRING || a.type == GPK_GEOM_MULTIPOINT) {
        *seq_off = a.geom_off;
        *n_seq = a.n_geoms;
    } else {
        *seq_off = a.ring_off;
        *n_seq = a.n_rings;
    }
}

// POLYGON column: does every polygon own exactly the ring with its own index?  (flag[0] stays 1 when so)
__global__ void one_ring_each_kernel(const int32_t* __restrict__ geom_off, int64_t n, int32_t* __restrict__ flag) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g <= n && geom_off[g] != (int32_t)g) atomicAnd(flag, 0);
}

// classification of an array's sequences, built once per handle (the lock also orders concurrent first uses)
static std::mutex g_classes_mu;
static int32_t seq_classes_of(const gpk_geoarray* a, hipStream_t s, const gpk_seq_classes** out) {
    std::lock_guard<std::mutex> lk(g_classes_mu);
    if (a->classes) {
        *out = a->classes;
        return GPK_OK;
    }
    const int32_t* seq_off;
    int64_t n_seq;
    seq_view(a->d, &seq_off, &n_seq);
    gpk_seq_classes* c = new gpk_seq_classes;
    memset(c, 0, sizeof *c);
    auto bail = [&](int32_t rc) {
        if (c->lists) (void)hipFree(c->lists);
        if (c->chunk_begin) (void)hipFree(c->chunk_begin);
        if (c->strip_first) (void)hipFree(c->strip_first);
        if (c->strip_cross) (void)hipFree(c->strip_cross);
        if (c->strip_desc) (void)hipFree(c->strip_desc);
        delete c;
        return rc;
    };
    if (n_seq > 0) {
        int32_t* cursor = nullptr;
        hipError_t e = device_malloc((void**)&cursor, 4 * sizeof(int32_t));
        if (e != hipSuccess) return bail(fail(GPK_ERR_OOM, "size classes: %s", hipGetErrorString(e)));
        int32_t h[4] = {0, 0, 0, 0};
        auto run = [&]() -> int32_t {
            GPK_HIP(hipMemsetAsync(cursor, 0, 4 * sizeof(int32_t), s));
            const unsigned cls_blocks = (unsigned)((n_seq + 256 * SEQ_CLS_ROUNDS - 1) / (256 * SEQ_CLS_ROUNDS));
            GPK_LAUNCH("gpk_seq_classify", seq_classify_kernel<false>, dim3(cls_blocks), dim3(256), 0, s, seq_off, n_seq, cursor,
                       (int32_t*)nullptr);
            GPK_HIP(hipMemcpyAsync(h, cursor, sizeof h, hipMemcpyDeviceToHost, s));
            GPK_HIP(hipStreamSynchronize(s));
            int64_t run_begin = 0;
            bool single = false;
            for (int k = 0; k < 4; ++k) {
                c->count[k] = h[k];
                c->begin[k] = run_begin;
                run_begin += h[k];
                single |= (int64_t)h[k] == n_seq;
            }
            if (single) return GPK_OK;  // one class holds everything: kernels walk the sequences in order, no list
            GPK_HIP(device_malloc((void**)&c->lists, sizeof(int32_t) * (size_t)n_seq));
            int32_t b32[4] = {(int32_t)c->begin[0], (int32_t)c->begin[1], (int32_t)c->begin[2], (int32_t)c->begin[3]};
            GPK_HIP(hipMemcpyAsync(cursor, b32, sizeof b32, hipMemcpyHostToDevice, s));
            GPK_LAUNCH("gpk_seq_classify", seq_classify_kernel<true>, dim3(cls_blocks), dim3(256), 0, s, seq_off, n_seq, cursor,
                       c->lists);
            GPK_HIP(hipStreamSynchronize(s));
            return GPK_OK;
        };
        int32_t rc = run();
        if (rc == GPK_OK) {
            if (a->d.type == GPK_GEOM_LINESTRING || a->d.type == GPK_GEOM_MULTIPOINT) {
                c->one_to_one = true;
            } else if (a->d.type == GPK_GEOM_POLYGON && a->d.n_rings == a->d.n_geoms) {
                auto run3 = [&]() -> int32_t {
                    int32_t one = 1;
                    GPK_HIP(hipMemcpyAsync(cursor, &one, sizeof one, hipMemcpyHostToDevice, s));
                    GPK_LAUNCH("gpk_one_ring_each", one_ring_each_kernel, dim3((unsigned)((a->d.n_geoms + 256) / 256)), dim3(256), 0, s, a->d.geom_off,
                               a->d.n_geoms, cursor);
                    GPK_HIP(hipMemcpyAsync(&one, cursor, sizeof one, hipMemcpyDeviceToHost, s));
                    GPK_HIP(hipStreamSynchronize(s));
                    c->one_to_one = one != 0;
                    return GPK_OK;
                };
                rc = run3();
            }
        }
        (void)hipFree(cursor);
        if (rc == GPK_OK && c->count[3] > 0) {  // chunks of the long sequences (exclusive scan over the long list)
            const int64_t nl = c->count[3], nb = (nl + 255) / 256;
            int32_t* chunks = nullptr;
            unsigned long long* btot = nullptr;
            auto run2 = [&]() -> int32_t {
                GPK_HIP(device_malloc((void**)&chunks, sizeof(int32_t) * (size_t)(nl + 1)));
                GPK_HIP(device_malloc((void**)&btot, sizeof(unsigned long long) * (size_t)(nb + 2)));
                GPK_HIP(device_malloc((void**)&c->chunk_begin, sizeof(int32_t) * (size_t)(nl + 1)));
                const int32_t* list = c->lists ? c->lists + c->begin[3] : nullptr;
                GPK_LAUNCH("gpk_seq_long_chunks", long_chunk_count_kernel, dim3((unsigned)nb), dim3(256), 0, s, seq_off, list, nl, chunks);
                GPK_TRY(exclusive_scan_i32(chunks, nl, c->chunk_begin, nullptr, btot, s));
                int32_t total = 0;
                GPK_HIP(hipMemcpyAsync(&total, c->chunk_begin + nl, sizeof total, hipMemcpyDeviceToHost, s));
                GPK_HIP(hipStreamSynchronize(s));
                c->n_chunks = total;
                return GPK_OK;
            };
            rc = run2();
            if (chunks) (void)hipFree(chunks);
            if (btot) (void)hipFree(btot);
        }
        // Which polygonal columns take the one-pass form (gpk_ringstream.hip): those it is measured faster on — ragged columns (no single size
        // class) and columns of rings of at most 16 coordinates; a column whose rings all sit in one of the larger classes (2M x 65 coordinates:
        // 0.44 against 0.38 ms) keeps the two-stage form.  GPK_RING_STREAM=0: no column takes it; =1: every eligible one (tests, A/B runs).
        static const int stream_mode = [] {
            const char* e = getenv("GPK_RING_STREAM");
            return e ? (e[0] == '0' ? 0 : 1) : 2;
        }();
        const bool uniform_large = (c->count[1] == n_seq || c->count[2] == n_seq || c->count[3] == n_seq) && n_seq > 0;
        if (rc == GPK_OK && stream_mode != 0 && !(stream_mode == 2 && uniform_large) && is_polygonal(a->d.type) &&
            a->d.n_coords < (int64_t)0x7FFFFFFF - RS_STRIP) {
            // the strip table of the one-pass form (gpk_ringstream.hip) and whether the column is eligible for it
            auto run4 = [&]() -> int32_t {
                const int64_t n_strips = ring_stream_strips(a->d.n_coords);
                GPK_HIP(device_malloc((void**)&c->strip_first, sizeof(int32_t) * (size_t)(2 * (n_strips + 1) + 1)));
                int32_t* flags = c->strip_first + 2 * (n_strips + 1);
                GPK_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t), s));
                GPK_TRY(ring_stream_build_table(a->d, c->strip_first, c->strip_first + n_strips + 1, flags, s));
                int32_t h_flags = -1;
                GPK_HIP(hipMemcpyAsync(&h_flags, flags, sizeof h_flags, hipMemcpyDeviceToHost, s));
                GPK_HIP(hipStreamSynchronize(s));
                c->strips_ok = h_flags == 0;
                if (c->strips_ok) GPK_TRY(ring_stream_build_cross(a->d, c->strip_first, c->strip_first + n_strips + 1, s, &c->strip_cross, &c->strip_desc));
                return GPK_OK;
            };
            rc = run4();
        }
        if (rc != GPK_OK) return bail(rc);
    }
    const_cast<gpk_geoarray*>(a)->classes = c;
    *out = c;
    return GPK_OK;
}

template <unsigned MASK, int FIN = FIN_NONE>
static int32_t launch_seq_stats(const gpk_geoarray* arr, double* stats, double* long_part, hipStream_t s, const char* name,
                                FinalOut fin = FinalOut{nullptr, nullptr, nullptr, nullptr}) {
    const DevGeo& a = arr->d;
    const int32_t* seq_off;
    int64_t n_seq;
    seq_view(a, &seq_off, &n_seq);
    if (n_seq == 0) return GPK_OK;
    const gpk_seq_classes* c;
    GPK_TRY(seq_classes_of(arr, s, &c));
    const int64_t cap = (int64_t)cu_count() * 16;
    SeqPlan p;
    int total_blocks = 0;
    for (int k = 0; k < 4; ++k) {
        p.list[k] = c->lists ? (const int32_t*)(c->lists + c->begin[k]) : (const int32_t*)nullptr;
        p.count[k] = c->count[k];
        int64_t blocks;
        if (k == 3) {
            blocks = c->n_chunks < 4096 ? c->n_chunks : 4096;
        } else {
            const int64_t groups_per_block = 256 / SEQ_LANES[k];
            blocks = (c->count[k] + groups_per_block - 1) / groups_per_block;
            if (blocks > cap) blocks = cap;
        }
        p.blocks[k] = c->count[k] > 0 ? (int)blocks : 0;
        total_blocks += p.blocks[k];
    }
    p.chunk_begin = c->chunk_begin;
    p.n_chunks = c->n_chunks;
    p.long_part = long_part;
    static_assert(SEQ_LANES[0] == 2 && SEQ_LANES[1] == 8 && SEQ_LANES[2] == 16, "seq_stats_kernel's dispatch");
    const size_t tiny_lds = (p.blocks[0] > 0 && !p.list[0]) ? sizeof(double2) * 4 * TINY_SPAN : 0;
    if (total_blocks > 0)
        GPK_LAUNCH(name, (seq_stats_kernel<MASK, FIN>), dim3((unsigned)total_blocks), dim3(256), tiny_lds, s, a.xy, seq_off, n_seq, stats, p, fin);
    if (c->n_chunks > c->count[3])  // some sequence spans several chunks
        GPK_LAUNCH("gpk_seq_long_combine", (seq_long_combine_kernel<MASK, FIN>), dim3((unsigned)((c->count[3] + 255) / 256)), dim3(256), 0, s, p.list[3],
                   c->count[3], (const int32_t*)c->chunk_begin, (const double*)long_part, n_seq, stats, fin, seq_off, a.xy);
    return GPK_OK;
}

static inline dim3 grid_for(int64_t n, int block = 256) {
    int64_t b = (n + block - 1) / block;
    return dim3((unsigned)(b > 0 ? b : 1));
}

// common prologue: stats + output staging in the workspace
struct UnaryCtx {
    double* stats = nullptr;
    double* long_part = nullptr;  // chunk partials of the long sequences (seq_stats_long_kernel)
    void* out_dev = nullptr;
    void* out2_dev = nullptr;
    int64_t n_seq = 0;
    int32_t* flag = nullptr;  // one device word for the operator (centroid: "some ring has zero area")
    const gpk_seq_classes* classes = nullptr;
    double* strip_part = nullptr;  // the one-pass form's sums across strip boundaries (gpk_ringstream.hip), when the column is eligible
};
static int32_t unary_begin(const gpk_geoarray* a, size_t out_bytes, size_t out2_bytes, void* out,
                           void* out2, int32_t out_space, UnaryCtx* c) {
    if (!a || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    const int32_t* so;
    seq_view(a->d, &so, &c->n_seq);
    if (a->d.type == GPK_GEOM_POINT) c->n_seq = 0;
    const size_t stats_bytes = sizeof(double) * ST_COUNT * (size_t)c->n_seq;
    const bool stage = out_space != GPK_MEM_DEVICE;
    size_t part_bytes = 0;
    if (c->n_seq > 0) {
        const gpk_seq_classes* cl;
        GPK_TRY(seq_classes_of(a, (hipStream_t) nullptr, &cl));
        part_bytes = sizeof(double) * 12 * (size_t)cl->n_chunks;
        c->classes = cl;
    }
    const size_t strip_bytes = c->classes && c->classes->strips_ok ? sizeof(double) * 8 * (size_t)ring_stream_strips(a->d.n_coords) : 0;
    GPK_TRY(workspace().begin(align256(stats_bytes) + align256(part_bytes) + align256(strip_bytes) +
                              (stage ? align256(out_bytes) + align256(out2_bytes) : 0) + 1024));
    c->stats = (double*)workspace().take(stats_bytes ? stats_bytes : 8);
    c->long_part = part_bytes ? (double*)workspace().take(part_bytes) : nullptr;
    c->strip_part = strip_bytes ? (double*)workspace().take(strip_bytes) : nullptr;
    c->out_dev = stage ? workspace().take(out_bytes ? out_bytes : 8) : out;
    c->out2_dev = out2 ? (stage ? workspace().take(out2_bytes ? out2_bytes : 8) : out2) : nullptr;
    c->flag = (int32_t*)workspace().take(64);
    return GPK_OK;
}

__global__ void stats_to_bbox_kernel(const double* __restrict__ stats, const int32_t* __restrict__ seq_off, int64_t n_seq,
                                     double4* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seq) return;
    if (seq_off[s + 1] == seq_off[s])
        out[s] = make_double4(NAN, NAN, NAN, NAN);
    else
        out[s] = bbox_records(stats, n_seq)[s];
}

int32_t ring_bboxes(const gpk_geoarray* a, double4* out_dev, hipStream_t s) {
    const int32_t* seq_off;
    int64_t n_seq;
    seq_view(a->d, &seq_off, &n_seq);
    if (n_seq == 0) return GPK_OK;
    const gpk_seq_classes* cl;
    GPK_TRY(seq_classes_of(a, s, &cl));
    const size_t part_bytes = sizeof(double) * 12 * (size_t)cl->n_chunks;
    GPK_TRY(workspace().begin(align256(sizeof(double) * ST_COUNT * (size_t)n_seq) + align256(part_bytes) + 1024));
    double* stats = (double*)workspace().take(sizeof(double) * ST_COUNT * (size_t)n_seq);
    double* long_part = part_bytes ? (double*)workspace().take(part_bytes) : nullptr;
    GPK_TRY(launch_seq_stats<M_BBOX>(a, stats, long_part, s, "gpk_seq_bbox"));
    GPK_LAUNCH("gpk_stats_to_bbox", stats_to_bbox_kernel, grid_for(n_seq), dim3(256), 0, s, stats, seq_off, n_seq, out_dev);
    return GPK_OK;
}

}  // namespace gpk

using namespace gpk;

extern "C" {

// polygonal columns eligible for the one-pass form (gpk_ringstream.hip): the whole operator in one streaming launch + the strip-boundary fix
static bool ring_stream_ok(const gpk_geoarray* a, const UnaryCtx& c) {
    return is_polygonal(a->d.type) && c.classes && c.classes->strips_ok && c.strip_part;
}
// bounds (four running values a coordinate, four LDS updates a flush) is measured slower in the one-pass form on every test column
// (2M x 64: 0.73 against 0.44 ms; power-law: 0.23 against 0.19): it stays on the two-stage form unless GPK_RING_STREAM_BOUNDS=1 (A/B runs)
static bool ring_stream_bounds_on() {
    static const bool on = [] {
        const char* e = getenv("GPK_RING_STREAM_BOUNDS");
        return e && e[0] == '1';
    }();
    return on;
}
static int32_t ring_stream_run(int op, const gpk_geoarray* a, const UnaryCtx& c, hipStream_t s) {
    const int64_t n_strips = ring_stream_strips(a->d.n_coords);
    return ring_stream_launch(op, a->d, c.classes->strip_first, c.classes->strip_first + n_strips + 1, c.classes->strip_cross, c.classes->strip_desc, c.stats,
                              c.strip_part, (double*)c.out_dev, s);
}

// see the header: a caller that REWRITES the offsets of a handle over borrowed device buffers drops what the handle has derived from them
int32_t gpk_geoarray_invalidate(gpk_geoarray* a) {
    if (!a) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    GPK_HIP(hipDeviceSynchronize());  // (launches in flight may still read the tables)
    std::lock_guard<std::mutex> lk(g_classes_mu);
    if (a->classes) {
        if (a->classes->lists) (void)hipFree(a->classes->lists);
        if (a->classes->chunk_begin) (void)hipFree(a->classes->chunk_begin);
        if (a->classes->strip_first) (void)hipFree(a->classes->strip_first);
        if (a->classes->strip_cross) (void)hipFree(a->classes->strip_cross);
        if (a->classes->strip_desc) (void)hipFree(a->classes->strip_desc);
        delete a->classes;
        a->classes = nullptr;
    }
    return GPK_OK;
}

static int32_t area_impl(const gpk_geoarray* a, double* out, int32_t out_space, void* stream, bool is_signed) {
    UnaryCtx c;
    hipStream_t s = (hipStream_t)stream;
    const size_t ob = a ? sizeof(double) * (size_t)a->d.n_geoms : 0;
    GPK_TRY(unary_begin(a, ob, 0, out, nullptr, out_space, &c));
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    if (a->d.type == GPK_GEOM_POINT) {
        GPK_LAUNCH("gpk_point_unary", point_unary_kernel, grid_for(n), dim3(256), 0, s, a->d, 0, (double*)c.out_dev, (uint8_t*)nullptr);
    } else if (ring_stream_ok(a, c)) {
        GPK_TRY(ring_stream_run(is_signed ? RS_SIGNED_AREA : RS_AREA, a, c, s));
    } else {
        const gpk_seq_classes* cl = nullptr;
        if (is_polygonal(a->d.type)) GPK_TRY(seq_classes_of(a, s, &cl));
        if (cl && cl->one_to_one) {  // single-ring polygons: stage 1 writes the areas, no combine pass
            const FinalOut fin{(double*)c.out_dev, a->d.validity, nullptr};
            if (is_signed)
                GPK_TRY((launch_seq_stats<M_AREA, FIN_SIGNED_AREA>(a, c.stats, c.long_part, s, "gpk_ring_area", fin)));
            else
                GPK_TRY((launch_seq_stats<M_AREA, FIN_AREA>(a, c.stats, c.long_part, s, "gpk_ring_area", fin)));
            return copy_out(out, out_space, c.out_dev, ob, s);
        }
        if (is_polygonal(a->d.type)) GPK_TRY(launch_seq_stats<M_AREA>(a, c.stats, c.long_part, s, "gpk_ring_area"));
        if (is_signed)
            GPK_LAUNCH("gpk_area_combine", area_combine_kernel<true>, grid_for(n), dim3(256), 0, s, a->d, c.stats, c.n_seq, (double*)c.out_dev);
        else
            GPK_LAUNCH("gpk_area_combine", area_combine_kernel<false>, grid_for(n), dim3(256), 0, s, a->d, c.stats, c.n_seq, (double*)c.out_dev);
    }
    return copy_out(out, out_space, c.out_dev, ob, s);
}

int32_t gpk_area(const gpk_geoarray* a, double* out, int32_t out_space, void* stream) {
    return area_impl(a, out, out_space, stream, false);
}
int32_t gpk_signed_area(const gpk_geoarray* a, double* out, int32_t out_space, void* stream) {
    return area_impl(a, out, out_space, stream, true);
}

int32_t gpk_euclidean_length(const gpk_geoarray* a, double* out, int32_t out_space, void* stream) {
    UnaryCtx c;
    hipStream_t s = (hipStream_t)stream;
    const size_t ob = a ? sizeof(double) * (size_t)a->d.n_geoms : 0;
    GPK_TRY(unary_begin(a, ob, 0, out, nullptr, out_space, &c));
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    if (a->d.type == GPK_GEOM_POINT) {
        GPK_LAUNCH("gpk_point_unary", point_unary_kernel, grid_for(n), dim3(256), 0, s, a->d, 0, (double*)c.out_dev, (uint8_t*)nullptr);
    } else if (ring_stream_ok(a, c)) {
        GPK_TRY(ring_stream_run(RS_LENGTH, a, c, s));
    } else {
        const gpk_seq_classes* cl = nullptr;
        if (a->d.type != GPK_GEOM_MULTIPOINT) GPK_TRY(seq_classes_of(a, s, &cl));
        if (cl && cl->one_to_one) {
            GPK_TRY((launch_seq_stats<M_LEN, FIN_LENGTH>(a, c.stats, c.long_part, s, "gpk_seq_length", FinalOut{(double*)c.out_dev, a->d.validity, nullptr, nullptr})));
            return copy_out(out, out_space, c.out_dev, ob, s);
        }
        if (a->d.type != GPK_GEOM_MULTIPOINT) GPK_TRY(launch_seq_stats<M_LEN>(a, c.stats, c.long_part, s, "gpk_seq_length"));
        GPK_LAUNCH("gpk_length_combine", length_combine_kernel, grid_for(n), dim3(256), 0, s, a->d, c.stats, c.n_seq, (double*)c.out_dev);
    }
    return copy_out(out, out_space, c.out_dev, ob, s);
}

int32_t gpk_bounds(const gpk_geoarray* a, double* out4, int32_t out_space, void* stream) {
    UnaryCtx c;
    hipStream_t s = (hipStream_t)stream;
    const size_t ob = a ? sizeof(double) * 4 * (size_t)a->d.n_geoms : 0;
    GPK_TRY(unary_begin(a, ob, 0, out4, nullptr, out_space, &c));
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    if (a->d.type == GPK_GEOM_POINT) {
        GPK_LAUNCH("gpk_point_unary", point_unary_kernel, grid_for(n), dim3(256), 0, s, a->d, 2, (double*)c.out_dev, (uint8_t*)nullptr);
    } else if (ring_stream_ok(a, c) && ring_stream_bounds_on()) {
        GPK_TRY(ring_stream_run(RS_BOUNDS, a, c, s));
    } else {
        const gpk_seq_classes* cl;
        GPK_TRY(seq_classes_of(a, s, &cl));
        if (cl->one_to_one) {  // stage 1 writes the boxes, no combine pass
            GPK_TRY((launch_seq_stats<M_BBOX, FIN_BOUNDS>(a, c.stats, c.long_part, s, "gpk_seq_bbox", FinalOut{(double*)c.out_dev, a->d.validity, nullptr, nullptr})));
        } else {
            GPK_TRY(launch_seq_stats<M_BBOX>(a, c.stats, c.long_part, s, "gpk_seq_bbox"));
            GPK_LAUNCH("gpk_bounds_combine", bounds_combine_kernel, grid_for(n), dim3(256), 0, s, a->d, c.stats, c.n_seq, (double*)c.out_dev);
        }
    }
    return copy_out(out4, out_space, c.out_dev, ob, s);
}

int32_t gpk_centroid(const gpk_geoarray* a, double* out_xy, uint8_t* out_valid, int32_t out_space, void* stream) {
    UnaryCtx c;
    hipStream_t s = (hipStream_t)stream;
    const size_t ob = a ? sizeof(double) * 2 * (size_t)a->d.n_geoms : 0;
    const size_t vb = a ? (size_t)a->d.n_geoms : 0;
    GPK_TRY(unary_begin(a, ob, vb, out_xy, out_valid, out_space, &c));
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    if (a->d.type == GPK_GEOM_POINT) {
        GPK_LAUNCH("gpk_point_unary", point_unary_kernel, grid_for(n), dim3(256), 0, s, a->d, 1, (double*)c.out_dev, (uint8_t*)c.out2_dev);
    } else {
        const gpk_seq_classes* cl;
        GPK_TRY(seq_classes_of(a, s, &cl));
        GPK_HIP(hipMemsetAsync(c.flag, 0, sizeof(int32_t), s));
        if (cl->one_to_one) {  // stage 1 writes the centroids: no stats round trip, no combine pass
            const FinalOut fin{(double*)c.out_dev, a->d.validity, (uint8_t*)c.out2_dev, c.flag};
            if (a->d.type == GPK_GEOM_MULTIPOINT) {
                GPK_TRY((launch_seq_stats<M_SUM, FIN_CENT_MPOINT>(a, c.stats, c.long_part, s, "gpk_seq_sum", fin)));
            } else if (a->d.type == GPK_GEOM_POLYGON) {
                GPK_TRY((launch_seq_stats<M_CENT, FIN_CENT_POLY>(a, c.stats, c.long_part, s, "gpk_ring_centroid", fin)));
                GPK_TRY((launch_seq_stats<M_LENC | M_DEGEN, FIN_CENT_DEGEN>(a, c.stats, c.long_part, s, "gpk_ring_centroid_degenerate", fin)));
            } else {
                GPK_TRY((launch_seq_stats<M_LENC, FIN_CENT_LINE>(a, c.stats, c.long_part, s, "gpk_line_centroid", fin)));
            }
            if (out_valid) GPK_TRY(copy_out(out_valid, out_space, c.out2_dev, vb, s));
            return copy_out(out_xy, out_space, c.out_dev, ob, s);
        }
        if (a->d.type == GPK_GEOM_MULTIPOINT)
            GPK_TRY(launch_seq_stats<M_SUM>(a, c.stats, c.long_part, s, "gpk_seq_sum"));
        else if (is_polygonal(a->d.type)) {
            const FinalOut flag_only{nullptr, nullptr, nullptr, c.flag};
            GPK_TRY(launch_seq_stats<M_CENT>(a, c.stats, c.long_part, s, "gpk_ring_centroid", flag_only));
            // zero-area rings degrade to their linestring centroid: a second pass that skips everything else (and leaves at
            // once when the first pass saw no such ring)
            GPK_TRY(launch_seq_stats<M_LENC | M_DEGEN>(a, c.stats, c.long_part, s, "gpk_ring_centroid_degenerate", flag_only));
        } else {
            GPK_TRY(launch_seq_stats<M_LENC>(a, c.stats, c.long_part, s, "gpk_line_centroid"));
        }
        GPK_LAUNCH("gpk_centroid_combine", centroid_combine_kernel, grid_for(n), dim3(256), 0, s, a->d, c.stats, c.n_seq, (double*)c.out_dev, (uint8_t*)c.out2_dev);
    }
    if (out_valid) GPK_TRY(copy_out(out_valid, out_space, c.out2_dev, vb, s));
    return copy_out(out_xy, out_space, c.out_dev, ob, s);
}

int32_t gpk_affine_transform(const gpk_geoarray* a, const double m[6], double* out_xy, int32_t out_space, void* stream) {
    if (!a || !m || !out_xy) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_coords;
    if (n == 0) return GPK_OK;
    const size_t ob = sizeof(double) * 2 * (size_t)n;
    void* out_dev = out_xy;
    if (out_space != GPK_MEM_DEVICE) {
        GPK_TRY(workspace().begin(ob + 512));
        out_dev = workspace().take(ob);
    }
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    GPK_LAUNCH("gpk_affine", affine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a->d.xy, n, m[0], m[1], m[2], m[3], m[4], m[5], (double2*)out_dev);
    return copy_out(out_xy, out_space, out_dev, ob, s);
}

int32_t gpk_affine_transform_rows(const gpk_geoarray* a, const double* matrices, double* out_xy, int32_t out_space, void* stream) {
    return gpk::affine_rows_impl(a, matrices, out_space, out_xy, out_space, (hipStream_t)stream);
}

}  // extern "C"

// matrices in `mat_space`, output in `out_space` (gpk_affine_about_origin builds its matrices on the device whatever the output space)
int32_t gpk::affine_rows_impl(const gpk_geoarray* a, const double* matrices, int32_t mat_space, double* out_xy, int32_t out_space, hipStream_t s) {
    if (!a || !matrices || !out_xy) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    const int64_t n = a->d.n_coords, ng = a->d.n_geoms;
    if (n == 0 || ng == 0) return GPK_OK;
    const size_t ob = sizeof(double) * 2 * (size_t)n, mb = sizeof(double) * 6 * (size_t)ng;
    void* out_dev = out_xy;
    const double* mats_dev = matrices;
    if (out_space != GPK_MEM_DEVICE || mat_space != GPK_MEM_DEVICE) {
        GPK_TRY(workspace().begin(align256(ob) + align256(mb) + 512));
        if (out_space != GPK_MEM_DEVICE) out_dev = workspace().take(ob);
        if (mat_space != GPK_MEM_DEVICE) {
            double* m = (double*)workspace().take(mb);
            GPK_HIP(hipMemcpyAsync(m, matrices, mb, hipMemcpyHostToDevice, s));
            mats_dev = m;
        }
    }
    const int G = pick_group(n, ng);
    const int64_t per_block = 256 / G;
    int64_t blocks = (ng + per_block - 1) / per_block;
    const int64_t cap = (int64_t)cu_count() * 16;
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks), block(256);
    switch (G) {
    case 4: GPK_LAUNCH("gpk_affine_rows", affine_rows_kernel<4>, grid, block, 0, s, a->d, mats_dev, (double2*)out_dev); break;
    case 8: GPK_LAUNCH("gpk_affine_rows", affine_rows_kernel<8>, grid, block, 0, s, a->d, mats_dev, (double2*)out_dev); break;
    case 16: GPK_LAUNCH("gpk_affine_rows", affine_rows_kernel<16>, grid, block, 0, s, a->d, mats_dev, (double2*)out_dev); break;
    case 32: GPK_LAUNCH("gpk_affine_rows", affine_rows_kernel<32>, grid, block, 0, s, a->d, mats_dev, (double2*)out_dev); break;
    default: GPK_LAUNCH("gpk_affine_rows", affine_rows_kernel<64>, grid, block, 0, s, a->d, mats_dev, (double2*)out_dev); break;
    }
    return copy_out(out_xy, out_space, out_dev, ob, s);
}
