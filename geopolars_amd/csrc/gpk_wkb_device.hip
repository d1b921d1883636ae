// gpk_wkb_device.hip — WKB -> GeoArrow decoding on the GPU (the step right before the
// hot path; the reference pays a per-row, per-op WKB parse, geopolars/geopolars-geo/src/util.rs:27-37, and
// calls it "expensive", README.md:83).  The raw WKB column (Arrow BinaryArray<i32>: values + offsets) is
// copied to HBM once and decoded there, so the GeoArrow SoA never exists on the host:
//   scan  : one lane per row parses the headers and counts parts / rings / coordinates;
//   scans : three exclusive scans give every row its output positions;
//   fill  : one lane per row re-parses and writes coordinates and offsets.
// Handles little-endian ISO WKB and EWKB with SRID, 2D, types 1-6 with the same promotion rules as the host
// decoder (gpk_wkb.cpp); anything else (big-endian, Z/M, mixed families, truncation) is reported, never guessed.
#include <vector>

#include "gpk_device.h"
#include "gpk_scan.h"

namespace gpk {

struct WkbCursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok;
    __device__ __forceinline__ uint32_t u8() {
        if (p + 1 > end) {
            ok = false;
            return 0;
        }
        return *p++;
    }
    __device__ __forceinline__ uint32_t u32() {
        if (p + 4 > end) {
            ok = false;
            return 0;
        }
        uint32_t v;
        __builtin_memcpy(&v, p, 4);  // one (unaligned) dword load: legal on gfx9+ under HSA, little-endian like the payload
        p += 4;
        return v;
    }
    __device__ __forceinline__ double f64() {
        if (p + 8 > end) {
            ok = false;
            return 0.0;
        }
        double v;
        __builtin_memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    __device__ __forceinline__ void skip(size_t n) {
        if (p + n > end)
            ok = false;
        else
            p += n;
    }
    // header -> base type 1..6, 0 on error (big-endian, Z/M, unknown)
    __device__ __forceinline__ int header() {
        const uint32_t bo = u8();
        if (!ok || bo != 1u) {
            ok = false;
            return 0;
        }
        uint32_t t = u32();
        if (t & 0x20000000u) (void)u32();
        if (!ok || (t & 0xC0000000u)) {
            ok = false;
            return 0;
        }
        t &= 0x0FFFFFFFu;
        if (t < 1u || t > 6u) {
            ok = false;
            return 0;
        }
        return (int)t;
    }
};

struct RowCount {
    int32_t type;    // 0 = null row
    int32_t k;       // sub-geometries of a multi type (points / linestrings / polygons); 1 otherwise
    int32_t rings;   // rings (polygon family) or linestrings (line family)
    int32_t coords;  // coordinates, POINT EMPTY members of a MULTIPOINT not counted
};

// `emit` < 0: count only.  Otherwise write coordinates from position `emit` and ring / part end offsets.
template <bool FILL>
__device__ inline bool parse_row(WkbCursor& r, RowCount& rc, double2* __restrict__ xy, int64_t cpos, int32_t* __restrict__ ring_off,
                                 int64_t rpos, int32_t* __restrict__ part_off, int64_t ppos, bool out_point,
                                 const uint8_t* __restrict__ values = nullptr, int32_t* __restrict__ seq_src = nullptr,
                                 int64_t seq_pos = 0) {
    const int t = r.header();
    if (!t) return false;
    rc.type = t;
    rc.k = 1;
    rc.rings = 0;
    rc.coords = 0;
    auto put = [&](double x, double y) {
        if (FILL) xy[cpos + rc.coords] = make_double2(x, y);
        ++rc.coords;
    };
    auto end_ring = [&]() {
        ++rc.rings;
        if (FILL && ring_off) ring_off[rpos + rc.rings] = (int32_t)(cpos + rc.coords);
    };
    int parts_done = 0;
    auto end_part = [&]() {
        ++parts_done;
        if (FILL && part_off) part_off[ppos + parts_done] = (int32_t)(rpos + rc.rings);
    };
    // a run of n coordinates (linestring / ring body) is never read here: the scan only needs its length, and the fill
    // records where it starts so that wkb_copy_kernel moves it with many lanes (a 100k-vertex ring is one header to
    // the lane that owns the row)
    int seqs_done = 0;
    auto coords_run = [&](uint32_t n) {
        if (FILL && seq_src) seq_src[seq_pos + seqs_done] = (int32_t)(r.p - values);
        ++seqs_done;
        r.skip(16 * (size_t)n);
        if (r.ok) rc.coords += (int32_t)n;
    };
    auto polygon_body = [&]() {
        const uint32_t nr = r.u32();
        for (uint32_t q = 0; q < nr && r.ok; ++q) {
            const uint32_t n = r.u32();
            coords_run(n);
            end_ring();
        }
    };
    switch (t) {
    case 1: {
        const double x = r.f64(), y = r.f64();
        if (out_point || !(isnan(x) && isnan(y))) put(x, y);
        break;
    }
    case 2: {
        const uint32_t n = r.u32();
        coords_run(n);
        end_ring();  // counted as one linestring; only used when the column is MULTILINESTRING
        break;
    }
    case 3:
        polygon_body();
        end_part();
        break;
    default: {
        const uint32_t k = r.u32();
        rc.k = (int32_t)k;
        for (uint32_t m = 0; m < k && r.ok; ++m) {
            const int ct = r.header();
            if (ct != t - 3) {
                r.ok = false;
                break;
            }
            if (ct == 1) {
                const double x = r.f64(), y = r.f64();
                if (!(isnan(x) && isnan(y))) put(x, y);
            } else if (ct == 2) {
                const uint32_t n = r.u32();
                coords_run(n);
                end_ring();
            } else {
                polygon_body();
                end_part();
            }
        }
    }
    }
    return r.ok;
}

// the three quantities a row contributes to the output's offset levels.  `parts` is what a MULTIPOLYGON column counts (a polygon row is
// one part, a multipolygon row its k members); the other column types never read it.
__device__ __forceinline__ int row_parts(const RowCount& rc) { return rc.type == 0 ? 0 : (rc.type == 3 ? 1 : rc.k); }
constexpr int WKB_BLOCK = 256;
// One lane per row parses the headers and counts; the work-group also leaves its TOTALS of coordinates / rings / parts
// (block_tot[ch * (n_blocks + 1) + block]): after one small scan of those, wkb_fill_kernel finds every row's output positions with a
// block scan of the same counts — no per-row position arrays, no extent pass, no three grid-wide scans (round 3: nine more launches
// and a second read-back between them).
__global__ __launch_bounds__(WKB_BLOCK) void wkb_scan_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ offsets, int64_t n_rows,
                                                             const uint8_t* __restrict__ validity, RowCount* __restrict__ rows, uint32_t* __restrict__ flags,
                                                             unsigned long long* __restrict__ block_tot, int64_t n_blocks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    RowCount rc{0, 0, 0, 0};
    uint32_t bits = 0;
    bool parsed = false;
    if (i < n_rows && dev::valid_row(validity, i)) {
        // A plain little-endian one-ring POLYGON or LINESTRING row (no SRID word, no Z / M flags) is settled from ONE 16-byte request:
        // byte order, type, ring count / length and the ring's length sit in its first 13 bytes.  parse_row reads them with four
        // dependent loads — four L1 requests a row, and the rate of those, not the bytes, set this kernel's time on columns of small
        // polygons.  Anything else (and a row whose coordinates would not fit its bytes) goes through parse_row as before.
        const int32_t o0 = offsets[i], o1 = offsets[i + 1];
        const int64_t len = (int64_t)o1 - o0;
        if (len >= 16) {
            uint32_t w[4];
            __builtin_memcpy(w, values + o0, 16);
            const uint32_t type = (w[0] >> 8) | (w[1] << 24), a = (w[1] >> 8) | (w[2] << 24), b = (w[2] >> 8) | (w[3] << 24);
            if ((w[0] & 0xFFu) == 1u && type == 3u && a == 1u && 13 + 16 * (int64_t)b <= len && b <= 0x7FFFFFFFu / 16u) {
                rc = RowCount{3, 1, 1, (int32_t)b};
                parsed = true;
            } else if ((w[0] & 0xFFu) == 1u && type == 2u && 9 + 16 * (int64_t)a <= len && a <= 0x7FFFFFFFu / 16u) {
                rc = RowCount{2, 1, 1, (int32_t)a};
                parsed = true;
            }
        }
        if (parsed) bits = 1u << rc.type;
    }
    if (i < n_rows && dev::valid_row(validity, i) && !parsed) {
        WkbCursor r{values + offsets[i], values + offsets[i + 1], true};
        if (!parse_row<false>(r, rc, nullptr, 0, nullptr, 0, nullptr, 0, false)) {
            bits = 0x80000000u;  // malformed / unsupported
            rc = RowCount{0, 0, 0, 0};
        } else {
            bits = 1u << rc.type;
        }
    }
    if (i < n_rows) rows[i] = rc;
    // the column's type flags: OR-ed across the wave, and only a wave that brings a NEW bit touches the word (one atomicOr per row on
    // one address was 8M serialised atomics for a column of 8M small polygons — most of this kernel's 0.73 ms)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, o, 64);
    if ((threadIdx.x & 63) == 0 && bits && (bits & ~__atomic_load_n(flags, __ATOMIC_RELAXED)) != 0u) atomicOr(flags, bits);
    {  // the longest row (flags[1]): what picks the form of the coordinate copy — again only a wave that raises it touches the word
        uint32_t mx = (uint32_t)rc.coords;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        if ((threadIdx.x & 63) == 0 && mx > __atomic_load_n(flags + 1, __ATOMIC_RELAXED)) atomicMax(flags + 1, mx);
    }
    // totals of the work-group's rows
    __shared__ unsigned long long s_tot[WKB_BLOCK / 64][3];
    unsigned long long c = (unsigned long long)rc.coords, g = (unsigned long long)rc.rings, q = (unsigned long long)row_parts(rc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o, 64);
        g += __shfl_xor(g, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_tot[threadIdx.x >> 6][0] = c;
        s_tot[threadIdx.x >> 6][1] = g;
        s_tot[threadIdx.x >> 6][2] = q;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < WKB_BLOCK / 64; ++w) t += s_tot[w][threadIdx.x];
        block_tot[(int64_t)threadIdx.x * (n_blocks + 1) + blockIdx.x] = t;
    }
}
// the three channels' block totals -> exclusive offsets in place; channel ch's grand total lands in its slot n_blocks and in
// totals_host[ch] (device-mapped host memory: read after the stream sync, no copy)
__global__ __launch_bounds__(1024) void wkb_totals_kernel(unsigned long long* __restrict__ block_tot, int64_t n_blocks, unsigned long long* __restrict__ totals_host,
                                                          const uint32_t* __restrict__ flags) {
    __shared__ unsigned long long lds[17];
    unsigned long long* v = block_tot + (int64_t)blockIdx.x * (n_blocks + 1);
    unsigned long long carry = 0;
    for (int64_t base = 0; base < n_blocks; base += 1024) {
        const int64_t j = base + threadIdx.x;
        const unsigned long long x = j < n_blocks ? v[j] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = dev::block_exclusive_scan<unsigned long long, 1024>(x, lds, &tot);
        if (j < n_blocks) v[j] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        v[n_blocks] = carry;
        totals_host[blockIdx.x] = carry;
        if (blockIdx.x == 0) totals_host[3] = (unsigned long long)flags[0] | ((unsigned long long)flags[1] << 32);
    }
}

// One lane per row parses again and writes offsets (and, for point columns, coordinates).  A row's output positions = the
// work-group's offsets (wkb_totals_kernel) + a block scan of the rows' counts (wkb_scan_kernel left them in `rows`).
__global__ __launch_bounds__(WKB_BLOCK) void wkb_fill_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ offsets, int64_t n_rows,
                                                             const uint8_t* __restrict__ validity, int out_type, const RowCount* __restrict__ rows,
                                                             const unsigned long long* __restrict__ block_off, int64_t n_blocks, double2* __restrict__ xy,
                                                             int32_t* __restrict__ geom_off, int32_t* __restrict__ part_off, int32_t* __restrict__ ring_off,
                                                             int32_t* __restrict__ seq_src) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool out_point = out_type == GPK_GEOM_POINT;
    const bool has_ring = out_type == GPK_GEOM_MULTILINESTRING || out_type == GPK_GEOM_POLYGON || out_type == GPK_GEOM_MULTIPOLYGON;
    const bool has_part = out_type == GPK_GEOM_MULTIPOLYGON;
    const RowCount mine = i < n_rows ? rows[i] : RowCount{0, 0, 0, 0};
    __shared__ unsigned long long lds[WKB_BLOCK / 64 + 1];
    unsigned long long tot;
    const int nc = mine.coords, nr = has_ring ? mine.rings : 0, np = has_part ? row_parts(mine) : 0;
    const int64_t cpos = out_point ? i : (int64_t)(block_off[blockIdx.x] + dev::block_exclusive_scan<unsigned long long, WKB_BLOCK>((unsigned long long)nc, lds, &tot));
    const int64_t rpos = has_ring ? (int64_t)(block_off[(n_blocks + 1) + blockIdx.x] + dev::block_exclusive_scan<unsigned long long, WKB_BLOCK>((unsigned long long)nr, lds, &tot)) : 0;
    const int64_t ppos = has_part ? (int64_t)(block_off[2 * (n_blocks + 1) + blockIdx.x] + dev::block_exclusive_scan<unsigned long long, WKB_BLOCK>((unsigned long long)np, lds, &tot)) : 0;
    if (i >= n_rows) return;
    const int32_t o0 = offsets[i], o1 = offsets[i + 1];
    const int64_t len = (int64_t)o1 - o0;
    // One-ring polygons and linestrings whose row is exactly header + coordinates need no second look at their bytes: the scan pass
    // parsed the row inside [o0, o1), so a length of 13 (9) + 16 n leaves no room for an SRID word or anything else — the ring's
    // coordinates start at o0 + 13 (o0 + 9).  (Columns of building footprints: the fill pass read a cache line per row for this.)
    if (dev::valid_row(validity, i) && mine.type == 3 && mine.rings == 1 && has_ring && len == 13 + 16 * (int64_t)nc) {
        ring_off[rpos + 1] = (int32_t)(cpos + nc);
        if (has_part) part_off[ppos + 1] = (int32_t)(rpos + 1);
        if (seq_src) seq_src[rpos] = o0 + 13;
    } else if (dev::valid_row(validity, i) && mine.type == 2 && !out_point && len == 9 + 16 * (int64_t)nc) {
        if (has_ring) ring_off[rpos + 1] = (int32_t)(cpos + nc);
        if (seq_src) seq_src[has_ring ? rpos : i] = o0 + 9;
    } else if (dev::valid_row(validity, i)) {
        WkbCursor r{values + o0, values + o1, true};
        RowCount rc;
        // sequences are numbered like the rings (ring types) or like the rows (a LINESTRING column: one per row)
        (void)parse_row<true>(r, rc, xy, cpos, has_ring ? ring_off : nullptr, rpos, has_part ? part_off : nullptr, ppos, out_point, values, seq_src,
                              has_ring ? rpos : i);
    } else if (seq_src && !has_ring) {
        seq_src[i] = 0;  // null row of a LINESTRING column: an empty sequence
    } else if (out_point) {
        xy[cpos] = make_double2(NAN, NAN);
    }
    if (i == 0) {
        if (geom_off) geom_off[0] = 0;
        if (has_ring) ring_off[0] = 0;
        if (has_part) part_off[0] = 0;
    }
    if (geom_off)  // level-1 offsets: end position of row i at the column's first nesting level
        geom_off[i + 1] = (int32_t)(has_part ? ppos + np : (has_ring ? rpos + nr : cpos + nc));
}

// ---- coordinate runs: WKB bytes -> xy ---------------------------------------------------------------------------
// sequence q (a ring / member line, or the row of a LINESTRING column) starts at values + seq_src[q] and fills
// xy[seq_off[q] .. seq_off[q + 1]).  The work is cut by OUTPUT coordinate, not by sequence: a work-group owns WKB_CP_TILE
// consecutive coordinates of xy whatever sequences they belong to (a 100k-vertex ring and a thousand triangles cost the same per
// coordinate — eight lanes per sequence left half the lanes idle on 4-coordinate rings and one group alone on a long one: 1.8 TB/s
// on a power-law column against 3.7 on 64-vertex rings).  The sequences that meet the tile are found with two binary searches,
// their offsets staged in LDS, and every lane finds the sequence of each of its coordinates there.
__device__ __forceinline__ double2 load_xy_unaligned(const uint8_t* p) {
    double2 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
constexpr int WKB_CP_BLOCK = 256, WKB_CP_PER = 8, WKB_CP_TILE = WKB_CP_BLOCK * WKB_CP_PER, WKB_CP_SEQS = 2560;
// tile_seq[t] = the sequence that holds the first coordinate of tile t (tile_seq[n_tiles] = the last sequence), one LANE per tile:
// the copy's work-groups used to open with two binary searches over all the offsets — some twenty DEPENDENT loads before a work-group
// of a few microseconds' work could start (the chain, not the bytes, set the kernel's time on columns of many short sequences)
__global__ void wkb_tile_seq_kernel(const int32_t* __restrict__ seq_off, int64_t n_seq, int64_t n_tiles, int32_t* __restrict__ tile_seq) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) {
        tile_seq[t] = (int32_t)(n_seq - 1);
        return;
    }
    const int64_t c = t * WKB_CP_TILE;
    int64_t lo = 0, hi = n_seq;  // invariant: seq_off[lo] <= c < seq_off[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)seq_off[mid] <= c)
            lo = mid;
        else
            hi = mid;
    }
    tile_seq[t] = (int32_t)lo;
}
__global__ __launch_bounds__(WKB_CP_BLOCK) void wkb_copy_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ seq_src,
                                                                const int32_t* __restrict__ seq_off, int64_t n_seq, int64_t n_coords, double2* __restrict__ xy,
                                                                const int32_t* __restrict__ tile_seq) {
    __shared__ int32_t s_off[WKB_CP_SEQS + 1], s_src[WKB_CP_SEQS];
    const int64_t c_lo = (int64_t)blockIdx.x * WKB_CP_TILE, c_hi = c_lo + WKB_CP_TILE < n_coords ? c_lo + WKB_CP_TILE : n_coords;
    // last sequence that starts at or before coordinate c: upper_bound(seq_off[0 .. n_seq], c) - 1 (empty sequences share an offset
    // with their successor: the LAST of them is the one that holds the coordinate)
    auto seq_of = [&](int64_t c) {
        int64_t lo = 0, hi = n_seq;  // invariant: seq_off[lo] <= c < seq_off[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)seq_off[mid] <= c)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    };
    // the sequences that meet the tile: from its first coordinate's to the next tile's first coordinate's (the last sequence for the
    // last tile) — at most one more than meet it, and s_off[nq] = the start of the sequence after them stays beyond every coordinate
    const int q0 = tile_seq[blockIdx.x], q1 = tile_seq[blockIdx.x + 1], nq = q1 - q0 + 1;
    const bool staged = nq <= WKB_CP_SEQS;  // (more: a run of empty sequences inside the tile — the lanes search the global offsets)
    if (staged) {
        for (int t = threadIdx.x; t <= nq; t += WKB_CP_BLOCK) s_off[t] = seq_off[q0 + t];
        for (int t = threadIdx.x; t < nq; t += WKB_CP_BLOCK) s_src[t] = seq_src[q0 + t];
    }
    __syncthreads();
    double2 v[WKB_CP_PER];
    int64_t cs[WKB_CP_PER];
#pragma unroll
    for (int j = 0; j < WKB_CP_PER; ++j) {  // all loads of a lane in flight before the first store
        const int64_t c = c_lo + j * WKB_CP_BLOCK + threadIdx.x;
        cs[j] = c;
        if (c >= c_hi) continue;
        int64_t src;
        if (staged) {
            // (marking every sequence's first coordinate and carrying the marks forward with a running maximum — no search — was
            // measured: the four barriers of that scan cost more than eleven LDS reads per coordinate, 1.11 against 0.94 ms on 2M x 64)
            int lo = 0, hi = nq;  // s_off[lo] <= c < s_off[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((int64_t)s_off[mid] <= c)
                    lo = mid;
                else
                    hi = mid;
            }
            src = (int64_t)s_src[lo] + 16 * (c - (int64_t)s_off[lo]);
        } else {
            const int64_t q = seq_of(c);
            src = (int64_t)seq_src[q] + 16 * (c - (int64_t)seq_off[q]);
        }
        v[j] = load_xy_unaligned(values + src);
    }
#pragma unroll
    for (int j = 0; j < WKB_CP_PER; ++j)
        if (cs[j] < c_hi) xy[cs[j]] = v[j];
}

// Columns of SHORT sequences (building footprints: a dozen coordinates per ring) keep the lane-group form — 8 lanes per sequence, no
// search: 0.49 ms against 0.60 on 8M x 8-vertex polygons; sequences beyond WKB_LONG are listed for wkb_copy_long_kernel.
constexpr int WKB_COPY_GS = 8, WKB_LONG = 4096;
__global__ __launch_bounds__(256) void wkb_copy_groups_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ seq_src,
                                                        const int32_t* __restrict__ seq_off, int64_t n_seq, double2* __restrict__ xy,
                                                        int32_t* __restrict__ long_list) {
    const int lane = threadIdx.x & (WKB_COPY_GS - 1);
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WKB_COPY_GS;
    if (q >= n_seq) return;
    const int c0 = seq_off[q], n = seq_off[q + 1] - c0;
    if (n <= 0) return;
    if (n > WKB_LONG) {
        if (lane == 0) long_list[1 + atomicAdd(&long_list[0], 1)] = (int32_t)q;
        return;
    }
    const uint8_t* src = values + seq_src[q];
    for (int i = lane; i < n; i += WKB_COPY_GS) xy[c0 + i] = load_xy_unaligned(src + 16 * (size_t)i);
}
__global__ __launch_bounds__(256) void wkb_copy_long_kernel(const uint8_t* __restrict__ values, const int32_t* __restrict__ seq_src,
                                                             const int32_t* __restrict__ seq_off, const int32_t* __restrict__ long_list,
                                                             double2* __restrict__ xy) {
    const int n_long = long_list[0];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int k = 0; k < n_long; ++k) {
        const int q = long_list[1 + k];
        const int c0 = seq_off[q], n = seq_off[q + 1] - c0;
        const uint8_t* src = values + seq_src[q];
        for (int64_t i = tid; i < n; i += stride) xy[c0 + i] = load_xy_unaligned(src + 16 * (size_t)i);
    }
}

}  // namespace gpk

using namespace gpk;

extern "C" int32_t gpk_geoarray_from_wkb(const uint8_t* wkb_values, const int32_t* wkb_offsets, int64_t n_rows, const uint8_t* validity,
                                         int32_t mem_space, void* stream, gpk_geoarray** out, int32_t* out_geom_type) {
    if (!out || !wkb_offsets || (n_rows > 0 && !wkb_values)) return fail(GPK_ERR_INVALID_ARGUMENT, "gpk_geoarray_from_wkb: NULL argument");
    *out = nullptr;
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    if (n_rows > INT32_MAX) return fail(GPK_ERR_INVALID_OFFSETS, "column exceeds i32 offsets");

    // temporaries of this call (freed on every path); the decoded buffers are owned by the returned handle
    void* tmp[16] = {nullptr};
    int n_tmp = 0;
    gpk_geoarray* a = nullptr;
    auto done = [&](int32_t rc) {
        if (n_tmp) (void)hipStreamSynchronize(s);
        for (int i = 0; i < n_tmp; ++i) cached_free(tmp[i]);
        if (rc != GPK_OK && a) gpk_geoarray_free(a);
        return rc;
    };
    // temporaries come from the thread's auxiliary arena while they fit (capped: mapping gigabytes costs more than the
    // allocations it replaces); the decoded buffers are the handle's own allocations
    {
        size_t est = (size_t)n_rows * 64 + (1u << 20);
        if (est > (size_t(256) << 20)) est = size_t(256) << 20;
        (void)workspace_aux(1).begin(est);
    }
    auto dalloc = [&](void** p, size_t bytes, bool temporary) -> int32_t {
        if (temporary) {
            *p = workspace_aux(1).take(bytes ? bytes : 8);
            if (*p) return GPK_OK;
        }
        // (the decoded buffers come from the library's block cache like index tables do: a dataframe pipeline decodes and drops
        // columns all the time, and five hipMalloc + five hipFree were half of this call's wall time)
        hipError_t e = cached_malloc(p, bytes ? bytes : 8);
        if (e != hipSuccess) return fail(GPK_ERR_OOM, "gpk_geoarray_from_wkb: hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        if (temporary) tmp[n_tmp++] = *p;
        return GPK_OK;
    };
#define W_TRY(x)                              \
    do {                                      \
        int32_t _rc = (x);                    \
        if (_rc != GPK_OK) return done(_rc);  \
    } while (0)
#define W_HIP(x)                                                                                        \
    do {                                                                                                \
        hipError_t _e = (x);                                                                            \
        if (_e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "%s: %s", #x, hipGetErrorString(_e)));   \
    } while (0)

    const uint8_t* values_dev = wkb_values;
    const int32_t* offsets_dev = wkb_offsets;
    const uint8_t* validity_dev = validity;
    if (mem_space == GPK_MEM_HOST) {
        const size_t vbytes = n_rows > 0 ? (size_t)wkb_offsets[n_rows] : 0;
        void *v = nullptr, *o = nullptr, *vd = nullptr;
        W_TRY(dalloc(&v, vbytes, true));
        W_TRY(dalloc(&o, sizeof(int32_t) * (size_t)(n_rows + 1), true));
        W_HIP(hipMemcpyAsync(v, wkb_values, vbytes, hipMemcpyHostToDevice, s));
        W_HIP(hipMemcpyAsync(o, wkb_offsets, sizeof(int32_t) * (size_t)(n_rows + 1), hipMemcpyHostToDevice, s));
        values_dev = (const uint8_t*)v;
        offsets_dev = (const int32_t*)o;
        if (validity) {
            W_TRY(dalloc(&vd, (size_t)((n_rows + 7) / 8), true));
            W_HIP(hipMemcpyAsync(vd, validity, (size_t)((n_rows + 7) / 8), hipMemcpyHostToDevice, s));
            validity_dev = (const uint8_t*)vd;
        }
    }
    RowCount* rows = nullptr;
    uint32_t* flags = nullptr;
    unsigned long long* btot = nullptr;
    const int64_t n_blocks = (n_rows + WKB_BLOCK - 1) / WKB_BLOCK > 0 ? (n_rows + WKB_BLOCK - 1) / WKB_BLOCK : 1;
    W_TRY(dalloc((void**)&rows, sizeof(RowCount) * (size_t)(n_rows ? n_rows : 1), true));
    W_TRY(dalloc((void**)&flags, 64, true));
    W_TRY(dalloc((void**)&btot, sizeof(unsigned long long) * (size_t)(3 * (n_blocks + 1)), true));
    W_HIP(hipMemsetAsync(flags, 0, 64, s));
    // the one read-back of the call: the column's type flags and the three totals, through device-mapped host memory
    static thread_local unsigned long long* totals_host = nullptr;
    if (!totals_host && hipHostMalloc((void**)&totals_host, 64, hipHostMallocMapped) != hipSuccess) totals_host = nullptr;
    if (!totals_host) return done(fail(GPK_ERR_OOM, "gpk_geoarray_from_wkb: hipHostMalloc failed"));
    const dim3 grid((unsigned)n_blocks), block(WKB_BLOCK);
    totals_host[0] = totals_host[1] = totals_host[2] = totals_host[3] = 0;
    auto launch1 = [&]() -> int32_t {
        if (n_rows > 0) {
            GPK_LAUNCH("gpk_wkb_scan", wkb_scan_kernel, grid, block, 0, s, values_dev, offsets_dev, n_rows, validity_dev, rows, flags, btot, n_blocks);
            GPK_LAUNCH("gpk_wkb_totals", wkb_totals_kernel, dim3(3), dim3(1024), 0, s, btot, n_blocks, totals_host, (const uint32_t*)flags);
        }
        return GPK_OK;
    };
    W_TRY(launch1());
    W_HIP(hipStreamSynchronize(s));
    const uint32_t hflags = (uint32_t)((volatile unsigned long long*)totals_host)[3];
    const uint32_t longest_row = (uint32_t)(((volatile unsigned long long*)totals_host)[3] >> 32);
    if (hflags & 0x80000000u) {
        // The column holds rows this decoder does not read — big-endian records, Z / M ordinates (EWKB flags or ISO 1000-codes) — or
        // malformed ones.  A HOST column is parsed by the host decoder instead (gpk_wkb_decode reads both byte orders and drops Z / M, as
        // geozero's to_geo does for the reference, util.rs:27-37) and uploaded: the caller gets its handle either way, the exotic
        // encodings just do not get the GPU's parse rate.  A DEVICE column is reported: its bytes are not the host's to read.
        if (mem_space != GPK_MEM_HOST)
            return done(fail(GPK_ERR_MISMATCHED_GEOMETRY, "gpk_geoarray_from_wkb: malformed, big-endian or Z/M WKB in a device column (decode it with gpk_wkb_decode on the host)"));
        int64_t counts[5] = {0, 0, 0, 0, 0};
        W_TRY(gpk_wkb_decode(wkb_values, wkb_offsets, n_rows, validity, counts, nullptr, nullptr, nullptr, nullptr));
        const int ht = (int)counts[0];
        std::vector<double> hxy((size_t)(2 * counts[4] + 2));
        std::vector<int32_t> hg((size_t)(n_rows + 1)), hp((size_t)(counts[2] + 1)), hr((size_t)(counts[3] + 1));
        W_TRY(gpk_wkb_decode(wkb_values, wkb_offsets, n_rows, validity, counts, hxy.data(), hg.data(), hp.data(), hr.data()));
        gpk_geoarrow_desc hd;
        memset(&hd, 0, sizeof hd);
        hd.geom_type = ht;
        hd.mem_space = GPK_MEM_HOST;
        hd.n_geoms = n_rows;
        hd.n_coords = counts[4];
        hd.xy = hxy.data();
        hd.validity = validity;
        const bool h_ring = ht == GPK_GEOM_MULTILINESTRING || ht == GPK_GEOM_POLYGON || ht == GPK_GEOM_MULTIPOLYGON;
        if (ht != GPK_GEOM_POINT) hd.geom_offsets = hg.data();
        if (ht == GPK_GEOM_MULTIPOLYGON) {
            hd.part_offsets = hp.data();
            hd.n_parts = counts[2];
        }
        if (h_ring) {
            hd.ring_offsets = hr.data();
            hd.n_rings = counts[3];
        }
        W_TRY(gpk_geoarray_upload(&hd, stream, out));
        if (out_geom_type) *out_geom_type = ht;
        return done(GPK_OK);
    }
    const bool fp = hflags & ((1u << 1) | (1u << 4)), fl = hflags & ((1u << 2) | (1u << 5)), fg = hflags & ((1u << 3) | (1u << 6));
    if ((int)fp + (int)fl + (int)fg > 1)
        return done(fail(GPK_ERR_MISMATCHED_GEOMETRY, "gpk_geoarray_from_wkb: mixed geometry families in one column"));
    const bool multi = hflags & ((1u << 4) | (1u << 5) | (1u << 6));
    int out_type = GPK_GEOM_POINT;
    if (fl) out_type = multi ? GPK_GEOM_MULTILINESTRING : GPK_GEOM_LINESTRING;
    if (fg) out_type = multi ? GPK_GEOM_MULTIPOLYGON : GPK_GEOM_POLYGON;
    if (fp) out_type = multi ? GPK_GEOM_MULTIPOINT : GPK_GEOM_POINT;
    const unsigned long long t_c = ((volatile unsigned long long*)totals_host)[0], t_r = ((volatile unsigned long long*)totals_host)[1],
                             t_p = ((volatile unsigned long long*)totals_host)[2];
    if (t_c > (unsigned long long)INT32_MAX || t_r > (unsigned long long)INT32_MAX || t_p > (unsigned long long)INT32_MAX)
        return done(fail(GPK_ERR_INVALID_OFFSETS, "gpk_geoarray_from_wkb: the decoded column exceeds i32 offsets"));
    // (a POINT column owns one coordinate slot per row, null and empty rows included)
    const int32_t tot_c = out_type == GPK_GEOM_POINT ? (int32_t)n_rows : (int32_t)t_c, tot_r = (int32_t)t_r, tot_p = (int32_t)t_p;

    a = new gpk_geoarray;
    memset(a, 0, sizeof *a);
    W_HIP(hipGetDevice(&a->device));
    const bool has_ring = out_type == GPK_GEOM_MULTILINESTRING || out_type == GPK_GEOM_POLYGON || out_type == GPK_GEOM_MULTIPOLYGON;
    const bool has_part = out_type == GPK_GEOM_MULTIPOLYGON;
    a->d.type = out_type;
    a->d.n_geoms = n_rows;
    a->d.n_coords = tot_c;
    a->d.n_rings = has_ring ? tot_r : 0;
    a->d.n_parts = has_part ? tot_p : (is_polygonal(out_type) ? n_rows : 0);
    double2* xy = nullptr;
    int32_t *go = nullptr, *po = nullptr, *ro = nullptr;
    W_TRY(dalloc((void**)&xy, sizeof(double2) * (size_t)tot_c, false));
    a->owned[0] = xy;
    if (out_type != GPK_GEOM_POINT) {
        W_TRY(dalloc((void**)&go, sizeof(int32_t) * (size_t)(n_rows + 1), false));
        a->owned[1] = go;
    }
    if (has_part) {
        W_TRY(dalloc((void**)&po, sizeof(int32_t) * (size_t)(tot_p + 1), false));
        a->owned[2] = po;
    }
    if (has_ring) {
        W_TRY(dalloc((void**)&ro, sizeof(int32_t) * (size_t)(tot_r + 1), false));
        a->owned[3] = ro;
    }
    uint8_t* vcopy = nullptr;
    if (validity_dev) {
        W_TRY(dalloc((void**)&vcopy, (size_t)((n_rows + 7) / 8), false));
        a->owned[4] = vcopy;
        W_HIP(hipMemcpyAsync(vcopy, validity_dev, (size_t)((n_rows + 7) / 8), hipMemcpyDeviceToDevice, s));
    }
    // coordinate runs of line / polygon columns are moved by wkb_copy_kernel from the positions the fill records
    const bool has_seq = out_type == GPK_GEOM_LINESTRING || has_ring;
    const int64_t n_seq = has_ring ? (int64_t)tot_r : n_rows;
    int32_t *seq_src = nullptr, *long_list = nullptr, *tile_seq = nullptr;
    // the copy's form: by output coordinate (balanced whatever the lengths), or — columns whose sequences average at most 16
    // coordinates and hold no long row — 8 lanes per sequence
    // (and no row longer than 64: a power-law column averages 9 coordinates a ring and still wants the balanced form — 0.22 against 0.54 ms)
    const bool short_seqs = has_seq && n_seq > 0 && (int64_t)tot_c <= 16 * n_seq && longest_row <= 64u;
    if (has_seq && n_rows > 0) {
        W_TRY(dalloc((void**)&seq_src, sizeof(int32_t) * (size_t)(n_seq + 1), true));
        if (short_seqs) {
            W_TRY(dalloc((void**)&long_list, sizeof(int32_t) * (size_t)(n_seq + 2), true));
            W_HIP(hipMemsetAsync(long_list, 0, sizeof(int32_t), s));
        } else {
            W_TRY(dalloc((void**)&tile_seq, sizeof(int32_t) * (size_t)(((int64_t)tot_c + WKB_CP_TILE - 1) / WKB_CP_TILE + 2), true));
        }
    }
    auto launch3 = [&]() -> int32_t {
        if (n_rows > 0) {
            GPK_LAUNCH("gpk_wkb_fill", wkb_fill_kernel, grid, block, 0, s, values_dev, offsets_dev, n_rows, validity_dev, out_type, (const RowCount*)rows,
                       (const unsigned long long*)btot, n_blocks, xy, go, po, ro, seq_src);
            if (has_seq && n_seq > 0 && tot_c > 0) {
                const int32_t* seq_off = has_ring ? (const int32_t*)ro : (const int32_t*)go;
                if (short_seqs) {
                    GPK_LAUNCH("gpk_wkb_copy", wkb_copy_groups_kernel, dim3((unsigned)((n_seq * WKB_COPY_GS + 255) / 256)), dim3(256), 0, s, values_dev,
                               (const int32_t*)seq_src, seq_off, n_seq, xy, long_list);
                    GPK_LAUNCH("gpk_wkb_copy_long", wkb_copy_long_kernel, dim3((unsigned)(cu_count() * 8)), dim3(256), 0, s, values_dev,
                               (const int32_t*)seq_src, seq_off, (const int32_t*)long_list, xy);
                } else {
                    const int64_t n_tiles = ((int64_t)tot_c + WKB_CP_TILE - 1) / WKB_CP_TILE;
                    GPK_LAUNCH("gpk_wkb_tile_seq", wkb_tile_seq_kernel, dim3((unsigned)((n_tiles + 256) / 256)), dim3(256), 0, s, seq_off, n_seq, n_tiles, tile_seq);
                    GPK_LAUNCH("gpk_wkb_copy", wkb_copy_kernel, dim3((unsigned)n_tiles), dim3(WKB_CP_BLOCK), 0, s, values_dev, (const int32_t*)seq_src, seq_off, n_seq,
                               (int64_t)tot_c, xy, (const int32_t*)tile_seq);
                }
            }
        } else if (go) {
            GPK_HIP(hipMemsetAsync(go, 0, sizeof(int32_t), s));
        }
        return GPK_OK;
    };
    W_TRY(launch3());
    W_HIP(hipStreamSynchronize(s));
#undef W_TRY
#undef W_HIP
    a->d.xy = xy;
    a->d.geom_off = go;
    a->d.part_off = po;
    a->d.ring_off = ro;
    a->d.validity = vcopy;
    a->nbytes = (int64_t)(sizeof(double2) * (size_t)tot_c + (go ? sizeof(int32_t) * (size_t)(n_rows + 1) : 0) +
                          (po ? sizeof(int32_t) * (size_t)(tot_p + 1) : 0) + (ro ? sizeof(int32_t) * (size_t)(tot_r + 1) : 0));
    if (out_geom_type) *out_geom_type = out_type;
    *out = a;
    return done(GPK_OK);
}

// Copy a device-resident array back as host GeoArrow buffers (tests, and callers that want the decoded column).
// sizes[4] = {n_coords, n_parts, n_rings, n_geoms}; pass NULL buffers to query sizes only.
extern "C" int32_t gpk_geoarray_download(const gpk_geoarray* a, int64_t sizes[4], double* xy, int32_t* geom_offsets, int32_t* part_offsets,
                                         int32_t* ring_offsets, void* stream) {
    if (!a || !sizes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    sizes[0] = a->d.n_coords;
    sizes[1] = a->d.type == GPK_GEOM_MULTIPOLYGON ? a->d.n_parts : 0;
    sizes[2] = a->d.n_rings;
    sizes[3] = a->d.n_geoms;
    if (xy && a->d.n_coords) GPK_HIP(hipMemcpyAsync(xy, a->d.xy, sizeof(double2) * (size_t)a->d.n_coords, hipMemcpyDeviceToHost, s));
    if (geom_offsets && a->d.geom_off)
        GPK_HIP(hipMemcpyAsync(geom_offsets, a->d.geom_off, sizeof(int32_t) * (size_t)(a->d.n_geoms + 1), hipMemcpyDeviceToHost, s));
    if (part_offsets && a->d.part_off)
        GPK_HIP(hipMemcpyAsync(part_offsets, a->d.part_off, sizeof(int32_t) * (size_t)(a->d.n_parts + 1), hipMemcpyDeviceToHost, s));
    if (ring_offsets && a->d.ring_off)
        GPK_HIP(hipMemcpyAsync(ring_offsets, a->d.ring_off, sizeof(int32_t) * (size_t)(a->d.n_rings + 1), hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}
