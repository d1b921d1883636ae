// gpk_pipindex.hip — builds the point-in-polygon accelerator of a polygonal right side (PipView,
// gpk_index.h): per-ring edge slabs on the raster rows, then the fine raster whose cells say
// "nothing here" / "strictly inside part p" / "an edge of part p may pass: test exactly".
//
// It belongs to SpatialIndex::try_from(&Series) (geopolars/src/spatial_index.rs:320-334): it is
// built once per right side and shared, like `Arc<SpatialIndex>` (spatial_index.rs:20-21).
//
// Why the raster labels are exact (not approximate):
//   * rows/columns are assigned to points and to edge endpoints by the SAME monotone function
//     (dev::cell_of), so an edge whose closed y-range contains p.y is registered in p's row slab;
//   * a cell is marked "boundary" for part q whenever an edge of q is not strictly on one side of the
//     cell's rectangle padded by 2^-16 of a cell (exact orientation tests on the four corners); the
//     padding dwarfs the rounding of the cell function (guarded: the accelerator is not built when
//     coordinates are so large relative to the extent that it would not);
//   * an unmarked cell is connected and meets no edge of q, so the exact position of its centre with
//     respect to q (holes included) is the position of every point that maps to the cell;
//   * border cells (where out-of-extent points are clamped) never get an "inside" label.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <chrono>

#include "gpk_device.h"
#include "gpk_index.h"
#include "gpk_pip.h"
#include "gpk_scan.h"

namespace gpk {

struct FineGrid {
    int R;
    double rx0, ry0, fw, fh, inv_fw, inv_fh, pad_x, pad_y;
};
__device__ __forceinline__ int fcol(const FineGrid& f, double x) { return dev::cell_of(x, f.rx0, f.inv_fw, f.R); }
__device__ __forceinline__ int frow(const FineGrid& f, double y) { return dev::cell_of(y, f.ry0, f.inv_fh, f.R); }

__global__ void part_geom_kernel(DevGeo a, uint32_t* __restrict__ part_geom) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_geoms) return;
    for (int p = a.geom_off[g]; p < a.geom_off[g + 1]; ++p) part_geom[p] = (uint32_t)g;
}
__global__ void ring_part_kernel(DevGeo a, int64_t n_parts, int32_t* __restrict__ ring_part) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts) return;
    int r0, r1;
    dev::part_rings(a, (int)p, r0, r1);
    for (int r = r0; r < r1; ++r) ring_part[r] = (int32_t)p;
}

__global__ void part_info_kernel(DevGeo a, int64_t n_parts, const int32_t* __restrict__ row0, const int32_t* __restrict__ slab_base,
                                 PartInfo* __restrict__ info) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts) return;
    int r0, r1;
    dev::part_rings(a, (int)p, r0, r1);
    PartInfo pi{0, 0, 0, r1 - r0};
    if (r1 > r0) {
        pi.slab_base = slab_base[r0];
        pi.row0 = row0[r0];
        pi.nrows = slab_base[r0 + 1] - slab_base[r0];
    }
    info[p] = pi;
}

__global__ void part_box_kernel(DevGeo a, int64_t n_parts, const double4* __restrict__ ring_bbox, float4* __restrict__ box) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts) return;
    int r0, r1;
    dev::part_rings(a, (int)p, r0, r1);
    const float nan = __builtin_nanf("");
    float4 b = make_float4(nan, nan, nan, nan);  // no exterior / empty exterior: no point is inside
    if (r1 > r0) {
        const double4 d = ring_bbox[r0];
        auto down = [](double v) {
            float f = (float)v;
            return (double)f > v ? nextafterf(f, -INFINITY) : f;
        };
        auto up = [](double v) {
            float f = (float)v;
            return (double)f < v ? nextafterf(f, INFINITY) : f;
        };
        b = make_float4(down(d.x), down(d.y), up(d.z), up(d.w));
    }
    box[p] = b;
}

// Slab rows of every ring.  f = the FINEST row grid (R * PIP_SLAB_MUL << PIP_FINE_LOG2 rows).  The ring's shift is the
// smallest one (up to max_shift) that brings the expected slab — edges / rows, exact for a ring whose edges are short
// against a row — down to SLAB_TARGET edges.
constexpr int SLAB_TARGET = 12;
__global__ void ring_rows_kernel(const double4* __restrict__ ring_bbox, int64_t n_rings, const int32_t* __restrict__ ring_off, FineGrid f,
                                 int max_shift, int32_t* __restrict__ row0, int32_t* __restrict__ nrows, int32_t* __restrict__ n_refined) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rings) return;
    const double4 b = ring_bbox[r];
    if (!(b.x == b.x) || !(b.y == b.y) || !(b.w == b.w)) {  // empty ring (or NaN coordinates): no slabs
        row0[r] = 0;
        nrows[r] = 0;
        return;
    }
    const int f0 = frow(f, b.y), f1 = frow(f, b.w);
    const int64_t n_edges = (int64_t)ring_off[r + 1] - ring_off[r] - 1;
    const int64_t base_rows = (f1 >> PIP_FINE_LOG2) - (f0 >> PIP_FINE_LOG2) + 1;
    int sh = 0;
    while (sh < max_shift && n_edges > (int64_t)SLAB_TARGET * (base_rows << sh)) ++sh;
    const int j0 = f0 >> (PIP_FINE_LOG2 - sh), j1 = f1 >> (PIP_FINE_LOG2 - sh);
    // "some ring has refined rows" (the join picks its lean kernel when there is none): a plain store of 1 by one lane of every wave
    // that holds such a ring — a count by atomics, even one per wave, was 90k atomics on ONE word for the power-law column, 1.9 ms of
    // a kernel that otherwise takes 0.1
    if (sh > 0 && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(sh > 0)) - 1) *n_refined = 1;
    row0[r] = j0 | (sh << 24);
    nrows[r] = j1 - j0 + 1;
}

// ring that owns coordinate i: largest r with ring_off[r] <= i
__device__ __forceinline__ int ring_of_coord(const int32_t* __restrict__ ring_off, int n_rings, int i) {
    int lo = 0, hi = n_rings;  // invariant: ring_off[lo] <= i < ring_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ring_off[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
// edge starting at coordinate i (a single-coordinate ring contributes one degenerate edge so that the
// "ring of one coordinate" arm of coord_pos_relative_to_ring is reproduced by the edge walk)
// coord_ring: the ring of every coordinate, one load (ring_start_kernel + a scan, built once per index build) — the per-edge
// kernels below ran 23 DEPENDENT loads of a binary search over the ring offsets per coordinate and were bound by that chain
// (C5: 143.9M coordinates, 5.0M rings; six launches, 46 of the build's 92 ms).
__device__ __forceinline__ bool edge_at(const DevGeo& a, const int32_t* __restrict__ coord_ring, int i, int& r, double2& s, double2& e) {
    r = coord_ring[i];
    const int c0 = a.ring_off[r], c1 = a.ring_off[r + 1];
    if (i >= c1) return false;  // coordinate belongs to an empty-ring gap (cannot happen with valid offsets)
    s = a.xy[i];
    if (i + 1 < c1) {
        e = a.xy[i + 1];
        return true;
    }
    if (c1 - c0 == 1) {
        e = s;
        return true;
    }
    return false;
}

// flag[ring_off[r]] += 1 for every ring r >= 1 that starts below n_coords: the inclusive prefix sum at coordinate i is then the
// largest r with ring_off[r] <= i (ring_of_coord's answer, empty rings included) — the build scans it in place
__global__ void ring_start_kernel(const int32_t* __restrict__ ring_off, int64_t n_rings, int64_t n_coords, int32_t* __restrict__ flag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < 1 || r >= n_rings) return;
    const int32_t o = ring_off[r];
    if (o >= 0 && (int64_t)o < n_coords) atomicAdd(&flag[o], 1);
}

template <bool FILL>
__global__ void slab_register_kernel(DevGeo a, const int32_t* __restrict__ coord_ring, FineGrid f, const int32_t* __restrict__ row0,
                                     const int32_t* __restrict__ slab_base, int32_t* __restrict__ cnt_or_cursor,
                                     double4* __restrict__ edges, int32_t* __restrict__ vidx = nullptr) {
    // vidx (optional, FILL only): the coordinate index every slab entry's edge starts at (chain_aux_kernel; PipView::slab_vidx when the
    // index keeps no edge copies: `edges` is nullptr then)
    //
    // One atomic per RUN of lanes that register in the same slab, not one per edge: consecutive edges of a ring mostly lie in one
    // slab row (SLAB_TARGET of them share it), and 144M same-address atomics in a row were what bound this kernel (C5: 6.8 + 10.2 ms
    // for the two passes after the ring search had gone).  The head lane of a run adds the run's length and hands every lane its
    // slot; an edge that spans several rows registers the first one this way and the others one atomic each, as before.
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    int r = 0;
    double2 s = make_double2(0, 0), e = s;
    bool live = i < a.n_coords && edge_at(a, coord_ring, (int)i, r, s, e);
    if (live) live = slab_base[r + 1] != slab_base[r];
    const double ylo = s.y < e.y ? s.y : e.y, yhi = s.y > e.y ? s.y : e.y;
    if (!(ylo == ylo) || !(yhi == yhi)) live = false;
    int j0 = 0, j1 = -1, sl0 = -1 - lane;  // dead lanes: keys no neighbour shares
    int rr = 0;
    if (live) {
        rr = row0[r];
        const int down = PIP_FINE_LOG2 - slab_shift_of(rr);  // f is the finest row grid
        j0 = frow(f, ylo) >> down;
        j1 = frow(f, yhi) >> down;
        sl0 = slab_base[r] + (j0 - slab_row0_of(rr));
    }
    const int prev = __shfl_up(sl0, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || sl0 != prev);
    const int head = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));  // the run's first lane (heads has bit 0 set)
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
    const int run = above ? __ffsll((long long)above) : 64 - lane;  // (meaningful on head lanes)
    int base = 0;
    if (live && head == lane) base = atomicAdd(&cnt_or_cursor[sl0], run);
    base = __shfl(base, head, 64);
    if (!live) return;
    const bool degenerate = FILL && a.ring_off[r + 1] - a.ring_off[r] == 1;  // edge_at: (s, s) of a one-coordinate ring
    auto put = [&](int slot) {
        if (edges) edges[slot] = make_double4(s.x, s.y, e.x, e.y);
        if (vidx) vidx[slot] = degenerate ? ~(int32_t)i : (int32_t)i;
    };
    if (FILL) put(base + (lane - head));
    for (int j = j0 + 1; j <= j1; ++j) {
        const int sl = slab_base[r] + (j - slab_row0_of(rr));
        const int slot = atomicAdd(&cnt_or_cursor[sl], 1);
        if (FILL) put(slot);
    }
}

// cells an edge may touch (see the header comment); f(i, j) is called for each
// Does the segment s -> e's supporting LINE separate the closed rectangle [xl, xh] x [yl, yh] strictly from itself — are all four
// corners strictly on one side?  orient2d(s, e, c) is the sign of a function that is LINEAR in the corner c:
//     sx ey - sy ex + cx (sy - ey) + cy (ex - sx),
// so its minimum and maximum over the rectangle sit at the two corners picked by the signs of (sy - ey) and (ex - sx): two exact
// orientations answer what four were computed for (round 4: the labelling kernels of the index build are bound by this arithmetic).
// Exact: dev::orient2d returns the sign of the true value, and the extreme corners of the true function are the ones picked.
__device__ __forceinline__ bool line_clear_of_rect(double sx, double sy, double ex, double ey, double xl, double yl, double xh, double yh) {
    const bool x_up = sy > ey, y_up = ex > sx;  // the function grows with cx / with cy (a zero coefficient: either corner)
    const double max_x = x_up ? xh : xl, min_x = x_up ? xl : xh, max_y = y_up ? yh : yl, min_y = y_up ? yl : yh;
    return dev::orient2d(sx, sy, ex, ey, min_x, min_y) > 0 || dev::orient2d(sx, sy, ex, ey, max_x, max_y) < 0;
}
template <typename F>
__device__ __forceinline__ void for_each_touched_cell(const FineGrid& g, double2 s, double2 e, F&& f) {
    const double xlo = s.x < e.x ? s.x : e.x, xhi = s.x > e.x ? s.x : e.x;
    const double ylo = s.y < e.y ? s.y : e.y, yhi = s.y > e.y ? s.y : e.y;
    if (!(xlo == xlo) || !(xhi == xhi) || !(ylo == ylo) || !(yhi == yhi)) return;
    // widen by the padding so that a vertex sitting within `pad` of a cell border also claims the neighbour
    const int i0 = fcol(g, xlo - g.pad_x), i1 = fcol(g, xhi + g.pad_x);
    const int j0 = frow(g, ylo - g.pad_y), j1 = frow(g, yhi + g.pad_y);
    if (i0 == i1 && j0 == j1) {
        f(i0, j0);
        return;
    }
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            if (i == 0 || j == 0 || i == g.R - 1 || j == g.R - 1) {  // clamped cells are unbounded: always claim
                f(i, j);
                continue;
            }
            const double xl = g.rx0 + (double)i * g.fw - g.pad_x, xh = g.rx0 + (double)(i + 1) * g.fw + g.pad_x;
            const double yl = g.ry0 + (double)j * g.fh - g.pad_y, yh = g.ry0 + (double)(j + 1) * g.fh + g.pad_y;
            if (!line_clear_of_rect(s.x, s.y, e.x, e.y, xl, yl, xh, yh)) f(i, j);
        }
}

template <bool FILL>
__global__ void mark_kernel(DevGeo a, const int32_t* __restrict__ coord_ring, FineGrid g, const int32_t* __restrict__ ring_part,
                            int32_t* __restrict__ cnt_or_off, unsigned long long* __restrict__ keys) {
    // count pass: cnt[i] = keys of the edge that starts at coordinate i; fill pass: the edge writes its keys at off[i].. (exclusive
    // scan of the counts) — no atomics, deterministic layout.  An edge that touches ONE cell and whose predecessor in the wave (the
    // previous edge of the same ring, nearly always) touched that one cell of the same part writes nothing: runs of short edges
    // inside a cell were most of the raw marks (C5: a part of thousands of vertices marks its two or three cells thousands of
    // times), and every one of them went through the sort and the unique pass.  Both passes see the same lanes, so they agree.
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    int r = 0;
    double2 s = make_double2(0, 0), e = s;
    const bool live = i < a.n_coords && edge_at(a, coord_ring, (int)i, r, s, e);
    const unsigned long long none = ~0ull;
    unsigned long long first = none;
    const int64_t o0 = FILL && i < a.n_coords ? (int64_t)cnt_or_off[i] : 0;
    int64_t n = 0;
    if (live) {
        const unsigned long long part = (unsigned long long)ring_part[r];
        for_each_touched_cell(g, s, e, [&](int ci, int cj) {
            const unsigned long long key = ((unsigned long long)(cj * g.R + ci) << 32) | part;
            if (n == 0) {
                first = key;  // held back: written below unless it repeats the previous lane's
            } else if (FILL) {
                if (n == 1) keys[o0] = first;
                keys[o0 + n] = key;
            }
            ++n;
        });
    }
    const unsigned long long single = n == 1 ? first : none;
    const unsigned long long prev = __shfl_up(single, 1, 64);
    const bool repeat = lane > 0 && single != none && single == prev;
    if (i >= a.n_coords) return;
    if (!FILL) {
        cnt_or_off[i] = repeat ? 0 : (int32_t)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF);
    } else if (n == 1 && !repeat) {
        keys[o0] = first;
    }
}

__global__ void unique_flags_kernel(const unsigned long long* __restrict__ sorted, int64_t n, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0;
}
// (+ cell_marks[c] += 1 for every unique mark of cell c: scanned, the cell kernels read their marks' range instead of running two
// 24-step binary searches over the marks per cell)
__global__ void unique_compact_kernel(const unsigned long long* __restrict__ sorted, int64_t n, const int32_t* __restrict__ pos,
                                      unsigned long long* __restrict__ out, int32_t* __restrict__ cell_marks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = sorted[i];
    if (i == 0 || key != sorted[i - 1]) {
        out[pos[i]] = key;
        atomicAdd(&cell_marks[key >> 32], 1);
    }
}

constexpr int CELL_SCRATCH = 8;  // entries per cell the count pass keeps for the fill pass (cell_build_kernel)
// One thread per raster cell: merge (a) the parts whose edges may touch the cell and (b) the parts that
// strictly contain the cell's centre, in ascending part order.
template <bool FILL>
__global__ void cell_build_kernel(DevGeo a, IndexView ix, PipView pv, FineGrid g,
                                  const unsigned long long* __restrict__ marks, const int32_t* __restrict__ mark_start,
                                  int32_t* __restrict__ need, const int32_t* __restrict__ list_off,
                                  uint32_t* __restrict__ cell, uint32_t* __restrict__ list, uint32_t* __restrict__ scratch,
                                  const double4* __restrict__ ring_bbox) {
    // scratch (n_cells x CELL_SCRATCH words, entry k of cell c at [k * n_cells + c]; may be null): the count pass leaves the first
    // entries of every cell there and the fill pass copies the lists that fit instead of walking the candidates' slabs a second time
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_cells = (int64_t)g.R * g.R;
    if (c >= n_cells) return;
    if (FILL && need[c] == 0) return;  // empty and one-entry cells got their word from the count pass: only list cells walk again
    if (FILL && scratch && need[c] - 1 <= CELL_SCRATCH) {
        const int m = need[c] - 1;
        uint32_t* out = list + list_off[c];
        out[0] = (uint32_t)m;
        for (int k = 0; k < m; ++k) out[1 + k] = scratch[(int64_t)k * n_cells + c];
        cell[c] = (CELL_TAG_LIST << 30) | (uint32_t)list_off[c];
        return;
    }
    const int ci = (int)(c % g.R), cj = (int)(c / g.R);
    const double cx = g.rx0 + ((double)ci + 0.5) * g.fw, cy = g.ry0 + ((double)cj + 0.5) * g.fh;
    const bool border = ci == 0 || cj == 0 || ci == g.R - 1 || cj == g.R - 1;
    const bool centre_ok = !border && fcol(g, cx) == ci && frow(g, cy) == cj;
    int64_t m = mark_start[c];  // the cell's marks: exclusive scan of the per-cell counts (unique_compact_kernel)
    const int64_t m_end = mark_start[c + 1];

    int n = 0;
    uint32_t first_entry = 0;
    uint32_t* out = nullptr;
    if (FILL && need[c] > 0) {
        out = list + list_off[c];
        out[0] = (uint32_t)(need[c] - 1);
        ++out;
    }
    auto emit = [&](uint32_t part, uint32_t boundary) {
        const uint32_t e = (part << 1) | boundary;
        if (n == 0) first_entry = e;
        if (out) out[n] = e;
        if (!FILL && scratch && n < CELL_SCRATCH) scratch[(int64_t)n * n_cells + c] = e;
        ++n;
    };
    auto flush_marks_below = [&](unsigned long long part_limit) {  // emit marks with part < part_limit
        while (m < m_end && (marks[m] & 0xFFFFFFFFull) < part_limit) {
            emit((uint32_t)(marks[m] & 0xFFFFFFFFull), 1u);
            ++m;
        }
    };

    // candidates whose bbox contains the centre, through the coarse directory (ascending geometry id)
    const GridParams cg = *ix.grid;
    const int gx = dev::cell_of(cx, cg.x0, cg.inv_w, cg.gx), gy = dev::cell_of(cy, cg.y0, cg.inv_h, cg.gy);
    const int gc = gy * cg.gx + gx;
    for (int k = ix.cell_off[gc]; k < ix.cell_off[gc + 1]; ++k) {
        const int j = ix.items[k];
        const double4 bb = ix.bbox[j];
        if (!(cx >= bb.x && cx <= bb.z && cy >= bb.y && cy <= bb.w)) continue;
        if (!dev::valid_row(a.validity, j)) continue;
        int p0, p1;
        dev::geom_parts(a, j, p0, p1);
        for (int p = p0; p < p1; ++p) {
            flush_marks_below((unsigned long long)p);
            if (m < m_end && (marks[m] & 0xFFFFFFFFull) == (unsigned long long)p) {
                emit((uint32_t)p, 1u);
                ++m;
                continue;
            }
            if (p1 - p0 > 1) {
                // a member of a multipolygon: the centre passed the GEOMETRY's box, which spans all members — outside this member's
                // exterior box it is outside the member, and the walk of its slab (PartInfo, row offsets, a dozen edges) is skipped
                int r0, r1;
                dev::part_rings(a, p, r0, r1);
                if (r1 <= r0) continue;
                const double4 rb = ring_bbox[r0];
                if (!(cx >= rb.x && cx <= rb.z && cy >= rb.y && cy <= rb.w)) continue;
            }
            const int pos = pip::part_pos_single(pv, a, p, cx, cy);
            if (pos == dev::POS_OUTSIDE) continue;
            // Inside an unmarked, well-formed interior cell: the whole cell is inside.  Anything else
            // (border cell, centre not representable in the cell, centre on a boundary) stays exact.
            emit((uint32_t)p, (centre_ok && pos == dev::POS_INSIDE) ? 0u : 1u);
        }
    }
    flush_marks_below(~0ull);

    if (!FILL) {
        need[c] = n >= 2 ? n + 1 : 0;
        if (n <= 1) cell[c] = n == 1 ? ((CELL_TAG_SINGLE << 30) | first_entry) : 0u;
        return;
    }
    cell[c] = (CELL_TAG_LIST << 30) | (uint32_t)list_off[c];
}

// ---- level 2: cells crossed by exactly one part -----------------------------------------------------
// flag[c] = 1: the cell is crossed by exactly one part and nothing else is there (SubCell); flag2[c] = 1: the cell
// holds exactly two parts and both cross it (SubCell2).  (A covering part + a crossing one was tried as a second record
// kind: no gain on the power-law multipolygon column, 12 ms more build time.)
__global__ void sub_flag_kernel(const uint32_t* __restrict__ cell, const uint32_t* __restrict__ list, int64_t n_cells,
                                int32_t* __restrict__ flag, int32_t* __restrict__ flag2) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t w = cell[c];
    const uint32_t tag = w >> 30;
    flag[c] = (tag == CELL_TAG_SINGLE && (w & 1u)) ? 1 : 0;
    int f2 = 0;
    if (tag == CELL_TAG_LIST) {
        const uint32_t off = w & 0x3FFFFFFFu;
        if (list[off] == 2u) f2 = (list[off + 1] & list[off + 2] & 1u) ? 1 : 0;  // two entries, both crossing the cell
    }
    flag2[c] = f2;
}

// exterior slabs of a cell's PIP_SLAB_MUL (= 2) base slab rows are adjacent in slab_off: [e0,e1) and [e1,e2); a part
// whose exterior has refined rows has more than two per cell: its record says SUB_INDIRECT and the join kernel goes
// through PartInfo for it.  (cj = raster row of the cell)
__device__ __forceinline__ void record_slabs(const PipView& pv, int p, int cj, uint32_t& flags, uint32_t& e0, uint32_t& e1, uint32_t& e2) {
    const PartInfo pi = pv.part_info[p];
    e0 = e1 = e2 = 0;
    flags = (uint32_t)p | (pi.n_rings > 1 ? 0x80000000u : 0u);
    if (slab_shift_of(pi.row0) != 0) {
        flags |= SUB_INDIRECT;
        return;
    }
    const int j0 = PIP_SLAB_MUL * cj - slab_row0_of(pi.row0), j1 = j0 + 1;
    const bool lo_ok = j0 >= 0 && j0 < pi.nrows, hi_ok = j1 >= 0 && j1 < pi.nrows;
    if (lo_ok) {
        e0 = (uint32_t)pv.slab_off[pi.slab_base + j0];
        e1 = (uint32_t)pv.slab_off[pi.slab_base + j0 + 1];
        e2 = e1;
    }
    if (hi_ok) {
        if (!lo_ok) e0 = e1 = (uint32_t)pv.slab_off[pi.slab_base + j1];
        e2 = (uint32_t)pv.slab_off[pi.slab_base + j1 + 1];
    }
}
// The head of every work item's record (part word + the exterior's slab ranges in the cell's raster row), one LANE per item: the
// five dependent gathers behind it (part -> PartInfo -> slab offsets) run 64 to a wave here instead of one to a wave at the top of
// sub_build_kernel, whose waves then start from their record head (see its fast path).
__global__ void sub_head_kernel(PipView pv, FineGrid g, const int32_t* __restrict__ work_cell, const uint32_t* __restrict__ work_part,
                                int64_t n_work, SubCell* __restrict__ sub) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_work) return;
    uint32_t flags, e0, e1, e2;
    record_slabs(pv, (int)work_part[i], (int)(work_cell[i] / g.R), flags, e0, e1, e2);
    *reinterpret_cast<uint4*>(&sub[i].part_flags) = make_uint4(flags, e0, e1, e2);
}

// PIP_SUB^2 lanes per flagged cell: lane k labels sub-cell (k % PIP_SUB, k / PIP_SUB).  A sub-cell is "test
// exactly" when any edge of ANY ring of a part that crosses the cell (taken from the rings' slabs of this raster row)
// is not strictly on one side of the sub-cell's padded rectangle; otherwise it inherits the exact position of its
// centre.  NP = 1: SubCell records (one crossing part); NP = 2: SubCell2 records (two parts, see gpk_index.h).
constexpr int SUB_EDGE_CAP = 48;  // edges of a cell's slab rows kept in LDS per wave (sub_build_kernel)
// WORK (NP = 1 only): the records are those of (cell, part) work items — the boundary entries of list cells,
// work_cell[w] / work_part[w] -> sub[w] — instead of one per flagged cell.
// MODE (WORK only): 0 = every record; 1 = only the records of the fast path below (one-ring parts without refined rows and at most
// 64 edges in the raster row); 2 = only the others.  The build launches 1 then 2: the fast path alone needs half the registers of the
// general one, so twice the waves stand behind its two dependent loads (a small right side's records all take it).
template <int NP, bool WORK = false, int MODE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 1 ? 8 : 5))) void sub_build_kernel(DevGeo a, PipView pv, FineGrid g, const int32_t* __restrict__ flag,
                                                        const int32_t* __restrict__ pos, int64_t n_cells, const uint32_t* __restrict__ cell,
                                                        const uint32_t* __restrict__ list, SubCell* __restrict__ sub,
                                                        SubCell2* __restrict__ sub2, const int32_t* __restrict__ work_cell = nullptr,
                                                        const uint32_t* __restrict__ work_part = nullptr, int64_t n_work = 0) {
    static_assert(!WORK || NP == 1, "work items carry one part");
    constexpr int S = PIP_SUB, SS = PIP_SUB * PIP_SUB;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = t / SS;  // a cell, or a work item
    const int k = (int)(t % SS);
    if (WORK ? item >= n_work : (item >= n_cells || !flag[item])) return;
    const int64_t c = WORK ? (int64_t)work_cell[item] : item;
    const uint32_t w = cell[c];  // still the level-1 word: sub_commit_kernel rewrites it afterwards
    int part[2] = {0, 0};
    bool crosses[2] = {true, true};
    if (WORK) {
        part[0] = (int)work_part[item];
    } else if (NP == 1) {
        part[0] = (int)((w & 0x3FFFFFFFu) >> 1);
    } else {
        const uint32_t off = w & 0x3FFFFFFFu;
        const uint32_t ea = list[off + 1], eb = list[off + 2];
        part[0] = (int)(ea >> 1);
        part[1] = (int)(eb >> 1);
    }
    const int ci = (int)(c % g.R), cj = (int)(c / g.R);
    const int si = S * ci + (k % S), sj = S * cj + (k / S);
    const double fw2 = g.fw / S, fh2 = g.fh / S, px2 = g.pad_x / S, py2 = g.pad_y / S;
    const double xl = g.rx0 + (double)si * fw2 - px2, xh = g.rx0 + (double)(si + 1) * fw2 + px2;
    const double yl = g.ry0 + (double)sj * fh2 - py2, yh = g.ry0 + (double)(sj + 1) * fh2 + py2;
    // The 64 lanes of a wave label the 64 sub-cells of ONE cell, so they share its slab rows.  First the wave picks the
    // edges whose box meets the (padded) cell at all — lane j looks at edge j, survivors are compacted by ballot into an
    // LDS list — then every lane tests only those few against its own sub-cell (a slab row holds every edge of the ring
    // that crosses the row anywhere in x; a cell sees one to three of them).
    __shared__ double4 s_edges[256 / 64][SUB_EDGE_CAP];
    __shared__ double4 s_edges_all[WORK ? 256 / 64 : 1][WORK ? 64 : 1];  // fast path: every edge of the part in this raster row
    const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
    // the cell = the union of its padded sub-cells, written with the very expressions the corner lanes use below
    const double cxl = g.rx0 + (double)(S * ci) * fw2 - px2, cxh = g.rx0 + (double)(S * ci + S) * fw2 + px2;
    const double cyl = g.ry0 + (double)(S * cj) * fh2 - py2, cyh = g.ry0 + (double)(S * cj + S) * fh2 + py2;
    auto edge_touches = [&](const double4 ed) {
        // cheap reject: edge bbox vs padded rectangle (closed)
        if (fmax(ed.x, ed.z) < xl || fmin(ed.x, ed.z) > xh || fmax(ed.y, ed.w) < yl || fmin(ed.y, ed.w) > yh) return false;
        return !line_clear_of_rect(ed.x, ed.y, ed.z, ed.w, xl, yl, xh, yh);
    };
    // the record's labels from the two ballots (low / high label bit): lanes 0..3 interleave their 16 sub-cells' bits into one word
    // each and store it — 64 global atomics on four addresses per record were most of this kernel's time
    auto store_labels = [&](SubCell* rec, uint32_t label) {
        const unsigned long long lo = __ballot((label & 1u) != 0), hi = __ballot((label & 2u) != 0);
        if (k < 4) {
            auto spread16 = [](uint32_t x) {
                x = (x | (x << 8)) & 0x00FF00FFu;
                x = (x | (x << 4)) & 0x0F0F0F0Fu;
                x = (x | (x << 2)) & 0x33333333u;
                x = (x | (x << 1)) & 0x55555555u;
                return x;
            };
            rec->labels[k] = spread16((uint32_t)(lo >> (16 * k)) & 0xFFFFu) | (spread16((uint32_t)(hi >> (16 * k)) & 0xFFFFu) << 1);
        }
    };
    if (WORK) {
        // Fast path (sub_head_kernel wrote the record head): a one-ring part without refined rows has ALL its edges of this raster row
        // in [e0, e2) — lower base slab row [e0, e1), upper [e1, e2).  One load per lane stages them in LDS; the touch test and the
        // centre's winding walk (its slab row is the lower or the upper one: sub-rows 0..3 / 4..7, as the join kernel picks it) both
        // run from there.  Two dependent loads per record instead of nine.
        const uint4 head = *reinterpret_cast<const uint4*>(&sub[item].part_flags);
        const uint32_t n_all = head.w - head.y;
        const bool fast = !(head.x & (SUB_INDIRECT | 0x80000000u)) && n_all <= 64u;
        if (MODE == 1 && !fast) return;
        if (MODE == 2 && fast) return;
        if (MODE != 2 && fast) {
            if ((uint32_t)lane64 < n_all) s_edges_all[wave][lane64] = pip::slab_edge(pv, (int)(head.y + (uint32_t)lane64));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            bool touched = false;
            for (uint32_t e = 0; e < n_all && !touched; ++e) touched = edge_touches(s_edges_all[wave][e]);
            uint32_t label = 2u;
            if (!touched) {
                const double cx = g.rx0 + ((double)si + 0.5) * fw2, cy = g.ry0 + ((double)sj + 0.5) * fh2;
                const bool ok = dev::cell_of(cx, g.rx0, g.inv_fw * S, g.R * S) == si && dev::cell_of(cy, g.ry0, g.inv_fh * S, g.R * S) == sj;
                if (ok) {
                    const uint32_t mid = head.z - head.y;
                    const uint32_t a0 = (k / S) >= S / 2 ? mid : 0u, a1 = (k / S) >= S / 2 ? n_all : mid;
                    int wn = 0;
                    bool on = false;
                    for (uint32_t e = a0; e < a1; ++e) {
                        const double4 ed = s_edges_all[wave][e];
                        on |= dev::ring_edge(ed.x, ed.y, ed.z, ed.w, cx, cy, wn);
                    }
                    label = on ? 2u : (wn != 0 ? 1u : 0u);
                }
            }
            store_labels(sub + item, label);
            return;
        }
        if (MODE == 1) return;  // (nothing but fast records reaches this point in that instance)
    }
    // General path: every edge of every ring of the part(s) registered in this raster row goes past the wave once, 64 to a load —
    // lane j first resolves the slab span of ring j (all rings' spans behind ONE chain of dependent gathers, not one chain per ring),
    // the spans are laid end to end, and lane q of each round takes edge q of that sequence.  Survivors of the cell's box are
    // compacted by ballot into the LDS list; a full list is drained (every lane tests its sub-cell against it) and refilled, so
    // no cell falls back to 64 private walks.  (Round 4: one serial ring after the other with its own span lookup and edge rounds
    // was 20 of this kernel's 26 ms on the power-law multipolygon column, whose large parts carry holes.)
    __shared__ int s_span_e0[256 / 64][64], s_span_at[256 / 64][64];
    int n_list = 0;  // wave-uniform
    bool touched = false;
    auto wave_min = [&](double v) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v = fmin(v, __shfl_xor(v, o, 64));
        return v;
    };
    // boxed (the drains of a list that overflowed): a part with thousands of vertices inside this raster row sends hundreds of
    // full lists past the wave, each a run of consecutive short edges — the list's own box (one wave reduction) lets every
    // sub-cell away from that run skip it, where it would have box-tested each of its edges (64 x edges tests per cell otherwise)
    auto drain = [&](bool boxed = false) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bool skip = false;
        if (boxed) {
            const double inf = __builtin_inf();
            double4 b = make_double4(inf, inf, inf, inf);  // (min x, min y, -max x, -max y) of lane e's edge
            bool odd = false;                               // a NaN coordinate: no box, every lane walks the list
            if (lane64 < n_list) {
                const double4 ed = s_edges[wave][lane64];
                b = make_double4(fmin(ed.x, ed.z), fmin(ed.y, ed.w), -fmax(ed.x, ed.z), -fmax(ed.y, ed.w));
                odd = !(ed.x == ed.x) || !(ed.y == ed.y) || !(ed.z == ed.z) || !(ed.w == ed.w);
            }
            if (!__ballot(odd)) {
                const double bx0 = wave_min(b.x), by0 = wave_min(b.y), bx1 = -wave_min(b.z), by1 = -wave_min(b.w);
                skip = bx1 < xl || bx0 > xh || by1 < yl || by0 > yh;  // every edge of the list fails the same comparisons
            }
        }
        for (int e = 0; e < n_list && !touched && !skip; ++e) touched = edge_touches(s_edges[wave][e]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        n_list = 0;
    };
    for (int q = 0; q < NP; ++q) {
        if (!crosses[q]) continue;
        int r0, r1;
        dev::part_rings(a, part[q], r0, r1);
        for (int rb = r0; rb < r1; rb += 64) {
            const int n_r = r1 - rb < 64 ? r1 - rb : 64;
            int e0 = 0, cnt = 0;
            if (lane64 < n_r) {
                int a0, a1;
                if (pip::slab_span_of_raster_row(pv, rb + lane64, cj, a0, a1)) {
                    e0 = a0;
                    cnt = a1 - a0;
                }
            }
            const int incl = dev::wave_inclusive_scan(cnt);
            const int total = __shfl(incl, 63, 64);
            s_span_e0[wave][lane64] = e0;
            s_span_at[wave][lane64] = incl - cnt;  // where ring j's edges start in the sequence (lanes past n_r: `total`)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int base = 0; base < total; base += 64) {
                const int t = base + lane64;
                bool keep = false;
                double4 ed = make_double4(0, 0, 0, 0);
                if (t < total) {
                    int lo = 0, hi = n_r;  // the last ring whose start is <= t (rings without edges here share their successor's start)
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_span_at[wave][mid] <= t)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    ed = pip::slab_edge(pv, s_span_e0[wave][lo] + (t - s_span_at[wave][lo]));
                    keep = !(fmax(ed.x, ed.z) < cxl || fmin(ed.x, ed.z) > cxh || fmax(ed.y, ed.w) < cyl || fmin(ed.y, ed.w) > cyh);
                }
                unsigned long long pending = __ballot(keep);
                while (pending) {  // (wave-uniform) survivors go to the list as far as it has room; a full list is drained and refilled
                    const int rank = __popcll(pending & ((1ull << lane64) - 1ull));
                    const bool put = keep && rank < SUB_EDGE_CAP - n_list;
                    if (put) {
                        s_edges[wave][n_list + rank] = ed;
                        keep = false;
                    }
                    const unsigned long long taken = __ballot(put);
                    n_list += __popcll(taken);
                    pending &= ~taken;
                    if (pending) drain(true);
                }
            }
            // (the span tables are rewritten by the next round of rings: every lane is past its reads — the wave runs in lockstep
            // through the uniform loops above, and the barrier at the top of the next round's reads orders the rest)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    drain();
    const uint32_t test_label = NP == 2 ? 3u : 2u;
    uint32_t label = test_label;
    if (!touched) {
        const double cx = g.rx0 + ((double)si + 0.5) * fw2, cy = g.ry0 + ((double)sj + 0.5) * fh2;
        // the centre must map to this very sub-cell under the point-side function (S x the level-1 scale)
        const bool ok = dev::cell_of(cx, g.rx0, g.inv_fw * S, g.R * S) == si && dev::cell_of(cy, g.ry0, g.inv_fh * S, g.R * S) == sj;
        if (ok) {
            if (NP == 1) {
                const int p = pip::part_pos_single(pv, a, part[0], cx, cy);
                label = p == dev::POS_INSIDE ? 1u : (p == dev::POS_OUTSIDE ? 0u : 2u);
            } else {
                const int pa = pip::part_pos_single(pv, a, part[0], cx, cy), pb = pip::part_pos_single(pv, a, part[1], cx, cy);
                if (pa == dev::POS_BOUNDARY || pb == dev::POS_BOUNDARY)
                    label = 3u;
                else if (pa == dev::POS_INSIDE && pb == dev::POS_INSIDE)
                    label = 3u;  // untouched by either boundary yet inside both: overlapping parts, leave it to the exact test
                else
                    label = pa == dev::POS_INSIDE ? 1u : (pb == dev::POS_INSIDE ? 2u : 0u);
            }
        }
    }
    auto slabs_of = [&](int p, uint32_t& flags, uint32_t& e0, uint32_t& e1, uint32_t& e2) { record_slabs(pv, p, cj, flags, e0, e1, e2); };
    SubCell* rec = WORK ? sub + item : (NP == 1 ? sub + pos[c] : &sub2[pos[c]].a);
    if (k == 0) {
        slabs_of(part[0], rec->part_flags, rec->e0, rec->e1, rec->e2);
        if (NP == 2) {
            SubCell2* r2 = sub2 + pos[c];
            slabs_of(part[1], r2->b_part_flags, r2->b_e0, r2->b_e1, r2->b_e2);
        }
    }
    store_labels(rec, label);
}
__global__ void sub_commit_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ pos, const int32_t* __restrict__ flag2,
                                  const int32_t* __restrict__ pos2, int64_t n_cells, uint32_t* __restrict__ cell) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    if (flag[c])
        cell[c] = (CELL_TAG_SUB << 30) | (uint32_t)pos[c];
    else if (flag2 && flag2[c])
        cell[c] = (CELL_TAG_SUB << 30) | SUB2_BIT | (uint32_t)pos2[c];
}

// ---- level 2 for list cells: one SubCell per boundary entry ------------------------------------------------------------
// Right sides whose parts overlap (or are about as large as a raster cell) have most points in list cells, and at the
// level-1 resolution nearly every entry there is a boundary entry: PartInfo + slab gathers + an exact walk per entry
// and point, nine times out of ten to learn "outside".  With a record per boundary entry the point reads one 32-byte
// record per entry and only its "test" sub-cells reach the queue.
__global__ void lrec_count_kernel(const uint32_t* __restrict__ cell, const uint32_t* __restrict__ list, int64_t n_cells,
                                  int32_t* __restrict__ cnt) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t w = cell[c];
    int n = 0;
    if ((w >> 30) == CELL_TAG_LIST) {
        const uint32_t off = w & 0x3FFFFFFFu, m = list[off];
        for (uint32_t t = 0; t < m; ++t) n += (int)(list[off + 1 + t] & 1u);
    }
    cnt[c] = n;
}
// ---- local chains of the `test` sub-cells of a lean index (gpk_index.h: ChainAux) ------------------------------------------
__device__ __forceinline__ int test_labels_of(uint32_t w) { return __popc((w >> 1) & ~w & 0x55555555u); }  // 2-bit fields equal to 2
__global__ void chain_count_kernel(const SubCell* __restrict__ sub, int64_t n_sub, int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sub) return;
    const SubCell rc = sub[i];
    cnt[i] = test_labels_of(rc.labels[0]) + test_labels_of(rc.labels[1]) + test_labels_of(rc.labels[2]) + test_labels_of(rc.labels[3]);
}
// One wave per record (work item = (cell, part) of sub_work_kernel), lane = sub-cell.  The wave lists the edges of the part's
// slab rows in this raster row that meet the padded cell (with the coordinate index each starts at), every `test` lane then
//   1. takes the listed edges that meet ITS padded sub-cell (exact: the test that labelled it) and the shortest arc of the
//      ring — a ring is a cycle: the arc may run over the closing vertex — that covers them,
//   2. grows the arc at both ends while the end vertex's y lies in the sub-cell's closed y-interval,
//   3. sums the contributions, at the sub-cell centre, of the edges of the centre's slab row that are NOT in the arc: `base`,
//   4. copies the arc's first four vertices into the entry (vertices 4 .. follow in chain_ext: chain_ext_kernel).
// An arc of more than CHAIN_MAX edges, a part with holes, or a cell whose edge list overflowed gets count = 0 (such rows are
// decided by the generic walk).
__global__ __launch_bounds__(256) void chain_aux_kernel(DevGeo a, PipView pv, FineGrid g, const int32_t* __restrict__ work_cell,
                                                        const uint32_t* __restrict__ work_part, int64_t n_work,
                                                        const int32_t* __restrict__ slab_vidx, const SubCell* __restrict__ sub,
                                                        const int32_t* __restrict__ aux_base, ChainAux* __restrict__ aux,
                                                        uint32_t* __restrict__ head, uint32_t* __restrict__ first_at,
                                                        int32_t* __restrict__ ext_need) {
    constexpr int S = PIP_SUB, SS = PIP_SUB * PIP_SUB;
    static_assert(SUB_EDGE_CAP <= 64, "the touched edges of a sub-cell are a 64-bit mask over the cell's list");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = t / SS;
    const int k = (int)(t % SS);
    if (item >= n_work) return;  // (whole waves: SS == 64)
    const int64_t c = work_cell[item];
    const int part = (int)work_part[item];
    const SubCell rc = sub[item];
    const uint32_t lw = rc.labels[k >> 4];
    const bool test = ((lw >> (2 * (k & 15))) & 3u) == 2u;
    const unsigned long long tm = __ballot(test);
    if (tm == 0ull) return;  // uniform
    const int ci = (int)(c % g.R), cj = (int)(c / g.R);
    const int si = S * ci + (k % S), sj = S * cj + (k / S);
    const double fw2 = g.fw / S, fh2 = g.fh / S, px2 = g.pad_x / S, py2 = g.pad_y / S;
    const double xl = g.rx0 + (double)si * fw2 - px2, xh = g.rx0 + (double)(si + 1) * fw2 + px2;
    const double yl = g.ry0 + (double)sj * fh2 - py2, yh = g.ry0 + (double)(sj + 1) * fh2 + py2;
    const double cxl = g.rx0 + (double)(S * ci) * fw2 - px2, cxh = g.rx0 + (double)(S * ci + S) * fw2 + px2;
    const double cyl = g.ry0 + (double)(S * cj) * fh2 - py2, cyh = g.ry0 + (double)(S * cj + S) * fh2 + py2;
    __shared__ double4 s_edges[256 / 64][SUB_EDGE_CAP];
    __shared__ int32_t s_vidx[256 / 64][SUB_EDGE_CAP];
    const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
    int r0, r1;
    dev::part_rings(a, part, r0, r1);
    bool list_ok = r1 - r0 == 1;  // a part with holes: no chains (uniform)
    int n_list = 0;
    if (list_ok) {
        int e0, e1;
        if (pip::slab_span_of_raster_row(pv, r0, cj, e0, e1)) {
            for (int eb = e0; eb < e1 && list_ok; eb += 64) {
                const int e = eb + lane64;
                bool keep = false;
                double4 ed = make_double4(0, 0, 0, 0);
                if (e < e1) {
                    ed = pip::slab_edge(pv, e);
                    keep = !(fmax(ed.x, ed.z) < cxl || fmin(ed.x, ed.z) > cxh || fmax(ed.y, ed.w) < cyl || fmin(ed.y, ed.w) > cyh);
                }
                const unsigned long long m = __ballot(keep);
                const int add = __popcll(m);
                if (n_list + add > SUB_EDGE_CAP) {
                    list_ok = false;
                    break;
                }
                if (keep) {
                    const int at = n_list + __popcll(m & ((1ull << lane64) - 1ull));
                    s_edges[wave][at] = ed;
                    s_vidx[wave][at] = pip::slab_vertex(slab_vidx[e]);
                }
                n_list += add;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!test) return;
    const int rank = __popcll(tm & ((1ull << lane64) - 1ull));
    const int64_t slot = (int64_t)aux_base[item] + rank;
    ChainAux out;
    out.v[0] = out.v[1] = out.v[2] = out.v[3] = make_double2(0.0, 0.0);
    uint32_t out_head = 0u, out_first = 0u;
    int need = 0;
    const int c0 = list_ok ? a.ring_off[r0] : 0, ne = list_ok ? a.ring_off[r0 + 1] - c0 - 1 : 0;  // the ring's edges: 0 .. ne - 1 (closed ring)
    if (list_ok && ne >= 1) {
        // 1. the listed edges that meet this padded sub-cell (an edge listed for both slab rows of the cell appears twice: harmless)
        unsigned long long touched = 0ull;
        for (int e = 0; e < n_list; ++e) {
            const double4 ed = s_edges[wave][e];
            if (fmax(ed.x, ed.z) < xl || fmin(ed.x, ed.z) > xh || fmax(ed.y, ed.w) < yl || fmin(ed.y, ed.w) > yh) continue;
            if (line_clear_of_rect(ed.x, ed.y, ed.z, ed.w, xl, yl, xh, yh)) continue;
            touched |= 1ull << e;
        }
        // the shortest arc [lo, lo + len) of the cycle 0 .. ne - 1 that covers the touched edges: the complement of the widest
        // gap between one touched edge and the next one after it
        int lo = 0, len = 0;
        if (touched) {
            int best_gap = -1, best_a = 0, best_next = 0;
            for (unsigned long long ma = touched; ma; ma &= ma - 1ull) {
                const int ea = s_vidx[wave][__ffsll((long long)ma) - 1] - c0;
                int nd = ne, nb = ea;  // distance to / index of the next touched edge after ea, cyclically (ne: ea is the only one)
                for (unsigned long long mb = touched; mb; mb &= mb - 1ull) {
                    const int eb = s_vidx[wave][__ffsll((long long)mb) - 1] - c0;
                    int d = eb - ea;
                    if (d < 0) d += ne;
                    if (d > 0 && d < nd) {
                        nd = d;
                        nb = eb;
                    }
                }
                if (nd > best_gap) {
                    best_gap = nd;
                    best_a = ea;
                    best_next = nb;
                }
            }
            lo = best_next;                    // the arc starts right after the widest gap ...
            len = ne - best_gap + 1;           // ... and ends at the edge before it (one touched edge: best_gap = ne, len = 1)
        }
        bool ok = len >= 1 && len <= CHAIN_MAX;
        // 2. grow: the arc's last edge ends at vertex lo + len, its first edge starts at vertex lo (indices modulo ne)
        while (ok && len < ne) {
            int hv = lo + len;
            if (hv >= ne) hv -= ne;
            const double y = a.xy[c0 + hv].y;
            if (!(y >= yl && y <= yh)) break;
            ++len;
            ok = len <= CHAIN_MAX;
        }
        while (ok && len < ne) {
            const double y = a.xy[c0 + lo].y;
            if (!(y >= yl && y <= yh)) break;
            lo = lo == 0 ? ne - 1 : lo - 1;
            ++len;
            ok = len <= CHAIN_MAX;
        }
        if (ok) {
            // 3. base: the other edges' winding at the centre — they all sit in the centre's slab row
            const double cx = g.rx0 + ((double)si + 0.5) * fw2, cy = g.ry0 + ((double)sj + 0.5) * fh2;
            int e0, e1, wn = 0;
            bool on = false;
            if (pip::slab_range(pv, r0, pip::row_of(pv, cy), e0, e1)) {
                for (int e = e0; e < e1; ++e) {
                    int d = pip::slab_vertex(slab_vidx[e]) - c0 - lo;
                    if (d < 0) d += ne;
                    if (d < len) continue;  // an edge of the arc
                    const double4 ed = pip::slab_edge(pv, e);
                    on |= dev::ring_edge(ed.x, ed.y, ed.z, ed.w, cx, cy, wn);
                }
            }
            if (!on && wn >= -127 && wn <= 127) {  // (an edge outside the arc cannot pass through the centre; guard anyway)
                out_head = (uint32_t)len | ((uint32_t)(uint8_t)(int8_t)wn << CHAIN_BASE_SHIFT);
                out_first = (uint32_t)(c0 + lo);
                for (int j = 0; j < 4; ++j) {  // 4. vertices 0 .. 3 (a shorter chain repeats its last vertex)
                    const int v = (lo + (j <= len ? j : len)) % ne;
                    out.v[j] = a.xy[c0 + v];
                }
                need = len > 3 ? len - 3 : 0;
            }
        }
    }
    aux[slot] = out;
    head[slot] = out_head;
    first_at[slot] = out_first;
    ext_need[slot] = need;
}
// vertices 4 .. count of the chains longer than three edges (one thread per chain entry), and where they are in the entry's head
__global__ void chain_ext_kernel(DevGeo a, uint32_t* __restrict__ head, const uint32_t* __restrict__ first_at, int64_t n_aux,
                                 const int32_t* __restrict__ ext_off, double2* __restrict__ ext) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_aux) return;
    const int len = (int)(head[i] & CHAIN_COUNT_MASK);
    if (len <= 3) return;
    if ((uint32_t)ext_off[i] >= (1u << (32 - CHAIN_EXT_SHIFT))) {  // the offset does not fit the head word: no chain entry (the generic walk decides)
        head[i] = 0u;
        return;
    }
    const int r = ring_of_coord(a.ring_off, (int)a.n_rings, (int)first_at[i]);
    const int c0 = a.ring_off[r], ne = a.ring_off[r + 1] - c0 - 1;
    const int lo = (int)first_at[i] - c0;
    head[i] |= (uint32_t)ext_off[i] << CHAIN_EXT_SHIFT;
    for (int j = 4; j <= len; ++j) ext[ext_off[i] + (j - 4)] = a.xy[c0 + (lo + j) % ne];
}
// ---- half-cell chains (gpk_index.h: GPK_HALF_CHAINS) -------------------------------------------------------------------------
// extended ring coordinates: ring r's n coordinates followed by CHAIN_MAX more, entry k = coordinate k % (n - 1)
__global__ void chain_xy_kernel(DevGeo a, double2* __restrict__ ext) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_ext = a.n_coords + a.n_rings * CHAIN_MAX;
    if (k >= n_ext) return;
    // ring of extended entry k: ext_off[r] = ring_off[r] + r * CHAIN_MAX is increasing in r
    int lo = 0, hi = (int)a.n_rings - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)a.ring_off[mid] + (int64_t)mid * CHAIN_MAX <= k)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int c0 = a.ring_off[lo], n = a.ring_off[lo + 1] - c0;
    const int j = (int)(k - ((int64_t)c0 + (int64_t)lo * CHAIN_MAX));
    const int ne = n - 1;
    ext[k] = n <= 0 ? make_double2(0.0, 0.0) : (ne >= 1 ? a.xy[c0 + j % ne] : a.xy[c0]);
}
// One LANE per half record (two per record): the half's chain word.  The lane walks the edges of the part's slab row of ITS half
// (PIP_SLAB_MUL = 2 base slab rows per raster row: a half cell is exactly one slab row, so the row's slab holds every edge whose
// y-range meets the half) straight from the slab table — about seven edges — instead of a wave staging the cell's edges for two
// working lanes (86 k waves of 2 busy lanes for the C2 right side: 0.23 ms; this form: a twentieth of the waves).
__global__ __launch_bounds__(256) void half_chain_kernel(DevGeo a, PipView pv, FineGrid g, const int32_t* __restrict__ work_cell,
                                                         const uint32_t* __restrict__ work_part, int64_t n_work,
                                                         const int32_t* __restrict__ slab_vidx, const SubCell* __restrict__ sub,
                                                         uint32_t* __restrict__ hword) {
    constexpr int S = PIP_SUB;
    static_assert(S == 8 && PIP_SLAB_MUL == 2, "a record is two halves of four sub-cell rows, one base slab row each");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = t >> 1;
    const int half = (int)(t & 1);
    if (item >= n_work) return;
    const int64_t c = work_cell[item];
    const int part = (int)work_part[item];
    const uint32_t lw0 = sub[item].labels[2 * half], lw1 = sub[item].labels[2 * half + 1];
    uint32_t word = 0u;
    int r0, r1;
    dev::part_rings(a, part, r0, r1);
    if (test_labels_of(lw0) + test_labels_of(lw1) > 0 && r1 - r0 == 1) {  // (a part with holes: no chains)
        const int ci = (int)(c % g.R), cj = (int)(c / g.R);
        const double fw2 = g.fw / S, fh2 = g.fh / S, px2 = g.pad_x / S, py2 = g.pad_y / S;
        const int sj0 = S * cj + (S / 2) * half;
        const double xl = g.rx0 + (double)(S * ci) * fw2 - px2, xh = g.rx0 + (double)(S * ci + S) * fw2 + px2;
        const double yl = g.ry0 + (double)sj0 * fh2 - py2, yh = g.ry0 + (double)(sj0 + S / 2) * fh2 + py2;
        const int c0 = a.ring_off[r0], ne = a.ring_off[r0 + 1] - c0 - 1;  // the ring's edges: 0 .. ne - 1
        bool closed = false;
        if (ne >= 1) {
            const double2 f = a.xy[c0], l = a.xy[c0 + ne];
            closed = f.x == l.x && f.y == l.y;  // (an unclosed ring has no closing edge for the other walks: no chain, they decide)
        }
        // the half's slab row (shift 0: lean indexes have no refined rings) — and, because the padding reaches into the neighbouring
        // rows by less than a row, the rows below and above it: an edge that only touches the PADDED half lives there
        const double cy = g.ry0 + ((double)sj0 + 0.25 * S) * fh2;
        const int row_c = pip::row_of(pv, cy);                     // finest row of the half's centre
        const int step = 1 << PIP_FINE_LOG2;                       // finest rows per base slab row
        int e0 = 0, e1 = 0;
        bool have = false;
        // 1. the edges that meet this padded half cell (exact: the test that labels sub-cells), as (first, last) ring positions of
        //    the shortest covering arc — found in one pass: the arc is grown edge by edge (edges arrive in no order)
        int lo = 0, len = 0;
        bool ok = closed;
        int n_touched = 0;
        int touched[CHAIN_MAX + 1];
        for (int dr = -1; dr <= 1 && ok; ++dr) {
            if (!pip::slab_range(pv, r0, row_c + dr * step, e0, e1)) continue;
            have = true;
            for (int e = e0; e < e1 && ok; ++e) {
                const double4 ed = pip::slab_edge(pv, e);
                if (fmax(ed.x, ed.z) < xl || fmin(ed.x, ed.z) > xh || fmax(ed.y, ed.w) < yl || fmin(ed.y, ed.w) > yh) continue;
                if (line_clear_of_rect(ed.x, ed.y, ed.z, ed.w, xl, yl, xh, yh)) continue;
                const int ei = pip::slab_vertex(slab_vidx[e]) - c0;
                bool seen = false;
                for (int q = 0; q < n_touched; ++q) seen = seen || touched[q] == ei;  // (an edge spanning rows is listed in each)
                if (seen) continue;
                if (n_touched >= CHAIN_MAX) {
                    ok = false;
                    break;
                }
                touched[n_touched++] = ei;
            }
        }
        ok = ok && have && n_touched > 0;
        if (ok) {
            // the shortest arc [lo, lo + len) of the cycle 0 .. ne - 1 that covers the touched edges: the complement of the widest
            // gap between one touched edge and the next one after it
            int best_gap = -1, best_next = 0;
            for (int p = 0; p < n_touched; ++p) {
                const int ea = touched[p];
                int nd = ne, nb = ea;
                for (int q = 0; q < n_touched; ++q) {
                    int dd = touched[q] - ea;
                    if (dd < 0) dd += ne;
                    if (dd > 0 && dd < nd) {
                        nd = dd;
                        nb = touched[q];
                    }
                }
                if (nd > best_gap) {
                    best_gap = nd;
                    best_next = nb;
                }
            }
            lo = best_next;
            len = ne - best_gap + 1;
            ok = len >= 1 && len <= CHAIN_MAX;
        }
        // 2. grow while the end vertex's y lies in the half's closed y-interval
        while (ok && len < ne) {
            int hv = lo + len;
            if (hv >= ne) hv -= ne;
            const double y = a.xy[c0 + hv].y;
            if (!(y >= yl && y <= yh)) break;
            ++len;
            ok = len <= CHAIN_MAX;
        }
        while (ok && len < ne) {
            const double y = a.xy[c0 + lo].y;
            if (!(y >= yl && y <= yh)) break;
            lo = lo == 0 ? ne - 1 : lo - 1;
            ++len;
            ok = len <= CHAIN_MAX;
        }
        if (ok) {
            // 3. base: the other edges' winding at the half's centre — they all sit in the centre's slab row
            const double cx = g.rx0 + ((double)(S * ci) + 0.5 * S) * fw2;
            int wn = 0;
            bool on = false;
            if (pip::slab_range(pv, r0, row_c, e0, e1)) {
                for (int e = e0; e < e1; ++e) {
                    int dd = pip::slab_vertex(slab_vidx[e]) - c0 - lo;
                    if (dd < 0) dd += ne;
                    if (dd < len) continue;  // an edge of the arc
                    const double4 ed = pip::slab_edge(pv, e);
                    on |= dev::ring_edge(ed.x, ed.y, ed.z, ed.w, cx, cy, wn);
                }
            }
            const int64_t start = (int64_t)c0 + (int64_t)r0 * CHAIN_MAX + lo;
            if (!on && wn >= -8 && wn <= 7 && start < ((int64_t)1 << (32 - HCHAIN_START_SHIFT)))
                word = (uint32_t)len | ((uint32_t)(wn & 15) << HCHAIN_BASE_SHIFT) | ((uint32_t)start << HCHAIN_START_SHIFT);
        }
    }
    hword[2 * item + half] = word;
}
__global__ void half_chain_commit_kernel(SubCell* __restrict__ sub, int64_t n_sub, const uint32_t* __restrict__ hword) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sub) return;
    const SubCell rc = sub[i];
    HalfCell* h = reinterpret_cast<HalfCell*>(sub + i);
    h[0] = HalfCell{{rc.labels[0], rc.labels[1]}, rc.part_flags & ~SUB_INDIRECT, hword[2 * i]};
    h[1] = HalfCell{{rc.labels[2], rc.labels[3]}, rc.part_flags & ~SUB_INDIRECT, hword[2 * i + 1]};
}
// an index with chains: every one-part record is rewritten as two half-cell records (gpk_index.h: HalfCell) — after chain_aux_kernel,
// which reads the SubCell form
__global__ void chain_commit_kernel(SubCell* __restrict__ sub, int64_t n_sub, const int32_t* __restrict__ aux_base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sub) return;
    const SubCell rc = sub[i];
    const uint32_t lower_tests = (uint32_t)(test_labels_of(rc.labels[0]) + test_labels_of(rc.labels[1]));
    HalfCell* h = reinterpret_cast<HalfCell*>(sub + i);
    h[0] = HalfCell{{rc.labels[0], rc.labels[1]}, rc.part_flags & ~SUB_INDIRECT, (uint32_t)aux_base[i]};
    h[1] = HalfCell{{rc.labels[2], rc.labels[3]}, rc.part_flags & ~SUB_INDIRECT, (uint32_t)aux_base[i] + lower_tests};
}
// LDS image of the level-1 routing (gpk_index.h: RouteWord): one thread per 32 cells of a raster row, after sub_commit_kernel
__global__ void route_build_kernel(const uint32_t* __restrict__ cell, int64_t n_words, RouteWord* __restrict__ route) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    RouteWord rw{0u, 0u, 0u, 0u};
    bool have = false;
    for (int i = 0; i < 32; ++i) {
        const uint32_t cw = cell[32 * w + i];
        const uint32_t tag = cw >> 30, payload = cw & 0x3FFFFFFFu;
        if (tag == CELL_TAG_SUB && !(payload & SUB2_BIT)) {
            rw.bmask |= 1u << i;
            if (!have) {
                rw.rec0 = payload;
                have = true;
            }
        } else if (cw != 0u) {
            rw.gmask |= 1u << i;
        }
    }
    route[w] = rw;
}

// RouteWord::pad = the 16-bit rank of rec0 among its raster row's records (rec0 - the rec0 of the row's first word that has records): the
// persistent point-join kernels keep 16-bit ranks + one base per row in LDS, and with the rank in the word their image load is one
// pass of independent 16-byte reads (the row's base = rec0 - pad of any of its words with records).  One thread per raster row.
__global__ void route_rank_kernel(RouteWord* __restrict__ route, int R) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    const int per_row = R / 32;
    RouteWord* rw = route + (int64_t)row * per_row;
    uint32_t base = 0u;
    bool have = false;
    for (int j = 0; j < per_row; ++j) {
        if (rw[j].bmask && !have) {
            base = rw[j].rec0;
            have = true;
        }
        rw[j].pad = rw[j].bmask ? rw[j].rec0 - base : 0u;
    }
}

// one-part records: the flagged cells as a work list (cell, part), record pos[c] — the build then launches one wave per
// RECORD instead of one per raster cell (two thirds of the cells of the C2 raster, nineteen in twenty of a 2048 x 2048 one,
// carry no record)
__global__ void sub_work_kernel(const uint32_t* __restrict__ cell, const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                int64_t n_cells, int32_t* __restrict__ work_cell, uint32_t* __restrict__ work_part) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells || !flag[c]) return;
    work_cell[pos[c]] = (int32_t)c;
    work_part[pos[c]] = (cell[c] & 0x3FFFFFFFu) >> 1;
}
__global__ void lrec_assign_kernel(const uint32_t* __restrict__ cell, uint32_t* __restrict__ list, int64_t n_cells,
                                   const int32_t* __restrict__ pos, int32_t* __restrict__ work_cell, uint32_t* __restrict__ work_part) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t w = cell[c];
    if ((w >> 30) != CELL_TAG_LIST) return;
    const uint32_t off = w & 0x3FFFFFFFu, m = list[off];
    int32_t o = pos[c];
    for (uint32_t t = 0; t < m; ++t) {
        const uint32_t e = list[off + 1 + t];
        if (!(e & 1u)) continue;
        work_cell[o] = (int32_t)c;
        work_part[o] = e >> 1;
        list[off + 1 + t] = ((uint32_t)o << 1) | 1u;  // the entry now names its record
        ++o;
    }
}

}  // namespace gpk

using namespace gpk;

namespace {
struct Temps {  // scratch of the build: carved from the thread's auxiliary arena, hipMalloc'ed (and released on every
                // exit path) only when the arena's estimate was too small
    void* p[48] = {nullptr};
    int n = 0;
    double malloc_ms = 0.0;  // GPK_DEBUG_INDEX: time spent in hipMalloc for temporaries the arena could not hold
    size_t malloc_bytes = 0;
    template <typename T>
    int32_t alloc(T** out, size_t count) {
        const size_t bytes = sizeof(T) * (count ? count : 1);
        void* q = gpk::workspace_aux(1).take(bytes);
        if (!q) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = cached_malloc(&q, bytes);
            malloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            malloc_bytes += bytes;
            if (e != hipSuccess) return fail(GPK_ERR_OOM, "index build: device_malloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            if (n < 48) p[n++] = q;
        }
        *out = (T*)q;
        return GPK_OK;
    }
    ~Temps() {
        const auto t0 = std::chrono::steady_clock::now();
        if (n) (void)hipDeviceSynchronize();  // (what hipFree did implicitly: no kernel of the build still reads a temporary; the build
                                              // runs on the calling thread's current device, which is the one that allocated them)
        for (int i = 0; i < n; ++i) cached_free(p[i]);
        if (getenv("GPK_DEBUG_INDEX"))
            fprintf(stderr, "[gpk] index build: %d temporaries beyond the arena (%.2f GB): hipMalloc %.3f ms, hipFree %.3f ms\n", n, (double)malloc_bytes / 1e9,
                    malloc_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};
inline dim3 blocks_for(int64_t n) { return dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)); }
}  // namespace

namespace gpk {

int32_t build_pip_index(const gpk_geoarray* a, gpk_index* ix, hipStream_t s, int list_records_mode) {
    memset(&ix->pip, 0, sizeof ix->pip);
    const DevGeo& d = a->d;
    if (!is_polygonal(d.type) || d.n_geoms == 0 || d.n_coords == 0 || d.n_rings == 0) return GPK_OK;
    const GridParams hg = ix->host_grid;
    const double w = hg.inv_w > 0.0 ? (double)hg.gx / hg.inv_w : 0.0, h = hg.inv_h > 0.0 ? (double)hg.gy / hg.inv_h : 0.0;
    if (!(w > 0.0) || !(h > 0.0) || !std::isfinite(w) || !std::isfinite(h)) return GPK_OK;  // degenerate extent

    int R = 64, r_max = 2048;
    if (const char* e = getenv("GPK_PIP_RMAX")) {  // tuning knob: cap of the level-1 raster side (power of two, 64..4096)
        const int v = atoi(e);
        if (v >= 64 && v <= 4096 && (v & (v - 1)) == 0) r_max = v;
    }
    while (R < r_max && (double)R < 2.0 * sqrt((double)d.n_coords)) R <<= 1;
    // a column of MANY small parts (C5: 6.5M parts of 5M multipolygons) puts 1.5 parts into every cell of the 2048 raster and every
    // point walks an entry list; one more doubling: tile 1.91 -> 1.54 ms, level-1 words 16 -> 64 MB
    // At that resolution the per-entry records of the remaining list cells (29M of them for C5) cost 97 ms of a 215 ms build to
    // take 2.03 -> 1.57 ms per 6.25M-point join: they are built on request only (GPK_INDEX_PIP_FULL), the entry lists get part boxes
    bool many_parts = false;
    if (!getenv("GPK_PIP_RMAX") && R == 2048 && d.n_parts > (int64_t)R * R / 2) {
        R = 4096;
        many_parts = true;
    }
    bool list_records = list_records_mode == 2 || (list_records_mode == 1 && !many_parts);
    if (const char* e = getenv("GPK_LIST_RECORDS")) list_records = atoi(e) != 0;  // tuning knob
    FineGrid g;
    g.R = R;
    g.fw = w / (double)(R - 3);
    g.fh = h / (double)(R - 3);
    g.rx0 = hg.x0 - 1.5 * g.fw;
    g.ry0 = hg.y0 - 1.5 * g.fh;
    g.inv_fw = 1.0 / g.fw;
    g.inv_fh = 1.0 / g.fh;
    g.pad_x = g.fw * (1.0 / 65536.0);
    g.pad_y = g.fh * (1.0 / 65536.0);
    // the padding must dwarf the rounding of (v - v0) * inv: 64 ulps of the largest coordinate magnitude
    const double max_abs = fmax(fmax(fabs(hg.x0), fabs(hg.x0 + w)), fmax(fabs(hg.y0), fabs(hg.y0 + h)));
    const double ulp64 = max_abs * 1.4210854715202004e-14;  // 64 * 2^-52
    if (!(g.pad_x > ulp64) || !(g.pad_y > ulp64)) return GPK_OK;

    const int64_t n_rings = d.n_rings, n_parts = d.n_parts, n_cells = (int64_t)R * R;
    // arena estimate for the temporaries: per-ring and per-cell arrays plus ~3 boundary marks per coordinate (24-byte
    // keys x 3 copies + flags); anything beyond it falls back to hipMalloc inside Temps
    // (capped: a multi-gigabyte arena costs more to map than the individual allocations it replaces)
    {
        // (nine per-cell arrays, the mark keys x 3, the sort's scratch, the chain pass: a temporary that misses the arena costs a
        // hipMalloc + a hipFree — 0.2 ms each on this runtime, a third of the whole build of a 1000-polygon right side)
        size_t est = (size_t)n_rings * 64 + (size_t)n_cells * 96 + (size_t)d.n_coords * 168 + (8u << 20);
        if (est > (size_t(256) << 20)) est = size_t(256) << 20;
        (void)workspace_aux(1).begin(est);
    }
    Temps t;
    // GPK_DEBUG_INDEX=1: wall time of every phase of the build (the stream is drained at each stamp)
    const bool dbg_time = getenv("GPK_DEBUG_INDEX") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (!dbg_time) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gpk] index build: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    int slot = 4;  // ix->owned[0..3] belong to the coarse directory
    auto keep = [&](void* p) {
        if (slot < 24) ix->owned[slot++] = p;
    };

    // ---- part / ring maps ----------------------------------------------------------------------
    uint32_t* part_geom = nullptr;
    if (d.type == GPK_GEOM_MULTIPOLYGON) {
        GPK_HIP(cached_malloc((void**)&part_geom, sizeof(uint32_t) * (size_t)(n_parts ? n_parts : 1)));
        keep(part_geom);
        GPK_LAUNCH("gpk_pipidx_part_geom", part_geom_kernel, blocks_for(d.n_geoms), dim3(256), 0, s, d, part_geom);
    }
    int32_t* ring_part;
    GPK_TRY(t.alloc(&ring_part, (size_t)n_rings));
    GPK_LAUNCH("gpk_pipidx_ring_part", ring_part_kernel, blocks_for(n_parts), dim3(256), 0, s, d, n_parts, ring_part);

    // ---- slabs ---------------------------------------------------------------------------------
    double4* ring_bbox;
    GPK_TRY(t.alloc(&ring_bbox, (size_t)n_rings));
    GPK_TRY(ring_bboxes(a, ring_bbox, s));
    stamp("ring maps + ring boxes");
    int32_t *row0 = nullptr, *nrows, *slab_base = nullptr;
    GPK_HIP(cached_malloc((void**)&row0, sizeof(int32_t) * (size_t)n_rings));
    keep(row0);
    GPK_HIP(cached_malloc((void**)&slab_base, sizeof(int32_t) * (size_t)(n_rings + 1)));
    keep(slab_base);
    GPK_TRY(t.alloc(&nrows, (size_t)n_rings));
    unsigned long long* btot;
    int64_t max_scan = n_cells > d.n_coords ? n_cells : d.n_coords;  // longest array scanned with btot (grown below for the slabs)
    if (n_rings > max_scan) max_scan = n_rings;
    GPK_TRY(t.alloc(&btot, (size_t)((max_scan + 255) / 256 + 4)));
    // the ring of every coordinate (edge_at): ring starts scattered, scanned in place; coord_ring[i] = the INCLUSIVE sum at i
    int32_t* ring_starts;
    GPK_TRY(t.alloc(&ring_starts, (size_t)d.n_coords + 2));
    GPK_HIP(hipMemsetAsync(ring_starts, 0, sizeof(int32_t) * (size_t)(d.n_coords + 2), s));
    GPK_LAUNCH("gpk_pipidx_ring_start", ring_start_kernel, blocks_for(n_rings), dim3(256), 0, s, d.ring_off, n_rings, d.n_coords, ring_starts);
    GPK_TRY(exclusive_scan_i32(ring_starts, d.n_coords + 1, ring_starts, nullptr, btot, s));
    const int32_t* coord_ring = ring_starts + 1;
    FineGrid gs = g;  // the FINEST slab-row grid (an exact power-of-two refinement of the raster row function); a ring's
                      // own rows are a right shift of it (ring_rows_kernel)
    gs.R = g.R * (PIP_SLAB_MUL << PIP_FINE_LOG2);
    gs.fw = g.fw / (PIP_SLAB_MUL << PIP_FINE_LOG2);
    gs.fh = g.fh / (PIP_SLAB_MUL << PIP_FINE_LOG2);
    gs.inv_fw = g.inv_fw * (double)(PIP_SLAB_MUL << PIP_FINE_LOG2);
    gs.inv_fh = g.inv_fh * (double)(PIP_SLAB_MUL << PIP_FINE_LOG2);
    int32_t n_slabs = 0, n_edges = 0;
    int32_t *slab_cnt = nullptr, *slab_off = nullptr, *cursor = nullptr;
    // refined rows assume short edges; a ring of many LONG edges (a comb) would register each of them in every refined
    // row, so a total far beyond the coordinate count sends the build back to the base rows
    const int slab_off_slot = slot++;  // owned by the index from the moment it exists (released with it on any error path)
    int32_t n_refined = 0, *n_refined_dev;
    GPK_TRY(t.alloc(&n_refined_dev, 64));
    for (int max_shift = PIP_FINE_LOG2;; max_shift = 0) {
        GPK_HIP(hipMemsetAsync(n_refined_dev, 0, sizeof(int32_t), s));
        GPK_LAUNCH("gpk_pipidx_ring_rows", ring_rows_kernel, blocks_for(n_rings), dim3(256), 0, s, ring_bbox, n_rings, d.ring_off, gs, max_shift,
                   row0, nrows, n_refined_dev);
        GPK_TRY(exclusive_scan_i32(nrows, n_rings, slab_base, nullptr, btot, s));
        GPK_HIP(d2h_small(&n_slabs, slab_base + n_rings, sizeof n_slabs, s));
        GPK_HIP(d2h_small(&n_refined, n_refined_dev, sizeof n_refined, s));
        GPK_HIP(sync_small(s));
        if ((int64_t)n_slabs > max_scan) {  // few-vertex rings spanning many slab rows: more slabs than coordinates
            max_scan = n_slabs;
            GPK_TRY(t.alloc(&btot, (size_t)((max_scan + 255) / 256 + 4)));
        }
        GPK_TRY(t.alloc(&slab_cnt, (size_t)n_slabs + 1));
        GPK_TRY(t.alloc(&cursor, (size_t)n_slabs + 1));
        if (slab_off) cached_free(slab_off);  // (the stream was drained just above)
        ix->owned[slab_off_slot] = slab_off = nullptr;
        GPK_HIP(cached_malloc((void**)&slab_off, sizeof(int32_t) * (size_t)(n_slabs + 1)));
        ix->owned[slab_off_slot] = slab_off;
        GPK_HIP(hipMemsetAsync(slab_cnt, 0, sizeof(int32_t) * (size_t)(n_slabs + 1), s));
        GPK_HIP(hipMemsetAsync(slab_off, 0, sizeof(int32_t) * (size_t)(n_slabs + 1), s));
        GPK_LAUNCH("gpk_pipidx_slab_count", slab_register_kernel<false>, blocks_for(d.n_coords), dim3(256), 0, s, d, coord_ring, gs, row0, slab_base,
                   slab_cnt, (double4*)nullptr);
        n_edges = 0;
        if (n_slabs > 0) {
            GPK_TRY(exclusive_scan_i32(slab_cnt, n_slabs, slab_off, cursor, btot, s));
            GPK_HIP(d2h_small(&n_edges, slab_off + n_slabs, sizeof n_edges, s));
            GPK_HIP(sync_small(s));
        }
        if (max_shift == 0 || (int64_t)n_edges <= 8 * d.n_coords + (1 << 20)) break;
    }
    stamp("slab rows + counts");
    // slab entries: 32-byte edge copies for small right sides (one load level in the exact walk), 4-byte coordinate indices beyond
    // GPK_SLAB_COPY_MAX_MB (default 256) of copies — the copies of a 144M-coordinate column were 5 of its index's 6.4 GB
    double4* edges = nullptr;
    int32_t* slab_vidx = nullptr;  // also read by the chain build of lean indexes
    size_t copy_max = size_t(256) << 20;
    if (const char* e = getenv("GPK_SLAB_COPY_MAX_MB")) copy_max = (size_t)atoll(e) << 20;
    const bool edge_copies = sizeof(double4) * (size_t)n_edges <= copy_max;
    if (edge_copies) {
        GPK_HIP(cached_malloc((void**)&edges, sizeof(double4) * (size_t)(n_edges ? n_edges : 1)));
        keep(edges);
        if ((int64_t)n_edges <= ((int64_t)64 << 20)) GPK_TRY(t.alloc(&slab_vidx, (size_t)(n_edges ? n_edges : 1)));
    } else {
        GPK_HIP(cached_malloc((void**)&slab_vidx, sizeof(int32_t) * (size_t)n_edges));
        keep(slab_vidx);
    }
    GPK_LAUNCH("gpk_pipidx_slab_fill", slab_register_kernel<true>, blocks_for(d.n_coords), dim3(256), 0, s, d, coord_ring, gs, row0, slab_base,
               cursor, edges, slab_vidx);

    stamp("slab fill (+ edges malloc)");
    PartInfo* part_info = nullptr;
    GPK_HIP(cached_malloc((void**)&part_info, sizeof(PartInfo) * (size_t)(n_parts ? n_parts : 1)));
    keep(part_info);
    GPK_LAUNCH("gpk_pipidx_part_info", part_info_kernel, blocks_for(n_parts), dim3(256), 0, s, d, n_parts, row0, slab_base, part_info);

    PipView pv;
    memset(&pv, 0, sizeof pv);
    pv.R = R;
    pv.rx0 = g.rx0;
    pv.ry0 = g.ry0;
    pv.fw = g.fw;
    pv.fh = g.fh;
    pv.inv_fw = g.inv_fw;
    pv.inv_fh = g.inv_fh;
    pv.part_geom = part_geom;
    pv.part_info = part_info;
    pv.ring_row0 = row0;
    pv.ring_slab_base = slab_base;
    pv.slab_off = slab_off;
    pv.slab_edges = edges;
    pv.slab_vidx = edge_copies ? nullptr : slab_vidx;
    pv.slab_xy = d.xy;

    stamp("part info");
    // ---- boundary marks: (cell, part) keys, sorted + unique ---------------------------------------
    int32_t *mark_cnt, *mark_off;
    GPK_TRY(t.alloc(&mark_cnt, (size_t)d.n_coords + 1));
    GPK_TRY(t.alloc(&mark_off, (size_t)d.n_coords + 1));
    GPK_LAUNCH("gpk_pipidx_mark_count", mark_kernel<false>, blocks_for(d.n_coords), dim3(256), 0, s, d, coord_ring, g, ring_part, mark_cnt,
               (unsigned long long*)nullptr);
    GPK_TRY(exclusive_scan_i32(mark_cnt, d.n_coords, mark_off, nullptr, btot, s));
    unsigned long long n_marks_raw = 0;  // the scan keeps its grand total in 64 bits
    GPK_HIP(d2h_small(&n_marks_raw, btot + (d.n_coords + 255) / 256, sizeof n_marks_raw, s));
    GPK_HIP(sync_small(s));
    if (n_marks_raw >= (1ull << 31))
        return GPK_OK;  // pathological (huge edges over a fine raster): leave the accelerator off
    stamp("mark count + scan");
    unsigned long long *keys, *sorted, *marks;
    GPK_TRY(t.alloc(&keys, (size_t)n_marks_raw));
    GPK_TRY(t.alloc(&sorted, (size_t)n_marks_raw));
    marks = keys;  // the unsorted keys are dead once the sort has run: the unique marks are compacted into their buffer
    int64_t n_marks = 0;
    int32_t* mark_start;  // marks of cell c: [mark_start[c], mark_start[c + 1])
    GPK_TRY(t.alloc(&mark_start, (size_t)n_cells + 2));
    GPK_HIP(hipMemsetAsync(mark_start, 0, sizeof(int32_t) * (size_t)(n_cells + 2), s));
    if (n_marks_raw > 0) {
        GPK_LAUNCH("gpk_pipidx_mark_fill", mark_kernel<true>, blocks_for(d.n_coords), dim3(256), 0, s, d, coord_ring, g, ring_part, mark_off, keys);
        size_t tmp_bytes = 0;
        // (cell << 32 | part: the cell number has cell_bits significant bits — one radix pass fewer than over all 64)
        unsigned cell_bits = 1;
        while (cell_bits < 32 && (1ll << cell_bits) < n_cells) ++cell_bits;
        const unsigned sort_end = 32 + cell_bits;
        GPK_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, sorted, (size_t)n_marks_raw, 0, sort_end, s));
        char* tmp;
        GPK_TRY(t.alloc(&tmp, tmp_bytes));
        GPK_HIP(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, sorted, (size_t)n_marks_raw, 0, sort_end, s));
        int32_t *flag, *pos;
        GPK_TRY(t.alloc(&flag, (size_t)n_marks_raw + 1));
        GPK_TRY(t.alloc(&pos, (size_t)n_marks_raw + 1));
        unsigned long long* btot2;
        GPK_TRY(t.alloc(&btot2, (size_t)((n_marks_raw + 255) / 256 + 4)));
        GPK_LAUNCH("gpk_pipidx_unique_flags", unique_flags_kernel, blocks_for((int64_t)n_marks_raw), dim3(256), 0, s, sorted,
                   (int64_t)n_marks_raw, flag);
        GPK_TRY(exclusive_scan_i32(flag, (int64_t)n_marks_raw, pos, nullptr, btot2, s));
        GPK_LAUNCH("gpk_pipidx_unique_compact", unique_compact_kernel, blocks_for((int64_t)n_marks_raw), dim3(256), 0, s, sorted,
                   (int64_t)n_marks_raw, pos, marks, mark_start);
        GPK_TRY(exclusive_scan_i32(mark_start, n_cells, mark_start, nullptr, btot, s));
        int32_t nu = 0;
        GPK_HIP(d2h_small(&nu, pos + n_marks_raw, sizeof nu, s));
        GPK_HIP(sync_small(s));
        n_marks = nu;
    }

    stamp("mark fill + sort + unique");
    // ---- cells ----------------------------------------------------------------------------------
    uint32_t* cell = nullptr;
    GPK_HIP(cached_malloc((void**)&cell, sizeof(uint32_t) * (size_t)n_cells));
    keep(cell);
    int32_t *need, *list_off;
    GPK_TRY(t.alloc(&need, (size_t)n_cells + 1));
    GPK_TRY(t.alloc(&list_off, (size_t)n_cells + 1));
    uint32_t* cell_scratch = nullptr;  // 32 bytes a cell (C5: 0.5 GB of temporaries against a second pass over every candidate's slabs)
    GPK_TRY(t.alloc(&cell_scratch, (size_t)n_cells * CELL_SCRATCH));
    GPK_LAUNCH("gpk_pipidx_cell_count", cell_build_kernel<false>, blocks_for(n_cells), dim3(256), 0, s, d, ix->v, pv, g, marks, mark_start,
               need, (const int32_t*)nullptr, cell, (uint32_t*)nullptr, cell_scratch, (const double4*)ring_bbox);
    GPK_TRY(exclusive_scan_i32(need, n_cells, list_off, nullptr, btot, s));
    int32_t list_len = 0;
    GPK_HIP(d2h_small(&list_len, list_off + n_cells, sizeof list_len, s));
    GPK_HIP(sync_small(s));
    if ((unsigned)list_len >= (1u << 30)) return GPK_OK;
    uint32_t* list = nullptr;
    GPK_HIP(cached_malloc((void**)&list, sizeof(uint32_t) * (size_t)(list_len ? list_len : 1)));
    keep(list);
    GPK_LAUNCH("gpk_pipidx_cell_fill", cell_build_kernel<true>, blocks_for(n_cells), dim3(256), 0, s, d, ix->v, pv, g, marks, mark_start,
               need, list_off, cell, list, cell_scratch, (const double4*)ring_bbox);
    GPK_HIP(hipStreamSynchronize(s));
    pv.cell = cell;
    pv.list = list;
    ix->pip_list_heavy = (int64_t)list_len * 4 > n_cells ? 1 : 0;

    stamp("cells");
    // ---- level 2 ------------------------------------------------------------------------------------
    int32_t n_sub = 0, n_sub2 = 0;
    bool sub_overflow = false;
    int32_t* swork_cell = nullptr;   // the one-part records' (cell, part) work list: reused by the chain pass below
    uint32_t* swork_part = nullptr;
    SubCell* sub = nullptr;
    SubCell2* sub2 = nullptr;
    static_assert(PIP_SLAB_MUL == 2, "SubCell stores exactly two adjacent slab ranges");
    const bool level2_ok = g.pad_x / PIP_SUB > ulp64 && g.pad_y / PIP_SUB > ulp64 && R * PIP_SUB <= 32768;
    if (level2_ok) {
        int32_t *sflag, *spos, *sflag2, *spos2;
        GPK_TRY(t.alloc(&sflag, (size_t)n_cells + 1));
        GPK_TRY(t.alloc(&spos, (size_t)n_cells + 1));
        GPK_TRY(t.alloc(&sflag2, (size_t)n_cells + 1));
        GPK_TRY(t.alloc(&spos2, (size_t)n_cells + 1));
        GPK_LAUNCH("gpk_pipidx_sub_flag", sub_flag_kernel, blocks_for(n_cells), dim3(256), 0, s, cell, (const uint32_t*)list, n_cells, sflag, sflag2);
        GPK_TRY(exclusive_scan_i32(sflag, n_cells, spos, nullptr, btot, s));
        GPK_TRY(exclusive_scan_i32(sflag2, n_cells, spos2, nullptr, btot, s));
        GPK_HIP(d2h_small(&n_sub, spos + n_cells, sizeof n_sub, s));
        GPK_HIP(d2h_small(&n_sub2, spos2 + n_cells, sizeof n_sub2, s));
        GPK_HIP(sync_small(s));
        if ((unsigned)n_sub >= SUB2_BIT || (unsigned)n_sub2 >= SUB2_BIT) {  // would not fit the cell word: no level 2
            n_sub = n_sub2 = 0;
            sub_overflow = true;
        }
        // two-part records make the join kernel carry a second part per point and cost a second labelling pass: they pay off
        // when shared borders are THE boundary shape (a tessellation: 17 two-part cells per one-part cell); columns of
        // overlapping polygons have about as many of either kind, most of their sub-cells end up "test both", and the pass
        // only added 12 ms to a 36 ms build — those keep the entry lists
        if (getenv("GPK_DEBUG_INDEX")) fprintf(stderr, "[gpk] level 2: %d one-part cells, %d two-part cells of %lld\n", n_sub, n_sub2, (long long)n_cells);
        if ((int64_t)n_sub2 < 4 * (int64_t)n_sub) n_sub2 = 0;
        if (n_sub > 0) {
            GPK_HIP(cached_malloc((void**)&sub, sizeof(SubCell) * (size_t)n_sub));
            keep(sub);
            GPK_HIP(hipMemsetAsync(sub, 0, sizeof(SubCell) * (size_t)n_sub, s));
            GPK_TRY(t.alloc(&swork_cell, (size_t)n_sub));
            GPK_TRY(t.alloc(&swork_part, (size_t)n_sub));
            GPK_LAUNCH("gpk_pipidx_sub_work", sub_work_kernel, blocks_for(n_cells), dim3(256), 0, s, (const uint32_t*)cell, (const int32_t*)sflag,
                       (const int32_t*)spos, n_cells, swork_cell, swork_part);
            GPK_LAUNCH("gpk_pipidx_sub_head", sub_head_kernel, blocks_for(n_sub), dim3(256), 0, s, pv, g, (const int32_t*)swork_cell,
                       (const uint32_t*)swork_part, (int64_t)n_sub, sub);
            // no ring with refined rows: nearly every record takes the fast path — its own instance (twice the waves per SIMD), then one
            // for the stragglers (parts with holes); a column WITH refined rings (C5: 15 % of the records) keeps the single launch,
            // where a second pass over all records to find the others costs more than the occupancy gives (measured: +1 ms / -10 %)
            static const bool force_split = getenv("GPK_SUB_SPLIT") != nullptr;  // A/B runs: the two instances also for columns with refined rings
            if (n_refined == 0 || force_split) {
                GPK_LAUNCH("gpk_pipidx_sub_build_fast", (sub_build_kernel<1, true, 1>), blocks_for((int64_t)n_sub * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n_cells, (const uint32_t*)cell, (const uint32_t*)list, sub, (SubCell2*)nullptr,
                           (const int32_t*)swork_cell, (const uint32_t*)swork_part, (int64_t)n_sub);
                GPK_LAUNCH("gpk_pipidx_sub_build_rest", (sub_build_kernel<1, true, 2>), blocks_for((int64_t)n_sub * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n_cells, (const uint32_t*)cell, (const uint32_t*)list, sub, (SubCell2*)nullptr,
                           (const int32_t*)swork_cell, (const uint32_t*)swork_part, (int64_t)n_sub);
            } else {
                GPK_LAUNCH("gpk_pipidx_sub_build", (sub_build_kernel<1, true, 0>), blocks_for((int64_t)n_sub * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n_cells, (const uint32_t*)cell, (const uint32_t*)list, sub, (SubCell2*)nullptr,
                           (const int32_t*)swork_cell, (const uint32_t*)swork_part, (int64_t)n_sub);
            }
        }
        if (n_sub2 > 0) {
            GPK_HIP(cached_malloc((void**)&sub2, sizeof(SubCell2) * (size_t)n_sub2));
            keep(sub2);
            GPK_HIP(hipMemsetAsync(sub2, 0, sizeof(SubCell2) * (size_t)n_sub2, s));
            GPK_LAUNCH("gpk_pipidx_sub2_build", sub_build_kernel<2>, blocks_for(n_cells * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g, sflag2, spos2,
                       n_cells, (const uint32_t*)cell, (const uint32_t*)list, (SubCell*)nullptr, sub2);
        }
        if (n_sub > 0 || n_sub2 > 0) {
            GPK_LAUNCH("gpk_pipidx_sub_commit", sub_commit_kernel, blocks_for(n_cells), dim3(256), 0, s, sflag, spos, n_sub2 > 0 ? sflag2 : nullptr,
                       spos2, n_cells, cell);
            GPK_HIP(hipStreamSynchronize(s));
        }
    }
    stamp("level-2 records");
    pv.sub = sub;
    pv.sub2 = sub2;
    // records for the boundary entries of the cells that stayed lists (after the commit: two-part cells are gone)
    int32_t n_lrec = 0;
    SubCell* lrec = nullptr;
    if (level2_ok && list_len > 0 && list_records && !getenv("GPK_NO_LIST_RECORDS")) {
        int32_t *lcnt, *lpos;
        GPK_TRY(t.alloc(&lcnt, (size_t)n_cells + 1));
        GPK_TRY(t.alloc(&lpos, (size_t)n_cells + 1));
        GPK_LAUNCH("gpk_pipidx_lrec_count", lrec_count_kernel, blocks_for(n_cells), dim3(256), 0, s, (const uint32_t*)cell, (const uint32_t*)list, n_cells,
                   lcnt);
        GPK_TRY(exclusive_scan_i32(lcnt, n_cells, lpos, nullptr, btot, s));
        GPK_HIP(d2h_small(&n_lrec, lpos + n_cells, sizeof n_lrec, s));
        GPK_HIP(sync_small(s));
        if (getenv("GPK_DEBUG_INDEX")) fprintf(stderr, "[gpk] level 2: %d boundary entries in list cells (list length %d)\n", n_lrec, list_len);
        if (n_lrec > 0 && (int64_t)n_lrec < (int64_t)1 << 28) {  // 2^28 records = 8 GB: beyond that the lists stay plain
            int32_t* work_cell;
            uint32_t* work_part;
            GPK_TRY(t.alloc(&work_cell, (size_t)n_lrec));
            GPK_TRY(t.alloc(&work_part, (size_t)n_lrec));
            GPK_HIP(cached_malloc((void**)&lrec, sizeof(SubCell) * (size_t)n_lrec));
            keep(lrec);
            GPK_HIP(hipMemsetAsync(lrec, 0, sizeof(SubCell) * (size_t)n_lrec, s));
            GPK_LAUNCH("gpk_pipidx_lrec_assign", lrec_assign_kernel, blocks_for(n_cells), dim3(256), 0, s, (const uint32_t*)cell, list, n_cells,
                       (const int32_t*)lpos, work_cell, work_part);
            GPK_LAUNCH("gpk_pipidx_lrec_head", sub_head_kernel, blocks_for(n_lrec), dim3(256), 0, s, pv, g, (const int32_t*)work_cell,
                       (const uint32_t*)work_part, (int64_t)n_lrec, lrec);
            if (n_refined == 0) {  // (as for the one-part records above)
                GPK_LAUNCH("gpk_pipidx_lrec_build", (sub_build_kernel<1, true, 1>), blocks_for((int64_t)n_lrec * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n_cells, (const uint32_t*)cell, (const uint32_t*)list, lrec, (SubCell2*)nullptr,
                           (const int32_t*)work_cell, (const uint32_t*)work_part, (int64_t)n_lrec);
                GPK_LAUNCH("gpk_pipidx_lrec_build", (sub_build_kernel<1, true, 2>), blocks_for((int64_t)n_lrec * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n_cells, (const uint32_t*)cell, (const uint32_t*)list, lrec, (SubCell2*)nullptr,
                           (const int32_t*)work_cell, (const uint32_t*)work_part, (int64_t)n_lrec);
            } else {
                GPK_LAUNCH("gpk_pipidx_lrec_build", (sub_build_kernel<1, true, 0>), blocks_for((int64_t)n_lrec * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n_cells, (const uint32_t*)cell, (const uint32_t*)list, lrec, (SubCell2*)nullptr,
                           (const int32_t*)work_cell, (const uint32_t*)work_part, (int64_t)n_lrec);
            }
            GPK_HIP(hipStreamSynchronize(s));
        } else {
            n_lrec = 0;
        }
    }
    stamp("list-cell records");
    pv.lrec = lrec;
    if (list_len > 0 && !lrec && !getenv("GPK_NO_PART_BOX")) {  // plain entry lists: a box per part rejects most (point, entry) pairs early
        float4* part_box = nullptr;
        GPK_HIP(cached_malloc((void**)&part_box, sizeof(float4) * (size_t)(n_parts ? n_parts : 1)));
        keep(part_box);
        GPK_LAUNCH("gpk_pipidx_part_box", part_box_kernel, blocks_for(n_parts), dim3(256), 0, s, d, n_parts, (const double4*)ring_bbox, part_box);
        pv.part_box = part_box;
        ix->nbytes += (int64_t)(sizeof(float4) * (size_t)n_parts);
    }
    // (Nearly) every cell is empty / strictly inside one part / crossed by one part with an inline level-2 record: a point has
    // one candidate part and the join runs its lean kernel (gpk_join.hip: pip_tile_lean_kernel).  Disjoint polygons
    // (parcels, the C2 right side) look like this; overlaps, shared borders and refined rings keep the general kernel.
    const bool boundary_cells_have_records = level2_ok && !sub_overflow && n_sub2 == 0;  // every `part << 1 | 1` cell became a record
    // (a few list cells are tolerated — polygons whose boxes touch share a cell here and there; the lean kernel sends
    // their points through the generic walk: at most 1 list word per 8 one-part records)
    ix->pip_lean = (int64_t)list_len * 8 <= (int64_t)n_sub && n_refined == 0 && boundary_cells_have_records ? 1 : 0;
    if (getenv("GPK_DEBUG_INDEX")) fprintf(stderr, "[gpk] lean join kernel eligible: %d (list %d, refined rings: %s, one-part records %d)\n", ix->pip_lean, list_len, n_refined ? "some" : "none", n_sub);
    // local chains for the `test` sub-cells of a lean index (gpk_index.h: ChainAux): the join then decides them in the owning lane
    if (GPK_HALF_CHAINS && ix->pip_lean && slab_vidx && n_sub > 0 && swork_cell && !getenv("GPK_NO_CHAINS") &&
        d.n_coords + d.n_rings * CHAIN_MAX < ((int64_t)1 << (32 - HCHAIN_START_SHIFT))) {
        // half-cell chains (gpk_index.h: GPK_HALF_CHAINS): the extended coordinates, one chain word per half record, the records
        // rewritten as half-cell records carrying their word — three launches, no table sized by a read-back
        const int64_t n_ext = d.n_coords + d.n_rings * CHAIN_MAX;
        double2* cxy = nullptr;
        GPK_HIP(cached_malloc((void**)&cxy, sizeof(double2) * (size_t)n_ext));
        keep(cxy);
        uint32_t* hword;
        GPK_TRY(t.alloc(&hword, (size_t)n_sub * 2 + 2));
        pv.sub = sub;
        GPK_LAUNCH("gpk_pipidx_chain_xy", chain_xy_kernel, blocks_for(n_ext), dim3(256), 0, s, d, cxy);
        GPK_LAUNCH("gpk_pipidx_half_chain", half_chain_kernel, blocks_for((int64_t)n_sub * 2), dim3(256), 0, s, d, pv, g,
                   (const int32_t*)swork_cell, (const uint32_t*)swork_part, (int64_t)n_sub, (const int32_t*)slab_vidx, (const SubCell*)sub, hword);
        GPK_LAUNCH("gpk_pipidx_chain_commit", half_chain_commit_kernel, blocks_for(n_sub), dim3(256), 0, s, sub, (int64_t)n_sub, (const uint32_t*)hword);
        pv.chain_xy = cxy;
        ix->nbytes += (int64_t)(sizeof(double2) * (size_t)n_ext);
        if (R <= PIP_ROUTE_RMAX && !getenv("GPK_NO_ROUTE_IMAGE")) {
            RouteWord* route = nullptr;
            const int64_t n_words = n_cells / 32;
            GPK_HIP(cached_malloc((void**)&route, sizeof(RouteWord) * (size_t)n_words));
            keep(route);
            GPK_LAUNCH("gpk_pipidx_route", route_build_kernel, blocks_for(n_words), dim3(256), 0, s, (const uint32_t*)cell, n_words, route);
            if (R >= 32) GPK_LAUNCH("gpk_pipidx_route_rank", route_rank_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, s, route, R);
            pv.route = route;
            ix->nbytes += (int64_t)(sizeof(RouteWord) * (size_t)n_words);
        }
        GPK_HIP(hipStreamSynchronize(s));  // (the temporaries go back to the arena when this function returns)
        if (getenv("GPK_DEBUG_INDEX")) fprintf(stderr, "[gpk] half-cell chains: %d records, %lld extended coordinates%s\n", n_sub, (long long)n_ext, pv.route ? ", routing image" : "");
        stamp("local chains");
    } else if (!GPK_HALF_CHAINS && ix->pip_lean && slab_vidx && n_sub > 0 && swork_cell && !getenv("GPK_NO_CHAINS")) {
        int32_t *ccnt, *cbase_tmp;
        GPK_TRY(t.alloc(&ccnt, (size_t)n_sub + 1));
        GPK_TRY(t.alloc(&cbase_tmp, (size_t)n_sub + 1));
        pv.sub = sub;
        GPK_LAUNCH("gpk_pipidx_chain_count", chain_count_kernel, blocks_for(n_sub), dim3(256), 0, s, (const SubCell*)sub, (int64_t)n_sub, ccnt);
        GPK_TRY(exclusive_scan_i32(ccnt, n_sub, cbase_tmp, nullptr, btot, s));
        int32_t n_aux = 0;
        GPK_HIP(d2h_small(&n_aux, cbase_tmp + n_sub, sizeof n_aux, s));
        GPK_HIP(sync_small(s));
        if (n_aux > 0) {
            ChainAux* aux = nullptr;
            GPK_HIP(cached_malloc((void**)&aux, sizeof(ChainAux) * (size_t)n_aux));
            keep(aux);
            uint32_t* chead = nullptr;
            GPK_HIP(cached_malloc((void**)&chead, sizeof(uint32_t) * (size_t)n_aux));
            keep(chead);
            int32_t *ext_need, *ext_off;
            uint32_t* first_at;
            GPK_TRY(t.alloc(&ext_need, (size_t)n_aux + 1));
            GPK_TRY(t.alloc(&ext_off, (size_t)n_aux + 1));
            GPK_TRY(t.alloc(&first_at, (size_t)n_aux + 1));
            unsigned long long* btot3;
            GPK_TRY(t.alloc(&btot3, (size_t)((n_aux + 255) / 256 + 4)));
            GPK_LAUNCH("gpk_pipidx_chain_aux", chain_aux_kernel, blocks_for((int64_t)n_sub * PIP_SUB * PIP_SUB), dim3(256), 0, s, d, pv, g,
                       (const int32_t*)swork_cell, (const uint32_t*)swork_part, (int64_t)n_sub, (const int32_t*)slab_vidx, (const SubCell*)sub,
                       (const int32_t*)cbase_tmp, aux, chead, first_at, ext_need);
            GPK_TRY(exclusive_scan_i32(ext_need, n_aux, ext_off, nullptr, btot3, s));
            int32_t n_ext = 0;
            GPK_HIP(d2h_small(&n_ext, ext_off + n_aux, sizeof n_ext, s));
            GPK_HIP(sync_small(s));
            double2* ext = nullptr;
            GPK_HIP(cached_malloc((void**)&ext, sizeof(double2) * (size_t)(n_ext > 0 ? n_ext : 1)));
            keep(ext);
            if (n_ext > 0)
                GPK_LAUNCH("gpk_pipidx_chain_ext", chain_ext_kernel, blocks_for(n_aux), dim3(256), 0, s, d, chead, (const uint32_t*)first_at, (int64_t)n_aux,
                           (const int32_t*)ext_off, ext);
            pv.chain_head = chead;
            pv.chain_ext = ext;
            ix->nbytes += (int64_t)(sizeof(double2) * (size_t)n_ext + sizeof(uint32_t) * (size_t)n_aux);
            // (after chain_aux_kernel, which still reads the records' labels only: e0 now names the record's first chain entry)
            GPK_LAUNCH("gpk_pipidx_chain_commit", chain_commit_kernel, blocks_for(n_sub), dim3(256), 0, s, sub, (int64_t)n_sub, (const int32_t*)cbase_tmp);
            pv.sub_aux = aux;
            ix->nbytes += (int64_t)(sizeof(ChainAux) * (size_t)n_aux);
            if (R <= PIP_ROUTE_RMAX && !getenv("GPK_NO_ROUTE_IMAGE")) {
                RouteWord* route = nullptr;
                const int64_t n_words = n_cells / 32;
                GPK_HIP(cached_malloc((void**)&route, sizeof(RouteWord) * (size_t)n_words));
                keep(route);
                GPK_LAUNCH("gpk_pipidx_route", route_build_kernel, blocks_for(n_words), dim3(256), 0, s, (const uint32_t*)cell, n_words, route);
            if (R >= 32) GPK_LAUNCH("gpk_pipidx_route_rank", route_rank_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, s, route, R);
                pv.route = route;
                ix->nbytes += (int64_t)(sizeof(RouteWord) * (size_t)n_words);
            }
            GPK_HIP(hipStreamSynchronize(s));
            if (getenv("GPK_DEBUG_INDEX")) fprintf(stderr, "[gpk] local chains: %d test sub-cells in %d records%s\n", n_aux, n_sub, pv.route ? ", routing image" : "");
        }
        stamp("local chains");
    }
    ix->nbytes += (int64_t)(sizeof(SubCell) * (size_t)n_sub + sizeof(SubCell2) * (size_t)n_sub2 + sizeof(SubCell) * (size_t)n_lrec);
    ix->pip = pv;
    ix->nbytes += (int64_t)(sizeof(uint32_t) * (size_t)n_cells + sizeof(uint32_t) * (size_t)list_len + (pv.slab_edges ? sizeof(double4) : sizeof(int32_t)) * (size_t)n_edges +
                            sizeof(int32_t) * (size_t)(n_slabs + 1) + sizeof(int32_t) * (size_t)(2 * n_rings + 1) +
                            (part_geom ? sizeof(uint32_t) * (size_t)n_parts : 0) + sizeof(PartInfo) * (size_t)n_parts);
    return GPK_OK;
}

}  // namespace gpk
