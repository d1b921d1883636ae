// gpk_join.hip — spatial index build + the join refine of geopolars/src/spatial_index.rs:37-143.
//
//   gpk_index_build   == SpatialIndex::try_from(&Series)            spatial_index.rs:320-334
//   gpk_spatial_join  == intersection_candidates_with_other_tree     spatial_index.rs:74-76
//                        + the exact refine loop                     spatial_index.rs:83-143
//
// Point x polygonal rows (the 10M x 1k headline) run as
//   pip_tile   : a work-group classifies a tile of points (raster routing + LDS-compacted exact phase over
//                edge slabs, see the kernel); emits the optional per-point hit count, a 4-byte result
//                code per point and one 64-bit total per work-group.
//   pip_write  : codes -> (l, r) pairs in sorted order; its global offsets come from a two-level sum of the
//                work-group totals (no scan kernel).
// (A single-pass variant with decoupled look-back was measured and was slower on MI355X: the in-order
// commit makes finished work-groups hold their LDS/wave slots while they wait — see DESIGN.md.)
#include <atomic>
#include <cstring>
#include <mutex>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include <chrono>

#include "gpk_device.h"
#include "gpk_index.h"
#include "gpk_pip.h"
#include "gpk_polypoly.h"
#include "gpk_contains.h"
#include "gpk_lineal.h"
#include "gpk_scan.h"
#include "gpk_pipshared.h"

namespace gpk {

// ================================= index build ==================================================
// stage 1 of the extent: one closed box per work-group (NaN = nothing but empty geometries), in the boxes' own format, so
// that extent_kernel folds them like boxes (min / max are exact: the result does not depend on the split)
__global__ __launch_bounds__(256) void extent_partial_kernel(const double4* __restrict__ bbox, int64_t n, double4* __restrict__ part) {
    __shared__ double red[4][4];
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double4 b = bbox[i];
        if (b.x == b.x) {
            mnx = fmin(mnx, b.x);
            mny = fmin(mny, b.y);
            mxx = fmax(mxx, b.z);
            mxy = fmax(mxy, b.w);
        }
    }
    mnx = dev::wave_min(mnx);
    mny = dev::wave_min(mny);
    mxx = dev::wave_max(mxx);
    mxy = dev::wave_max(mxy);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = mnx;
        red[1][wave] = mny;
        red[2][wave] = mxx;
        red[3][wave] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mnx = fmin(mnx, red[0][w]);
            mny = fmin(mny, red[1][w]);
            mxx = fmax(mxx, red[2][w]);
            mxy = fmax(mxy, red[3][w]);
        }
        part[blockIdx.x] = mnx <= mxx ? make_double4(mnx, mny, mxx, mxy) : make_double4(NAN, NAN, NAN, NAN);
    }
}
__global__ __launch_bounds__(1024) void extent_kernel(const double4* __restrict__ bbox, int64_t n,
                                                      int gx, int gy, GridParams* __restrict__ out) {
    __shared__ double red[4][16];
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double4 b = bbox[i];
        if (b.x == b.x) {  // NaN marks an empty geometry
            mnx = fmin(mnx, b.x);
            mny = fmin(mny, b.y);
            mxx = fmax(mxx, b.z);
            mxy = fmax(mxy, b.w);
        }
    }
    mnx = dev::wave_min(mnx);
    mny = dev::wave_min(mny);
    mxx = dev::wave_max(mxx);
    mxy = dev::wave_max(mxy);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = mnx;
        red[1][wave] = mny;
        red[2][wave] = mxx;
        red[3][wave] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            mnx = fmin(mnx, red[0][w]);
            mny = fmin(mny, red[1][w]);
            mxx = fmax(mxx, red[2][w]);
            mxy = fmax(mxy, red[3][w]);
        }
        GridParams g;
        const bool any = mnx <= mxx;
        g.x0 = any ? mnx : 0.0;
        g.y0 = any ? mny : 0.0;
        const double w = any ? mxx - mnx : 0.0, h = any ? mxy - mny : 0.0;
        g.inv_w = w > 0.0 ? (double)gx / w : 0.0;
        g.inv_h = h > 0.0 ? (double)gy / h : 0.0;
        g.gx = gx;
        g.gy = gy;
        *out = g;
    }
}

template <bool FILL>
__global__ void grid_register_kernel(const double4* __restrict__ bbox, int64_t n,
                                     const GridParams* __restrict__ gp, int32_t* __restrict__ cell_cnt,
                                     int32_t* __restrict__ items) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double4 b = bbox[j];
    if (!(b.x == b.x)) return;
    const GridParams g = *gp;
    const int cx0 = dev::cell_of(b.x, g.x0, g.inv_w, g.gx), cx1 = dev::cell_of(b.z, g.x0, g.inv_w, g.gx);
    const int cy0 = dev::cell_of(b.y, g.y0, g.inv_h, g.gy), cy1 = dev::cell_of(b.w, g.y0, g.inv_h, g.gy);
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const int c = cy * g.gx + cx;
            const int slot = atomicAdd(&cell_cnt[c], 1);
            if (FILL) items[slot] = (int32_t)j;
        }
}

// ascending ids within each cell -> deterministic candidate order, hence sorted (l, r) output
__global__ void cell_sort_kernel(const int32_t* __restrict__ cell_off, int64_t n_cells, int32_t* __restrict__ items) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const int b = cell_off[c], e = cell_off[c + 1];
    for (int i = b + 1; i < e; ++i) {
        const int32_t key = items[i];
        int k = i - 1;
        while (k >= b && items[k] > key) {
            items[k + 1] = items[k];
            --k;
        }
        items[k + 1] = key;
    }
}

// ================================= point-in-polygon join =========================================
#ifndef GPK_PIP_PPT
#define GPK_PIP_PPT 2  // points per lane of pip_tile_kernel on ordinary indexes; list-heavy ones (gpk_index::pip_list_heavy) run the
                       // PPT = 1 instance: the flattened passes carry the memory-level parallelism there, and a 256-point tile keeps 84
                       // registers and 19 KB of LDS per work-group (C5 1.41 -> 1.31 ms); a tessellation, which streams, loses 20 % with it
#endif
// GPK_ABLATE (tuning builds only, tools/pmc_ablate.sh, tools/ablate_time.py): 1 = no exact phase, 2 = every cell empty, 3 = level-1 interiors only,
// 5 = level-2 labels without queue pushes, 6 = list cells skipped, 7 = list entries end after the box test, 8 = no hole rings.  0 in the shipped library.
#ifndef GPK_ABLATE
#define GPK_ABLATE 0
#endif
#ifndef GPK_PIP_GS
#define GPK_PIP_GS 8
#endif
#ifndef GPK_PIP_BLOCK
#define GPK_PIP_BLOCK 256
#endif
constexpr int PIP_BLOCK = GPK_PIP_BLOCK;       // threads per work-group of pip_tile
#ifndef GPK_WR_BLOCK
#define GPK_WR_BLOCK 256
#endif
constexpr int WR_BLOCK = GPK_WR_BLOCK;                  // threads per work-group of pip_write
constexpr int PIP_PPT = GPK_PIP_PPT;           // points per thread, strided by PIP_BLOCK (coalesced 16-byte loads)
constexpr int PIP_TILE = PIP_BLOCK * PIP_PPT;  // points per work-group
constexpr int PIP_GS = GPK_PIP_GS;             // lanes cooperating on one queued (point, part) pair
constexpr int PIP_OVF = 256;                   // per-tile overflow list for points with more than PIP_KHIT hits
#ifndef GPK_PIP_QCAP
#define GPK_PIP_QCAP (GPK_PIP_BLOCK * GPK_PIP_PPT)
#endif
constexpr int PIP_QCAP = GPK_PIP_QCAP;             // LDS queue capacity (overflow is resolved inline, still exact)
constexpr int PIP_SUPER_SHIFT = 6;             // 64 tiles per super-tile (two-level prefix of the tile totals)
#ifndef GPK_WR_WPT
#define GPK_WR_WPT 8
#endif
constexpr int PIP_WPT = GPK_WR_WPT;            // writer: consecutive points per thread (a multiple of 4: 16-byte code loads)
constexpr int PIP_WTILE = WR_BLOCK * PIP_WPT;  // writer: points per work-group (a multiple of PIP_TILE)
constexpr int WR_CAP = PIP_WTILE;              // writer: pairs of one tile compacted in LDS (16 KB) before the coalesced copy-out
static_assert(PIP_WTILE % PIP_TILE == 0, "a writer tile is a whole number of pip_tile tiles");
constexpr int PIP_KHIT = 2;  // part hits remembered per point in LDS; rows with more take the generic walk

// Visits every right-side row whose closed bbox contains the point, in ascending id order.
template <typename F>
__device__ __forceinline__ void for_each_candidate(const IndexView& ix, const GridParams& g, double px,
                                                   double py, F&& f) {
    if (!(px == px) || !(py == py)) return;  // empty point
    const int cx = dev::cell_of(px, g.x0, g.inv_w, g.gx), cy = dev::cell_of(py, g.y0, g.inv_h, g.gy);
    const int c = cy * g.gx + cx;
    const int b = ix.cell_off[c], e = ix.cell_off[c + 1];
    for (int k = b; k < e; ++k) {
        const int j = ix.items[k];
        const double4 bb = ix.bbox[j];
        if (px >= bb.x && px <= bb.z && py >= bb.y && py <= bb.w) f(j);
    }
}

// Generic (always exact, never fast) evaluation of one point: directory candidates -> full ring walks.
// It is the reference every accelerated route falls back to: rows with several hits, queue overflow,
// right sides without a raster.
__device__ inline void generic_point(const DevGeo& polys, const IndexView& ix, double px, double py, uint32_t& cnt,
                                     uint32_t& first) {
    cnt = 0;
    first = CODE_NONE;
    const GridParams g = *ix.grid;
    for_each_candidate(ix, g, px, py, [&](int j) {
        if (dev::valid_row(polys.validity, j) && dev::polygonal_hits_point<false>(polys, j, px, py)) {
            if (cnt == 0) first = (uint32_t)j;
            ++cnt;
        }
    });
}

struct QEntry {  // 32 bytes: one queued (point, part) pair with its exterior slab already located
    double px, py;
    uint32_t part, e0, cnt;
    uint32_t li_flags;  // local point index | (part has holes) << 31
};

// pip_tile: one work-group classifies PIP_TILE points (the GENERAL tile kernel: entry lists, two-part records, refined rings, holes,
// rows in several geometries; disjoint right sides run pip_tile_route / pip_tile_chain, further down).
//   phase 1 (lane = point): ONE 4-byte gather from the raster answers most points outright (no polygon
//            here / strictly inside part p).  A point whose cell one part crosses reads that cell's level-2 record (or PartInfo + slab
//            offsets) and is pushed to an LDS queue with a wave-aggregated slot grab (ballot + one LDS atomic per wave).  Loads are
//            issued in stages (all points, then all cells, then all records, ...) to keep PPT requests per lane in flight.
//   entry lists (round 3): the lanes whose cell is a LIST register (point, list) jobs; the jobs' entries are then one flat work
//            list — every lane takes a run of consecutive (point, entry) items, GPK_FLAT_B at a time with their gathers in flight
//            together (entry word + point, then the entry's record or its part's box), and pushes what survives.
//   phase 2 (round 3: lane = slab EDGE): the queued pairs' slabs, flattened the same way; an edge's winding / on-boundary contribution
//            goes to its pair's LDS accumulator (one atomicAdd of wn << 16 | on); a pair inside the exterior of a part with holes
//            queues its hole rings, which go through the same pass.  (Round 2 walked a pair with PIP_GS = 8 lanes, eight passes per
//            tile back to back, and lists / holes one lane each: DESIGN.md section 4.4 "Round 3, C5".)
//   finalize: rows with exactly one part hit map part -> geometry; rows in several geometries (overlapping polygons / multipolygon
//            parts) get a sorted segment of the multi-hit pool (its space reserved once per wave); only a tile that overflows its
//            lists falls back to the generic walk.
#ifndef GPK_PIP_MINWAVES
#define GPK_PIP_MINWAVES 1
#endif
#ifndef GPK_PIP_NT
#define GPK_PIP_NT 2  // bit 0: non-temporal point loads (measured slower), bit 1: non-temporal result stores
#endif
#ifndef GPK_FLAT_B
#define GPK_FLAT_B 2  // entries / edges a lane of the flattened passes keeps in flight (4: 123 registers instead of 92, no faster)
#endif
#define PIP_SYNC() __syncthreads()
// GPK_TILE_TRACE (diagnosis builds only; tools/c5_stage_clocks.py): thread 0 of every work-group adds the wall-clock ticks (100 MHz) it
// spent in each stage of pip_tile_kernel — barrier waits included, so a stage's total is the work-groups' critical path through it — to
// the statistics buffer's words 8 + stage.
#ifdef GPK_TILE_TRACE
#define PIP_STAGE_CLK(idx)                                                   \
    do {                                                                     \
        if (stats && tid == 0) {                                             \
            const unsigned long long _now = wall_clock64();                  \
            atomicAdd(&stats[8 + (idx)], _now - t_stage);                    \
            t_stage = _now;                                                  \
        }                                                                    \
    } while (0)
#else
#define PIP_STAGE_CLK(idx) \
    do {                   \
    } while (0)
#endif
// SUB2: the index holds two-part level-2 records (PipView::sub2); the variant without them stays as lean as it was
template <bool RASTER, bool SUB2, int PPT = GPK_PIP_PPT>
__global__ __launch_bounds__(PIP_BLOCK, GPK_PIP_MINWAVES) void pip_tile_kernel(DevGeo pts, DevGeo polys, IndexView ix, PipView pv,
                                                              uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ code,
                                                              unsigned long long* __restrict__ block_tot,
                                                              unsigned long long* __restrict__ super_tot,
                                                              uint32_t* __restrict__ multi_pool, uint32_t multi_cap,
                                                              uint32_t* __restrict__ multi_top, unsigned long long* __restrict__ stats) {
    constexpr int PIP_PPT = PPT, PIP_TILE = PIP_BLOCK * PPT, PIP_QCAP = PIP_BLOCK * PPT;  // (this instance's: they shadow the defaults)
    __shared__ QEntry q[RASTER ? PIP_QCAP : 1];
    __shared__ uint32_t s_pre[RASTER ? PIP_QCAP + 1 : 1];  // phase 2: first flattened edge of every queued pair (+ the total)
    constexpr int ACC_WORDS = PIP_QCAP > 2 * PIP_BLOCK ? PIP_QCAP : 2 * PIP_BLOCK;  // (the round's job list — two words per lane — shares it)
    __shared__ int s_acc[RASTER ? ACC_WORDS : 1];          // phase 2: the pair's winding sum << 16 | on-boundary count
    __shared__ uint32_t s_scan[PIP_BLOCK / 64 + 1];
    __shared__ uint32_t lj_n[PIP_PPT];  // (point, list) jobs registered in round k
    constexpr int PIP_HQCAP = 128;      // hole rings queued per drain of the pair queue (beyond it: the owning lane walks the hole)
    __shared__ uint4 s_hq[RASTER ? PIP_HQCAP : 1];  // (owning pair, first slab entry, entries, -)
    __shared__ int s_hacc[RASTER ? PIP_HQCAP : 1];
    __shared__ uint32_t hq_n;
    // a job = (list offset, point slot); the job list lives in s_acc's storage (phase 2 starts after the list pass)
    static_assert(sizeof(uint2) * PIP_BLOCK <= sizeof(int) * ACC_WORDS, "the job list fits s_acc");
    uint2* s_lj = reinterpret_cast<uint2*>(s_acc);
    __shared__ uint32_t s_cnt[PIP_TILE], s_hit[PIP_TILE * PIP_KHIT];
    __shared__ uint32_t q_n, ovf_n;
    __shared__ uint2 s_ovf[RASTER ? PIP_OVF : 1];  // (point slot, part) hits beyond a point's PIP_KHIT inline slots
    __shared__ unsigned long long s_tot;  // hits of the tile
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * PIP_TILE;
    const int64_t n = pts.n_geoms;
    const uint32_t rem = (uint32_t)(n - base < (int64_t)PIP_TILE ? n - base : (int64_t)PIP_TILE);  // points in this tile
    const double2* __restrict__ tile_xy = pts.xy + base;
    if (tid == 0) s_tot = 0;
#ifdef GPK_TILE_TRACE
    unsigned long long t_stage = wall_clock64();
#endif

    if (RASTER) {
#pragma unroll
        for (int k = 0; k < PIP_PPT; ++k) s_cnt[k * PIP_BLOCK + tid] = 0;
        if (tid == 0) q_n = 0, ovf_n = 0, hq_n = 0;
        if (tid < PIP_PPT) lj_n[tid] = 0;
        __syncthreads();

        const int lane64 = tid & 63;
        double2 p[PIP_PPT];
        uint32_t word[PIP_PPT];
        uint32_t fy[PIP_PPT];
        bool want[PIP_PPT];
        PartInfo pi[PIP_PPT];
        int e0[PIP_PPT], e1[PIP_PPT];
        // stage A: points (coalesced 16-byte loads)
#pragma unroll
        for (int k = 0; k < PIP_PPT; ++k) {
            const uint32_t t = (uint32_t)(k * PIP_BLOCK + tid);  // scalar tile base + 32-bit lane offset
            const bool ok = t < rem && dev::valid_row(pts.validity, base + t);
            p[k] = ok ? ((GPK_PIP_NT & 1) ? dev::load_stream(tile_xy + t) : tile_xy[t]) : make_double2(NAN, NAN);
        }
        // stage B: raster words (one 4-byte gather per point).  Columns/rows are computed at PIP_SUB x the
        // raster resolution (an exact power-of-two rescale of the same monotone function): `/ PIP_SUB` is the
        // level-1 cell, `% PIP_SUB` the level-2 sub-cell.
        // Rows are computed ONCE at the finest slab resolution (FINE per raster row); `>> FY_SUB` is the level-2 row
        // (the value the S-scaled function returns: power-of-two rescales commute with floor and the clamps),
        // `>> PIP_FINE_LOG2` the base slab row, `>> (PIP_FINE_LOG2 - shift)` the slab row of a refined ring.
        constexpr int S = PIP_SUB, FINE = PIP_SLAB_MUL << PIP_FINE_LOG2, FY_SUB = PIP_FINE_LOG2 - 2;
        static_assert(FINE == S << FY_SUB, "fine rows per level-2 row");
        uint32_t sx[PIP_PPT], fyf[PIP_PPT];
        // slab of the exterior ring of `part` for a point in finest row `fine`: false = p.y outside the ring's y-range
        auto part_slab = [&](const PartInfo& pq, uint32_t fine, int& a0, int& a1) -> bool {
            const int j = (int)(fine >> (PIP_FINE_LOG2 - slab_shift_of(pq.row0))) - slab_row0_of(pq.row0);
            if (j < 0 || j >= pq.nrows) return false;
            a0 = pv.slab_off[pq.slab_base + j];
            a1 = pv.slab_off[pq.slab_base + j + 1];
            return true;
        };
#pragma unroll
        for (int k = 0; k < PIP_PPT; ++k) {
            sx[k] = (uint32_t)dev::cell_of(p[k].x, pv.rx0, pv.inv_fw * S, pv.R * S);
            fyf[k] = (uint32_t)dev::cell_of(p[k].y, pv.ry0, pv.inv_fh * FINE, pv.R * FINE);
            fy[k] = fyf[k] >> FY_SUB;
            word[k] = (p[k].x == p[k].x && p[k].y == p[k].y) ? pv.cell[(fy[k] / S) * (uint32_t)pv.R + (sx[k] / S)] : 0u;
            if (GPK_ABLATE == 2) word[k] = 0u;  // tuning builds only
            if (GPK_ABLATE == 3) word[k] = (word[k] >> 30) == CELL_TAG_SINGLE && !(word[k] & 1u) ? word[k] : 0u;
        }
        // stage C: decided cells; level-2 record gather (32 B) or PartInfo gather for inline boundary entries
        SubCell sc[PIP_PPT];
        uint4 scb[PIP_PPT];  // second part of a two-part record (SubCell2): b_part_flags, b_e0, b_e1, b_e2
        bool has_sub[PIP_PPT], has_sub2[PIP_PPT], want_b[PIP_PPT];
#pragma unroll
        for (int k = 0; k < PIP_PPT; ++k) {
            const uint32_t tag = word[k] >> 30, payload = word[k] & 0x3FFFFFFFu;
            want[k] = false;
            want_b[k] = false;
            has_sub[k] = tag == CELL_TAG_SUB;
            has_sub2[k] = SUB2 && has_sub[k] && (payload & SUB2_BIT);
            scb[k] = make_uint4(0u, 0u, 0u, 0u);
            if (has_sub2[k]) {
                const SubCell2* __restrict__ r2 = pv.sub2 + (payload & (SUB2_BIT - 1u));
                sc[k] = r2->a;
                scb[k] = *reinterpret_cast<const uint4*>(&r2->b_part_flags);
            } else if (has_sub[k]) {
                sc[k] = pv.sub[payload];
            } else if (tag == CELL_TAG_SINGLE) {
                if (payload & 1u) {
                    want[k] = true;
                    pi[k] = pv.part_info[payload >> 1];
                } else {
                    const int li = k * PIP_BLOCK + tid;
                    s_cnt[li] = 1;  // only this lane touches s_cnt[li] before the barrier
                    s_hit[li * PIP_KHIT] = payload >> 1;
                }
            }
        }
        // stage D: level-2 label, or slab offsets for inline boundary entries
        uint32_t qpart[PIP_PPT], qflag[PIP_PPT];
#pragma unroll
        for (int k = 0; k < PIP_PPT; ++k) {
            qpart[k] = (word[k] & 0x3FFFFFFFu) >> 1;
            qflag[k] = 0;
            e0[k] = e1[k] = 0;
            if (has_sub[k]) {
                const int idx = (fy[k] % S) * S + (sx[k] % S);
                const int wsel = idx >> 4;  // select by compares: a runtime-indexed register array would go to scratch
                const uint32_t lw = wsel == 0 ? sc[k].labels[0] : (wsel == 1 ? sc[k].labels[1] : (wsel == 2 ? sc[k].labels[2] : sc[k].labels[3]));
                const uint32_t lab = (lw >> (2 * (idx & 15))) & 3u;
                qpart[k] = sc[k].part_flags & 0x3FFFFFFFu;
                // slab of slot A for this point: inline in the record, or through PartInfo when the ring has refined rows
                auto slot_a = [&]() {
                    want[k] = true;
                    qflag[k] = sc[k].part_flags & 0x80000000u;
                    if (sc[k].part_flags & SUB_INDIRECT) {
                        want[k] = part_slab(pv.part_info[qpart[k]], fyf[k], e0[k], e1[k]);
                    } else {
                        const bool upper = ((fyf[k] >> PIP_FINE_LOG2) & 1u) != 0;
                        e0[k] = (int)(upper ? sc[k].e1 : sc[k].e0);
                        e1[k] = (int)(upper ? sc[k].e2 : sc[k].e1);
                    }
                };
                if (!SUB2 || !has_sub2[k]) {
                    if (lab == 1u) {
                        const int li = k * PIP_BLOCK + tid;
                        s_cnt[li] = 1;
                        s_hit[li * PIP_KHIT] = qpart[k];
                    } else if (lab == 2u && GPK_ABLATE != 5) {
                        slot_a();
                    }
                } else {  // two parts share the cell (gpk_index.h: SubCell2)
                    const int li = k * PIP_BLOCK + tid;
                    want_b[k] = lab == 3u;
                    if (lab == 1u || lab == 2u) {
                        s_cnt[li] = 1;
                        s_hit[li * PIP_KHIT] = lab == 1u ? qpart[k] : (scb[k].x & 0x3FFFFFFFu);
                    }
                    if (lab == 3u) slot_a();
                }
            } else if (want[k]) {
                want[k] = part_slab(pi[k], fyf[k], e0[k], e1[k]);  // false: p.y outside the exterior's y-range: Outside
                qflag[k] = pi[k].n_rings > 1 ? 0x80000000u : 0u;
            }
        }
        // stage E, one round per k: queue pushes (wave-aggregated), then — when the queue is filling up, and after
        // the last round — phase 2 drains it.  Boundary-dominated right sides (small overlapping polygons) queue more
        // than one pair per point; draining between rounds keeps them on the cooperative path.
        auto record_any = [&](uint32_t li, uint32_t part) {  // any lane, any point of the tile (the flattened passes)
            const uint32_t sl = atomicAdd(&s_cnt[li], 1u);
            if (sl < (uint32_t)PIP_KHIT) {
                s_hit[li * PIP_KHIT + sl] = part;
            } else {
                const uint32_t o = atomicAdd(&ovf_n, 1u);
                if (o < (uint32_t)PIP_OVF) s_ovf[o] = make_uint2(li, part);
            }
        };
        auto record = [&](int li, uint32_t part) {  // phase 1: only this lane touches s_cnt[li] between barriers
            const uint32_t sl = s_cnt[li]++;
            if (sl < (uint32_t)PIP_KHIT) {
                s_hit[li * PIP_KHIT + sl] = part;
            } else {
                const uint32_t o = atomicAdd(&ovf_n, 1u);
                if (o < (uint32_t)PIP_OVF) s_ovf[o] = make_uint2((uint32_t)li, part);
            }
        };
        PIP_STAGE_CLK(0);
#pragma unroll
        for (int k = 0; k < PIP_PPT; ++k) {
            const int li = k * PIP_BLOCK + tid;
            const bool push = want[k] && e1[k] > e0[k];
            const unsigned long long mask = __ballot(push);
            if (mask) {
                uint32_t wbase = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if (lane64 == leader) wbase = atomicAdd(&q_n, (uint32_t)__popcll(mask));
                wbase = __shfl(wbase, leader, 64);
                if (push) {
                    const uint32_t part = qpart[k];
                    const uint32_t slot = wbase + (uint32_t)__popcll(mask & ((1ull << lane64) - 1ull));
                    if (slot < (uint32_t)PIP_QCAP) {
                        q[slot] = QEntry{p[k].x, p[k].y, part, (uint32_t)e0[k], (uint32_t)(e1[k] - e0[k]), (uint32_t)li | qflag[k]};
                    } else if (pip::part_pos_single(pv, polys, (int)part, p[k].x, p[k].y) == dev::POS_INSIDE) {
                        record(li, part);
                    }
                }
            }
            if (SUB2 && want_b[k]) {  // second part of a two-part cell: rare enough for one LDS atomic per lane
                const bool upper = ((fyf[k] >> PIP_FINE_LOG2) & 1u) != 0;
                uint32_t b0 = upper ? scb[k].z : scb[k].y, b1 = upper ? scb[k].w : scb[k].z;
                const uint32_t bpart = scb[k].x & 0x3FFFFFFFu;
                if (scb[k].x & SUB_INDIRECT) {
                    int a0 = 0, a1 = 0;
                    if (!part_slab(pv.part_info[bpart], fyf[k], a0, a1)) a0 = a1 = 0;
                    b0 = (uint32_t)a0;
                    b1 = (uint32_t)a1;
                }
                if (b1 > b0) {
                    const uint32_t slot = atomicAdd(&q_n, 1u);
                    if (slot < (uint32_t)PIP_QCAP)
                        q[slot] = QEntry{p[k].x, p[k].y, bpart, b0, b1 - b0, (uint32_t)li | (scb[k].x & 0x80000000u)};
                    else if (pip::part_pos_single(pv, polys, (int)bpart, p[k].x, p[k].y) == dev::POS_INSIDE)
                        record(li, bpart);
                }
            }
            // cells where several parts meet (three or more, or two that both cover it): the (point, entry) pairs of the round, FLATTENED.
            // A lane used to walk its own cell's list — five entries on average for C5, each a chain of dependent gathers (entry -> box
            // -> PartInfo -> slab offsets), the whole work-group waiting at the barrier for the lane with the longest list: 72 of the
            // 151 us a work-group lived.  Now the lanes register (point, list) jobs, and every lane takes a run of consecutive entries
            // of the flattened job list, four at a time with their gathers in flight together.
            {
                const bool has_list = (word[k] >> 30) == CELL_TAG_LIST && GPK_ABLATE != 6;
                const unsigned long long lm = __ballot(has_list);
                if (lm) {  // (wave-uniform)
                    const uint32_t off = word[k] & 0x3FFFFFFFu;
                    const int leader = __ffsll((long long)lm) - 1;
                    uint32_t wb = 0;
                    if (lane64 == leader) wb = atomicAdd(&lj_n[k], (uint32_t)__popcll(lm));
                    wb = __shfl(wb, leader, 64);
                    if (has_list) s_lj[wb + (uint32_t)__popcll(lm & ((1ull << lane64) - 1ull))] = make_uint2(off, (uint32_t)li);
                }
            }
            PIP_SYNC();
            const uint32_t nl = lj_n[k];  // (uniform; at most one job per lane and round)
            if (nl) {
                const uint32_t my_m = (uint32_t)tid < nl ? pv.list[s_lj[tid].x] : 0u;  // entries of the job's list
                uint32_t n_items;
                const uint32_t pre = dev::block_exclusive_scan<uint32_t, PIP_BLOCK>(my_m, s_scan, &n_items);
                if ((uint32_t)tid < nl) s_pre[tid] = pre;
                if (tid == 0) s_pre[nl] = n_items;
                __syncthreads();
                const uint32_t chunk = (n_items + PIP_BLOCK - 1) / PIP_BLOCK;
                const uint32_t j0 = (uint32_t)tid * chunk, j1 = j0 + chunk < n_items ? j0 + chunk : n_items;
                // (one instance per index kind: the records of the one and the boxes of the other never share registers)
                auto list_items = [&](auto LREC_T) {
                    constexpr bool LREC = decltype(LREC_T)::value;
                    if (j0 >= j1) return;
                    uint32_t jb = 0;  // largest job with s_pre[jb] <= j0
                    {
                        uint32_t lo = 0, hi = nl;
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_pre[mid] <= j0)
                                lo = mid;
                            else
                                hi = mid;
                        }
                        jb = lo;
                    }
                    constexpr int B = GPK_FLAT_B;
                    for (uint32_t j = j0; j < j1; j += B) {
                        bool ok[B];
                        uint32_t jli[B], ew[B];
                        double2 jp[B];
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            const uint32_t jj = j + u;
                            ok[u] = jj < j1;
                            if (ok[u])
                                while (jj >= s_pre[jb + 1]) ++jb;  // (a job with no entries is stepped over)
                            const uint2 job = s_lj[jb];
                            jli[u] = job.y;
                            ew[u] = pv.list[job.x + 1u + (ok[u] ? jj - s_pre[jb] : 0u)];  // (a lane past its run re-reads a valid entry and drops it)
                            jp[u] = tile_xy[jli[u]];
                        }
                        // second level, all four in flight: the entry's record (indexes with per-entry records) or its part's box
                        SubCell rc[LREC ? B : 1];
                        float4 bb[LREC ? 1 : B];
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            if constexpr (LREC)
                                rc[u] = pv.lrec[(ew[u] & 1u) ? ew[u] >> 1 : 0u];
                            else if (pv.part_box)
                                bb[u] = pv.part_box[ew[u] >> 1];
                        }
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            if (!ok[u]) continue;
                            const uint32_t e = ew[u], jl = jli[u];
                            const double jx = jp[u].x, jy = jp[u].y;
                            uint32_t part = e >> 1;
                            if (e & 1u) {
                                int a0 = 0, a1 = 0;
                                uint32_t holes = 0;
                                const uint32_t jfyf = (uint32_t)dev::cell_of(jy, pv.ry0, pv.inv_fh * FINE, pv.R * FINE);
                                if constexpr (LREC) {  // the entry names a level-2 record: most points finish on its label
                                    const uint32_t jsx = (uint32_t)dev::cell_of(jx, pv.rx0, pv.inv_fw * S, pv.R * S), jfy = jfyf >> FY_SUB;
                                    part = rc[u].part_flags & 0x3FFFFFFFu;
                                    const int idx = (jfy % S) * S + (jsx % S);
                                    const int wsel = idx >> 4;
                                    const uint32_t lw = wsel == 0 ? rc[u].labels[0] : (wsel == 1 ? rc[u].labels[1] : (wsel == 2 ? rc[u].labels[2] : rc[u].labels[3]));
                                    const uint32_t lab = (lw >> (2 * (idx & 15))) & 3u;
                                    if (lab == 0u) continue;
                                    if (lab == 1u) {
                                        record_any(jl, part);
                                        continue;
                                    }
                                    holes = rc[u].part_flags & 0x80000000u;
                                    if (rc[u].part_flags & SUB_INDIRECT) {
                                        if (!part_slab(pv.part_info[part], jfyf, a0, a1)) continue;
                                    } else {
                                        const bool upper = ((jfyf >> PIP_FINE_LOG2) & 1u) != 0;
                                        a0 = (int)(upper ? rc[u].e1 : rc[u].e0);
                                        a1 = (int)(upper ? rc[u].e2 : rc[u].e1);
                                    }
                                } else {
                                    // (closed box of the exterior, rounded outward: outside it = outside the part)
                                    if (pv.part_box && !(jx >= (double)bb[u].x && jx <= (double)bb[u].z && jy >= (double)bb[u].y && jy <= (double)bb[u].w)) continue;
                                    if (GPK_ABLATE == 7) continue;  // tuning builds only
                                    const PartInfo pq = pv.part_info[part];
                                    if (!part_slab(pq, jfyf, a0, a1)) continue;
                                    holes = pq.n_rings > 1 ? 0x80000000u : 0u;
                                }
                                if (a1 <= a0) continue;
                                const uint32_t slot = atomicAdd(&q_n, 1u);
                                if (slot < (uint32_t)PIP_QCAP) {
                                    q[slot] = QEntry{jx, jy, part, (uint32_t)a0, (uint32_t)(a1 - a0), jl | holes};
                                    continue;
                                }
                                if (pip::part_pos_single(pv, polys, (int)part, jx, jy) != dev::POS_INSIDE) continue;
                            }
                            record_any(jl, part);
                        }
                    }
                };
                if (pv.lrec)
                    list_items(std::true_type{});
                else
                    list_items(std::false_type{});
            }
            PIP_SYNC();
            PIP_STAGE_CLK(1);
            const uint32_t queued = q_n;  // uniform: read after the barrier
            if (k + 1 < PIP_PPT && queued <= (uint32_t)PIP_QCAP / 2) continue;
            // phase 2: the queued pairs' slab edges, FLATTENED — lane = edge, not 8 lanes = pair.  A pair has 7 edges on average and its
            // walk is three dependent gathers (queue entry -> slab entry -> coordinates): with 32 pairs per pass a tile needed eight
            // such passes back to back, and that chain — not the arithmetic — was 1.1 ms of the 1.8 ms C5 join.  Here every lane takes a
            // run of consecutive edges of the flattened list, four at a time with all their gathers in flight together, and adds the
            // edge's winding / on-boundary contribution to its pair's LDS accumulator; one more pass decides the pairs.
            const uint32_t nq = GPK_ABLATE == 1 ? 0u : (queued < (uint32_t)PIP_QCAP ? queued : (uint32_t)PIP_QCAP);
            if (stats && tid == 0 && queued) atomicAdd(&stats[0], (unsigned long long)queued);  // measurement runs only
            if (nq) {  // (uniform)
                uint32_t mine = 0, c_of[PIP_PPT];
#pragma unroll
                for (int u = 0; u < PIP_PPT; ++u) {
                    const uint32_t e = (uint32_t)tid * PIP_PPT + u;
                    c_of[u] = e < nq ? q[e].cnt : 0u;
                    mine += c_of[u];
                    if (e < nq) s_acc[e] = 0;
                }
                uint32_t n_flat;
                uint32_t run = dev::block_exclusive_scan<uint32_t, PIP_BLOCK>(mine, s_scan, &n_flat);
#pragma unroll
                for (int u = 0; u < PIP_PPT; ++u) {
                    const uint32_t e = (uint32_t)tid * PIP_PPT + u;
                    if (e <= nq) s_pre[e] = run;  // (entry nq: the total — written by the thread that owns slot nq, or below)
                    run += c_of[u];
                }
                if (tid == 0) s_pre[nq] = n_flat;  // (same value if already written: entries past nq count zero)
                __syncthreads();
                PIP_STAGE_CLK(2);
                if (stats && tid == 0) atomicAdd(&stats[1], (unsigned long long)n_flat);
                // the flattened pass over `n_ent` entries whose first edges are s_pre[0 .. n_ent]: ent(i) -> (first slab entry, point)
                auto flat_edges = [&](uint32_t n_ent, uint32_t n_edges, auto&& ent, int* acc) {
                    const uint32_t chunk = (n_edges + PIP_BLOCK - 1) / PIP_BLOCK;
                    const uint32_t j0 = (uint32_t)tid * chunk, j1 = j0 + chunk < n_edges ? j0 + chunk : n_edges;
                    if (j0 >= j1) return;
                    uint32_t e = 0;  // largest entry with s_pre[e] <= j0
                    {
                        uint32_t lo = 0, hi = n_ent;
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_pre[mid] <= j0)
                                lo = mid;
                            else
                                hi = mid;
                        }
                        e = lo;
                    }
                    constexpr int B = GPK_FLAT_B;
                    for (uint32_t j = j0; j < j1; j += B) {
                        uint32_t ee[B], at[B];
                        double2 pt[B];
                        bool ok[B];
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            const uint32_t jj = j + u;
                            ok[u] = jj < j1;
                            if (ok[u])
                                while (jj >= s_pre[e + 1]) ++e;  // (entries have at least one edge)
                            ee[u] = e;
                            uint32_t first;
                            ent(e, first, pt[u]);
                            at[u] = first + (ok[u] ? jj - s_pre[e] : 0u);  // (a lane past its run re-reads a valid entry and drops the result)
                        }
                        double4 ed[B];
#pragma unroll
                        for (int u = 0; u < B; ++u) ed[u] = pip::slab_edge(pv, (int)at[u]);
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            if (!ok[u]) continue;
                            int wn = 0;
                            const int on = (int)dev::ring_edge(ed[u].x, ed[u].y, ed[u].z, ed[u].w, pt[u].x, pt[u].y, wn);
                            if (wn != 0 || on) atomicAdd(&acc[ee[u]], wn * 65536 + on);
                        }
                    }
                };
                flat_edges(nq, n_flat, [&](uint32_t e, uint32_t& first, double2& pt) {
                    first = q[e].e0;
                    pt = make_double2(q[e].px, q[e].py);
                }, s_acc);
                __syncthreads();
                PIP_STAGE_CLK(3);
                // decide.  A pair inside the exterior of a part WITH holes queues one entry per hole ring whose rows reach the point
                // (three gathers per hole) and waits: the holes' slab edges go through the same flattened pass — a lane walking a hole
                // edge by edge (a dozen dependent gathers) kept its whole work-group at the barrier.
                auto hit = [&](const QEntry& en) {
                    const uint32_t li2 = en.li_flags & 0x7FFFFFFFu;
                    const uint32_t sl = atomicAdd(&s_cnt[li2], 1u);
                    if (sl < (uint32_t)PIP_KHIT) {
                        s_hit[li2 * PIP_KHIT + sl] = en.part;
                    } else {
                        const uint32_t o = atomicAdd(&ovf_n, 1u);
                        if (o < (uint32_t)PIP_OVF) s_ovf[o] = make_uint2(li2, en.part);
                    }
                };
                constexpr int PENDING = 0x7FFFFFFF;
                for (uint32_t e = (uint32_t)tid; e < nq; e += PIP_BLOCK) {
                    const int packed = s_acc[e];  // winding sum in the high half, on-boundary count in the low half
                    s_acc[e] = 0;
                    if ((packed & 0xFFFF) != 0 || (packed >> 16) == 0) continue;  // on the exterior ring, or outside it
                    const QEntry en = q[e];
                    if ((en.li_flags >> 31) && GPK_ABLATE != 8) {  // the part has holes: inside one (or on it) = not inside the part
                        int r0, r1;
                        dev::part_rings(polys, (int)en.part, r0, r1);
                        const int row = pip::row_of(pv, en.py);
                        bool out_of_part = false, pending = false;
                        for (int r = r0 + 1; r < r1 && !out_of_part; ++r) {
                            int a0, a1;
                            if (!pip::slab_range(pv, r, row, a0, a1) || a1 <= a0) continue;  // no edge of this hole reaches the point's row: outside it
                            const uint32_t slot = atomicAdd(&hq_n, 1u);
                            if (slot < (uint32_t)PIP_HQCAP) {
                                s_hq[slot] = make_uint4(e, (uint32_t)a0, (uint32_t)(a1 - a0), 0u);
                                pending = true;
                            } else {
                                out_of_part = pip::ring_pos_single(pv, r, en.px, en.py, row) != dev::POS_OUTSIDE;
                            }
                        }
                        if (out_of_part) continue;
                        if (pending) {
                            s_acc[e] = PENDING;
                            continue;
                        }
                    }
                    hit(en);
                }
                __syncthreads();
                const uint32_t nh = hq_n < (uint32_t)PIP_HQCAP ? hq_n : (uint32_t)PIP_HQCAP;  // (uniform)
                if (nh) {
                    static_assert(PIP_HQCAP <= PIP_BLOCK, "one hole entry per lane in the scan");
                    const uint32_t hc = (uint32_t)tid < nh ? s_hq[tid].z : 0u;
                    uint32_t n_hedges;
                    const uint32_t hpre = dev::block_exclusive_scan<uint32_t, PIP_BLOCK>(hc, s_scan, &n_hedges);
                    if ((uint32_t)tid < nh) {
                        s_pre[tid] = hpre;
                        s_hacc[tid] = 0;
                    }
                    if (tid == 0) s_pre[nh] = n_hedges;
                    __syncthreads();
                    if (stats && tid == 0) atomicAdd(&stats[1], (unsigned long long)n_hedges);
                    flat_edges(nh, n_hedges, [&](uint32_t h, uint32_t& first, double2& pt) {
                        const uint4 he = s_hq[h];
                        first = he.y;
                        pt = make_double2(q[he.x].px, q[he.x].py);
                    }, s_hacc);
                    __syncthreads();
                    if ((uint32_t)tid < nh && s_hacc[tid] != 0) s_acc[s_hq[tid].x] = 0;  // inside that hole, or on it (every writer stores 0)
                    __syncthreads();
                    for (uint32_t e = (uint32_t)tid; e < nq; e += PIP_BLOCK)
                        if (s_acc[e] == PENDING) hit(q[e]);
                    if (tid == 0) hq_n = 0;
                }
            }
            PIP_SYNC();
            PIP_STAGE_CLK(4);
            if (tid == 0) q_n = 0;
            PIP_SYNC();
        }
    }

    if (!RASTER) __syncthreads();  // s_tot = 0 visible
    uint32_t* __restrict__ tile_counts = counts ? counts + base : nullptr;
    uint32_t* __restrict__ tile_code = code + base;
    unsigned long long wave_hits = 0;  // rows with exactly one hit, counted by ballot (uniform per wave)
    // space in the multi-hit pool is reserved once per WAVE (a scan over the lanes' needs + one atomic): one atomic per multi-hit row on
    // the single pool cursor was 1.1 ms of the 1.8 ms C5 join (300k rows in two or more overlapping multipolygons per 6.25M points)
    auto pool_reserve = [&](uint32_t need) -> uint32_t {
        const unsigned long long wants = __ballot(need != 0u);
        if (!wants) return 0u;  // (uniform)
        const int lane = tid & 63;
        uint32_t incl = need;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t w = __shfl_up(incl, o, 64);
            if (lane >= o) incl += w;
        }
        const int last = 63 - __builtin_clzll(__ballot(true));  // highest active lane: holds the wave's total
        uint32_t wbase = 0;
        if (lane == last) wbase = atomicAdd(multi_top, incl);
        wbase = __shfl(wbase, last, 64);
        return wbase + incl - need;
    };
#pragma unroll
    for (int k = 0; k < PIP_PPT; ++k) {
        const int li = k * PIP_BLOCK + tid;
        const int64_t i = base + li;
        const bool in_tile = (uint32_t)li < rem;
        uint32_t cnt = 0, first = CODE_NONE, pool_code = CODE_MULTI;
        bool generic = !RASTER;
        uint32_t g0 = CODE_NONE, g1 = CODE_NONE, need = 0, novf = 0;
        int mode = 0;  // 1: two geometries (a 3-word pool segment), 2: more part hits than inline slots (collected from the overflow list)
        if (RASTER && in_tile) {
            cnt = s_cnt[li];
            if (cnt >= 1 && cnt <= (uint32_t)PIP_KHIT) {
                // part hits -> geometry hits: ascending, each geometry once, null geometries dropped (written out
                // for PIP_KHIT == 2 so that nothing is a runtime-indexed register array)
                static_assert(PIP_KHIT == 2, "the finalize step is written for two remembered hits");
                uint32_t m = 0;
                {
                    const uint32_t part = s_hit[li * PIP_KHIT];
                    const uint32_t geom = pv.part_geom ? pv.part_geom[part] : part;
                    if (dev::valid_row(polys.validity, geom)) {
                        g0 = geom;
                        m = 1;
                    }
                }
                if (cnt == 2) {
                    const uint32_t part = s_hit[li * PIP_KHIT + 1];
                    const uint32_t geom = pv.part_geom ? pv.part_geom[part] : part;
                    if (dev::valid_row(polys.validity, geom)) {
                        if (m == 0) {
                            g0 = geom;
                            m = 1;
                        } else if (geom != g0) {
                            g1 = geom > g0 ? geom : g0;
                            g0 = geom > g0 ? g0 : geom;
                            m = 2;
                        }
                    }
                }
                cnt = m;
                if (m >= 1) first = g0;
                if (m == 2) {
                    need = 3u;
                    mode = 1;
                }
            } else if (cnt > (uint32_t)PIP_KHIT) {
                // more hits than inline slots: the rest sit in the tile's overflow list.  Collect all of them into a
                // pool segment, then sort + dedup there (a handful of words, one lane).
                novf = ovf_n;
                if (novf <= (uint32_t)PIP_OVF) {
                    need = cnt + 1u;
                    mode = 2;
                } else {
                    generic = true;
                }
            }
        }
        const uint32_t at = pool_reserve(need);  // (lanes past the end of the tile take part with need = 0)
        if (mode == 1 && at + 3u <= multi_cap) {
            multi_pool[at] = 2u;
            multi_pool[at + 1] = g0;
            multi_pool[at + 2] = g1;
            pool_code = CODE_POOL | at;
        }
        if (mode == 2) {
            if (at + cnt + 1u > multi_cap) {
                generic = true;
            } else {
                uint32_t* seg = multi_pool + at + 1;
                uint32_t m = 0;
                auto put = [&](uint32_t part) {
                    const uint32_t geom = pv.part_geom ? pv.part_geom[part] : part;
                    if (!dev::valid_row(polys.validity, geom)) return;
                    uint32_t j = m;  // insertion into the ascending prefix; equal geometry: drop
                    while (j > 0 && seg[j - 1] > geom) --j;
                    if (j > 0 && seg[j - 1] == geom) return;
                    for (uint32_t t = m; t > j; --t) seg[t] = seg[t - 1];
                    seg[j] = geom;
                    ++m;
                };
                for (int h = 0; h < PIP_KHIT; ++h) put(s_hit[li * PIP_KHIT + h]);
                for (uint32_t o = 0; o < novf; ++o) {
                    const uint2 e = s_ovf[o];
                    if (e.x == (uint32_t)li) put(e.y);
                }
                multi_pool[at] = m;
                cnt = m;
                if (m >= 1) first = seg[0];
                if (m >= 2) pool_code = CODE_POOL | at;
            }
        }
        if (!in_tile) continue;
        if (generic) {
            cnt = 0;
            if (dev::valid_row(pts.validity, i)) {
                const double2 pp = pts.xy[i];
                generic_point(polys, ix, pp.x, pp.y, cnt, first);
            }
        }
        const uint32_t cd = cnt == 0 ? CODE_NONE : (cnt == 1 ? first : pool_code);
        if (GPK_PIP_NT & 2) {
            if (tile_counts) dev::store_stream(tile_counts + (uint32_t)li, cnt);
            dev::store_stream(tile_code + (uint32_t)li, cd);
        } else {
            if (tile_counts) tile_counts[(uint32_t)li] = cnt;
            tile_code[(uint32_t)li] = cd;
        }
        wave_hits += (unsigned long long)__popcll(__ballot(cnt == 1));
        if (cnt >= 2) atomicAdd(&s_tot, (unsigned long long)cnt);
    }
    if ((tid & 63) == 0 && wave_hits) atomicAdd(&s_tot, wave_hits);
    __syncthreads();
    PIP_STAGE_CLK(5);
    if (tid == 0) {
        const unsigned long long tot = s_tot;
        block_tot[blockIdx.x] = tot;
        if (tot) atomicAdd(&super_tot[blockIdx.x >> PIP_SUPER_SHIFT], tot);  // integer adds: order-independent
    }
}

// ---- pip_tile_lean: the same tile step for right sides whose raster names at most ONE part per cell ---------------
// (gpk_index::pip_lean: no entry lists, no two-part records, no refined rings — disjoint polygons, the C2 right side.)
// A point then has one candidate part at most, so nothing per point lives in LDS: the lane keeps its points' results in
// registers, the only shared structure is the queue of (point, part) pairs that need the exact walk, and a queued pair's
// verdict comes back through its queue entry.  Versus pip_tile_kernel: no per-point LDS arrays (their initialisation, the
// LDS atomics of phase 2, one of the three barriers per round), no list / overflow / multi-hit arms, PPT points per lane
// in flight before the first dependent gather, one drain of the queue per tile.
#ifndef GPK_LEAN_PPT
#define GPK_LEAN_PPT 4
#endif
#ifndef GPK_LEAN_MINWAVES
#define GPK_LEAN_MINWAVES 1
#endif
#ifndef GPK_LEAN_NT
#define GPK_LEAN_NT 0  // 1: non-temporal point loads — a gain of 3 % while a queued pair carried its point in the entry; now the lane groups read the
                       // point again from the tile, and that second read wants the line still in L2 (136.5 vs 138.7 us)
#endif
constexpr int LEAN_PPT = GPK_LEAN_PPT, LEAN_TILE = PIP_BLOCK * LEAN_PPT;  // (the queue holds a whole tile)
static_assert(PIP_WTILE % LEAN_TILE == 0, "a writer tile is a whole number of lean tiles");
// GPK_TILE_TRACE (diagnosis builds only): lane 0 of every 64th tile stamps the wall clock at the stage boundaries of the lean
// kernel (after forcing the stage's loads to land) into the statistics buffer; tools/tile_trace.py prints the stage times.
#ifdef GPK_TILE_TRACE
#define TILE_STAMP(i)                                                                                              \
    do {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                \
        if (stats && tid == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 8000) stats[8 + (blockIdx.x >> 6) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define TILE_STAMP(i) do {} while (0)
#endif
constexpr uint32_t LEAN_SLOW = 1u << 30;  // LeanEntry::li_flags: a point of a list cell — one lane runs the generic walk for it
// A queued (point, part) pair of the lean kernel: 16 bytes.  The point itself is NOT in the entry — the lane group reads it
// from the tile (its address is known from `li`, so the load goes out together with the edge loads) — which lets the
// queue hold a whole tile (no overflow rounds: nothing of phase 1 has to stay in registers across the exact phase) in
// 16 KB of LDS.
struct LeanEntry {
    uint32_t li_flags;  // point index within the tile | LEAN_SLOW | (part has holes) << 31
    uint32_t part;      // in: the part; out: the part when the point is inside it, CODE_NONE otherwise (slow path: the result code)
    uint32_t e0, cnt;   // in: the exterior slab's edge range; out (slow path): cnt = hits
};
__global__ __launch_bounds__(PIP_BLOCK, GPK_LEAN_MINWAVES) void pip_tile_lean_kernel(DevGeo pts, DevGeo polys, IndexView ix, PipView pv,
                                                                                    uint32_t* __restrict__ counts, uint32_t* __restrict__ code,
                                                                                    unsigned long long* __restrict__ block_tot,
                                                                                    unsigned long long* __restrict__ super_tot,
                                                                                    unsigned long long* __restrict__ stats) {
    constexpr int PPT = LEAN_PPT, S = PIP_SUB, FINE = PIP_SLAB_MUL << PIP_FINE_LOG2, FY_SUB = PIP_FINE_LOG2 - 2;
    static_assert(FINE == S << FY_SUB, "fine rows per level-2 row");
    __shared__ LeanEntry q[LEAN_TILE];
    __shared__ uint32_t q_n;
    __shared__ unsigned long long s_tot;
    const int tid = threadIdx.x, lane64 = tid & 63;
    const int64_t base = (int64_t)blockIdx.x * LEAN_TILE;
    const int64_t n = pts.n_geoms;
    const uint32_t rem = (uint32_t)(n - base < (int64_t)LEAN_TILE ? n - base : (int64_t)LEAN_TILE);
    const double2* __restrict__ tile_xy = pts.xy + base;
    TILE_STAMP(0);

    // stage A: every point of the lane is requested before anything waits (coalesced 16-byte loads)
    double2 p[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t t = (uint32_t)(k * PIP_BLOCK + tid);
        const bool ok = t < rem && dev::valid_row(pts.validity, base + t);
        p[k] = ok ? (GPK_LEAN_NT ? dev::load_stream(tile_xy + t) : tile_xy[t]) : make_double2(NAN, NAN);
    }
    if (tid == 0) {
        q_n = 0;
        s_tot = 0;
    }
    __syncthreads();  // the loads above are in flight while the work-group meets here
    TILE_STAMP(1);

    // stage B: level-1 words (one 4-byte gather per point; empty points read nothing)
    uint32_t sx[PPT], fyf[PPT], word[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        sx[k] = (uint32_t)dev::cell_of(p[k].x, pv.rx0, pv.inv_fw * S, pv.R * S);
        fyf[k] = (uint32_t)dev::cell_of(p[k].y, pv.ry0, pv.inv_fh * FINE, pv.R * FINE);
        const uint32_t fy = fyf[k] >> FY_SUB;
        word[k] = (p[k].x == p[k].x && p[k].y == p[k].y) ? pv.cell[(fy / S) * (uint32_t)pv.R + (sx[k] / S)] : 0u;
        // tuning builds only (tools/ablate_time.py; the answers are wrong on purpose)
        if (GPK_ABLATE == 2) word[k] = 0u;                                                  // no gather at all: the streaming floor
        if (GPK_ABLATE == 3) word[k] = (word[k] >> 30) == CELL_TAG_SINGLE ? word[k] : 0u;  // level-1 interiors only
        if (GPK_ABLATE == 6) word[k] = (word[k] >> 30) == CELL_TAG_SUB ? ((CELL_TAG_SINGLE << 30) | (word[k] & 0x3FFFFFu)) : word[k];  // records never read
    }
    TILE_STAMP(2);
    // stage C: level-2 records of the cells an edge crosses (32 bytes: two 16-byte gathers off one line)
    uint4 ra[PPT], rb[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        ra[k] = rb[k] = make_uint4(0u, 0u, 0u, 0u);
        if ((word[k] >> 30) == CELL_TAG_SUB && !(word[k] & SUB2_BIT)) {
            const uint4* __restrict__ r = reinterpret_cast<const uint4*>(pv.sub + (word[k] & 0x3FFFFFFFu));
            ra[k] = r[0];  // part_flags, e0, e1, e2
            rb[k] = r[1];  // labels
        }
    }
    TILE_STAMP(3);
    // stage D: decide, or mark for the exact walk
    uint32_t res[PPT];   // part that contains the point / CODE_NONE — or, while bit k of `pending` is set, its queue slot
    uint32_t qpart[PPT], qe0[PPT], qcnt[PPT];
    bool todo[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t tag = word[k] >> 30, payload = word[k] & 0x3FFFFFFFu;
        res[k] = CODE_NONE;
        todo[k] = false;
        qpart[k] = ra[k].x;  // part | (has holes) << 31
        qe0[k] = qcnt[k] = 0;
        // (what the host-side gate of a lean index rules out — boundary entries without a record, two-part records, records
        // that reach their slabs through PartInfo — takes the generic walk here instead of being trusted not to occur)
        if (tag == CELL_TAG_SINGLE && !(payload & 1u)) {
            res[k] = payload >> 1;  // lean index: single entries are interiors (boundary cells carry records)
        } else if (tag == CELL_TAG_SUB && !(payload & SUB2_BIT) && !(ra[k].x & SUB_INDIRECT)) {
            const uint32_t fy = fyf[k] >> FY_SUB;
            const int idx = (int)((fy % S) * S + (sx[k] % S));
            const int wsel = idx >> 4;
            const uint32_t lw = wsel == 0 ? rb[k].x : (wsel == 1 ? rb[k].y : (wsel == 2 ? rb[k].z : rb[k].w));
            const uint32_t lab = (lw >> (2 * (idx & 15))) & 3u;
            if (lab == 1u) res[k] = ra[k].x & 0x3FFFFFFFu;
            if (lab == 2u) {
                const bool upper = ((fyf[k] >> PIP_FINE_LOG2) & 1u) != 0;
                qe0[k] = upper ? ra[k].z : ra[k].y;
                qcnt[k] = (upper ? ra[k].w : ra[k].z) - qe0[k];
                todo[k] = qcnt[k] > 0 && GPK_ABLATE != 1;  // an empty slab: p.y is outside the exterior's y-range (ablation 1: no exact phase)
            }
        } else if (tag != CELL_TAG_EMPTY) {  // the few cells where parts meet (a lean index has next to none): generic walk
            qpart[k] = LEAN_SLOW;
            todo[k] = true;
        }
    }
    uint32_t slow_cnt[PPT];  // hit count of a list-cell point (its code then names a geometry, CODE_NONE or CODE_MULTI)
#pragma unroll
    for (int k = 0; k < PPT; ++k) slow_cnt[k] = 0;
    // Queue the marked points (the queue holds a whole tile: one round) -> PIP_GS lanes per queued pair walk the slab's edges ->
    // owners collect the verdicts.  From here on a lane keeps only res[] / the pending mask of its points.
    unsigned long long edges_walked = 0, pairs_walked = 0;
    const int glane = tid & (PIP_GS - 1);
    uint32_t pending = 0;  // bit k: res[k] holds a queue slot whose verdict is still to be collected; bit 8 + k: a slow-path point
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const unsigned long long mask = __ballot(todo[k]);
        if (mask) {
            uint32_t wbase = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane64 == leader) wbase = atomicAdd(&q_n, (uint32_t)__popcll(mask));
            wbase = __shfl(wbase, leader, 64);
            const uint32_t slot = wbase + (uint32_t)__popcll(mask & ((1ull << lane64) - 1ull));
            if (todo[k]) {
                q[slot] = LeanEntry{(uint32_t)(k * PIP_BLOCK + tid) | (qpart[k] & (0x80000000u | LEAN_SLOW)), qpart[k] & 0x3FFFFFFFu, qe0[k], qcnt[k]};
                res[k] = slot;
                pending |= (1u << k) | (qpart[k] == LEAN_SLOW ? 0x100u << k : 0u);
            }
        }
    }
    __syncthreads();
    TILE_STAMP(4);
    const uint32_t nq = q_n;  // uniform: read after the barrier (<= LEAN_TILE: one entry per point at most)
    for (uint32_t e = tid / PIP_GS; e < nq; e += PIP_BLOCK / PIP_GS) {
        const LeanEntry en = q[e];
        const double2 pt = tile_xy[en.li_flags & 0x3FFFFFFFu];  // (requested together with the edges below; an L2 hit: the tile was just read)
        if (en.li_flags & LEAN_SLOW) {  // uniform within the group (one entry per group)
            if (glane == 0) {
                uint32_t cnt, first;
                generic_point(polys, ix, pt.x, pt.y, cnt, first);
                q[e].part = cnt == 0 ? CODE_NONE : (cnt == 1 ? first : CODE_MULTI);
                q[e].cnt = cnt;
            }
            continue;
        }
        const int pos = pip::part_pos_group_from_edges<PIP_GS>(pv, polys, (int)en.part, (en.li_flags >> 31) ? 2 : 1, (int)en.e0, (int)en.cnt, pt.x,
                                                               pt.y, glane);
        if (glane == 0) {
            q[e].part = pos == dev::POS_INSIDE ? en.part : CODE_NONE;  // the verdict travels back in the entry
            edges_walked += en.cnt;
            ++pairs_walked;
        }
    }
    TILE_STAMP(5);
    __syncthreads();
    TILE_STAMP(6);
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (pending & (1u << k)) {
            const uint32_t slot = res[k];
            res[k] = q[slot].part;
            if (pending & (0x100u << k)) slow_cnt[k] = q[slot].cnt | 0x80000000u;  // top bit: "this point went the slow way"
        }
    if (stats && pairs_walked) {  // measurement runs only (gpk_join_stats_enable): uniform branch on the pointer
        atomicAdd(&stats[0], pairs_walked);
        atomicAdd(&stats[1], edges_walked);
    }
    // finalize: part -> geometry (null geometries dropped), count + code, tile total
    uint32_t* __restrict__ tile_counts = counts ? counts + base : nullptr;
    uint32_t* __restrict__ tile_code = code + base;
    unsigned long long wave_hits = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t li = (uint32_t)(k * PIP_BLOCK + tid);
        uint32_t r = res[k], cnt;
        if (slow_cnt[k]) {  // list-cell point: the generic walk already speaks in geometries (validity included)
            cnt = slow_cnt[k] & 0x7FFFFFFFu;
        } else {
            if (r != CODE_NONE) {
                const uint32_t geom = pv.part_geom ? pv.part_geom[r] : r;
                r = dev::valid_row(polys.validity, geom) ? geom : CODE_NONE;
            }
            cnt = r != CODE_NONE ? 1u : 0u;
        }
        if (li < rem) {
            if (tile_counts) dev::store_stream(tile_counts + li, cnt);
            dev::store_stream(tile_code + li, r);
        }
        wave_hits += (unsigned long long)__popcll(__ballot(li < rem && cnt == 1u));
        if (li < rem && cnt >= 2u) atomicAdd(&s_tot, (unsigned long long)cnt);
    }
    if (lane64 == 0 && wave_hits) atomicAdd(&s_tot, wave_hits);
    TILE_STAMP(7);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long tot = s_tot;
        block_tot[blockIdx.x] = tot;
        if (tot) atomicAdd(&super_tot[blockIdx.x >> PIP_SUPER_SHIFT], tot);
    }
}

// ---- pip_tile_chain: the lean tile step with `test` sub-cells decided from their local chains ----------------------------------------
// (gpk_index::pip_lean with chain tables, gpk_index.h.)  One WAVE owns a tile of 64 * P points and never meets another wave: no
// work-group queue, no barrier.  A point's way through the tables, one dependent memory round trip per step:
//   1. the point itself (coalesced 16-byte loads, non-temporal: read once);
//   2. its raster cell's level-1 word: empty -> done; strictly inside a part -> the word names it; an edge crosses the cell -> the
//      half-cell record of the point's sub-cell (ONE 16-byte request: 32 labels, the part, the half's chain word);
//   3. the label: outside / inside -> done; `test` -> the half cell's chain (gpk_index.h: GPK_HALF_CHAINS);
//   4. base + the contributions of the chain's edges; count + code + tile total (pip_write turns the codes into pairs).
// Step 3 / 4 concern one point in twenty, scattered over the lanes: the wave packs those points into a list in its own slice of
// LDS (no other wave sees it: wave-level ordering is enough) and lanes 0 .. T - 1 take one each — one round trip and one pass of
// the orientation code for all of the tile's `test` points, in dense lanes.
// This kernel serves chain indexes WITHOUT an LDS routing image (rasters beyond PIP_ROUTE_RMAX) and whatever gpk_pipflow.hip's
// one-launch join does not take (GPK_TILE_KERNEL=chain: A/B runs); with the image, pip_tile_flow_kernel is the join.  The round-3 .. 5 forms
// that sat between the two (level 1 from the image in persistent work-groups, hits in per-wave LDS lists / staging slots / chunks /
// a work-group pool) were retired in round 6: DESIGN.md 4.1 keeps their measurements.
//
// The kernel's arguments hold only what the hot path reads (ChainHot, gpk_pipshared.h); what needs the full views — a point of a
// list cell, a `test` point whose half cell has no chain, a point whose orientation against a chain edge Shewchuk's stage-A bound
// cannot certify: a handful per launch on real data — is settled at the end of the tile by the whole wave with the generic (always
// exact) walk, its arguments read from device memory (ChainCold) at that point.
#ifndef GPK_CHAIN_PPT
#define GPK_CHAIN_PPT 4
#endif
#ifndef GPK_CHAIN_NT
#define GPK_CHAIN_NT 1  // non-temporal point loads: a point is read exactly once by this kernel
#endif
#ifndef GPK_CHAIN_ABLATE
#define GPK_CHAIN_ABLATE 0  // tuning builds only (answers wrong on purpose): 1 = `test` points count as outside
#endif
constexpr int CHAIN_PPT = GPK_CHAIN_PPT;
static_assert(PIP_WTILE % (64 * CHAIN_PPT) == 0, "a writer tile is a whole number of chain tiles");

// one small launch in place of the memset of a join's totals: zeroes them and writes the rare arm's arguments
__global__ __launch_bounds__(256) void join_prep_kernel(unsigned long long* __restrict__ zero, int64_t n_zero, ChainCold* __restrict__ cold_out, DevGeo polys,
                                                        IndexView ix) {
    for (int64_t i = threadIdx.x; i < n_zero; i += 256) zero[i] = 0ull;
    if (threadIdx.x == 0) {
        cold_out->polys = polys;
        cold_out->ix = ix;
        cold_out->grid = *ix.grid;
    }
}
// one `test` point of a tile, in the wave's LDS list (24 bytes; reading the point again from memory instead was measured: the
// tile's lines are streamed with the non-temporal hint and are gone from the L2 — 20 us more per launch)
struct ChainItem {
    double px, py;
    uint32_t aux_at;  // in: the point's chain entry; out: the verdict (bit 0 inside, bit 1 defer)
    uint32_t pad;
};
// list slots per wave = one pass of the exact step: a quarter of the tile's points.  A tile with more `test` points than that (the
// raster is far too coarse for such a right side) hands the surplus to the generic walk like any other deferred row.
template <int P>
constexpr int chain_items() { return 16 * P; }

// FULL: the tile holds 64 * P points and the left column has no validity bitmap (wave-uniform, true for all tiles but the last
// of a plain column): no per-point guards on loads and stores
template <int P, bool FULL>
__device__ __forceinline__ void chain_tile(const ChainHot& h, ChainItem* s_items, int64_t tile, int lane) {
    constexpr int S = PIP_SUB, CHAIN_TILE = 64 * P, ITEMS = chain_items<P>();
    constexpr bool FUSED = false;  // (the exact step below is shared text with the round-4 / 5 kernels: no argument-segment reads here)
    const int64_t base = tile * CHAIN_TILE;
    const uint32_t rem = FULL ? (uint32_t)CHAIN_TILE : (uint32_t)(h.n_points - base < (int64_t)CHAIN_TILE ? h.n_points - base : (int64_t)CHAIN_TILE);
    const int logR = h.logR;
    // 1. the points (NaN for rows past the end and null rows)
    const double2* __restrict__ tile_xy = h.pts_xy + base;
    double px[P], py[P];
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        double2 v = make_double2(NAN, NAN);
        if (FULL || ((uint32_t)(k * 64 + lane) < rem && dev::valid_row(h.pts_validity, base + k * 64 + lane)))
            v = GPK_CHAIN_NT ? dev::load_stream(tile_xy + (k * 64 + lane)) : tile_xy[k * 64 + lane];
        px[k] = v.x;
        py[k] = v.y;
    });
    // (GPK_SCHED_FENCE: nothing is moved across — left to itself the scheduler interleaves the steps of different points until it
    // runs out of registers, then spills; the source order below IS the intended schedule: requests of a step back to back, their
    // uses in the next step)
    GPK_SCHED_FENCE();
    // 2. level 1, then the half-cell records
    uint32_t sidx4[(P + 3) / 4], w[P], gw[P];  // sub-cell within the cell (label index, x fastest; byte k % 4 of word k / 4); record index; level-1 word
    uint32_t recmask = 0u;                     // bit k: point k's cell carries a one-part record and w[k] is its index
    const double sub_max = (double)(((uint32_t)S << logR) - 1u);
    const double inv_w_s = h.inv_fw * S, inv_h_s = h.inv_fh * S;
    const uint32_t* __restrict__ const cell_words = h.cell;
    static_for<(P + 3) / 4>([&](auto J) { sidx4[decltype(J)::value] = 0u; });
#define SIDX(k) ((sidx4[(k) / 4] >> (8 * ((k) % 4))) & 0xFFu)
#define HOT_ARG(field) (h.field)
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        // dev::cell_of at sub-cell resolution (negative / NaN products clamp to 0, large ones to the last sub-cell)
        const double pxk = px[k], pyk = py[k];
        const uint32_t sx = (uint32_t)fmin(fmax((pxk - h.rx0) * inv_w_s, 0.0), sub_max);
        const uint32_t sy = (uint32_t)fmin(fmax((pyk - h.ry0) * inv_h_s, 0.0), sub_max);
        const bool real = pxk == pxk && pyk == pyk;
        const uint32_t cx = sx / S, cy = sy / S;
        sidx4[k / 4] |= ((sy % S) * S + (sx % S)) << (8 * (k % 4));
        w[k] = gw[k] = 0u;
        if (real) gw[k] = cell_words[(cy << logR) + cx];
        GPK_SCHED_FENCE();
    });
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const HalfCell* __restrict__ const half_recs = h.half;
    u32x4 rec[P];  // HalfCell: lw[0], lw[1], part, aux_base  (kept as the 16-byte tuple the request fills: copies out of it would sit
                   // in the requesting branch and wait for the request on the spot — P round trips one after the other)
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        rec[k] = u32x4{0u, 0u, 0u, 0u};
        if ((gw[k] >> 30) == CELL_TAG_SUB && !(gw[k] & SUB2_BIT)) {  // (the record's index is in the word just read)
            w[k] = gw[k] & 0x3FFFFFFFu;
            recmask |= 1u << k;
        }
        if ((recmask >> k) & 1u) rec[k] = *reinterpret_cast<const u32x4*>(half_recs + 2u * w[k] + (SIDX(k) >> 5));
    });
    GPK_SCHED_FENCE();
    // 3. labels; a `test` point goes into the wave's LDS list with its chain entry; anything a lean index should not hold is deferred
    static_assert(ITEMS <= 256, "a list slot is an 8-bit field");
    uint32_t res[P], slots[(P + 3) / 4];  // slots: the list slot of point k in byte k % 4 of word k / 4
    static_for<(P + 3) / 4>([&](auto J) { slots[decltype(J)::value] = 0u; });
    uint32_t tmask = 0u, dmask = 0u;  // bit k: point k is in the list / is deferred to the generic walk
    uint32_t n_items = 0;             // wave-uniform
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const bool has = ((recmask >> k) & 1u) != 0u;
        const uint32_t tag = gw[k] >> 30, payload = gw[k] & 0x3FFFFFFFu;
        res[k] = CODE_NONE;
        bool test = false;
        uint32_t aux_at = 0u;
        if (has) {
            const uint32_t si = SIDX(k), sh = 2u * (si & 15u), upper = (si >> 4) & 1u;
            const uint32_t lw = upper ? rec[k].y : rec[k].x;
            const uint32_t lab = (lw >> sh) & 3u;
            if (lab >= 1u) res[k] = rec[k].z & 0x3FFFFFFFu;
            if (lab >= 2u) {
#if GPK_HALF_CHAINS
                aux_at = rec[k].w;  // the half's chain word
#else
                // rank of this `test` label among the half's: the lower label word (if the label sits in the upper one), then
                // the fields below it in its own word
                const uint32_t tl = (lw >> 1) & ~lw & 0x55555555u, tl0 = (rec[k].x >> 1) & ~rec[k].x & 0x55555555u;
                aux_at = rec[k].w + (upper ? (uint32_t)__popc(tl0) : 0u) + (uint32_t)__popc(tl & ((1u << sh) - 1u));
#endif
                if (GPK_CHAIN_ABLATE == 1)
                    res[k] = CODE_NONE;
                else
                    test = true;
            }
        } else if (tag == CELL_TAG_SINGLE && !(payload & 1u)) {
            res[k] = payload >> 1;
        } else if (gw[k] != 0u) {  // list cells (a lean index has next to none), and whatever a lean index should not hold
            dmask |= 1u << k;
        }
        const unsigned long long m = __ballot(test);
        if (m) {  // (wave-uniform)
            const uint32_t at = n_items + (FUSED ? lanes_below(m) : (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
            n_items += (uint32_t)__popcll(m);
            if (test) {
                if (at < (uint32_t)ITEMS) {
                    slots[k / 4] |= at << (8 * (k % 4));
                    tmask |= 1u << k;
                    ChainItem* it = s_items + at;
                    it->px = px[k];
                    it->py = py[k];
                    it->aux_at = aux_at;
                } else {
                    dmask |= 1u << k;  // the list is full
                }
            }
        }
        GPK_SCHED_FENCE();
    });
#undef SIDX
    uint32_t* const counts_all = h.counts;
    uint32_t* __restrict__ tile_counts = counts_all ? counts_all + base : nullptr;
    uint32_t* __restrict__ tile_code = h.code + base;
    // 4. the exact step: one listed point per lane and pass
    n_items = n_items < (uint32_t)ITEMS ? n_items : (uint32_t)ITEMS;
    unsigned long long edges_walked = 0;
    if (n_items) {  // (wave-uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if GPK_HALF_CHAINS
        const double2* __restrict__ const cxy = HOT_ARG(chain_xy);
        for (uint32_t i = (uint32_t)lane; i < n_items; i += 64u) {
            const uint32_t hd = s_items[i].aux_at;  // the half's chain word: count, base, first vertex
            const double qx = s_items[i].px, qy = s_items[i].py;
            const int count = (int)(hd & HCHAIN_COUNT_MASK);
            const double2* __restrict__ v = cxy + (hd >> HCHAIN_START_SHIFT);
            // the first five vertices (four edges: 99 % of the chains of the C2 right side) are requested together; a short chain
            // repeats its last vertex's request, which costs nothing new
            double2 a0 = v[0], a1 = v[count >= 1 ? 1 : 0], a2 = v[count >= 2 ? 2 : (count >= 1 ? 1 : 0)];
            double2 a3 = a2, a4 = a2;
            if (count >= 3) {
                a3 = v[3];
                a4 = v[count >= 4 ? 4 : 3];
            }
            asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y), "+v"(a4.x), "+v"(a4.y));
            bool inside = false, defer = count == 0;  // no chain for this half cell: the generic walk decides
            if (count > 0) {
                int wn = ((int)(hd << (28 - HCHAIN_BASE_SHIFT))) >> 28;  // the signed 4-bit base
                bool on = dev::ring_edge_filtered(a0.x, a0.y, a1.x, a1.y, qx, qy, wn, defer);
                if (count >= 2) on |= dev::ring_edge_filtered(a1.x, a1.y, a2.x, a2.y, qx, qy, wn, defer);
                if (count >= 3) on |= dev::ring_edge_filtered(a2.x, a2.y, a3.x, a3.y, qx, qy, wn, defer);
                if (count >= 4) on |= dev::ring_edge_filtered(a3.x, a3.y, a4.x, a4.y, qx, qy, wn, defer);
                if (count > 4) {
                    double ax = a4.x, ay = a4.y;
                    for (int j = 4; j < count; ++j) {
                        const double2 b2 = v[j + 1];
                        on |= dev::ring_edge_filtered(ax, ay, b2.x, b2.y, qx, qy, wn, defer);
                        ax = b2.x;
                        ay = b2.y;
                    }
                }
                inside = !on && wn != 0;
                edges_walked += (unsigned long long)count;
            }
#else
        const ChainAux* __restrict__ const aux_all = HOT_ARG(sub_aux);
        const uint32_t* __restrict__ const head_all = HOT_ARG(chain_head);
        for (uint32_t i = (uint32_t)lane; i < n_items; i += 64u) {
            const uint32_t at = GPK_CHAIN_ABLATE == 2 ? (s_items[i].aux_at & 0x3FFFu) : s_items[i].aux_at;  // (2, tuning builds only: chain entries from a 1 MB corner of the table)
            const ChainAux* __restrict__ e = aux_all + at;
            double2 q = make_double2(s_items[i].px, s_items[i].py);
            uint32_t hd = head_all[at];
            double2 a0 = e->v[0], a1 = e->v[1], a2 = e->v[2], a3 = e->v[3];
            // (all five requests go out before the head is looked at: the compiler would otherwise sink the vertex requests
            // under `count > 0` — a second round trip)
            asm volatile("" : "+v"(hd), "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y));
            const double qx = q.x, qy = q.y;
            const int count = (int)(hd & CHAIN_COUNT_MASK);
            bool inside = false, defer = count == 0;  // no chain entry for this sub-cell: the generic walk decides
            if (count > 0) {
                int wn = (int)(int8_t)((hd >> CHAIN_BASE_SHIFT) & 0xFFu);
                bool on = dev::ring_edge_filtered(a0.x, a0.y, a1.x, a1.y, qx, qy, wn, defer);
                if (count >= 2) on |= dev::ring_edge_filtered(a1.x, a1.y, a2.x, a2.y, qx, qy, wn, defer);
                if (count >= 3) on |= dev::ring_edge_filtered(a2.x, a2.y, a3.x, a3.y, qx, qy, wn, defer);
                if (count > 3) {  // 0.2 % of the chains: the further vertices follow in chain_ext
                    const double2* __restrict__ ev = HOT_ARG(chain_ext) + (hd >> CHAIN_EXT_SHIFT);
                    double ax = a3.x, ay = a3.y;
                    for (int j = 3; j < count; ++j) {
                        const double2 b = ev[j - 3];
                        on |= dev::ring_edge_filtered(ax, ay, b.x, b.y, qx, qy, wn, defer);
                        ax = b.x;
                        ay = b.y;
                    }
                }
                inside = !on && wn != 0;
                edges_walked += (unsigned long long)count;
            }
#endif
            s_items[i].aux_at = (inside && !defer ? 1u : 0u) | (defer ? 2u : 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        static_for<P>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if ((tmask >> k) & 1u) {
                const uint32_t v = s_items[(slots[k / 4] >> (8 * (k % 4))) & 0xFFu].aux_at;
                if (!(v & 1u)) res[k] = CODE_NONE;
                if (v & 2u) dmask |= 1u << k;
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next tile overwrites the list)
        __builtin_amdgcn_wave_barrier();
    }
    // the rare rows (list-cell points, sub-cells without a chain entry, orientations the filter could not certify) are listed now —
    // the point's index within the tile, in the wave's LDS list, which holds a whole tile of them — and settled at the very end of
    // the tile, when nothing of the tile's state is live any more
    uint32_t n_rare = 0;  // wave-uniform
    uint32_t* s_rare = reinterpret_cast<uint32_t*>(s_items);
    static_assert(sizeof(ChainItem) * ITEMS >= sizeof(uint32_t) * 64 * P, "the list holds a tile of rare rows");
    if (__any(dmask != 0u)) {
        static_for<P>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const bool want = ((dmask >> k) & 1u) != 0u;
            const unsigned long long m = __ballot(want);
            if (want) s_rare[n_rare + (FUSED ? lanes_below(m) : (uint32_t)__popcll(m & ((1ull << lane) - 1ull)))] = (uint32_t)(k * 64 + lane);
            n_rare += (uint32_t)__popcll(m);
        });
    }
    if (h.stats && n_items) {  // measurement runs only (gpk_join_stats_enable): uniform branch on the pointer
        if (lane == 0) atomicAdd(&h.stats[0], (unsigned long long)n_items);
        if (edges_walked) atomicAdd(&h.stats[1], edges_walked);
    }
    // part -> geometry (null geometries dropped), count + code, the tile's total
    unsigned long long hits = 0;  // wave-uniform
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const uint32_t li = (uint32_t)(k * 64 + lane);
        uint32_t r = res[k];
        if (r != CODE_NONE) {
            const uint32_t geom = h.part_geom ? h.part_geom[r] : r;
            r = dev::valid_row(h.polys_validity, geom) ? geom : CODE_NONE;
        }
        const uint32_t cnt = r != CODE_NONE ? 1u : 0u;
        const bool mine = (FULL || li < rem) && !((dmask >> k) & 1u);  // (a rare row's count and code were written by its walk)
        if (mine) {
            if (tile_counts) dev::store_stream(tile_counts + li, cnt);
            dev::store_stream(tile_code + li, r);
        }
        hits += (unsigned long long)__popcll(__ballot(mine && cnt == 1u));
    });
    // the rare rows: the whole wave walks one row after the other with the generic walk; the walk's arguments come from device
    // memory (ChainCold).  Their counts and codes are written here (the owning lanes skipped their stores).
    if (n_rare) {  // (wave-uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = 0; i < n_rare; ++i) {
            const uint32_t li = s_rare[i];
            const double2 q = tile_xy[li];
            uint32_t first;
            const uint32_t cnt = chain_generic_row<false>(h.cold, q.x, q.y, lane, &first);
            if (lane == 0) {
                if (tile_counts) tile_counts[li] = cnt;
                tile_code[li] = cnt == 0 ? CODE_NONE : (cnt == 1 ? first : CODE_MULTI);
            }
            hits += cnt;
        }
        if (h.stats && lane == 0) atomicAdd(&h.stats[2], (unsigned long long)n_rare);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next tile overwrites the list)
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        h.block_tot[tile] = hits;
        if (hits) atomicAdd(&h.super_tot[tile >> PIP_SUPER_SHIFT], hits);  // integer adds: order-independent
    }
}
#undef HOT_ARG
__global__ __launch_bounds__(PIP_BLOCK) void pip_tile_chain_kernel(ChainHot h) {
    __shared__ ChainItem s_items[PIP_BLOCK / 64][chain_items<CHAIN_PPT>()];
    const int64_t tile = (int64_t)blockIdx.x * (PIP_BLOCK / 64) + (threadIdx.x >> 6);
    if (tile >= h.n_tiles) return;  // (whole waves)
    const int lane = threadIdx.x & 63;
    if (h.pts_validity == nullptr && (tile + 1) * (int64_t)(64 * CHAIN_PPT) <= h.n_points)  // (wave-uniform)
        chain_tile<CHAIN_PPT, true>(h, s_items[threadIdx.x >> 6], tile, lane);
    else
        chain_tile<CHAIN_PPT, false>(h, s_items[threadIdx.x >> 6], tile, lane);
}

// pip_write: turns the per-point codes into the sorted (l, r) pair list.  Reads 4 bytes per point, writes 8
// bytes per hit; only CODE_MULTI rows touch geometry again.  There is no separate scan kernel: a work-group
// gets its global offset from the two-level totals (<= 64 tile totals + the super-tile totals before them,
// summed by one wave), each thread owns PIP_WPT consecutive points, one block scan orders the threads.
__global__ __launch_bounds__(WR_BLOCK) void pip_write_kernel(DevGeo pts, DevGeo polys, IndexView ix,
                                                               const uint32_t* __restrict__ code,
                                                               const unsigned long long* __restrict__ block_tot,
                                                               const unsigned long long* __restrict__ super_tot,
                                                               const uint32_t* __restrict__ multi_pool,
                                                               int64_t n_tiles, int tile_points, uint32_t left_base,
                                                               uint2* __restrict__ pairs, int64_t capacity,
                                                               unsigned long long* __restrict__ grand,
                                                               unsigned long long* __restrict__ grand_host) {
    __shared__ unsigned long long lds[WR_BLOCK / 64 + 1];
    __shared__ unsigned long long s_base;
    __shared__ uint2 s_pairs[WR_CAP];
    const int tid = threadIdx.x;
    const int64_t first_tile = (int64_t)blockIdx.x * (PIP_WTILE / tile_points);  // tiles of the kernel that produced the totals
    if (tid < 64) {
        const int64_t sb = first_tile >> PIP_SUPER_SHIFT;
        unsigned long long acc = 0;
        for (int64_t i = tid; i < sb; i += 64) acc += super_tot[i];
        for (int64_t i = (sb << PIP_SUPER_SHIFT) + tid; i < first_tile; i += 64) acc += block_tot[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (tid == 0) s_base = acc;
    }
    const int64_t i0 = (int64_t)blockIdx.x * PIP_WTILE + (int64_t)tid * PIP_WPT;
    uint32_t c[PIP_WPT];
    static_assert(PIP_WPT % 4 == 0, "16-byte code loads");
    if (i0 + PIP_WPT <= pts.n_geoms) {
#pragma unroll
        for (int q = 0; q < PIP_WPT / 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4*>(code + i0 + 4 * q);
            c[4 * q] = a.x; c[4 * q + 1] = a.y; c[4 * q + 2] = a.z; c[4 * q + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PIP_WPT; ++k) c[k] = i0 + k < pts.n_geoms ? code[i0 + k] : CODE_NONE;
    }
    uint32_t cnt[PIP_WPT], mine = 0;
#pragma unroll
    for (int k = 0; k < PIP_WPT; ++k) {
        cnt[k] = c[k] == CODE_NONE ? 0u : 1u;
        if (c[k] != CODE_NONE && c[k] != CODE_MULTI && (c[k] & CODE_POOL)) cnt[k] = multi_pool[c[k] & ~CODE_POOL];
        if (c[k] == CODE_MULTI) {
            uint32_t first;
            const double2 p = pts.xy[i0 + k];
            generic_point(polys, ix, p.x, p.y, cnt[k], first);
        }
        mine += cnt[k];
    }
    unsigned long long tot;
    const unsigned long long ex = dev::block_exclusive_scan<unsigned long long, WR_BLOCK>((unsigned long long)mine, lds, &tot);
    // (block_exclusive_scan's barriers also publish s_base)
    int64_t o = (int64_t)(s_base + ex);
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        *grand = s_base + tot;
        if (grand_host) *grand_host = s_base + tot;
    }
    if (!pairs) return;
    // Pairs are compacted in LDS and copied out by consecutive lanes (full-line stores); a tile with more pairs than the
    // LDS list holds (heavily overlapping right sides) writes them straight from the owning lanes instead.
    const bool via_lds = tot <= (unsigned long long)WR_CAP;  // uniform across the work-group
    const int64_t out_base = (int64_t)s_base;
    auto emit = [&](int64_t at, uint2 v) {
        if (via_lds)
            s_pairs[at - out_base] = v;
        else if (at < capacity)
            pairs[at] = v;
    };
#pragma unroll
    for (int k = 0; k < PIP_WPT; ++k) {
        if (cnt[k] == 0) continue;
        const uint32_t l = left_base + (uint32_t)(i0 + k);
        if (c[k] != CODE_MULTI && (c[k] & CODE_POOL)) {  // several geometries, listed in the pool
            const uint32_t at = c[k] & ~CODE_POOL;
            for (uint32_t t = 0; t < cnt[k]; ++t) emit(o++, make_uint2(l, multi_pool[at + 1 + t]));
            continue;
        }
        if (c[k] != CODE_MULTI) {
            emit(o++, make_uint2(l, c[k]));
            continue;
        }
        const double2 p = pts.xy[i0 + k];
        const GridParams g = *ix.grid;
        for_each_candidate(ix, g, p.x, p.y, [&](int j) {
            if (dev::valid_row(polys.validity, j) && dev::polygonal_hits_point<false>(polys, j, p.x, p.y)) emit(o++, make_uint2(l, (uint32_t)j));
        });
    }
    if (via_lds) {
        __syncthreads();
        const int n_out = (int)tot;
        for (int t = tid; t < n_out; t += WR_BLOCK)
            if (out_base + t < capacity) pairs[out_base + t] = s_pairs[t];
    }
}

// ---- join statistics (bench.py's edge_tests/s; off unless enabled) ---------------------------------------------------
constexpr size_t JOIN_STATS_WORDS = 8 + 8 * 8000;  // + the stage stamps a GPK_TILE_TRACE build of gpk_pipflow.hip leaves (512 KB, only when statistics are on)
static unsigned long long* g_join_stats = nullptr;  // device: {queued (point, part) pairs, slab edges walked by the exact phase, 0, 0}
static bool g_join_stats_on = false;
static unsigned long long* join_stats_buffer() { return g_join_stats_on ? g_join_stats : nullptr; }

// ================================= polygonal x polygonal join ======================================
// Candidate generation of spatial_index.rs:74-76 for bbox-shaped left rows: every directory cell the left
// bbox touches is visited; a pair seen in several cells is processed only in the cell that holds the lower
// left corner of the two boxes' intersection (computed with the same monotone cell function, so that cell
// is in both registration ranges).  The exact refine is Intersects<Polygon> (gpk_polypoly.h).
template <typename F>
__device__ __forceinline__ void for_each_bbox_candidate(const IndexView& ix, const GridParams& g, const double4 lb, F&& f) {
    if (!(lb.x == lb.x)) return;  // empty left geometry
    const int cx0 = dev::cell_of(lb.x, g.x0, g.inv_w, g.gx), cx1 = dev::cell_of(lb.z, g.x0, g.inv_w, g.gx);
    const int cy0 = dev::cell_of(lb.y, g.y0, g.inv_h, g.gy), cy1 = dev::cell_of(lb.w, g.y0, g.inv_h, g.gy);
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const int c = cy * g.gx + cx;
            auto visit = [&](int j, const double4 rb) {
                if (lb.z < rb.x || lb.w < rb.y || rb.z < lb.x || rb.w < lb.y) return;  // closed-interval overlap test
                const double rx = lb.x > rb.x ? lb.x : rb.x, ry = lb.y > rb.y ? lb.y : rb.y;
                if (dev::cell_of(rx, g.x0, g.inv_w, g.gx) != cx || dev::cell_of(ry, g.y0, g.inv_h, g.gy) != cy) return;
                f(j);
            };
            const int k1 = ix.cell_off[c + 1];
            for (int k = ix.cell_off[c]; k < k1; k += 2) {  // two items per trip: both ids, then both boxes, in flight together
                const bool two = k + 1 < k1;
                const int j0 = ix.items[k], j1 = ix.items[two ? k + 1 : k];
                const double4 b0 = ix.bbox[j0], b1 = ix.bbox[j1];
                visit(j0, b0);
                if (two) visit(j1, b1);
            }
        }
}

// Stage 1: candidates.  One lane per left row lists the right rows whose closed bbox overlaps the row's bbox
// (count pass, then fill pass into the row's slice, sorted by right id: the hits then come out sorted).  Rows with a
// handful of candidates sort their slice in place; a row with more than CAND_INLINE_SORT (one country against a
// column of parcels) raises *big_rows in the count pass and every slice goes through one segmented radix sort instead.
constexpr int CAND_INLINE_SORT = 48;
template <bool WRITE>
__global__ __launch_bounds__(256) void bbox_cand_kernel(DevGeo left, DevGeo right, IndexView ix, const double4* __restrict__ lbbox,
                                                         int32_t* __restrict__ cand_cnt, const int32_t* __restrict__ cand_off,
                                                         uint32_t* __restrict__ cand_r, uint32_t* __restrict__ cand_l,
                                                         int32_t* __restrict__ big_rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= left.n_geoms) return;
    int cnt = 0;
    const int64_t o0 = WRITE ? (int64_t)cand_off[i] : 0;
    if (dev::valid_row(left.validity, i)) {
        const GridParams g = *ix.grid;
        for_each_bbox_candidate(ix, g, lbbox[i], [&](int j) {
            if (!dev::valid_row(right.validity, j)) return;
            if (WRITE) cand_r[o0 + cnt] = (uint32_t)j;
            ++cnt;
        });
    }
    if (!WRITE) {
        cand_cnt[i] = cnt;
        if (cnt > CAND_INLINE_SORT) *big_rows = 1;
        return;
    }
    for (int a = 1; a < cnt && cnt <= CAND_INLINE_SORT; ++a) {  // rows have a handful of candidates
        const uint32_t key = cand_r[o0 + a];
        int b = a - 1;
        while (b >= 0 && cand_r[o0 + b] > key) {
            cand_r[o0 + b + 1] = cand_r[o0 + b];
            --b;
        }
        cand_r[o0 + b + 1] = key;
    }
    for (int a = 0; a < cnt; ++a) cand_l[o0 + a] = (uint32_t)i;  // left row of every candidate: the refine reads it directly
}

// One search instead of two for ordinary rows: the count pass also leaves each row's first CAND_STAGE candidates (sorted) in a padded
// staging slice; when no row has more (nearly every join: rows have a handful), cand_compact_kernel moves the slices to their scanned
// offsets and the second directory walk (bbox_cand_kernel<true>: 0.90 ms of the 6.1 ms C4 join) does not run.
constexpr int CAND_STAGE = 16;
static_assert(CAND_STAGE <= CAND_INLINE_SORT, "a staged row is one that the fill pass would have sorted inline");
// Round 6: CAND_LANES lanes per left row.  One lane per row walked its cells' items as a chain of dependent requests — cell offsets, then
// ids two at a time, then their boxes — about eight round trips a row with the lanes of a wave on rows of different lengths (0.82 ms
// for the 1 M rows of C4); the lanes of a row now take the items of a cell side by side (ids together, boxes together: two round trips
// a cell) and append their finds to the row's slice with one ballot.
constexpr int CAND_LANES = 8;
__global__ __launch_bounds__(256) void bbox_cand_stage_kernel(DevGeo left, DevGeo right, IndexView ix, const double4* __restrict__ lbbox,
                                                               int32_t* __restrict__ cand_cnt, uint32_t* __restrict__ stage,
                                                               int32_t* __restrict__ flags /* [0]: big rows, [1]: rows beyond CAND_STAGE */) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / CAND_LANES;
    const int sub = (int)(threadIdx.x & (CAND_LANES - 1)), gbase = (int)(threadIdx.x & 63) & ~(CAND_LANES - 1);
    if (i >= left.n_geoms) return;  // (whole groups: CAND_LANES divides the block)
    int cnt = 0;
    uint32_t* mine = stage + i * CAND_STAGE;
    const double4 lb = lbbox[i];
    if (dev::valid_row(left.validity, i) && lb.x == lb.x) {
        const GridParams g = *ix.grid;
        const int cx0 = dev::cell_of(lb.x, g.x0, g.inv_w, g.gx), cx1 = dev::cell_of(lb.z, g.x0, g.inv_w, g.gx);
        const int cy0 = dev::cell_of(lb.y, g.y0, g.inv_h, g.gy), cy1 = dev::cell_of(lb.w, g.y0, g.inv_h, g.gy);
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) {
                const int c = cy * g.gx + cx;
                const int k0 = ix.cell_off[c], k1 = ix.cell_off[c + 1];
                for (int kb = k0; kb < k1; kb += CAND_LANES) {  // (group-uniform trip count)
                    const int k = kb + sub;
                    bool keep = false;
                    int j = 0;
                    if (k < k1) {
                        j = ix.items[k];
                        const double4 rb = ix.bbox[j];
                        // closed-interval overlap, and the pair belongs to THIS cell: the one that holds the lower-left corner of the two
                        // boxes' intersection (for_each_bbox_candidate)
                        if (!(lb.z < rb.x || lb.w < rb.y || rb.z < lb.x || rb.w < lb.y)) {
                            const double rx = lb.x > rb.x ? lb.x : rb.x, ry = lb.y > rb.y ? lb.y : rb.y;
                            keep = dev::cell_of(rx, g.x0, g.inv_w, g.gx) == cx && dev::cell_of(ry, g.y0, g.inv_h, g.gy) == cy && dev::valid_row(right.validity, j);
                        }
                    }
                    const uint32_t m = (uint32_t)((__ballot(keep) >> gbase) & ((1u << CAND_LANES) - 1u));
                    const int at = cnt + __popc(m & ((1u << sub) - 1u));
                    if (keep && at < CAND_STAGE) mine[at] = (uint32_t)j;  // (in directory order: cand_compact_kernel sorts the slice across its 16 lanes)
                    cnt += __popc(m);
                }
            }
    }
    if (sub == 0) {
        cand_cnt[i] = cnt;
        if (cnt > CAND_STAGE) flags[1] = 1;
        if (cnt > CAND_INLINE_SORT) flags[0] = 1;
    }
}
// (a row with more than CAND_STAGE candidates — a dense cluster — walks the directory again, like bbox_cand_kernel<true>: one lane of
// its CAND_STAGE; rows beyond CAND_INLINE_SORT send the whole join down the two-search path with its segmented sort)
__global__ __launch_bounds__(256) void cand_compact_kernel(DevGeo left, DevGeo right, IndexView ix, const double4* __restrict__ lbbox,
                                                            const int32_t* __restrict__ cand_cnt, const int32_t* __restrict__ cand_off,
                                                            const uint32_t* __restrict__ stage, uint32_t* __restrict__ cand_r,
                                                            uint32_t* __restrict__ cand_l) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / CAND_STAGE;
    const int j = (int)(t % CAND_STAGE);
    if (i >= left.n_geoms) return;
    const int cnt = cand_cnt[i];
    const int64_t o0 = (int64_t)cand_off[i];
    if (cnt <= CAND_STAGE) {
        // the row's slice, sorted by right id across the row's CAND_STAGE lanes: a bitonic network of ten shuffle steps (the count
        // pass used to keep the slice sorted by insertion — a chain of dependent global loads per candidate)
        static_assert(CAND_STAGE == 16, "the sorting network below is written for 16 lanes per row");
        uint32_t v = j < cnt ? stage[t] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 2; k <= CAND_STAGE; k <<= 1) {
#pragma unroll
            for (int d = k >> 1; d > 0; d >>= 1) {
                const uint32_t w = __shfl_xor(v, d, CAND_STAGE);
                const bool keep_min = ((j & d) == 0) == ((j & k) == 0);
                v = keep_min ? (v < w ? v : w) : (v > w ? v : w);
            }
        }
        if (j < cnt) {
            cand_r[o0 + j] = v;
            cand_l[o0 + j] = (uint32_t)i;
        }
        return;
    }
    if (j != 0) return;
    int m = 0;
    const GridParams g = *ix.grid;
    for_each_bbox_candidate(ix, g, lbbox[i], [&](int r) {
        if (!dev::valid_row(right.validity, r)) return;
        int b = m - 1;  // insertion into the ascending prefix
        while (b >= 0 && cand_r[o0 + b] > (uint32_t)r) {
            cand_r[o0 + b + 1] = cand_r[o0 + b];
            --b;
        }
        cand_r[o0 + b + 1] = (uint32_t)r;
        ++m;
    });
    for (int a = 0; a < m; ++a) cand_l[o0 + a] = (uint32_t)i;
}

// Stage 2: exact refine, JOIN_GS lanes per candidate pair (pairs are independent: the unit of parallelism is the
// pair, not the row, so ragged candidate lists do not unbalance waves).
#ifndef GPK_JOIN_GS
#define GPK_JOIN_GS 16
#endif
constexpr int JOIN_GS = GPK_JOIN_GS;
__device__ __forceinline__ int64_t row_of_candidate(const int32_t* __restrict__ off, int64_t n_rows, int64_t c) {
    int64_t lo = 0, hi = n_rows;  // largest row with off[row] <= c
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)off[mid] <= c)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
// (GPK_REFINE_MINWAVES=4 — 128 registers instead of 141, four waves per SIMD instead of three — was 4 % faster, 4.06 -> 3.90 ms, and wrote
// 1.1 GB of spilled registers per launch to scratch memory, WRITE_SIZE 9.7 MB -> 1.14 GB: not taken)
#ifndef GPK_REFINE_KEEP_A
#define GPK_REFINE_KEEP_A 1
#endif
#ifndef GPK_REFINE_MINWAVES
#define GPK_REFINE_MINWAVES 1
#endif
__global__ __launch_bounds__(256, GPK_REFINE_MINWAVES) void pair_refine_kernel(DevGeo left, DevGeo right, const uint32_t* __restrict__ cand_l,
                                                           const uint32_t* __restrict__ cand_r, int64_t n_cand,
                                                           const double4* __restrict__ lbbox, const double4* __restrict__ rbbox,
                                                           uint8_t* __restrict__ hit, bool l_one_ring, bool r_one_ring) {
    // (l_one_ring / r_one_ring: every polygon of that POLYGON column is known to have exactly one ring — ring r is geometry r)
    // per group: the staging slice of the small-pair path, which doubles as the two in-window segment lists of the general one
    static_assert(sizeof(PairSmallLds) >= 2 * PP_LIST * sizeof(double4), "the general routine's lists fit the small-pair slice");
    __shared__ PairSmallLds slices[256 / JOIN_GS];
    const int lane = threadIdx.x & (JOIN_GS - 1);
    PairSmallLds* slice = slices + threadIdx.x / JOIN_GS;
    const bool plain = left.type == GPK_GEOM_POLYGON && right.type == GPK_GEOM_POLYGON && lbbox && rbbox;  // (uniform)
    // A group takes a CONTIGUOUS run of candidates: they are ordered by left row, so consecutive ones mostly share it and its ring stays
    // staged (round 6; a group used to stride over the list, staging both rings of every pair)
    const int64_t groups = (int64_t)gridDim.x * (256 / JOIN_GS);
    const int64_t per = (n_cand + groups - 1) / groups, g_id = (int64_t)blockIdx.x * (256 / JOIN_GS) + threadIdx.x / JOIN_GS;
    const int64_t c_lo = g_id * per, c_hi = c_lo + per < n_cand ? c_lo + per : n_cand;
    int64_t staged_i = -1;  // the left row whose ring is in slice->a
    for (int64_t c = c_lo; c < c_hi; ++c) {
        const int64_t i = (int64_t)cand_l[c], j = (int64_t)cand_r[c];
        bool h;
        bool small = false;
        int ca = 0, na = 0, cb = 0, nb = 0;
        if (plain) {  // two single-ring polygons of at most PP_SMALL coordinates: the staged path (gpk_polypoly.h)
            int ra0 = (int)i, ra1 = (int)i + 1, rb0 = (int)j, rb1 = (int)j + 1;
            if (!l_one_ring) {
                ra0 = left.geom_off[i];
                ra1 = left.geom_off[i + 1];
            }
            if (!r_one_ring) {
                rb0 = right.geom_off[j];
                rb1 = right.geom_off[j + 1];
            }
            if (ra1 - ra0 == 1 && rb1 - rb0 == 1) {
                ca = left.ring_off[ra0];
                na = left.ring_off[ra0 + 1] - ca;
                cb = right.ring_off[rb0];
                nb = right.ring_off[rb0 + 1] - cb;
                small = na >= 1 && nb >= 1 && na <= PP_SMALL && nb <= PP_SMALL;
            }
        }
        if (small) {
            h = polygon_pair_small<JOIN_GS>(left.xy + ca, na, right.xy + cb, nb, lbbox[i], rbbox[j], lane, slice, GPK_REFINE_KEEP_A && staged_i == i);
            staged_i = i;
        } else {
            h = polygonal_intersects_polygonal_group<JOIN_GS>(left, i, right, j, lane, reinterpret_cast<double4*>(slice), lbbox, rbbox);
            staged_i = -1;  // (the general routine keeps its segment lists in the slice)
        }
        if (lane == 0) hit[c] = h;
    }
}

// Contains<Polygon> for Polygon / MultiPolygon (spatial_index.rs:99-101,107-111; gpk_contains.h): the right polygon can only
// lie in a left geometry whose box holds its box, which settles most candidates of the (closed-overlap) candidate list.
__global__ __launch_bounds__(256) void pair_contains_kernel(DevGeo left, DevGeo right, const uint32_t* __restrict__ cand_l,
                                                             const uint32_t* __restrict__ cand_r, int64_t n_cand,
                                                             const double4* __restrict__ lbbox, const double4* __restrict__ rbbox,
                                                             uint8_t* __restrict__ hit) {
    const int lane = threadIdx.x & (JOIN_GS - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / JOIN_GS);
    for (int64_t c = (int64_t)blockIdx.x * (256 / JOIN_GS) + threadIdx.x / JOIN_GS; c < n_cand; c += groups) {
        const int64_t i = (int64_t)cand_l[c], j = (int64_t)cand_r[c];
        const double4 lb = lbbox[i], rb = rbbox[j];
        bool h = false;
        if (rb.x >= lb.x && rb.y >= lb.y && rb.z <= lb.z && rb.w <= lb.w) h = cont::polygonal_contains_polygonal_group<JOIN_GS>(left, i, right, j, lane);
        if (lane == 0) hit[c] = h;
    }
}

__global__ __launch_bounds__(256) void lineal_point_refine_kernel(DevGeo left, DevGeo right, const uint32_t* __restrict__ cand_l,
                                                                   const uint32_t* __restrict__ cand_r, int64_t n_cand,
                                                                   uint8_t* __restrict__ hit) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cand) return;
    const int64_t i = cand_l[c], j = cand_r[c];
    const bool point_left = left.type == GPK_GEOM_POINT;
    const double2 p = point_left ? left.xy[i] : right.xy[j];
    bool h = false;
    if (p.x == p.x && p.y == p.y) h = point_left ? lineal_contains_point(right, j, p.x, p.y) : lineal_contains_point(left, i, p.x, p.y);
    hit[c] = h;
}

// Stage 3: per-row hit counts, then (after a scan) the (l, r) pairs in candidate order == sorted by (l, r).
template <bool WRITE>
__global__ __launch_bounds__(256) void pair_emit_kernel(int64_t n_rows, const int32_t* __restrict__ cand_off,
                                                         const uint32_t* __restrict__ cand_r, const uint8_t* __restrict__ hit,
                                                         int32_t* __restrict__ counts, const int32_t* __restrict__ offsets,
                                                         uint32_t left_base, uint2* __restrict__ pairs, int64_t capacity) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    int cnt = 0;
    const int64_t o0 = WRITE ? (int64_t)offsets[i] : 0;
    for (int c = cand_off[i]; c < cand_off[i + 1]; ++c) {
        if (!hit[c]) continue;
        if (WRITE && o0 + cnt < capacity) pairs[o0 + cnt] = make_uint2(left_base + (uint32_t)i, cand_r[c]);
        ++cnt;
    }
    if (!WRITE) counts[i] = cnt;
}

__global__ void i32_to_u32_copy_kernel(const int32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}

// ================================= host drivers ================================================
static inline dim3 grid_for(int64_t n, int block) {
    int64_t b = (n + block - 1) / block;
    return dim3((unsigned)(b > 0 ? b : 1));
}

// polygonal x polygonal: candidates (count, scan, fill) -> pair-parallel exact refine -> hits (count, scan, emit)
enum { REFINE_POLYGONAL = 0, REFINE_LINEAL_POINT = 1, REFINE_CONTAINS = 2, REFINE_ENVELOPE_INTERSECTS = 3, REFINE_ENVELOPE_CONTAINED = 4 };
// gpk_index_query_envelope's refine (rstar's locate_in_envelope_intersecting / locate_in_envelope, spatial_index.rs:385-387,424-426): a
// candidate's box already meets the query box (closed intervals: for_each_bbox_candidate); `contained` additionally asks that it lies
// inside it, bounds included (rstar AABB::contains_envelope)
__global__ __launch_bounds__(256) void query_boxes_kernel(const double4* __restrict__ in, int64_t n, double4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double4 b = in[i];
    const bool nan = !(b.x == b.x && b.y == b.y && b.z == b.z && b.w == b.w);
    out[i] = nan ? make_double4(NAN, NAN, NAN, NAN) : make_double4(fmin(b.x, b.z), fmin(b.y, b.w), fmax(b.x, b.z), fmax(b.y, b.w));
}
__global__ __launch_bounds__(256) void envelope_refine_kernel(const uint32_t* __restrict__ cand_l, const uint32_t* __restrict__ cand_r, int64_t n,
                                                               const double4* __restrict__ lbbox, const double4* __restrict__ rbbox, int contained,
                                                               uint8_t* __restrict__ hit) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double4 lb = lbbox[cand_l[i]], rb = rbbox[cand_r[i]];
    bool ok = rb.x == rb.x && lb.x == lb.x;
    if (contained) ok = ok && rb.x >= lb.x && rb.z <= lb.z && rb.y >= lb.y && rb.w <= lb.w;
    hit[i] = ok ? 1 : 0;
}
static int32_t bbox_join(const gpk_geoarray* left, const gpk_geoarray* right, const gpk_index* right_index, uint32_t left_row_base,
                         uint32_t* out_counts, uint32_t* out_pairs, int64_t pair_capacity, int64_t* n_pairs, int32_t out_space,
                         hipStream_t s, int refine = REFINE_POLYGONAL, const double4* given_lbbox = nullptr) {
    const int64_t n = left->d.n_geoms;
    const bool host_out = out_space != GPK_MEM_DEVICE;
    const bool want_pairs = pair_capacity > 0;
    // per-call allocations that outlive a workspace reset (gpk_bounds uses the workspace itself)
    void* owned[4] = {nullptr, nullptr, nullptr, nullptr};
    auto done = [&](int32_t rc) {
        for (void* p : owned)
            if (p) (void)hipFree(p);
        return rc;
    };
    // left boxes and the candidate buffers live in the thread's auxiliary arenas (no hipMalloc / hipFree per call)
    // (given_lbbox: gpk_index_query_envelope hands its query boxes over — in device memory — in the left boxes' place)
    double4* lbbox = const_cast<double4*>(given_lbbox);
    int32_t rc = GPK_OK;
    if (!given_lbbox) {
        GPK_TRY(workspace_aux(0).begin(sizeof(double4) * (size_t)n + 256));
        lbbox = (double4*)workspace_aux(0).take(sizeof(double4) * (size_t)n);
        rc = gpk_bounds(left, (double*)lbbox, GPK_MEM_DEVICE, (void*)s);
        if (rc != GPK_OK) return done(rc);
    }
    const int64_t nb = (n + 255) / 256;
    const size_t pairs_bytes = sizeof(uint32_t) * 2 * (size_t)pair_capacity;
    const size_t i32n = align256(sizeof(int32_t) * (size_t)(n + 1));
    size_t need = 4 * i32n + align256(sizeof(unsigned long long) * (size_t)(nb + 2)) + 256 + 1024;
    if (host_out && out_counts) need += align256(sizeof(uint32_t) * (size_t)n);
    if (host_out && want_pairs) need += align256(pairs_bytes);
    // (padded staging of the candidates: up to 512 MB — 8M left rows; beyond that the two-search path)
    const size_t stage_bytes = sizeof(uint32_t) * CAND_STAGE * (size_t)(n > 0 ? n : 1);
    static const bool no_stage = getenv("GPK_NO_CAND_STAGE") != nullptr;  // A/B runs
    const bool staged = !no_stage && stage_bytes <= (size_t(512) << 20);
    if (staged) need += align256(stage_bytes);
    rc = workspace().begin(need);
    if (rc != GPK_OK) return done(rc);
    int32_t* cand_cnt = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* cand_off = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* counts = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* offsets = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n + 1));
    unsigned long long* btot = (unsigned long long*)workspace().take(sizeof(unsigned long long) * (size_t)(nb + 2));
    int32_t* big_rows = (int32_t*)workspace().take(256);
    uint32_t* counts_out = out_counts ? (host_out ? (uint32_t*)workspace().take(sizeof(uint32_t) * (size_t)n) : out_counts) : nullptr;
    uint32_t* pairs_dev = want_pairs ? (host_out ? (uint32_t*)workspace().take(pairs_bytes) : out_pairs) : nullptr;
    uint32_t* stage = staged ? (uint32_t*)workspace().take(stage_bytes) : nullptr;

    int32_t n_cand = 0, has_big_rows = 0;
    unsigned long long cand_total = 0;  // the 64-bit grand total of the scan: cand_off[n] is its truncation to i32
    auto stage1 = [&]() -> int32_t {
        GPK_HIP(hipMemsetAsync(big_rows, 0, 2 * sizeof(int32_t), s));
        if (staged)
            GPK_LAUNCH("gpk_bbox_cand_count", bbox_cand_stage_kernel, dim3((unsigned)((n * CAND_LANES + 255) / 256)), dim3(256), 0, s, left->d, right->d,
                       right_index->v, lbbox, cand_cnt, stage, big_rows);
        else
            GPK_LAUNCH("gpk_bbox_cand_count", bbox_cand_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, left->d, right->d, right_index->v,
                       lbbox, cand_cnt, (const int32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, big_rows);
        GPK_TRY(exclusive_scan_i32(cand_cnt, n, cand_off, nullptr, btot, s));
        int32_t fl[2] = {0, 0};
        GPK_HIP(hipMemcpyAsync(&cand_total, btot + nb, sizeof cand_total, hipMemcpyDeviceToHost, s));
        GPK_HIP(hipMemcpyAsync(fl, big_rows, sizeof fl, hipMemcpyDeviceToHost, s));
        GPK_HIP(hipStreamSynchronize(s));
        has_big_rows = fl[0];
        return GPK_OK;
    };
    rc = stage1();
    if (rc != GPK_OK) return done(rc);
    // candidate offsets are i32 (one slice per left row): more than 2^31 - 1 bbox candidates cannot be addressed
    if (cand_total > (unsigned long long)INT32_MAX)
        return done(fail(GPK_ERR_CAPACITY, "spatial_join: %llu bbox candidates exceed the i32 candidate offsets: shard the left side", cand_total));
    n_cand = (int32_t)cand_total;
    uint32_t *cand_r = nullptr, *cand_l = nullptr, *cand_sorted = nullptr;
    uint8_t* hit = nullptr;
    void* seg_tmp = nullptr;
    size_t seg_bytes = 0;
    unsigned seg_bits = 1;
    while (seg_bits < 32 && ((int64_t)1 << seg_bits) < right->d.n_geoms) ++seg_bits;
    {
        const size_t nc1 = (size_t)(n_cand > 0 ? n_cand : 1);
        if (has_big_rows) {
            const hipError_t qe = rocprim::segmented_radix_sort_keys(nullptr, seg_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned)n_cand,
                                                                     (unsigned)n, (const int32_t*)cand_off, (const int32_t*)cand_off + 1, 0, seg_bits, s);
            if (qe != hipSuccess) return done(fail(GPK_ERR_DEVICE, "spatial_join: %s", hipGetErrorString(qe)));
        }
        rc = workspace_aux(1).begin((has_big_rows ? 3 : 2) * align256(sizeof(uint32_t) * nc1) + align256(nc1) + align256(seg_bytes) + 512);
        if (rc != GPK_OK) return done(rc);
        cand_r = (uint32_t*)workspace_aux(1).take(sizeof(uint32_t) * nc1);
        cand_l = (uint32_t*)workspace_aux(1).take(sizeof(uint32_t) * nc1);
        hit = (uint8_t*)workspace_aux(1).take(nc1);
        if (has_big_rows) {
            cand_sorted = (uint32_t*)workspace_aux(1).take(sizeof(uint32_t) * nc1);
            seg_tmp = workspace_aux(1).take(seg_bytes ? seg_bytes : 1);
        }
    }
    auto stage23 = [&]() -> int32_t {
        if (staged && !has_big_rows)
            GPK_LAUNCH("gpk_cand_compact", cand_compact_kernel, dim3((unsigned)((n * CAND_STAGE + 255) / 256)), dim3(256), 0, s, left->d, right->d,
                       right_index->v, (const double4*)lbbox, (const int32_t*)cand_cnt, (const int32_t*)cand_off, (const uint32_t*)stage, cand_r, cand_l);
        else
            GPK_LAUNCH("gpk_bbox_cand_fill", bbox_cand_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, left->d, right->d, right_index->v,
                       lbbox, cand_cnt, (const int32_t*)cand_off, cand_r, cand_l, big_rows);
        if (has_big_rows && n_cand > 0) {  // some slice is long: sort every slice by right id, segment = left row
            GPK_HIP(rocprim::segmented_radix_sort_keys(seg_tmp, seg_bytes, (const uint32_t*)cand_r, cand_sorted, (unsigned)n_cand, (unsigned)n,
                                                       (const int32_t*)cand_off, (const int32_t*)cand_off + 1, 0, seg_bits, s));
            cand_r = cand_sorted;
        }
        if (n_cand > 0) {
            int64_t blocks = ((int64_t)n_cand + (256 / JOIN_GS) - 1) / (256 / JOIN_GS);
            const int64_t cap = (int64_t)cu_count() * 64;
            if (blocks > cap) blocks = cap;
            if (refine == REFINE_ENVELOPE_INTERSECTS || refine == REFINE_ENVELOPE_CONTAINED)
                GPK_LAUNCH("gpk_envelope_refine", envelope_refine_kernel, dim3((unsigned)(((int64_t)n_cand + 255) / 256)), dim3(256), 0, s,
                           (const uint32_t*)cand_l, (const uint32_t*)cand_r, (int64_t)n_cand, (const double4*)lbbox, right_index->v.bbox,
                           refine == REFINE_ENVELOPE_CONTAINED ? 1 : 0, hit);
            else if (refine == REFINE_LINEAL_POINT)
                GPK_LAUNCH("gpk_lineal_point_refine", lineal_point_refine_kernel, dim3((unsigned)(((int64_t)n_cand + 255) / 256)), dim3(256), 0, s,
                           left->d, right->d, (const uint32_t*)cand_l, (const uint32_t*)cand_r, (int64_t)n_cand, hit);
            else if (refine == REFINE_CONTAINS)
                GPK_LAUNCH("gpk_pair_contains", pair_contains_kernel, dim3((unsigned)blocks), dim3(256), 0, s, left->d, right->d,
                           (const uint32_t*)cand_l, (const uint32_t*)cand_r, (int64_t)n_cand, (const double4*)lbbox,
                           right_index->v.bbox, hit);
            else
                GPK_LAUNCH("gpk_pair_refine", pair_refine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, left->d, right->d,
                           (const uint32_t*)cand_l, (const uint32_t*)cand_r, (int64_t)n_cand, (const double4*)lbbox,
                           right_index->v.bbox, hit, left->d.type == GPK_GEOM_POLYGON && left->classes && left->classes->one_to_one,
                           right->d.type == GPK_GEOM_POLYGON && right->classes && right->classes->one_to_one);
        }
        GPK_LAUNCH("gpk_pair_count", pair_emit_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, n, (const int32_t*)cand_off,
                   (const uint32_t*)cand_r, (const uint8_t*)hit, counts, (const int32_t*)nullptr, left_row_base, (uint2*)nullptr, (int64_t)0);
        GPK_TRY(exclusive_scan_i32(counts, n, offsets, nullptr, btot, s));
        if (counts_out)
            GPK_LAUNCH("gpk_counts_copy", i32_to_u32_copy_kernel, dim3((unsigned)nb), dim3(256), 0, s, counts, counts_out, n);
        if (want_pairs)
            GPK_LAUNCH("gpk_pair_emit", pair_emit_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, n, (const int32_t*)cand_off,
                       (const uint32_t*)cand_r, (const uint8_t*)hit, counts, (const int32_t*)offsets, left_row_base, (uint2*)pairs_dev,
                       pair_capacity);
        return GPK_OK;
    };
    rc = stage23();
    if (rc != GPK_OK) return done(rc);
    unsigned long long total = 0;  // 64-bit grand total of the hit scan (hits <= candidates <= INT32_MAX, checked above)
    hipError_t e = hipMemcpyAsync(&total, btot + nb, sizeof total, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "spatial_join: %s", hipGetErrorString(e)));
    *n_pairs = (int64_t)total;
    if (host_out) {
        if (out_counts) {
            rc = copy_out(out_counts, out_space, counts_out, sizeof(uint32_t) * (size_t)n, s);
            if (rc != GPK_OK) return done(rc);
        }
        if (want_pairs) {
            const int64_t w = (int64_t)total < pair_capacity ? (int64_t)total : pair_capacity;
            rc = copy_out(out_pairs, out_space, pairs_dev, sizeof(uint32_t) * 2 * (size_t)w, s);
            if (rc != GPK_OK) return done(rc);
        }
    }
    if (want_pairs && (int64_t)total > pair_capacity)
        return done(fail(GPK_ERR_CAPACITY, "spatial_join: %lld pairs but capacity %lld", (long long)total, (long long)pair_capacity));
    return done(GPK_OK);
}

// The epoch words of the fused point joins: one buffer per device (zeroed when created; a launch's tag is never 0), a process-wide
// launch counter, and an event that keeps launches from DIFFERENT streams apart — they share the words, and two such launches
// running side by side could each hold compute units the other's lower-numbered work-groups still wait for.
// Layout: n total words (one per work-group) | the `lost` word | the ticket counter (it only ever grows: the host knows where a
// launch's numbers begin).
static std::mutex g_fused_mu;
struct FusedDev {
    unsigned long long* slots = nullptr;
    unsigned long long tickets = 0;       // what the ticket counter holds once every launch queued so far has started its work-groups
    unsigned long long era = 0;           // launch counter >> 24 when the words were last cleared
    int n = 0;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool any = false;
};
static FusedDev g_fused[16];
static unsigned long long g_fused_epoch = 0;
struct FusedWords {
    unsigned long long *slots, *ticket, *lost;
    unsigned long long epoch, ticket_base;
};
// n_words: epoch words the launch needs; wgs: tickets it will draw
static int32_t fused_launch_begin(hipStream_t s, int64_t n_words, int wgs, FusedWords* out) {
    int dev = 0;
    GPK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail(GPK_ERR_DEVICE, "spatial_join: device %d out of range", dev);
    if (n_words > (int64_t)0x7FFFFFF0ll) return fail(GPK_ERR_INVALID_ARGUMENT, "spatial_join: %lld epoch words", (long long)n_words);
    g_fused_mu.lock();  // released by fused_launch_end: launch order = event order
    FusedDev& f = g_fused[dev];
    auto bail = [&](hipError_t e) {
        g_fused_mu.unlock();
        return fail(GPK_ERR_DEVICE, "spatial_join: %s", hipGetErrorString(e));
    };
    if ((int64_t)f.n < n_words) {
        if (f.slots) {
            (void)hipDeviceSynchronize();
            (void)hipFree(f.slots);
            f.slots = nullptr;
        }
        const int want = n_words < 4096 ? 4096 : (int)n_words;
        const size_t bytes = sizeof(unsigned long long) * (size_t)(want + 2);
        hipError_t e = device_malloc((void**)&f.slots, bytes);
        if (e != hipSuccess) return bail(e);
        e = hipMemset(f.slots, 0, bytes);
        if (e != hipSuccess) return bail(e);
        f.n = want;
        f.tickets = 0;
        f.era = g_fused_epoch >> (64 - FUSED_TOTAL_BITS);
    }
    if (!f.done) {
        const hipError_t e = hipEventCreateWithFlags(&f.done, hipEventDisableTiming);
        if (e != hipSuccess) return bail(e);
    }
    if (f.any && f.last != s) {  // another stream launched last: this launch starts after everything queued there so far
        hipError_t e = hipEventRecord(f.done, f.last);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, f.done, 0);
        if (e != hipSuccess) {  // (that stream is gone: whatever it held has run or is drained here)
            (void)hipGetLastError();
            e = hipDeviceSynchronize();
            if (e != hipSuccess) return bail(e);
        }
    }
    ++g_fused_epoch;
    if ((g_fused_epoch & ((1ull << (64 - FUSED_TOTAL_BITS)) - 1ull)) == 0ull) ++g_fused_epoch;  // (a tag of 0 is what a fresh word holds)
    // the tag is 24 bits of the launch counter: when it has wrapped since this device's words were last cleared, a word written 16.7 M
    // launches ago by a larger launch could carry the new launch's tag — clear them (stream-ordered, behind every earlier launch)
    if ((g_fused_epoch >> (64 - FUSED_TOTAL_BITS)) != f.era) {
        const hipError_t e = hipMemsetAsync(f.slots, 0, sizeof(unsigned long long) * (size_t)(f.n + 1), s);  // (+ the `lost` word)
        if (e != hipSuccess) return bail(e);
        f.era = g_fused_epoch >> (64 - FUSED_TOTAL_BITS);
    }
    out->epoch = g_fused_epoch & ((1ull << (64 - FUSED_TOTAL_BITS)) - 1ull);
    out->slots = f.slots;
    out->lost = f.slots + f.n;
    out->ticket = f.slots + f.n + 1;
    out->ticket_base = f.tickets;
    f.tickets += (unsigned long long)wgs;
    return GPK_OK;
}
static void fused_launch_end(hipStream_t s, bool launched) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16) {
        g_fused[dev].last = s;
        g_fused[dev].any = true;
        if (!launched) g_fused[dev].n = 0;  // the counters and the host's idea of them may differ now: fresh words for the next launch
    }
    g_fused_mu.unlock();
}

// Enqueues the point x polygonal join on `s` (memset of the totals, pip_tile, pip_write) and returns without waiting.
// Device scratch comes from the calling thread's workspace; *total_out (device or device-mapped, may be NULL) receives
// the number of hits when the stream gets there.
static int32_t pip_join_enqueue(const gpk_geoarray* left, const gpk_geoarray* right, const gpk_index* right_index, uint32_t left_row_base,
                                uint32_t* out_counts, uint32_t* out_pairs, int64_t pair_capacity, bool host_out,
                                unsigned long long* total_out, hipStream_t s, uint32_t** counts_dev_out, uint32_t** pairs_dev_out,
                                unsigned long long** grand_out) {
    const int64_t n = left->d.n_geoms;
    static const bool no_lean = [] {  // GPK_NO_LEAN=1: A/B runs of the general tile kernel on a lean-eligible index
        const char* e = getenv("GPK_NO_LEAN");
        return e && *e && *e != '0';
    }();
    const bool lean = right_index->pip.R > 0 && right_index->pip_lean && !no_lean;
    // an index with chains (gpk_index.h) is served by the chain kernel or — with an LDS routing image — by the one-launch join of
    // gpk_pipflow.hip: its records carry chain words where the queue kernels expect slab ranges.
    // GPK_TILE_KERNEL=chain: A/B runs of the chain kernel + writer on an index that has the image
    static const bool no_flow = [] {
        const char* e = getenv("GPK_TILE_KERNEL");
        return e && !strcmp(e, "chain");
    }();
    // slabs kept as coordinate indices read the coordinates of the array THIS call names (an index built over another upload of the
    // same column — SpatialIndex(series) next to the series itself — stays valid after that upload is gone)
    PipView pvj = right_index->pip;
    pvj.slab_xy = right->d.xy;
    const bool chain = right_index->pip.R > 0 && (GPK_HALF_CHAINS ? right_index->pip.chain_xy != nullptr : right_index->pip.sub_aux != nullptr);
    const bool one_per_lane = !chain && !lean && right_index->pip.R > 0 && right_index->pip_list_heavy;  // (pip_tile_kernel<., ., 1>)
    // `flow` (gpk_pipflow.hip, round 6): ONE launch — optimistic hits in a global pool, dense exact passes, dense emission, tiles of
    // 64 .. 512 rows by the column's length — for every chain index with a routing image, up to 4 M geometries and 201 M left rows
    const int flow_p = GPK_HALF_CHAINS && chain && !no_flow && right_index->pip.route != nullptr && right_index->pip.R <= PIP_ROUTE_RMAX
                           ? pip_flow_points_per_lane(n, right->d.n_geoms, right_index->pip.R, cu_count())
                           : 0;
    const bool flow = flow_p > 0;
    const int tile_points = flow ? 64 * flow_p : (chain ? 64 * CHAIN_PPT : (lean ? LEAN_TILE : (one_per_lane ? PIP_BLOCK : PIP_TILE)));
    const int64_t n_blocks = (n + tile_points - 1) / tile_points;
    const bool want_pairs = pair_capacity > 0;
    const size_t counts_bytes = sizeof(uint32_t) * (size_t)n;
    const size_t pairs_bytes = sizeof(uint32_t) * 2 * (size_t)pair_capacity;
    const int64_t n_super = (n_blocks >> PIP_SUPER_SHIFT) + 1;
    const int64_t n_wblocks = (n + PIP_WTILE - 1) / PIP_WTILE;
    // words in the multi-hit pool (a chain launch has no multi-hit rows in it: a rare row with several hits is CODE_MULTI)
    const uint32_t multi_cap = chain ? 1024u : (uint32_t)(n < (int64_t)0x18000000 ? 2 * n + 1024 : (int64_t)0x30000000);
    int64_t flow_wgs = (int64_t)cu_count();  // persistent work-groups, one per CU (the routing image takes most of a CU's LDS)
    {
        const int64_t want = (n_blocks + 15) / 16;  // (16 waves a work-group, a tile a wave)
        if (flow_wgs > want) flow_wgs = want;
        if (flow_wgs < 1) flow_wgs = 1;
    }
    const bool fused = flow;  // (no result codes, no tile totals, no writer launch)
    const size_t stage_bytes = flow ? pip_flow_pool_bytes(n) : 0;
    size_t need = align256(fused ? 64 : counts_bytes + 64) /*code*/ + align256(sizeof(unsigned long long) * (size_t)(n_blocks + n_super + 3)) +
                  align256(sizeof(uint32_t) * (size_t)multi_cap) + align256(sizeof(ChainCold)) + align256(stage_bytes) + 1024;
    if (host_out && out_counts) need += align256(counts_bytes);
    if (host_out && want_pairs) need += align256(pairs_bytes);
    // scratch of the stream-ordered join: one arena per (calling thread, stream), so that joins a thread enqueues on
    // DIFFERENT streams never share code / totals buffers (they used to: the header asked callers not to)
    Workspace& ws = workspace_for_stream(s);
    int32_t rc = ws.begin(need);
    if (rc != GPK_OK) return rc;
    uint32_t* code = (uint32_t*)ws.take(fused ? 64 : counts_bytes + 64);
    unsigned long long* btot = (unsigned long long*)ws.take(sizeof(unsigned long long) * (size_t)(n_blocks + n_super + 3));
    unsigned long long* stot = btot + n_blocks;     // n_super super-tile totals
    unsigned long long* grand = stot + n_super;     // total hits
    uint32_t* multi_top = (uint32_t*)(grand + 1);   // words used in the multi-hit pool (zeroed with the totals)
    uint32_t* multi_pool = (uint32_t*)ws.take(sizeof(uint32_t) * (size_t)multi_cap);
    ChainCold* cold = (ChainCold*)ws.take(sizeof(ChainCold));
    uint2* stage = stage_bytes ? (uint2*)ws.take(stage_bytes) : nullptr;
    uint32_t* counts_dev = out_counts ? (host_out ? (uint32_t*)ws.take(counts_bytes) : out_counts) : nullptr;
    uint32_t* pairs_dev = want_pairs ? (host_out ? (uint32_t*)ws.take(pairs_bytes) : out_pairs) : nullptr;


#define J_LAUNCH(...)                          \
    do {                                       \
        auto _f = [&]() -> int32_t {           \
            GPK_LAUNCH(__VA_ARGS__);           \
            return GPK_OK;                     \
        };                                     \
        int32_t _rc = _f();                    \
        if (_rc != GPK_OK) return _rc;         \
    } while (0)

    unsigned long long* stats = join_stats_buffer();  // nullptr unless gpk_join_stats_enable(1)
    if (fused) {  // nothing to zero; the rare arm's arguments are written when they differ from what this arena holds
        struct {
            DevGeo polys;
            IndexView ix;
            uint64_t serial;
        } want_cold;
        memset(&want_cold, 0, sizeof want_cold);
        want_cold.polys = right->d;
        want_cold.ix = right_index->v;
        want_cold.serial = right_index->serial;
        if (!ws.tag_matches(cold, &want_cold, sizeof want_cold)) {
            J_LAUNCH("gpk_join_prep", join_prep_kernel, dim3(1), dim3(256), 0, s, stot, 0, cold, right->d, right_index->v);
            ws.set_tag(cold, &want_cold, sizeof want_cold);
        }
    } else if (chain) {  // totals zeroed and the rare arm's arguments written by one small launch
        J_LAUNCH("gpk_join_prep", join_prep_kernel, dim3(1), dim3(256), 0, s, stot, n_super + 2, cold, right->d, right_index->v);
    } else {
        const hipError_t me = hipMemsetAsync(stot, 0, sizeof(unsigned long long) * (size_t)(n_super + 2), s);  // (+ grand, multi_top)
        if (me != hipSuccess) return fail(GPK_ERR_DEVICE, "spatial_join: %s", hipGetErrorString(me));
    }
    ChainHot hot;
    memset(&hot, 0, sizeof hot);
    if (chain) {
        const PipView& pv = right_index->pip;
        hot.pts_xy = left->d.xy;
        hot.pts_validity = left->d.validity;
        hot.n_points = n;
        hot.n_tiles = n_blocks;
        hot.polys_validity = right->d.validity;
        hot.R = pv.R;
        hot.logR = 0;
        while ((1 << hot.logR) < pv.R) ++hot.logR;
        hot.rx0 = pv.rx0;
        hot.ry0 = pv.ry0;
        hot.inv_fw = pv.inv_fw;
        hot.inv_fh = pv.inv_fh;
        hot.cell = pv.cell;
        hot.half = reinterpret_cast<const HalfCell*>(pv.sub);  // (an index with chains keeps its one-part records in half-cell form)
        hot.sub_aux = pv.sub_aux;
        hot.chain_head = pv.chain_head;
        hot.chain_ext = pv.chain_ext;
        hot.chain_xy = pv.chain_xy;
        hot.part_geom = pv.part_geom;
        hot.route = pv.route;
        hot.counts = counts_dev;
        hot.code = code;
        hot.block_tot = btot;
        hot.super_tot = stot;
        hot.stats = stats;
        hot.cold = cold;
        hot.stage = nullptr;
        hot.pool = reinterpret_cast<uint32_t*>(stage);
        hot.n_full_tiles = left->d.validity ? 0 : (int32_t)(n / tile_points);  // (tiles that need no guards: whole tiles of a column without a validity bitmap)
        hot.inv_fw_s = pv.inv_fw * (double)PIP_SUB;
        hot.inv_fh_s = pv.inv_fh * (double)PIP_SUB;
        hot.sub_max = (double)(((uint32_t)PIP_SUB << hot.logR) - 1u);
    }
    if (flow) {  // one launch: persistent work-groups (one per CU) draw tiles, rank optimistic hits, and write the pairs themselves
        const int64_t wgs = flow_wgs;
        FusedTail tail;
        memset(&tail, 0, sizeof tail);
        tail.pairs = (uint2*)pairs_dev;
        tail.capacity = pair_capacity;
        tail.grand = grand;
        tail.grand_host = total_out;
        tail.left_base = left_row_base;
        FusedWords fw;
        int32_t frc = fused_launch_begin(s, wgs, (int)wgs, &fw);
        if (frc != GPK_OK) return frc;
        tail.slots = fw.slots;
        tail.epoch = fw.epoch;
        tail.ticket = fw.ticket;
        tail.ticket_base = fw.ticket_base;
        tail.lost = fw.lost;
        frc = launch_pip_flow(hot, tail, (int)wgs, flow_p, s);
        fused_launch_end(s, frc == GPK_OK);
        if (frc != GPK_OK) return frc;
        *counts_dev_out = counts_dev;
        *pairs_dev_out = pairs_dev;
        *grand_out = grand;
        return GPK_OK;
    }
    if (chain)
        J_LAUNCH("gpk_pip_tile", pip_tile_chain_kernel, dim3((unsigned)((n_blocks + PIP_BLOCK / 64 - 1) / (PIP_BLOCK / 64))), dim3(PIP_BLOCK), 0, s, hot);
    else if (lean)
        J_LAUNCH("gpk_pip_tile", pip_tile_lean_kernel, dim3((unsigned)n_blocks), dim3(PIP_BLOCK), 0, s, left->d, right->d, right_index->v, pvj,
                 counts_dev, code, btot, stot, stats);
    else if (right_index->pip.R > 0)
        if (right_index->pip.sub2 && one_per_lane)
            J_LAUNCH("gpk_pip_tile", (pip_tile_kernel<true, true, 1>), dim3((unsigned)n_blocks), dim3(PIP_BLOCK), 0, s, left->d, right->d,
                     right_index->v, pvj, counts_dev, code, btot, stot, multi_pool, multi_cap, multi_top, stats);
        else if (right_index->pip.sub2)
            J_LAUNCH("gpk_pip_tile", (pip_tile_kernel<true, true>), dim3((unsigned)n_blocks), dim3(PIP_BLOCK), 0, s, left->d, right->d,
                     right_index->v, pvj, counts_dev, code, btot, stot, multi_pool, multi_cap, multi_top, stats);
        else if (one_per_lane)
            J_LAUNCH("gpk_pip_tile", (pip_tile_kernel<true, false, 1>), dim3((unsigned)n_blocks), dim3(PIP_BLOCK), 0, s, left->d, right->d,
                     right_index->v, pvj, counts_dev, code, btot, stot, multi_pool, multi_cap, multi_top, stats);
        else
            J_LAUNCH("gpk_pip_tile", (pip_tile_kernel<true, false>), dim3((unsigned)n_blocks), dim3(PIP_BLOCK), 0, s, left->d, right->d,
                     right_index->v, pvj, counts_dev, code, btot, stot, multi_pool, multi_cap, multi_top, stats);
    else
        J_LAUNCH("gpk_pip_tile_generic", (pip_tile_kernel<false, false>), dim3((unsigned)n_blocks), dim3(PIP_BLOCK), 0, s, left->d, right->d,
                 right_index->v, pvj, counts_dev, code, btot, stot, multi_pool, multi_cap, multi_top, stats);
    // the writer also produces the grand total; in count-only mode it runs without a pair buffer
    J_LAUNCH("gpk_pip_write", pip_write_kernel, dim3((unsigned)n_wblocks), dim3(WR_BLOCK), 0, s, left->d, right->d, right_index->v,
             code, btot, stot, (const uint32_t*)multi_pool, n_blocks, tile_points, left_row_base, (uint2*)pairs_dev, pair_capacity, grand, total_out);
#undef J_LAUNCH

    *counts_dev_out = counts_dev;
    *pairs_dev_out = pairs_dev;
    *grand_out = grand;
    return GPK_OK;
}

// ---- polygonal LEFT x point RIGHT (spatial_index.rs:92,96: `poly.contains(point)` whichever side the polygon is on) -----
// The same join with the roles swapped — the polygons get the index, the points stream through pip_tile — followed
// by a transpose: the (point, polygon) pairs come out sorted by point, the caller wants (l = polygon, r = point) sorted
// by (l, r): a radix sort of 64-bit keys.
__global__ __launch_bounds__(256) void swap_keys_kernel(const uint2* __restrict__ pairs, int64_t n, unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((unsigned long long)pairs[i].y << 32) | pairs[i].x;  // (polygon, point)
}
__global__ __launch_bounds__(256) void swap_emit_kernel(const unsigned long long* __restrict__ sorted, int64_t n, int64_t capacity,
                                                         uint32_t left_base, uint2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i < capacity) out[i] = make_uint2(left_base + (uint32_t)(sorted[i] >> 32), (uint32_t)sorted[i]);
}
__global__ __launch_bounds__(256) void swap_counts_kernel(const unsigned long long* __restrict__ sorted, int64_t n, int64_t n_left,
                                                           uint32_t* __restrict__ counts) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_left) return;
    auto lower = [&](unsigned long long key) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sorted[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    counts[g] = (uint32_t)(lower((unsigned long long)(g + 1) << 32) - lower((unsigned long long)g << 32));
}

static int32_t swapped_pip_join(const gpk_geoarray* polys, const gpk_geoarray* pts, uint32_t left_row_base, uint32_t* out_counts,
                                uint32_t* out_pairs, int64_t pair_capacity, int64_t* n_pairs, int32_t out_space, hipStream_t s) {
    const int64_t n_left = polys->d.n_geoms, n_pts = pts->d.n_geoms;
    const bool host_out = out_space != GPK_MEM_DEVICE;
    const bool want_pairs = pair_capacity > 0;
    *n_pairs = 0;
    if (n_left == 0) return GPK_OK;
    if (n_pts == 0) {  // nothing on the right: every polygon has zero hits
        if (out_counts) {
            if (host_out)
                memset(out_counts, 0, sizeof(uint32_t) * (size_t)n_left);
            else
                GPK_HIP(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * (size_t)n_left, s));
        }
        return GPK_OK;
    }
    gpk_index* lix = nullptr;
    GPK_TRY(gpk_index_build(polys, (void*)s, &lix));
    auto done = [&](int32_t rc) {
        gpk_index_free(lix);
        return rc;
    };
    int64_t total = 0;
    int32_t rc = gpk_spatial_join(pts, polys, lix, GPK_PRED_CONTAINS, 0, nullptr, nullptr, 0, &total, GPK_MEM_DEVICE, (void*)s);
    if (rc != GPK_OK) return done(rc);
    *n_pairs = total;
    // scratch in the thread's auxiliary arenas (the inner joins recycle the main workspace)
    const size_t t1 = (size_t)(total > 0 ? total : 1);
    size_t sort_bytes = 0;
    int bits = 33;
    while (bits < 64 && (1ll << (bits - 32)) < n_left) ++bits;
    GPK_HIP(rocprim::radix_sort_keys(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, t1, 0, bits, s));
    rc = workspace_aux(0).begin(align256(8 * t1) * 3 + align256(sort_bytes + 256) + align256(4 * (size_t)(n_left + 1)) +
                                (host_out && want_pairs ? align256(8 * (size_t)pair_capacity) : 0) + 1024);
    if (rc != GPK_OK) return done(rc);
    uint2* tmp_pairs = (uint2*)workspace_aux(0).take(8 * t1);
    unsigned long long* keys = (unsigned long long*)workspace_aux(0).take(8 * t1);
    unsigned long long* sorted = (unsigned long long*)workspace_aux(0).take(8 * t1);
    void* sort_tmp = workspace_aux(0).take(sort_bytes + 256);
    uint32_t* counts_dev = out_counts ? (host_out ? (uint32_t*)workspace_aux(0).take(4 * (size_t)(n_left + 1)) : out_counts) : nullptr;
    uint2* pairs_dev = want_pairs ? (host_out ? (uint2*)workspace_aux(0).take(8 * (size_t)pair_capacity) : (uint2*)out_pairs) : nullptr;
    auto run = [&]() -> int32_t {
        if (total > 0) {
            int64_t again = 0;
            GPK_TRY(gpk_spatial_join(pts, polys, lix, GPK_PRED_CONTAINS, 0, nullptr, (uint32_t*)tmp_pairs, total, &again, GPK_MEM_DEVICE, (void*)s));
            const dim3 g((unsigned)((total + 255) / 256));
            GPK_LAUNCH("gpk_swap_keys", swap_keys_kernel, g, dim3(256), 0, s, (const uint2*)tmp_pairs, total, keys);
            GPK_HIP(rocprim::radix_sort_keys(sort_tmp, sort_bytes, (const unsigned long long*)keys, sorted, (size_t)total, 0, bits, s));
            if (pairs_dev)
                GPK_LAUNCH("gpk_swap_emit", swap_emit_kernel, g, dim3(256), 0, s, (const unsigned long long*)sorted, total, pair_capacity, left_row_base,
                           pairs_dev);
        }
        if (counts_dev && n_left > 0)
            GPK_LAUNCH("gpk_swap_counts", swap_counts_kernel, dim3((unsigned)((n_left + 255) / 256)), dim3(256), 0, s,
                       (const unsigned long long*)sorted, total, n_left, counts_dev);
        if (host_out) {
            if (out_counts) GPK_TRY(copy_out(out_counts, out_space, counts_dev, 4 * (size_t)n_left, s));
            if (want_pairs) GPK_TRY(copy_out(out_pairs, out_space, pairs_dev, 8 * (size_t)(total < pair_capacity ? total : pair_capacity), s));
        }
        GPK_HIP(hipStreamSynchronize(s));
        return GPK_OK;
    };
    rc = run();
    if (rc != GPK_OK) return done(rc);
    if (want_pairs && total > pair_capacity)
        return done(fail(GPK_ERR_CAPACITY, "spatial_join: %lld pairs but capacity %lld", (long long)total, (long long)pair_capacity));
    return done(GPK_OK);
}

}  // namespace gpk

using namespace gpk;

extern "C" {

int32_t gpk_join_stats_enable(int32_t on) {
    if (on && !g_join_stats) {
        GPK_TRY(require_device());
        GPK_HIP(device_malloc((void**)&g_join_stats, JOIN_STATS_WORDS * sizeof(unsigned long long)));
        GPK_HIP(hipMemset(g_join_stats, 0, JOIN_STATS_WORDS * sizeof(unsigned long long)));
    }
    g_join_stats_on = on != 0;
    return GPK_OK;
}

int32_t gpk_join_stats(int64_t out[4], int32_t reset) {
    if (!out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!g_join_stats) return GPK_OK;
    GPK_HIP(hipDeviceSynchronize());
    unsigned long long h[4];
    GPK_HIP(hipMemcpy(h, g_join_stats, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out[i] = (int64_t)h[i];
    if (reset) GPK_HIP(hipMemset(g_join_stats, 0, sizeof h));
    return GPK_OK;
}

int32_t gpk_join_trace(unsigned long long* out, int64_t n_words) {  // the raw stage stamps of a GPK_TILE_TRACE build of gpk_pipflow.hip (zeros otherwise)
    if (!g_join_stats || !out) return GPK_ERR_INVALID_ARGUMENT;
    GPK_HIP(hipDeviceSynchronize());
    GPK_HIP(hipMemcpy(out, g_join_stats + 8, sizeof(unsigned long long) * (size_t)(n_words < (int64_t)(JOIN_STATS_WORDS - 8) ? n_words : (int64_t)(JOIN_STATS_WORDS - 8)), hipMemcpyDeviceToHost));
    GPK_HIP(hipMemset(g_join_stats, 0, JOIN_STATS_WORDS * sizeof(unsigned long long)));
    return GPK_OK;
}

int32_t gpk_index_query_envelope(const gpk_index* idx, const double* boxes4, int64_t n_boxes, int32_t mode, uint32_t* out_counts,
                                 uint32_t* out_pairs, int64_t pair_capacity, int64_t* n_pairs, int32_t space, void* stream) {
    if (!idx || !n_pairs || (n_boxes > 0 && !boxes4)) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mode != GPK_QUERY_CONTAINED && mode != GPK_QUERY_INTERSECTING) return fail(GPK_ERR_INVALID_ARGUMENT, "unknown envelope query mode %d", mode);
    if (n_boxes < 0 || pair_capacity < 0 || (pair_capacity > 0 && !out_pairs)) return fail(GPK_ERR_INVALID_ARGUMENT, "bad sizes");
    if (n_boxes > (int64_t)0x7FFFFFF0ll) return fail(GPK_ERR_INVALID_ARGUMENT, "more than 2^31 query boxes");
    *n_pairs = 0;
    GPK_TRY(require_device());
    if (n_boxes == 0) return GPK_OK;
    hipStream_t s = (hipStream_t)stream;
    // the queries take the left rows' place in the box join, the index's own array the right rows' (the index holds the leaves: no
    // geometry is read); null and empty rows of the indexed array have NaN boxes and are in no directory cell
    gpk_geoarray left, right;
    memset(&left, 0, sizeof left);
    memset(&right, 0, sizeof right);
    left.d.type = GPK_GEOM_POINT;
    left.d.n_geoms = n_boxes;
    left.device = right.device = idx->device;
    right.d.type = idx->geom_type;
    right.d.n_geoms = idx->n_geoms;
    // the queries in the library's own memory, corners ordered the way `AABB::from_corners` orders them (lower = the component-wise
    // minimum of the two corners, upper = the maximum)
    GPK_TRY(workspace_aux(0).begin(2 * (sizeof(double4) * (size_t)n_boxes + 256)));
    double4* boxes_dev = (double4*)workspace_aux(0).take(sizeof(double4) * (size_t)n_boxes);
    const double4* src = reinterpret_cast<const double4*>(boxes4);
    if (space != GPK_MEM_DEVICE) {
        double4* up = (double4*)workspace_aux(0).take(sizeof(double4) * (size_t)n_boxes);
        GPK_HIP(hipMemcpyAsync(up, boxes4, sizeof(double4) * (size_t)n_boxes, hipMemcpyHostToDevice, s));
        src = up;
    }
    GPK_LAUNCH("gpk_query_boxes", query_boxes_kernel, dim3((unsigned)((n_boxes + 255) / 256)), dim3(256), 0, s, src, n_boxes, boxes_dev);
    return bbox_join(&left, &right, idx, 0u, out_counts, out_pairs, pair_capacity, n_pairs, space, s,
                     mode == GPK_QUERY_CONTAINED ? REFINE_ENVELOPE_CONTAINED : REFINE_ENVELOPE_INTERSECTS, boxes_dev);
}

int32_t gpk_index_free(gpk_index* idx) {
    if (!idx) return GPK_OK;
    // (hipFree's implicit wait, once: a join enqueued against this index may still be running — on the device that OWNS the tables,
    // which need not be the calling thread's current one: the blocks go back to a process-wide cache tagged by device)
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != idx->device) (void)hipSetDevice(idx->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 24; ++i)
        if (idx->owned[i]) cached_free(idx->owned[i]);
    if (cur >= 0 && cur != idx->device) (void)hipSetDevice(cur);
    delete idx;
    return GPK_OK;
}

int32_t gpk_index_describe(const gpk_index* idx, int64_t out[8]) {
    if (!idx || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = idx->pip.R;
    out[1] = idx->pip_lean;
    out[2] = GPK_HALF_CHAINS ? idx->pip.chain_xy != nullptr : idx->pip.sub_aux != nullptr;
    out[3] = idx->pip.route != nullptr;
    out[4] = idx->pip_list_heavy;
    return GPK_OK;
}

int32_t gpk_index_nbytes(const gpk_index* idx, int64_t* out_bytes) {
    if (!idx || !out_bytes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_bytes = idx->nbytes;
    return GPK_OK;
}

int32_t gpk_index_build(const gpk_geoarray* a, void* stream, gpk_index** out) {
    return gpk_index_build_ex(a, GPK_INDEX_BBOX_GRID | GPK_INDEX_PIP, nullptr, stream, out);
}

int32_t gpk_index_build_ex(const gpk_geoarray* a, int32_t parts, const double* bbox4_dev, void* stream, gpk_index** out) {
    if (!a || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;

    gpk_index* ix = new gpk_index;
    memset(ix, 0, sizeof *ix);
    ix->device = a->device;
    ix->n_geoms = n;
    ix->geom_type = a->d.type;
    ix->n_coords = a->d.n_coords;
    ix->n_rings = a->d.n_rings;
    {
        static std::atomic<uint64_t> next_serial{1};
        ix->serial = next_serial.fetch_add(1);
    }

    // grid resolution: ~2 cells per geometry along each axis of a square layout
    int gdim = (int)ceil(2.0 * sqrt((double)(n > 0 ? n : 1)));
    if (gdim < 1) gdim = 1;
    if (gdim > 2048) gdim = 2048;
    const int64_t n_cells = (int64_t)gdim * gdim;

    auto cleanup = [&](int32_t rc) {
        gpk_index_free(ix);
        return rc;
    };
#define IX_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return cleanup(fail(_e == hipErrorOutOfMemory ? GPK_ERR_OOM : GPK_ERR_DEVICE,         \
                                "%s failed: %s", #expr, hipGetErrorString(_e)));                  \
    } while (0)
#define IX_TRY(expr)                             \
    do {                                         \
        int32_t _rc = (expr);                    \
        if (_rc != GPK_OK) return cleanup(_rc);  \
    } while (0)
// a failed launch must release the half-built index too (GPK_LAUNCH returns from the enclosing function)
#define IX_LAUNCH(...)                           \
    do {                                         \
        auto _f = [&]() -> int32_t {             \
            GPK_LAUNCH(__VA_ARGS__);             \
            return GPK_OK;                       \
        };                                       \
        int32_t _rc = _f();                      \
        if (_rc != GPK_OK) return cleanup(_rc);  \
    } while (0)

    const bool dbg_time = getenv("GPK_DEBUG_INDEX") != nullptr;  // wall time of the directory phases (the stream is drained per stamp)
    auto t_last = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (!dbg_time) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gpk] index build: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    double4* bbox = nullptr;
    GridParams* grid = nullptr;
    int32_t* cell_off = nullptr;
    IX_HIP(cached_malloc((void**)&bbox, sizeof(double4) * (size_t)(n > 0 ? n : 1)));
    ix->owned[0] = bbox;
    IX_HIP(cached_malloc((void**)&grid, sizeof(GridParams)));
    ix->owned[1] = grid;
    IX_HIP(cached_malloc((void**)&cell_off, sizeof(int32_t) * (size_t)(n_cells + 1)));
    ix->owned[2] = cell_off;

    // 1. bounding boxes (NodeEnvelope, spatial_index.rs:212-312) — or the caller's (the leaves another rank built and
    //    sent over xGMI: dist.all_gather_leaves)
    if (bbox4_dev)
        IX_HIP(hipMemcpyAsync(bbox, bbox4_dev, sizeof(double4) * (size_t)n, hipMemcpyDeviceToDevice, s));
    else
        IX_TRY(gpk_bounds(a, (double*)bbox, GPK_MEM_DEVICE, stream));

    stamp("boxes (gpk_bounds)");
    // 2. extent + grid parameters, all on device (two stages beyond a few thousand boxes: one work-group walked 5M of them in 4.7 ms)
    const int64_t n_blocks = (n_cells + 255) / 256;
    const int64_t ext_blocks = n > 65536 ? 1024 : 0;
    IX_TRY(workspace().begin(align256(sizeof(int32_t) * (size_t)(n_cells + 1)) * 2 +
                             align256(sizeof(unsigned long long) * (size_t)(n_blocks + 1)) + align256(sizeof(double4) * 1024) + 1024));
    int32_t* cell_cnt = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_cells + 1));
    int32_t* cursor = (int32_t*)workspace().take(sizeof(int32_t) * (size_t)(n_cells + 1));
    unsigned long long* btot = (unsigned long long*)workspace().take(sizeof(unsigned long long) * (size_t)(n_blocks + 1));
    double4* ext_part = (double4*)workspace().take(sizeof(double4) * 1024);
    if (ext_blocks) {
        IX_LAUNCH("gpk_index_extent_partial", extent_partial_kernel, dim3((unsigned)ext_blocks), dim3(256), 0, s, bbox, n, ext_part);
        IX_LAUNCH("gpk_index_extent", extent_kernel, dim3(1), dim3(1024), 0, s, (const double4*)ext_part, ext_blocks, gdim, gdim, grid);
    } else {
        IX_LAUNCH("gpk_index_extent", extent_kernel, dim3(1), dim3(1024), 0, s, bbox, n, gdim, gdim, grid);
    }

    stamp("extent");
    // 3. count, scan, fill, sort
    IX_HIP(hipMemsetAsync(cell_cnt, 0, sizeof(int32_t) * (size_t)(n_cells + 1), s));
    if (n > 0)
        IX_LAUNCH("gpk_index_count", grid_register_kernel<false>, grid_for(n, 256), dim3(256), 0, s, bbox, n, grid, cell_cnt, (int32_t*)nullptr);
    IX_TRY(exclusive_scan_i32(cell_cnt, n_cells, cell_off, cursor, btot, s));
    unsigned long long total = 0;
    IX_HIP(d2h_small(&total, btot + n_blocks, sizeof total, s));
    IX_HIP(d2h_small(&ix->host_grid, grid, sizeof(GridParams), s));
    IX_HIP(sync_small(s));
    if (total > (unsigned long long)INT32_MAX)
        return cleanup(fail(GPK_ERR_INVALID_OFFSETS, "spatial index directory overflows i32 (%llu entries)", total));
    int32_t* items = nullptr;
    IX_HIP(cached_malloc((void**)&items, sizeof(int32_t) * (size_t)(total > 0 ? total : 1)));
    ix->owned[3] = items;
    if (n > 0) {
        IX_LAUNCH("gpk_index_fill", grid_register_kernel<true>, grid_for(n, 256), dim3(256), 0, s, bbox, n, grid, cursor, items);
        IX_LAUNCH("gpk_index_sort", cell_sort_kernel, grid_for(n_cells, 256), dim3(256), 0, s, cell_off, n_cells, items);
    }
    IX_HIP(hipStreamSynchronize(s));  // the workspace may be recycled by the next call on another stream
    stamp("directory");
#undef IX_HIP
#undef IX_TRY
#undef IX_LAUNCH

    ix->v.bbox = bbox;
    ix->v.grid = grid;
    ix->v.cell_off = cell_off;
    ix->v.items = items;
    ix->v.gx = gdim;
    ix->v.gy = gdim;
    ix->nbytes = (int64_t)(sizeof(double4) * (size_t)n + sizeof(GridParams) + sizeof(int32_t) * (size_t)(n_cells + 1) +
                           sizeof(int32_t) * (size_t)total);
    if (parts & GPK_INDEX_PIP) {
        const int32_t rc = build_pip_index(a, ix, s, (parts & GPK_INDEX_PIP_LIGHT) ? 0 : ((parts & GPK_INDEX_PIP_FULL) ? 2 : 1));  // raster + slabs for polygonal arrays
        if (rc != GPK_OK) {
            gpk_index_free(ix);
            return rc;
        }
    }
    *out = ix;
    return GPK_OK;
}

int32_t gpk_spatial_join(const gpk_geoarray* left, const gpk_geoarray* right, const gpk_index* right_index,
                         int32_t predicate, uint32_t left_row_base, uint32_t* out_counts, uint32_t* out_pairs,
                         int64_t pair_capacity, int64_t* n_pairs, int32_t out_space, void* stream) {
    if (!left || !right || !n_pairs) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (predicate != GPK_PRED_INTERSECTS && predicate != GPK_PRED_CONTAINS && predicate != GPK_PRED_WITHIN)
        return fail(GPK_ERR_INVALID_ARGUMENT, "unknown predicate %d", predicate);
    if (pair_capacity < 0 || (pair_capacity > 0 && !out_pairs))
        return fail(GPK_ERR_INVALID_ARGUMENT, "pair_capacity without out_pairs");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    *n_pairs = 0;

    // dispatch table of spatial_index.rs:89-137
    const bool pip = left->d.type == GPK_GEOM_POINT && is_polygonal(right->d.type);
    const bool polypoly = is_polygonal(left->d.type) && is_polygonal(right->d.type);
    if (is_polygonal(left->d.type) && right->d.type == GPK_GEOM_POINT)  // the same test with the polygon on the left (:92,96)
        return swapped_pip_join(left, right, left_row_base, out_counts, out_pairs, pair_capacity, n_pairs, out_space, s);
    // polygonal pairs: `intersects` (:102-104,112-123) and `contains` with a POLYGON on the right (:99-101,107-111) have arms
    const bool polypoly_arm = polypoly && (predicate == GPK_PRED_INTERSECTS || (predicate == GPK_PRED_CONTAINS && right->d.type == GPK_GEOM_POLYGON));
    auto lineal = [](int32_t t) { return t == GPK_GEOM_LINESTRING || t == GPK_GEOM_MULTILINESTRING; };
    const bool lineal_point = (left->d.type == GPK_GEOM_POINT && lineal(right->d.type)) || (lineal(left->d.type) && right->d.type == GPK_GEOM_POINT);
    if (!pip && !polypoly_arm && !lineal_point) {  // `_ => false` (spatial_index.rs:136): an empty join, not an error
        if (out_counts && left->d.n_geoms > 0) {
            if (out_space == GPK_MEM_DEVICE)
                GPK_HIP(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * (size_t)left->d.n_geoms, s));
            else
                memset(out_counts, 0, sizeof(uint32_t) * (size_t)left->d.n_geoms);
        }
        return GPK_OK;
    }

    gpk_index* tmp_index = nullptr;
    if (!right_index) {  // built on the fly, like spatial_index.rs:60-71 — only the tables this arm reads
        // (an index that serves ONE join skips the per-entry records of list cells: they cost more to build than one join saves)
        // The built index stays on the right-side handle (gpk_geoarray::auto_index): the reference's default call shape repeated
        // against the same series — a dataframe joined batch after batch against one geometry column — pays for the build once.
        static const bool memo_on = [] {
            const char* e = getenv("GPK_AUTO_INDEX");
            return !(e && *e == '0');
        }();
        static const int64_t memo_max = [] {
            const char* e = getenv("GPK_AUTO_INDEX_MAX_MB");
            return (int64_t)(e && *e ? atoll(e) : 256) << 20;
        }();
        static std::mutex memo_mu;
        const int slot = pip ? 1 : 0;
        gpk_geoarray* rw = const_cast<gpk_geoarray*>(right);
        // (only a handle that owns every buffer it reads: a borrowed device view can be rewritten by its owner between two calls, and
        // an index kept over it would answer for the old bytes)
        const void* const reads[5] = {right->d.xy, right->d.geom_off, right->d.part_off, right->d.ring_off, right->d.validity};
        bool owns_all = true;
        for (int i = 0; i < 5; ++i) owns_all = owns_all && (reads[i] == nullptr || right->owned[i] != nullptr);
        const bool memo = memo_on && owns_all;
        std::lock_guard<std::mutex> lk(memo_mu);
        if (memo && rw->auto_index[slot]) {
            right_index = rw->auto_index[slot];
        } else {
            GPK_TRY(gpk_index_build_ex(right, GPK_INDEX_BBOX_GRID | (pip ? GPK_INDEX_PIP | GPK_INDEX_PIP_LIGHT : 0), nullptr, stream, &tmp_index));
            right_index = tmp_index;
            if (memo && tmp_index->nbytes <= memo_max) {
                rw->auto_index[slot] = tmp_index;
                tmp_index = nullptr;  // (owned by the handle from here on)
            }
        }
    } else if (right_index->n_geoms != right->d.n_geoms || right_index->n_coords != right->d.n_coords || right_index->n_rings != right->d.n_rings) {
        // (an index whose slabs name coordinates by index reads THIS array's coordinates: rows alone do not identify the column)
        return fail(GPK_ERR_INVALID_ARGUMENT, "right_index was built over a different array");
    }
    auto done = [&](int32_t rc) {
        if (tmp_index) gpk_index_free(tmp_index);
        return rc;
    };

    const int64_t n = left->d.n_geoms;
    if (n == 0) return done(GPK_OK);
    if (polypoly)
        return done(bbox_join(left, right, right_index, left_row_base, out_counts, out_pairs, pair_capacity, n_pairs, out_space, s,
                              predicate == GPK_PRED_CONTAINS ? REFINE_CONTAINS : REFINE_POLYGONAL));
    if (lineal_point)
        return done(bbox_join(left, right, right_index, left_row_base, out_counts, out_pairs, pair_capacity, n_pairs, out_space, s, REFINE_LINEAL_POINT));
    const bool host_out = out_space != GPK_MEM_DEVICE;
    const bool want_pairs = pair_capacity > 0;
    const size_t counts_bytes = sizeof(uint32_t) * (size_t)n;
    static thread_local unsigned long long* pinned_total = nullptr;  // device-mapped host word: no D2H copy per call
    if (!pinned_total && hipHostMalloc((void**)&pinned_total, 64, hipHostMallocMapped) != hipSuccess) pinned_total = nullptr;
    uint32_t *counts_dev = nullptr, *pairs_dev = nullptr;
    unsigned long long* grand = nullptr;
    int32_t rc = pip_join_enqueue(left, right, right_index, left_row_base, out_counts, out_pairs, pair_capacity, host_out, pinned_total, s,
                                  &counts_dev, &pairs_dev, &grand);
    if (rc != GPK_OK) return done(rc);

    unsigned long long total = 0;
    hipError_t e = hipSuccess;
    if (!pinned_total) e = hipMemcpyAsync(&total, grand, sizeof total, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "spatial_join: %s", hipGetErrorString(e)));
    if (pinned_total) total = *(volatile unsigned long long*)pinned_total;
    if (total == ~0ull) return done(fail(GPK_ERR_DEVICE, "spatial_join: the fused point join gave up waiting for one of its work-groups"));
    *n_pairs = (int64_t)total;
    if (host_out) {
        if (out_counts) {
            rc = copy_out(out_counts, out_space, counts_dev, counts_bytes, s);
            if (rc != GPK_OK) return done(rc);
        }
        if (want_pairs) {
            const int64_t w = (int64_t)total < pair_capacity ? (int64_t)total : pair_capacity;
            rc = copy_out(out_pairs, out_space, pairs_dev, sizeof(uint32_t) * 2 * (size_t)w, s);
            if (rc != GPK_OK) return done(rc);
        }
    }
    if (want_pairs && (int64_t)total > pair_capacity)
        return done(fail(GPK_ERR_CAPACITY, "spatial_join: %lld pairs but capacity %lld", (long long)total,
                         (long long)pair_capacity));
    return done(GPK_OK);
}

int32_t gpk_spatial_join_async(const gpk_geoarray* left, const gpk_geoarray* right, const gpk_index* right_index, int32_t predicate,
                               uint32_t left_row_base, uint32_t* out_counts, uint32_t* out_pairs, int64_t pair_capacity,
                               int64_t* n_pairs_dev, void* stream) {
    if (!left || !right || !right_index) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument (the stream-ordered join needs a prebuilt index)");
    if (predicate != GPK_PRED_INTERSECTS && predicate != GPK_PRED_CONTAINS && predicate != GPK_PRED_WITHIN)
        return fail(GPK_ERR_INVALID_ARGUMENT, "unknown predicate %d", predicate);
    if (pair_capacity < 0 || (pair_capacity > 0 && !out_pairs))
        return fail(GPK_ERR_INVALID_ARGUMENT, "pair_capacity without out_pairs");
    GPK_TRY(require_device());
    if (!(left->d.type == GPK_GEOM_POINT && is_polygonal(right->d.type)))
        return fail(GPK_ERR_MISMATCHED_GEOMETRY,
                    "spatial_join_async: only point x polygon/multipolygon is stream-ordered (left type %d x right type %d)",
                    left->d.type, right->d.type);
    if (right_index->n_geoms != right->d.n_geoms || right_index->n_coords != right->d.n_coords || right_index->n_rings != right->d.n_rings)
        return fail(GPK_ERR_INVALID_ARGUMENT, "right_index was built over a different array");
    hipStream_t s = (hipStream_t)stream;
    if (left->d.n_geoms == 0) {
        if (n_pairs_dev) GPK_HIP(hipMemsetAsync(n_pairs_dev, 0, sizeof(int64_t), s));
        return GPK_OK;
    }
    uint32_t *counts_dev, *pairs_dev;
    unsigned long long* grand;
    return pip_join_enqueue(left, right, right_index, left_row_base, out_counts, out_pairs, pair_capacity, /*host_out=*/false,
                            (unsigned long long*)n_pairs_dev, s, &counts_dev, &pairs_dev, &grand);
}

}  // extern "C"
