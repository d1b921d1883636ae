// gpk_pipflow.hip — the point x polygonal join of an index with an LDS routing image in ONE launch: the headline kernel
// (C2: 10 M points x 1000 64-vertex polygons).  Reference: the refine loop of geopolars/src/spatial_index.rs:83-143 with the
// `Contains<Point>` arms :91-96; output = per-row hit counts, the sorted (l, r) pair list, the total.
//
// Round 6.  The round-5 kernel (pip_tile_pool_kernel, gpk_join.hip) was bound by the instructions it issued — 1350 vector + 790
// scalar per 512-point tile, `valu_busy` 0.49, a 45 us issue floor under a 92 us launch — and two stages of it ran with most lanes idle:
//   * the exact step (the `test` points of a tile, 5.2 % on C2, walk their half cell's chain): once per tile, 27 of 64 lanes busy,
//     the tile waiting for the verdicts before it could rank its hits;
//   * emission: one pass per point ROW (22 of 64 lanes hold a hit), 35 instructions a row.
// Here a tile never waits for a verdict and emission never sees a miss:
//   * OPTIMISTIC hits.  A `test` point names its polygon before the test (the half-cell record does), so the tile ranks it as a
//     hit at once, writes its 4-byte entry — geometry << 9 | row within the tile — to the tile's slots of a GLOBAL pool (4 bytes per
//     left row, a tile's entries packed at the start of its 512 slots: written and read back by the same CU, they live in its L2) and
//     pushes the point onto the wave's LDS list.  The list is walked when it holds FLOW_PASS_AT points — one pass per two tiles on C2,
//     54 of 64 lanes busy — and a point that fails its test stores DEAD over its entry, 0 over its count, and takes one off its
//     tile's total.
//   * DENSE emission.  After the work-group's barrier a tile's entries are read back 64 at a time (every lane a hit but the dead
//     ones: one ballot), their loads in flight while the work-group waits for the totals of the work-groups before it.
//   * rows the tables cannot settle (list cells, half cells without a chain, orientations the filter cannot certify) are walked by the
//     whole wave where they turn up; a row in SEVERAL geometries is one entry that carries its hit count, expanded at emission.
// No LDS pool, no 16-bit geometry ids, no cap on a work-group's tiles short of FLOW_MAX_TILES: the forms of rounds 4 and 5 (hits in
// per-wave LDS lists, staging slots, chunks, the work-group pool) are gone.
#include "gpk_pipshared.h"

namespace gpk {

// A tile = 64 * P consecutive left rows, decided by one wave; P = 8 for long columns (most work per request round trip), 4 / 2 / 1 for
// short ones — a shard of a column spread over several GPUs — so that every wave of the chip still gets a tile or two and a tile's
// chain of dependent round trips is short (pip_flow_points_per_lane)
constexpr int FLOW_P_MAX = 8;
#ifndef GPK_FLOW_BLOCK
#define GPK_FLOW_BLOCK 1024
#endif
constexpr int FLOW_BLOCK = GPK_FLOW_BLOCK, FLOW_W = FLOW_BLOCK / 64;
constexpr int FLOW_ITEMS = 128;                // list slots per wave
#ifndef GPK_FLOW_PASS_AT
#define GPK_FLOW_PASS_AT 40
#endif
#ifndef GPK_FLOW_PREFETCH
#define GPK_FLOW_PREFETCH 0  // 1: a guard-free tile's points are requested by the tile before it, row by row as that tile's rows are done with
                             // their registers.  Measured 10 us SLOWER on C2 (99.7 against 90.1 us): see DESIGN.md 4.1, round 6
#endif
#ifndef GPK_FLOW_POOL_NT
#define GPK_FLOW_POOL_NT 0  // (1: measured 2 us slower)
#endif
#ifndef GPK_FLOW_VERTEX_MASK
#define GPK_FLOW_VERTEX_MASK 0  // (1: vertices 3 .. 5 requested only by chains that have them — no change on C2: 91.3 against 90.6 us)
#endif
#ifndef GPK_FLOW_FLAT_EDGES
#define GPK_FLOW_FLAT_EDGES 1
#endif
#ifndef GPK_FLOW_ABLATE
#define GPK_FLOW_ABLATE 0  // tuning builds only (answers wrong on purpose): 1 no record / level-1 requests, 2 no exact passes, 3 no count stores,
                           // 4 no pool stores / emission reads nothing, 5 = 1 + 2 + 3 + 4 (points in, nothing decided)
#endif
#ifndef GPK_FLOW_ROUTE_BATCH
#define GPK_FLOW_ROUTE_BATCH 1
#endif
constexpr int FLOW_PASS_AT = GPK_FLOW_PASS_AT;  // a list this long is walked before the next tile is decided
constexpr int FLOW_MAX_TILES = 1536;           // tile records per work-group (201 M rows on 256 CUs at P = 8)
constexpr int FLOW_ID_BITS = 22;               // geometry ids (and the hit count of a row in several geometries) in an entry
constexpr uint32_t FLOW_DEAD = 0xFFFFFFFFu;    // entry of a `test` point that failed
constexpr uint32_t FLOW_MULTI = 0x80000000u;   // entry of a row in several geometries: MULTI | count << 9 | row
constexpr uint32_t FLOW_PEND = FLOW_MULTI;     // ... with count 0: a listed row that found the list full (settled at the end of its tile)
static_assert(64 * FLOW_P_MAX == 512, "an entry keeps the row within its tile in 9 bits");
static_assert(FLOW_MAX_TILES <= (1 << 14), "FlowItem::loc keeps the tile in 14 bits");

// one `test` point on a wave's list
struct FlowItem {
    double px, py;
    uint32_t aux;  // the half cell's chain word (HCHAIN_*)
    uint32_t loc;  // tile (of the work-group's range) << 18 | rank among the tile's entries << 9 | row within the tile
};
static_assert(sizeof(FlowItem) == 24, "list slot");

#ifdef GPK_TILE_TRACE
// diagnosis builds (tools/flow_trace.py): lane 0 of waves 0, 5, 10, 15 of every work-group stamps the 100 MHz wall clock at the
// stage boundaries: 16 words per traced wave at stats[8 + ((blockIdx.x * 4 + wave / 5) * 16 + i)]
#define FLOW_STAMP(i)                                                                                                    \
    do {                                                                                                                 \
        if (h.stats && lane == 0 && wave % 5 == 0 && blockIdx.x < 900u) h.stats[8 + (blockIdx.x * 4 + wave / 5) * 16 + (i)] = wall_clock64(); \
    } while (0)
// ... and a wave's tile time split by phase (stamps 3 .. 8 hold SUMS over the wave's tiles, in clock ticks, instead of tile ends when
// GPK_TILE_TRACE is 2): the phase boundaries wait for the phase's requests, which the shipped kernel does not
#if GPK_TILE_TRACE == 2
#define FLOW_PHASE(i)                                        \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long _n = wall_clock64();        \
        g_phase[i] += _n - g_last;                           \
        g_last = _n;                                         \
    } while (0)
#define FLOW_PHASE_ARGS , unsigned long long (&g_phase)[6], unsigned long long& g_last
#define FLOW_PHASE_PASS , g_phase, g_last
#else
#define FLOW_PHASE(i) do {} while (0)
#define FLOW_PHASE_ARGS
#define FLOW_PHASE_PASS
#endif
#else
#define FLOW_STAMP(i) do {} while (0)
#define FLOW_PHASE(i) do {} while (0)
#define FLOW_PHASE_ARGS
#define FLOW_PHASE_PASS
#endif

// What is read once per pass / per launch (or hardly ever) comes from the kernel-argument segment where it is used — a scalar load that hits
// the scalar cache — instead of living in scalar registers across the tile loop, which has none to spare (a spilled scalar costs
// v_writelane / v_readlane pairs, and past 64 of them another vector register: the hot loop then spills vectors).
#define H_COLD(field) (kernel_arg_at<decltype(ChainHot::field)>((uint32_t)offsetof(ChainHot, field)))
#define T_COLD(field) (kernel_arg_at<decltype(FusedTail::field)>((uint32_t)(sizeof(ChainHot) + offsetof(FusedTail, field))))
static_assert(sizeof(ChainHot) % 8 == 0, "FusedTail follows ChainHot in the argument segment");

__device__ __forceinline__ uint32_t cvt_u32_sat(double v) {  // negative and NaN -> 0, large -> 0xFFFFFFFF (the instruction saturates)
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ void wave_lds_fence() {  // this wave's LDS writes so far are seen by its other lanes' reads that follow
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dev::ring_edge_filtered (gpk_device.h: one edge of geo's coord_pos_relative_to_ring, the orientation certified by Shewchuk's stage-A
// bound or reported `unsure`) without its early exits: in a pass every lane holds a point and some lane takes every arm, so the arms'
// exec-mask branches (a dozen scalar instructions each) are all paid anyway; here every lane evaluates everything and the arms are
// mask logic.  Same verdicts: the winding number changes only where the original changed it, `on` / `unsure` are raised only for
// an edge whose closed y-range holds the point with the point inside its closed x-range.
__device__ __forceinline__ void ring_edge_flat(double sx, double sy, double ex, double ey, double cx, double cy, int& wn, bool& on, bool& unsure) {
    const bool up = sy <= cy && ey >= cy, down = sy > cy && ey <= cy;
    const bool left = cx < fmin(sx, ex), within = !left && cx <= fmax(sx, ex);
    const double detleft = (sx - cx) * (ey - cy), detright = (sy - cy) * (ex - cx);
    const double det = detleft - detright;
    const double errbound = 3.3306690738754716e-16 * (fabs(detleft) + fabs(detright));  // (3 + 16 eps) eps
    const bool opposite = (detleft > 0.0 && detright <= 0.0) || (detleft < 0.0 && detright >= 0.0) || detleft == 0.0;
    const bool certain = opposite || fabs(det) >= errbound;
    const bool eval = (up || down) && within;
    unsure |= eval && !certain;
    on |= eval && certain && det == 0.0;
    wn += (up && ey != cy && (left || (within && det > 0.0))) ? 1 : 0;
    wn -= (down && (left || (within && det < 0.0))) ? 1 : 0;
}

// The verdict of one walked row at its owner: `cnt` hits, the first one `first`.
//   0 hits: the entry dies; 1: the entry names the geometry; several: the entry carries the count (expanded at emission)
__device__ __forceinline__ void flow_settle(const ChainHot& h, uint32_t* s_ttot, uint32_t t, uint32_t slot, uint32_t row, uint32_t li, uint32_t cnt,
                                            uint32_t first) {
    if (cnt == 0u) {
        if (GPK_FLOW_ABLATE != 8 && GPK_FLOW_ABLATE != 9) h.pool[slot] = FLOW_DEAD;
        if (h.counts && GPK_FLOW_ABLATE != 8 && GPK_FLOW_ABLATE != 10) h.counts[row] = 0u;
        atomicSub(&s_ttot[t], 1u);
    } else if (cnt == 1u) {
        h.pool[slot] = (first << 9) | li;
    } else {
        h.pool[slot] = FLOW_MULTI | (cnt << 9) | li;
        if (h.counts) h.counts[row] = cnt;
        atomicAdd(&s_ttot[t], cnt - 1u);
    }
}

// The exact step: the first min(n_list, 64) points of the wave's list, one per lane.  base + the contributions of the chain's edges
// (coord_pos_relative_to_ring's arms, gpk_device.h) decide the point; the five vertices of the first four edges — 99.4 % of the C2
// chains — are requested together (a shorter chain repeats its last vertex: a zero-length edge contributes nothing and reports
// `on` only for a point ON that vertex, which the edge before it has reported already).
template <int TS>  // (a tile = 1 << TS rows)
__device__ __forceinline__ void flow_exact_pass(const ChainHot& h, FlowItem* items, uint32_t& n_list, int lane, uint32_t* s_ttot, uint32_t tile0) {
    const uint32_t n = n_list < 64u ? n_list : 64u;  // (wave-uniform)
    wave_lds_fence();
    const bool act = (uint32_t)lane < n;
    double qx = 0.0, qy = 0.0;
    uint32_t hd = 0u, loc = 0u;
    if (act) {
        const FlowItem* it = items + lane;
        qx = it->px;
        qy = it->py;
        hd = it->aux;
        loc = it->loc;
    }
    const int count = (int)(hd & HCHAIN_COUNT_MASK);
    const double2* __restrict__ v = H_COLD(chain_xy) + (hd >> HCHAIN_START_SHIFT);
#if GPK_FLOW_VERTEX_MASK
    // (every lane-load is a request to the L2 whatever line it names — the L1 does not merge the five: a chain of one or two edges (76 % of the
    // C2 half cells) asks for three vertices, not five)
    double2 a0 = v[0], a1 = v[count < 1 ? count : 1], a2 = a1, a3, a4;
    if (count >= 2) a2 = v[2];
    a3 = a2;
    if (count >= 3) a3 = v[3];
    a4 = a3;
    if (count >= 4) a4 = v[4];
#else
    const int i1 = count < 1 ? count : 1, i2 = count < 2 ? count : 2, i3 = count < 3 ? count : 3, i4 = count < 4 ? count : 4;
    double2 a0 = v[0], a1 = v[i1], a2 = v[i2], a3 = v[i3], a4 = v[i4];
#endif
    asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y), "+v"(a4.x), "+v"(a4.y));
    int wn = ((int)(hd << (28 - HCHAIN_BASE_SHIFT))) >> 28;  // the signed 4-bit base
    bool unsure = false, on = false;
#if GPK_FLOW_FLAT_EDGES
    ring_edge_flat(a0.x, a0.y, a1.x, a1.y, qx, qy, wn, on, unsure);
    ring_edge_flat(a1.x, a1.y, a2.x, a2.y, qx, qy, wn, on, unsure);
    ring_edge_flat(a2.x, a2.y, a3.x, a3.y, qx, qy, wn, on, unsure);
    ring_edge_flat(a3.x, a3.y, a4.x, a4.y, qx, qy, wn, on, unsure);
#else
    on |= dev::ring_edge_filtered(a0.x, a0.y, a1.x, a1.y, qx, qy, wn, unsure);
    on |= dev::ring_edge_filtered(a1.x, a1.y, a2.x, a2.y, qx, qy, wn, unsure);
    on |= dev::ring_edge_filtered(a2.x, a2.y, a3.x, a3.y, qx, qy, wn, unsure);
    on |= dev::ring_edge_filtered(a3.x, a3.y, a4.x, a4.y, qx, qy, wn, unsure);
#endif
    if (count > 4) {
        double ax = a4.x, ay = a4.y;
        for (int j = 4; j < count; ++j) {
            const double2 b = v[j + 1];
            on |= dev::ring_edge_filtered(ax, ay, b.x, b.y, qx, qy, wn, unsure);
            ax = b.x;
            ay = b.y;
        }
    }
    const bool walk = act && (count == 0 || unsure);  // no chain for this half cell, or the filter could not certify a sign
    const bool inside = !on && wn != 0;
    const uint32_t t = loc >> 18, li = loc & 511u;
    const uint32_t slot = ((tile0 + t) << TS) + ((loc >> 9) & 511u), row = ((tile0 + t) << TS) + li;
    if (act && !walk && !inside) flow_settle(h, s_ttot, t, slot, row, li, 0u, 0u);
    unsigned long long wm = __ballot(walk);
#ifdef GPK_TILE_TRACE
    unsigned long long* const stats = nullptr;  // (the trace build stamps the wall clock into that buffer: no counting)
#else
    unsigned long long* const stats = H_COLD(stats);
#endif
    if (stats) {
        const unsigned long long edges = wave_sum_u64((unsigned long long)(act ? count : 0));
        if (lane == 0) {
            atomicAdd(&stats[0], (unsigned long long)n);
            atomicAdd(&stats[1], edges);
            if (wm) atomicAdd(&stats[2], (unsigned long long)__popcll(wm));
        }
    }
    while (wm) {  // (wave-uniform; a handful per launch on real data)
        const int j = __builtin_ctzll(wm);
        wm &= wm - 1ull;
        const double x = __shfl(qx, j, 64), y = __shfl(qy, j, 64);
        const unsigned long long cf = chain_generic_row_first_call(H_COLD(cold), x, y, lane);
        if (lane == j) flow_settle(h, s_ttot, t, slot, row, li, (uint32_t)cf, (uint32_t)(cf >> 32));
    }
    n_list -= n;
    if (n_list) {  // (wave-uniform) the rest of the list moves to its front (at most 64 slots: FLOW_ITEMS = 128)
        FlowItem keep{};
        if ((uint32_t)lane < n_list) keep = items[64 + lane];
        wave_lds_fence();
        if ((uint32_t)lane < n_list) items[lane] = keep;
    }
    wave_lds_fence();
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The entries of tile t that found no room on the list (FLOW_PEND): the whole wave walks each row and its verdict settles the entry.
// Only a tile with more than FLOW_ITEMS - FLOW_PASS_AT + 1 listed rows gets here (the raster is far too coarse for such a right side,
// or the points hug the boundaries).
static __device__ __noinline__ void flow_settle_pending(const double2* pts_xy, const ChainCold* cold, uint32_t* pool, uint32_t* counts, unsigned long long* stats,
                                                        uint32_t* s_ttot, uint32_t t, uint32_t tile, uint32_t n_ent, int lane, int ts) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tile's entries have left this wave)
    uint32_t* seg = pool + ((size_t)tile << ts);
    for (uint32_t i = 0; i < n_ent; i += 64u) {
        const uint32_t e = i + (uint32_t)lane < n_ent ? __hip_atomic_load(seg + i + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        unsigned long long pm = __ballot((e & ~511u) == FLOW_PEND);
        while (pm) {
            const int j = __builtin_ctzll(pm);
            pm &= pm - 1ull;
            const uint32_t li = (uint32_t)__shfl((int)e, j, 64) & 511u, row = (tile << ts) + li, slot = i + (uint32_t)j;
            const double2 q = pts_xy[row];
            const unsigned long long cf = chain_generic_row_first_call(cold, q.x, q.y, lane);
            const uint32_t cnt = (uint32_t)cf, first = (uint32_t)(cf >> 32);
            if (lane == 0) {
                if (stats) atomicAdd(&stats[2], 1ull);
                if (cnt == 0u) {
                    seg[slot] = FLOW_DEAD;
                    if (counts) counts[row] = 0u;
                    atomicSub(&s_ttot[t], 1u);
                } else if (cnt == 1u) {
                    seg[slot] = (first << 9) | li;
                } else {
                    seg[slot] = FLOW_MULTI | (cnt << 9) | li;
                    if (counts) counts[row] = cnt;
                    atomicAdd(&s_ttot[t], cnt - 1u);
                }
            }
        }
    }
}

// One tile: 64 * P consecutive left rows by one wave.
//   SIMPLE  the right side is a POLYGON column without nulls: part == geometry, nothing to look up
//   FULL    the tile holds 64 * P rows of a column without a validity bitmap: no guards on loads and stores
// A row is a hit — one entry in the tile's pool slots, count 1 — as soon as it has a polygon to be in: strictly inside by the tables,
// or LISTED: a `test` point (its half cell's chain decides) or a row the tables cannot settle (list cells, whatever a lean index
// should not hold: chain word 0, the whole wave walks it).  flow_exact_pass takes listed rows back.
template <bool SIMPLE, bool FULL, int P>
__device__ __forceinline__ void flow_tile(const ChainHot& h, const uint2* s_mask, const uint32_t* s_rec0, FlowItem* items, uint32_t& n_list, uint32_t t,
                                          uint32_t tile0, int lane, uint32_t* s_tcand, uint32_t* s_ttot, double (&px)[P], double (&py)[P],
                                          const double2* next_xy FLOW_PHASE_ARGS) {
    constexpr int S = PIP_SUB, FLOW_TILE = 64 * P, TS = P == 8 ? 9 : (P == 4 ? 8 : (P == 2 ? 7 : 6));
    static_assert(P == 1 || P == 2 || P == 4 || P == 8, "tile sizes");
    static_assert(S == 8, "the sub-cell arithmetic below is written for 8 x 8 sub-cells");
    asm volatile("" : "+v"(lane));  // (opaque per tile: k * 64 + lane is then computed where it is used — hoisted out of the tile loop as eight
                                    // loop-invariant registers it is spilled)
    const uint32_t tile = tile0 + t;
    const int64_t base = (int64_t)tile * FLOW_TILE;
    const int64_t n_points = FULL ? 0 : H_COLD(n_points);
    const uint32_t rem = FULL ? (uint32_t)FLOW_TILE : (uint32_t)(n_points - base < (int64_t)FLOW_TILE ? n_points - base : (int64_t)FLOW_TILE);
    // 1. the points.  FULL: px / py hold them already — the tile before requested them row by row as its own rows were done with their
    // registers, so a wave's next 8 KB are on their way from HBM while it works (with one tile's loads per wave in flight only while
    // that wave waits for them, a CU kept ~30 KB in flight where 6.3 TB/s x 2 us needs 49) — and receive the NEXT tile's below.
    // Guarded tiles (the last of a column, columns with nulls): loaded here, NaN for rows past the end and null rows (they route to nothing)
    if (!FULL || !GPK_FLOW_PREFETCH) {
        const double2* __restrict__ tile_xy = h.pts_xy + base;
        static_for<P>([&](auto K) {
            constexpr int k = decltype(K)::value;
            double2 v = make_double2(NAN, NAN);
            if (FULL || ((uint32_t)(k * 64 + lane) < rem && dev::valid_row(H_COLD(pts_validity), base + k * 64 + lane))) v = dev::load_stream(tile_xy + (k * 64 + lane));
            px[k] = v.x;
            py[k] = v.y;
        });
    }
    GPK_SCHED_FENCE();
    FLOW_PHASE(1);  // points here
    // 2. level 1 from the LDS image (two independent reads per point), then the half-cell record / the level-1 word of an interior
    const int logR = h.logR;
    const uint32_t sub_max = ((uint32_t)S << logR) - 1u;
    uint32_t sidx4[(P + 3) / 4], gw[P];
    u32x4 rec[P];
    static_for<(P + 3) / 4>([&](auto J) { sidx4[decltype(J)::value] = 0u; });
#if GPK_FLOW_ROUTE_BATCH
    // (all sixteen image reads first, then all requests: one LDS round trip a tile instead of eight one after the other)
    uint32_t sxs[P], sys_[P], r0s[P];
    uint2 ms[P];
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        uint32_t sx = cvt_u32_sat((px[k] - h.rx0) * h.inv_fw_s), sy = cvt_u32_sat((py[k] - h.ry0) * h.inv_fh_s);
        sx = sx < sub_max ? sx : sub_max;
        sy = sy < sub_max ? sy : sub_max;
        const uint32_t at = ((sy >> 3) << (logR - 5)) + (sx >> 8);
        ms[k] = s_mask[at];
        r0s[k] = s_rec0[at];
        sxs[k] = sx;
        sys_[k] = sy;
    });
    GPK_SCHED_FENCE();
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const uint32_t sx = sxs[k], sy = sys_[k], bit = (sx >> 3) & 31u;
        const bool real = !__builtin_isunordered(px[k], py[k]);
        const bool has = real && ((ms[k].x >> bit) & 1u) != 0u && GPK_FLOW_ABLATE != 1 && (GPK_FLOW_ABLATE < 5 || GPK_FLOW_ABLATE > 7);
        const bool want = real && !has && ((ms[k].y >> bit) & 1u) != 0u && GPK_FLOW_ABLATE != 1 && (GPK_FLOW_ABLATE < 5 || GPK_FLOW_ABLATE > 7);
        sidx4[k / 4] |= (((sy & 3u) << 3) | (sx & 7u)) << (8 * (k % 4));
        rec[k] = u32x4{0u, 0u, 0u, 0u};
        gw[k] = 0u;
        if (has) {
            const uint32_t w = r0s[k] + (uint32_t)__popc(__builtin_amdgcn_ubfe(ms[k].x, 0u, bit));
            rec[k] = *reinterpret_cast<const u32x4*>(h.half + (2u * w + ((sy >> 2) & 1u)));
        }
        if (want) gw[k] = h.cell[((sy >> 3) << logR) + (sx >> 3)];
    });
    GPK_SCHED_FENCE();
#else
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        uint32_t sx = cvt_u32_sat((px[k] - h.rx0) * h.inv_fw_s), sy = cvt_u32_sat((py[k] - h.ry0) * h.inv_fh_s);
        sx = sx < sub_max ? sx : sub_max;
        sy = sy < sub_max ? sy : sub_max;
        const bool real = !__builtin_isunordered(px[k], py[k]);
        const uint32_t cy = sy >> 3, at = (cy << (logR - 5)) + (sx >> 8), bit = (sx >> 3) & 31u;
        const uint2 m = s_mask[at];
        const uint32_t r0 = s_rec0[at];
        const bool has = real && ((m.x >> bit) & 1u) != 0u;
        const bool want = real && !has && ((m.y >> bit) & 1u) != 0u;
        sidx4[k / 4] |= (((sy & 3u) << 3) | (sx & 7u)) << (8 * (k % 4));
        rec[k] = u32x4{0u, 0u, 0u, 0u};
        gw[k] = 0u;
        if (has) {
            const uint32_t w = r0 + (uint32_t)__popc(__builtin_amdgcn_ubfe(m.x, 0u, bit));
            rec[k] = *reinterpret_cast<const u32x4*>(h.half + (2u * w + ((sy >> 2) & 1u)));
        }
        if (want) gw[k] = h.cell[(cy << logR) + (sx >> 3)];
        GPK_SCHED_FENCE();
    });
#endif
    FLOW_PHASE(2);  // routed, records here
    // 3. labels -> hits, ranked row by row; entries, counts and the list's new points leave as they are known
    uint32_t run = 0u, n_new = 0u;  // (wave-uniform) entries so far; points pushed
    bool pend = false;              // (wave-uniform) some listed row found the list full
    uint32_t* const tile_counts = h.counts ? h.counts + base : nullptr;
    uint32_t* const tile_pool = h.pool + ((size_t)tile << TS);
    static_for<P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const uint32_t li = (uint32_t)(k * 64 + lane);
        const uint32_t s5 = (sidx4[k / 4] >> (8 * (k % 4))) & 31u;
        const uint32_t lw = (s5 & 16u) ? rec[k].y : rec[k].x;
        const uint32_t lab = (lw >> (2u * (s5 & 15u))) & 3u;  // 0 outside (and every lane without a record), 1 strictly inside, 2 test
        uint32_t r = lab ? (rec[k].z & 0x3FFFFFFFu) : CODE_NONE;
        const uint32_t g = gw[k];
        const bool single = (g & 0xC0000001u) == 0x40000000u;  // CELL_TAG_SINGLE, strictly inside
        if (single) r = (g >> 1) & 0x1FFFFFFFu;
        const bool walk = g != 0u && !single;  // list cells, whatever a lean index should not hold
        if (!SIMPLE) {
            if (r != CODE_NONE) {
                const uint32_t geom = h.part_geom ? h.part_geom[r] : r;
                r = dev::valid_row(h.polys_validity, geom) ? geom : CODE_NONE;
            }
        }
        if (walk) r = 0u;  // (any geometry: the walk names the real one)
        if (GPK_FLOW_ABLATE == 6 || GPK_FLOW_ABLATE == 7) r = px[k] > 650.0 ? 7u : CODE_NONE;  // (6, 7: the skeleton — points in, 35 % "hits" out, nothing looked up)
        const bool mine = FULL || li < rem;
        const bool cand = mine && r != CODE_NONE;
        bool listed = cand && (lab >= 2u || walk) && GPK_FLOW_ABLATE != 2 && GPK_FLOW_ABLATE != 5;
        const unsigned long long mcand = __ballot(cand);
        const uint32_t rank = run + lanes_below(mcand);
        uint32_t e = (r << 9) | li;
        unsigned long long ml = __ballot(listed);
        if (ml) {  // (wave-uniform)
            if (__builtin_expect(n_list + n_new + (uint32_t)__popcll(ml) > (uint32_t)FLOW_ITEMS, 0)) {  // (wave-uniform) the list is full
                if (listed && n_list + n_new + lanes_below(ml) >= (uint32_t)FLOW_ITEMS) {
                    listed = false;
                    e = FLOW_PEND | li;
                }
                ml = __ballot(listed);
                pend = true;
            }
            if (listed) {
                FlowItem* it = items + (n_list + n_new + lanes_below(ml));
                it->px = px[k];
                it->py = py[k];
                it->aux = walk ? 0u : rec[k].w;
                it->loc = (t << 18) | (rank << 9) | li;
            }
            n_new += (uint32_t)__popcll(ml);
        }
        if (tile_counts && mine && GPK_FLOW_ABLATE != 3 && GPK_FLOW_ABLATE != 5) dev::store_stream(tile_counts + li, cand ? 1u : 0u);
        if (cand && GPK_FLOW_ABLATE != 4 && GPK_FLOW_ABLATE != 5 && GPK_FLOW_ABLATE != 7) {
            if (GPK_FLOW_POOL_NT)
                __builtin_nontemporal_store(e, tile_pool + rank);
            else
                tile_pool[rank] = e;
        }
        run += (uint32_t)__popcll(mcand);
        if (FULL && GPK_FLOW_PREFETCH) {
            if (next_xy) {  // (wave-uniform) this row's registers are free: the next tile's row k
                const double2 v = dev::load_stream(next_xy + (k * 64 + lane));
                px[k] = v.x;
                py[k] = v.y;
            }
        }
        GPK_SCHED_FENCE();
    });
    if (lane == 0) {
        s_tcand[t] = run;
        s_ttot[t] = run;
    }
    n_list += n_new;
    FLOW_PHASE(3);  // ranked, stored
    if (__builtin_expect(pend, 0)) flow_settle_pending(h.pts_xy, H_COLD(cold), h.pool, h.counts, H_COLD(stats), s_ttot, t, tile, run, lane, TS);
}

// 64 entries of a tile -> pairs.  `done`: pairs of the tile written so far; out = the tile's first pair slot
// row0: the tile's first row as the pairs name it (the caller's row base added), pt0: the same row in this call's point array
__device__ __forceinline__ void flow_emit_chunk(const ChainHot& h, uint32_t e, uint32_t row0, uint32_t pt0, unsigned long long* out64, uint32_t room, uint32_t& done,
                                                int lane) {
    const bool live = e != FLOW_DEAD;
    const bool multi = live && (e & FLOW_MULTI) != 0u;
    const unsigned long long mm = __ballot(multi);
    if (__builtin_expect(mm == 0ull, 1)) {  // (wave-uniform)
        const unsigned long long m = __ballot(live);
        const uint32_t at = done + lanes_below(m);
        if (live && at < room) __builtin_nontemporal_store(((unsigned long long)(e >> 9) << 32) | (unsigned long long)(row0 + (e & 511u)), out64 + at);
        done += (uint32_t)__popcll(m);
        return;
    }
    // a row in several geometries among them: every entry's pair count, scanned; the row's hits come from the whole wave's walk
    const uint32_t n_out = !live ? 0u : (multi ? (e >> 9) & ((1u << FLOW_ID_BITS) - 1u) : 1u);
    uint32_t incl = n_out;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const uint32_t at = done + incl - n_out;
    if (live && !multi && at < room) __builtin_nontemporal_store(((unsigned long long)(e >> 9) << 32) | (unsigned long long)(row0 + (e & 511u)), out64 + at);
    unsigned long long wm = mm;
    while (wm) {
        const int j = __builtin_ctzll(wm);
        wm &= wm - 1ull;
        const uint32_t lj = (uint32_t)__shfl((int)e, j, 64) & 511u, aj = (uint32_t)__shfl((int)at, j, 64);
        const double2 q = h.pts_xy[pt0 + lj];
        chain_generic_row_call(H_COLD(cold), q.x, q.y, lane, reinterpret_cast<uint2*>(out64 + aj), aj < room ? room - aj : 0u, row0 + lj);
    }
    done += (uint32_t)__shfl((int)incl, 63, 64);
}

template <bool SIMPLE, int P>
__global__ __launch_bounds__(FLOW_BLOCK) void pip_tile_flow_kernel(ChainHot h, FusedTail tail) {
    constexpr int WORDS = PIP_ROUTE_RMAX * PIP_ROUTE_RMAX / 32, W = FLOW_W, TS = P == 8 ? 9 : (P == 4 ? 8 : (P == 2 ? 7 : 6));
    __shared__ uint2 s_mask[WORDS];     // RouteWord::bmask, gmask
    __shared__ uint32_t s_rec0[WORDS];  // RouteWord::rec0
    __shared__ FlowItem s_items[W][FLOW_ITEMS];
    __shared__ uint32_t s_tcand[FLOW_MAX_TILES], s_ttot[FLOW_MAX_TILES];  // a tile's entries; its pairs (after the scan: the pairs before it)
    __shared__ uint32_t s_next, s_next2, s_wg, s_wgtot, s_gave_up;
    __shared__ unsigned long long s_part[W];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    FLOW_STAMP(0);
    // (the work-group's ticket is asked for before the image is read and used after: its round trip is the image's)
    unsigned long long my_ticket = 0ull;
    if (threadIdx.x == 0) my_ticket = __hip_atomic_fetch_add(T_COLD(ticket), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        const int R = H_COLD(R), words = R * R / 32;  // (R >= 32: the host checks)
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(H_COLD(route));
        for (int i = threadIdx.x; i < words; i += FLOW_BLOCK) {
            const uint4 rw = src[i];
            s_mask[i] = make_uint2(rw.x, rw.y);
            s_rec0[i] = rw.z;
        }
    }
    FLOW_STAMP(1);
    if (threadIdx.x == 0) {
        s_wg = (uint32_t)(my_ticket - T_COLD(ticket_base));
        s_next = 0u;
        s_next2 = 0u;
        s_gave_up = 0u;
    }
    __syncthreads();
    const uint32_t wg = __builtin_amdgcn_readfirstlane(s_wg);
    const uint32_t T0 = (uint32_t)((int64_t)wg * H_COLD(n_tiles) / (int64_t)gridDim.x), T1 = (uint32_t)((int64_t)(wg + 1u) * H_COLD(n_tiles) / (int64_t)gridDim.x);
    const uint32_t nt = T1 - T0;  // (<= FLOW_MAX_TILES: the host checks)
    const bool want_pairs = T_COLD(pairs) != nullptr;
    // ---- the tiles, one at a time from the work-group's counters: the guard-free tiles of the range first (a prefix of it), then the
    // guarded ones — two plain loops, each around ONE instance of the tile code
    FLOW_STAMP(2);
    const uint32_t nf = (int64_t)T1 <= (int64_t)H_COLD(n_full_tiles) ? nt : ((int64_t)T0 >= (int64_t)H_COLD(n_full_tiles) ? 0u : (uint32_t)(H_COLD(n_full_tiles) - (int32_t)T0));
    uint32_t n_list = 0u;
#ifdef GPK_TILE_TRACE
    int n_mine = 0;
#if GPK_TILE_TRACE == 2
    unsigned long long g_phase[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, g_last = wall_clock64();
#endif
#endif
    double px[P], py[P];
    // (the draw: EVERY lane adds 1 — the counter runs in units of 64 — and the wave barrier keeps the iterations apart: see DESIGN.md 4.1,
    // "two things the compiler did")
    auto draw = [&](uint32_t* counter) -> uint32_t {
        __builtin_amdgcn_wave_barrier();
        const uint32_t d = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(d) >> 6;
    };
    {
        uint32_t t = draw(&s_next);
        if (GPK_FLOW_PREFETCH && t < nf) {
            const double2* __restrict__ xy = h.pts_xy + ((size_t)(T0 + t) << TS);
            static_for<P>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const double2 v = dev::load_stream(xy + (k * 64 + lane));
                px[k] = v.x;
                py[k] = v.y;
            });
        }
        while (t < nf) {
            const uint32_t tn = draw(&s_next);  // (the tile after this one)
            FLOW_PHASE(4);  // drawn
            // (a list long enough for a dense pass is walked first)
            if (n_list >= (uint32_t)FLOW_PASS_AT) flow_exact_pass<TS>(h, s_items[wave], n_list, lane, s_ttot, T0);
            FLOW_PHASE(0);  // list walked
            flow_tile<SIMPLE, true, P>(h, s_mask, s_rec0, s_items[wave], n_list, t, T0, lane, s_tcand, s_ttot, px, py,
                                       tn < nf ? h.pts_xy + ((size_t)(T0 + tn) << TS) : nullptr FLOW_PHASE_PASS);
            t = tn;
#if defined(GPK_TILE_TRACE) && GPK_TILE_TRACE != 2
            if (n_mine < 6) FLOW_STAMP(3 + n_mine);
            ++n_mine;
#endif
        }
    }
    for (;;) {
        const uint32_t t = draw(&s_next2) + nf;
        if (t >= nt) break;
        if (n_list >= (uint32_t)FLOW_PASS_AT) flow_exact_pass<TS>(h, s_items[wave], n_list, lane, s_ttot, T0);
        flow_tile<SIMPLE, false, P>(h, s_mask, s_rec0, s_items[wave], n_list, t, T0, lane, s_tcand, s_ttot, px, py, nullptr FLOW_PHASE_PASS);
    }
    while (n_list) flow_exact_pass<TS>(h, s_items[wave], n_list, lane, s_ttot, T0);  // what is left on the list
#if defined(GPK_TILE_TRACE) && GPK_TILE_TRACE == 2
    FLOW_PHASE(5);
    if (h.stats && lane == 0 && wave % 5 == 0 && blockIdx.x < 900u)
        for (int i = 0; i < 6; ++i) h.stats[8 + (blockIdx.x * 4 + wave / 5) * 16 + 3 + i] = g_phase[i];
#endif
    FLOW_STAMP(9);
    __syncthreads();
    FLOW_STAMP(10);
    // ---- emission runs in ROUNDS of EM_T tiles a wave (wave w: tiles w, w + W, ...), up to EM_C * 64 entries of each requested together
    // (a tile with more reads the rest afterwards); the FIRST round is requested now — its round trip runs under the scan and the wait
    // for the work-groups before this one
    constexpr int EM_C = P < 4 ? P : 4, EM_T = 24 / EM_C;
    uint32_t ent[EM_T][EM_C], cand[EM_T];
    auto em_request = [&](uint32_t first) {
        static_for<EM_T>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            const uint32_t et = first + (uint32_t)(r * W);
            cand[r] = et < nt && want_pairs && GPK_FLOW_ABLATE != 4 && GPK_FLOW_ABLATE != 5 && GPK_FLOW_ABLATE != 7 ? s_tcand[et] : 0u;
            cand[r] = __builtin_amdgcn_readfirstlane(cand[r]);
            const uint32_t* seg = h.pool + ((size_t)(T0 + et) << TS);
            static_for<EM_C>([&](auto Cc) {
                constexpr int c = decltype(Cc)::value;
                ent[r][c] = FLOW_DEAD;
                if ((uint32_t)(c * 64 + lane) < cand[r]) ent[r][c] = seg[c * 64 + lane];
            });
        });
    };
    em_request((uint32_t)wave);
    // ---- the tiles' places within the work-group's pairs (wave 0: an in-place scan of the totals), the work-group's total
    if (wave == 0) {
        uint32_t carry = 0u;
        for (uint32_t c0 = 0; c0 < nt; c0 += 64u) {
            const uint32_t v = c0 + (uint32_t)lane < nt ? s_ttot[c0 + lane] : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t a = __shfl_up(incl, o, 64);
                if (lane >= o) incl += a;
            }
            if (c0 + (uint32_t)lane < nt) s_ttot[c0 + lane] = carry + incl - v;
            carry += (uint32_t)__shfl((int)incl, 63, 64);
        }
        if (lane == 0) s_wgtot = carry;
    }
    __syncthreads();
    const unsigned long long wg_tot = (unsigned long long)s_wgtot;
    const unsigned long long epoch = T_COLD(epoch);
    unsigned long long* const slots = T_COLD(slots);
    if (threadIdx.x == 0) __hip_atomic_store(&slots[wg], (epoch << FUSED_TOTAL_BITS) | wg_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long acc = 0;
    bool gave_up = false;
    for (unsigned b = threadIdx.x; b < wg; b += FLOW_BLOCK) {
        unsigned long long v;
        uint32_t spins = 0;
        while (((v = __hip_atomic_load(&slots[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> FUSED_TOTAL_BITS) != epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > FUSED_SPIN_LIMIT) {  // (seconds: a work-group before this one never ran — report, never hang the device)
                gave_up = true;
                break;
            }
        }
        acc += v & ((1ull << FUSED_TOTAL_BITS) - 1ull);
    }
    if (gave_up) __hip_atomic_store(&s_gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    acc = wave_sum_u64(acc);
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    const bool lost = s_gave_up != 0u;
    unsigned long long base_off = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) base_off += s_part[w];
    if ((wg == gridDim.x - 1 || lost) && threadIdx.x == 0) {  // (FUSED_LOST is sticky: whoever writes the total has seen every word anybody waited for)
        unsigned long long* const lostp = T_COLD(lost);
        if (lost) __hip_atomic_store(lostp, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool any_lost = lost || __hip_atomic_load(lostp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
        unsigned long long* const grand = T_COLD(grand);
        unsigned long long* const grand_host = T_COLD(grand_host);
        *grand = any_lost ? FUSED_LOST : base_off + wg_tot;
        if (grand_host) *grand_host = any_lost ? FUSED_LOST : base_off + wg_tot;
    }
    FLOW_STAMP(11);
    if (!want_pairs || lost) return;
    // ---- emission
    unsigned long long* const pairs64 = reinterpret_cast<unsigned long long*>(T_COLD(pairs));
    const int64_t capacity = T_COLD(capacity);
    const uint32_t left_base = T_COLD(left_base);
    for (uint32_t first = (uint32_t)wave; first < nt; first += (uint32_t)(EM_T * W)) {
        if (first != (uint32_t)wave) em_request(first);  // (columns with more than EM_T tiles a wave)
        static_for<EM_T>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            const uint32_t et = first + (uint32_t)(r * W);
            if (cand[r]) {  // (wave-uniform)
                const unsigned long long off = base_off + (unsigned long long)s_ttot[et];
                unsigned long long* const out64 = pairs64 + off;
                const int64_t left = capacity - (int64_t)off;
                const uint32_t room = left <= 0 ? 0u : (left > (int64_t)0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)left);
                const uint32_t pt0 = (T0 + et) << TS, row0 = pt0 + left_base;
                uint32_t done = 0u;
                static_for<EM_C>([&](auto Cc) {
                    constexpr int c = decltype(Cc)::value;
                    if ((uint32_t)(c * 64) < cand[r]) flow_emit_chunk(h, ent[r][c], row0, pt0, out64, room, done, lane);
                });
                if (EM_C * 64 < 64 * P) {
                    const uint32_t* seg = h.pool + ((size_t)(T0 + et) << TS);
                    for (uint32_t i = (uint32_t)(EM_C * 64); i < cand[r]; i += 64u) {
                        const uint32_t e = i + (uint32_t)lane < cand[r] ? seg[i + lane] : FLOW_DEAD;
                        flow_emit_chunk(h, e, row0, pt0, out64, room, done, lane);
                    }
                }
            }
        });
    }
    FLOW_STAMP(12);
}

// Points per lane for a column of n_left_rows rows on `wgs` work-groups: the largest tile that still leaves every wave a tile
// (a wave's tile is a chain of dependent round trips, ~13 us at P = 8: a short shard is spread thin instead); 0: not eligible
int pip_flow_points_per_lane(int64_t n_left_rows, int64_t n_right_geoms, int32_t R, int wgs) {
    if (R < 32 || R > PIP_ROUTE_RMAX || wgs < 1) return 0;
    if (n_right_geoms >= (int64_t)((1u << FLOW_ID_BITS) - 1u)) return 0;
    if (n_left_rows >= (int64_t)0xFFFFFE00ll) return 0;  // rows (and a tile's slots) are 32-bit indices
    static const int forced = [] {
        const char* e = getenv("GPK_FLOW_P");  // A/B runs: 1, 2, 4 or 8
        return e ? atoi(e) : 0;
    }();
    const int64_t waves = (int64_t)wgs * FLOW_W;
    for (int p = FLOW_P_MAX; p >= 1; p >>= 1) {
        const int64_t n_tiles = (n_left_rows + 64 * p - 1) / (64 * p);
        if ((n_tiles + wgs - 1) / wgs > (int64_t)FLOW_MAX_TILES) return p == FLOW_P_MAX ? 0 : 2 * p;
        if (forced ? p == forced : (n_tiles >= waves || p == 1)) return p;  // (measured, C2 right side: 1.25 M rows 26.2 / 22.7 / 23.9 / 26.1 us at P = 8 / 4 / 2 / 1; 2.5 M rows 32.6 / 32.5 / 33.3 / 37.4)
    }
    return 1;
}
size_t pip_flow_pool_bytes(int64_t n_left_rows) { return sizeof(uint32_t) * (size_t)((n_left_rows + 511) / 512) * 512u; }

template <int P>
static int32_t launch_pip_flow_p(const ChainHot& hot, const FusedTail& tail, int wgs, hipStream_t s) {
    if (hot.part_geom == nullptr && hot.polys_validity == nullptr)
        GPK_LAUNCH("gpk_pip_tile", (pip_tile_flow_kernel<true, P>), dim3((unsigned)wgs), dim3(FLOW_BLOCK), 0, s, hot, tail);
    else
        GPK_LAUNCH("gpk_pip_tile", (pip_tile_flow_kernel<false, P>), dim3((unsigned)wgs), dim3(FLOW_BLOCK), 0, s, hot, tail);
    return GPK_OK;
}
int32_t launch_pip_flow(const ChainHot& hot, const FusedTail& tail, int wgs, int points_per_lane, hipStream_t s) {
    switch (points_per_lane) {
        case 8: return launch_pip_flow_p<8>(hot, tail, wgs, s);
        case 4: return launch_pip_flow_p<4>(hot, tail, wgs, s);
        case 2: return launch_pip_flow_p<2>(hot, tail, wgs, s);
        case 1: return launch_pip_flow_p<1>(hot, tail, wgs, s);
    }
    return fail(GPK_ERR_INVALID_ARGUMENT, "spatial_join: %d points per lane", points_per_lane);
}

}  // namespace gpk
