// gpk_pipshared.h — what the point x polygonal join kernels of gpk_join.hip (general / lean / chain tile kernels + the writer) and of
// gpk_pipflow.hip (the one-launch join of an index with an LDS routing image: the headline kernel) have in common: result codes,
// the hot / cold argument records of the chain kernels, the whole-wave generic walk of a rare row, the words a one-launch join's
// work-groups exchange.  Reference: the refine loop of geopolars/src/spatial_index.rs:83-143 (Contains<Point>, :91-96).
#pragma once

#include <type_traits>

#include "gpk_device.h"
#include "gpk_index.h"

namespace gpk {

// per-point result code handed from pip_tile to pip_write: a geometry id (exactly one hit), CODE_NONE, or
// CODE_MULTI (several hits: the writer re-enumerates them with the generic walk)
constexpr uint32_t CODE_NONE = 0xFFFFFFFFu, CODE_MULTI = 0xFFFFFFFEu;
// a code with the top bit set (and not one of the two above) is 0x80000000 | offset into the multi-hit pool:
// pool[offset] = m, then the m geometry ids (ascending) of a point that lies in several geometries
constexpr uint32_t CODE_POOL = 0x80000000u;

#define GPK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// a loop over 0 .. N - 1 whose index is a compile-time constant in the body: per-point state lives in small arrays, and only constant
// indices from the start keep the compiler from turning such an array into one wide register tuple (or leaving it in scratch memory)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

struct ChainHot {
    const double2* pts_xy;
    const uint8_t* pts_validity;
    int64_t n_points, n_tiles;
    const uint8_t* polys_validity;
    int32_t R, logR;  // the raster side is a power of two
    double rx0, ry0, inv_fw, inv_fh;
    const uint32_t* cell;
    const HalfCell* half;
    const ChainAux* sub_aux;
    const uint32_t* chain_head;
    const double2* chain_ext;
    const double2* chain_xy;  // GPK_HALF_CHAINS: the vertices the records' chain words index
    const uint32_t* part_geom;
    const RouteWord* route;
    uint32_t* counts;
    uint32_t* code;
    unsigned long long* block_tot;
    unsigned long long* super_tot;
    unsigned long long* stats;
    const struct ChainCold* cold;  // what the generic walk of a rare row reads (written by join_prep_kernel)
    uint2* stage;                  // pip_tile_fused_kernel: one pair slot per left row, a wave's hits at the start of its rows' slots
    int32_t n_full_tiles, pad;     // pip_tile_fused_kernel: tiles 0 .. n_full_tiles - 1 need no guards (whole tiles of a column without a validity bitmap)
    double inv_fw_s, inv_fh_s, sub_max;  // pip_tile_fused_kernel: inv_fw * PIP_SUB, inv_fh * PIP_SUB (exact: a power of two), (PIP_SUB << logR) - 1
    uint32_t* pool;                // pip_tile_flow_kernel: one 4-byte hit slot per left row (a tile's hits at the start of its 512 slots)
};
// the arguments of the rare arm, in device memory: loaded where they are used — as kernel arguments they would be held in scalar
// registers across the hot loop (and spilled)
struct ChainCold {
    DevGeo polys;
    IndexView ix;
    GridParams grid;
};

// The generic (always exact) walk for ONE point, by a whole wave: directory candidates in ascending id order, each candidate's
// rings with the lanes striding over the edges (a row costs a handful of dependent loads, not one per edge), the exact
// orientation kernel inlined.  Returns the hit count; *first = the first hit.  Same answers as generic_point.
// INLINE_EXACT: the expansion arithmetic of the exact orientation unrolled into registers (no call, no scratch memory: what the
// persistent route kernel wants, which owns 128 registers per lane anyway) or reached by a call (the chain kernel: 84 registers
// instead of 113, i.e. one more wave per SIMD, for a 208-byte stack).
// emit != nullptr: hit number t of the row (ascending geometry id) is stored as (l, id) in emit[t] while t < emit_room
template <bool INLINE_EXACT>
__device__ __forceinline__ uint32_t chain_generic_row(const ChainCold* __restrict__ cold, double px, double py, int lane, uint32_t* first,
                                                      uint2* emit = nullptr, uint32_t emit_room = 0u, uint32_t l = 0u) {
    const DevGeo polys = cold->polys;
    const IndexView ix = cold->ix;
    const GridParams g = cold->grid;
    uint32_t cnt = 0;
    *first = CODE_NONE;
    if (!(px == px && py == py)) return 0u;
    const int cx = dev::cell_of(px, g.x0, g.inv_w, g.gx), cy = dev::cell_of(py, g.y0, g.inv_h, g.gy);
    const int cc = cy * g.gx + cx;
    for (int q = ix.cell_off[cc]; q < ix.cell_off[cc + 1]; ++q) {
        const int j = ix.items[q];
        const double4 bb = ix.bbox[j];
        if (!(px >= bb.x && px <= bb.z && py >= bb.y && py <= bb.w) || !dev::valid_row(polys.validity, j)) continue;
        int p0, p1;
        dev::geom_parts(polys, j, p0, p1);
        bool hit = false;
        for (int part = p0; part < p1 && !hit; ++part) {  // Contains<Point>: strictly inside some member polygon
            int r0, r1;
            dev::part_rings(polys, part, r0, r1);
            int pos = dev::POS_INSIDE;  // position w.r.t. the polygon: exterior first, then the holes
            for (int r = r0; r < r1 && pos == dev::POS_INSIDE; ++r) {
                const int c0 = polys.ring_off[r], n = polys.ring_off[r + 1] - c0;
                int wn = 0, on = 0;
                if (n == 1) {
                    const double2 s0 = polys.xy[c0];
                    on = (px == s0.x && py == s0.y) ? 1 : 0;
                }
                for (int i = lane; i + 1 < n; i += 64) {
                    const double2 s0 = polys.xy[c0 + i], s1 = polys.xy[c0 + i + 1];
                    on |= (int)(INLINE_EXACT ? dev::ring_edge_inline(s0.x, s0.y, s1.x, s1.y, px, py, wn) : dev::ring_edge(s0.x, s0.y, s1.x, s1.y, px, py, wn));
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    wn += __shfl_xor(wn, o, 64);
                    on |= __shfl_xor(on, o, 64);
                }
                const int rp = n == 0 ? dev::POS_OUTSIDE : (on ? dev::POS_BOUNDARY : (wn != 0 ? dev::POS_INSIDE : dev::POS_OUTSIDE));
                if (r == r0)
                    pos = rp;  // Outside / OnBoundary of the exterior ends it
                else if (rp == dev::POS_BOUNDARY)
                    pos = dev::POS_BOUNDARY;
                else if (rp == dev::POS_INSIDE)
                    pos = dev::POS_OUTSIDE;  // inside a hole
            }
            hit = r1 > r0 && pos == dev::POS_INSIDE;
        }
        if (hit) {
            if (cnt == 0) *first = (uint32_t)j;
            if (emit != nullptr && lane == 0 && cnt < emit_room) emit[cnt] = make_uint2(l, (uint32_t)j);
            ++cnt;
        }
    }
    return cnt;
}

// set bits of a wave mask below this lane (v_mbcnt: no lane-mask register pair to keep)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// What the fused kernel's tile loop reads once per tile (or hardly ever) is read from the kernel-argument segment where it is used
// — a scalar load that hits the scalar cache — instead of living in scalar registers across the loop, which has none to spare: every
// spilled scalar costs v_writelane / v_readlane pairs, and past 64 of them a second vector register.  (The pointer is made opaque:
// named directly the compiler loads every argument at the top of the kernel.)
template <typename T>
__device__ __forceinline__ T kernel_arg_at(uint32_t offset) {
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(const T __attribute__((address_space(4)))*)(ka + offset);
}

// The same walk as a CALL (pip_tile_fused_kernel): the fused tile loop keeps the next tile's points in registers across the rare arm;
// inlined, the walk's register needs are the loop's (everything live across it is spilled, on every path); called, only the call
// site saves what it must.
static __device__ __noinline__ uint32_t chain_generic_row_call(const ChainCold* cold, double px, double py, int lane, uint2* emit, uint32_t emit_room, uint32_t l) {
    uint32_t first;
    return chain_generic_row<false>(cold, px, py, lane, &first, emit, emit_room, l);
}
// (count in the low half, the first hit's geometry in the high half)
static __device__ __noinline__ unsigned long long chain_generic_row_first_call(const ChainCold* cold, double px, double py, int lane) {
    uint32_t first;
    const uint32_t cnt = chain_generic_row<false>(cold, px, py, lane, &first);
    return ((unsigned long long)first << 32) | cnt;
}

struct FusedTail {
    uint2* pairs;            // may be nullptr: counts and total only
    int64_t capacity;        // pair slots of `pairs`
    unsigned long long* slots;   // one word per work-group: epoch << FUSED_TOTAL_BITS | hit total
    unsigned long long epoch;
    unsigned long long* grand;
    unsigned long long* grand_host;
    uint32_t left_base, pad;
    unsigned long long* ticket;      // work-groups number themselves in the order they START: ticket - ticket_base
    unsigned long long ticket_base;  // (the counter only ever grows: the host knows where a launch's numbers begin)
    unsigned long long* lost;        // a work-group that gave up waiting stores the launch's epoch here; the work-group that writes the total
                                     // reads it AFTER its own waits (which cover every word anybody waited for): FUSED_LOST is sticky
};

constexpr int FUSED_TOTAL_BITS = 40;
constexpr uint32_t FUSED_SPIN_LIMIT = 1u << 22;
constexpr unsigned long long FUSED_LOST = ~0ull;  // in the total's place: the launch gave up waiting (gpk_spatial_join reports GPK_ERR_DEVICE)

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// gpk_pipflow.hip: the one-launch join (pip_tile_flow_kernel).  pip_flow_points_per_lane: 0 when the join cannot take it, else the tile size
// (64 * points per lane) that ChainHot::n_tiles / n_full_tiles must be counted in; launch_pip_flow: the caller holds the launch lock of
// fused_launch_begin and has filled the epoch words of `tail`.
int pip_flow_points_per_lane(int64_t n_left_rows, int64_t n_right_geoms, int32_t R, int wgs);
size_t pip_flow_pool_bytes(int64_t n_left_rows);
int32_t launch_pip_flow(const ChainHot& hot, const FusedTail& tail, int wgs, int points_per_lane, hipStream_t s);

}  // namespace gpk
