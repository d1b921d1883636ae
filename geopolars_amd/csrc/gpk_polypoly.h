// gpk_polypoly.h — polygon x polygon `intersects` (geo 0.27 algorithm/intersects/{line,polygon}.rs),
// reached from the join dispatch geopolars/src/spatial_index.rs:102-104,112-123 and from the row-wise
// north-star predicate.  One lane evaluates one pair; every orientation is exact (gpk_device.h).
#pragma once

#include "gpk_device.h"

#ifndef GPK_PP_BOX_SKIP
#define GPK_PP_BOX_SKIP 1  // a vertex outside the other polygon's exterior box is outside it: the containment walk is skipped
#endif
namespace gpk {

// Intersects<Line> for Line (geo 0.27 intersects/line.rs)
__device__ __forceinline__ bool point_in_rect(double px, double py, double ax, double ay, double bx, double by) {
    return dev::value_in_between(px, ax, bx) && dev::value_in_between(py, ay, by);
}
__device__ inline bool line_intersects_line(double2 a0, double2 a1, double2 b0, double2 b1) {
    if (a0.x == a1.x && a0.y == a1.y)
        return dev::orient2d(b0.x, b0.y, b1.x, b1.y, a0.x, a0.y) == 0 && point_in_rect(a0.x, a0.y, b0.x, b0.y, b1.x, b1.y);
    const int c11 = dev::orient2d(a0.x, a0.y, a1.x, a1.y, b0.x, b0.y);
    const int c12 = dev::orient2d(a0.x, a0.y, a1.x, a1.y, b1.x, b1.y);
    if (c11 != c12) {
        const int c21 = dev::orient2d(b0.x, b0.y, b1.x, b1.y, a0.x, a0.y);
        const int c22 = dev::orient2d(b0.x, b0.y, b1.x, b1.y, a1.x, a1.y);
        return c21 != c22;
    }
    if (c11 == 0)
        return point_in_rect(b0.x, b0.y, a0.x, a0.y, a1.x, a1.y) || point_in_rect(b1.x, b1.y, a0.x, a0.y, a1.x, a1.y) ||
               point_in_rect(a1.x, a1.y, b0.x, b0.y, b1.x, b1.y) || point_in_rect(a0.x, a0.y, b0.x, b0.y, b1.x, b1.y);
    return false;
}

__device__ inline bool exterior_bbox(const DevGeo& a, int r0, int r1, double4* bb) {
    if (r1 <= r0) return false;
    const int c0 = a.ring_off[r0], c1 = a.ring_off[r0 + 1];
    if (c1 == c0) return false;
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = c0; i < c1; ++i) {
        const double2 p = a.xy[i];
        mnx = p.x < mnx ? p.x : mnx;
        mny = p.y < mny ? p.y : mny;
        mxx = p.x > mxx ? p.x : mxx;
        mxy = p.y > mxy ? p.y : mxy;
    }
    *bb = make_double4(mnx, mny, mxx, mxy);
    return true;
}

// bbox over ALL rings of a polygon: used only to prune segment pairs (invalid input may have holes
// poking out of the exterior, and pruning must never change the answer)
__device__ inline double4 all_rings_bbox(const DevGeo& a, int r0, int r1) {
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = a.ring_off[r0]; i < a.ring_off[r1]; ++i) {
        const double2 p = a.xy[i];
        mnx = p.x < mnx ? p.x : mnx;
        mny = p.y < mny ? p.y : mny;
        mxx = p.x > mxx ? p.x : mxx;
        mxy = p.y > mxy ? p.y : mxy;
    }
    return make_double4(mnx, mny, mxx, mxy);
}

// Intersects<Polygon> for Polygon (geo 0.27 intersects/polygon.rs), one lane.  Equivalent boolean:
// bboxes not disjoint AND (some ring segment pair intersects OR a vertex of B is not Outside A OR a
// vertex of A's exterior is not Outside B).
__device__ inline bool polygon_intersects_polygon(const DevGeo& a, int ar0, int ar1, const DevGeo& b, int br0, int br1) {
    double4 ba, bb;
    if (!exterior_bbox(a, ar0, ar1, &ba) || !exterior_bbox(b, br0, br1, &bb)) return false;
    if (ba.z < bb.x || ba.w < bb.y || bb.z < ba.x || bb.w < ba.y) return false;
    const double4 fa = all_rings_bbox(a, ar0, ar1);
    for (int rb = br0; rb < br1; ++rb) {
        const int b0 = b.ring_off[rb], b1 = b.ring_off[rb + 1];
        for (int j = b0; j + 1 < b1; ++j) {
            const double2 q0 = b.xy[j], q1 = b.xy[j + 1];
            // cheap reject of this segment against A's bbox keeps the O(n*m) loop short in practice
            const double qlx = fmin(q0.x, q1.x), qhx = fmax(q0.x, q1.x), qly = fmin(q0.y, q1.y), qhy = fmax(q0.y, q1.y);
            if (qhx < fa.x || qlx > fa.z || qhy < fa.y || qly > fa.w) continue;
            for (int ra = ar0; ra < ar1; ++ra) {
                const int a0 = a.ring_off[ra], a1 = a.ring_off[ra + 1];
                for (int i = a0; i + 1 < a1; ++i) {
                    const double2 p0 = a.xy[i], p1 = a.xy[i + 1];
                    if (fmax(p0.x, p1.x) < qlx || fmin(p0.x, p1.x) > qhx || fmax(p0.y, p1.y) < qly || fmin(p0.y, p1.y) > qhy)
                        continue;  // disjoint segment boxes cannot intersect (closed test)
                    if (line_intersects_line(p0, p1, q0, q1)) return true;
                }
            }
        }
    }
    // no boundary crossing: containment of one in the other (every vertex is tested, as upstream does)
    for (int rb = br0; rb < br1; ++rb) {
        const int b0 = b.ring_off[rb], b1 = b.ring_off[rb + 1];
        for (int j = b0; j < b1; ++j) {
            const double2 q = b.xy[j];
            if (dev::polygon_pos(a, ar0, ar1, q.x, q.y) != dev::POS_OUTSIDE) return true;
        }
    }
    {
        const int a0 = a.ring_off[ar0], a1 = a.ring_off[ar0 + 1];
        for (int i = a0; i < a1; ++i) {
            const double2 p = a.xy[i];
            if (dev::polygon_pos(b, br0, br1, p.x, p.y) != dev::POS_OUTSIDE) return true;
        }
    }
    return false;
}

__device__ inline bool polygonal_intersects_polygonal(const DevGeo& a, int64_t ia, const DevGeo& b, int64_t ib) {
    int a0, a1, b0, b1;
    dev::geom_parts(a, ia, a0, a1);
    dev::geom_parts(b, ib, b0, b1);
    for (int p = a0; p < a1; ++p) {
        int ar0, ar1;
        dev::part_rings(a, p, ar0, ar1);
        for (int q = b0; q < b1; ++q) {
            int br0, br1;
            dev::part_rings(b, q, br0, br1);
            if (polygon_intersects_polygon(a, ar0, ar1, b, br0, br1)) return true;
        }
    }
    return false;
}



// ---- G lanes cooperating on one (A, B) pair ----------------------------------------------------------
// Same boolean as polygonal_intersects_polygonal.  Lane k owns segments k, k+G, ... of B (flattened over B's
// rings) and tests each against every segment of A with box pruning; a group-wide OR after every round allows
// early exit.  If no boundary pair touches, a ring never changes side of the other polygon's boundary, so ONE
// vertex per ring decides containment (upstream tests every endpoint; the outcome is identical).
template <int G>
__device__ __forceinline__ bool group_any(bool v) {
    return dev::group_or<G>(v ? 1 : 0) != 0;
}

// Ring-relative position of (cx, cy), G lanes striding over the ring's edges: winding contributions add up, the
// on-boundary flag ORs.  Same value on every lane of the group.
template <int G>
__device__ inline int coord_pos_ring_group(const double2* __restrict__ ring, int n, double cx, double cy, int lane) {
    if (n == 0) return dev::POS_OUTSIDE;
    if (n == 1) {
        const double2 s = ring[0];
        return (cx == s.x && cy == s.y) ? dev::POS_BOUNDARY : dev::POS_OUTSIDE;
    }
    int wn = 0, on = 0;
    for (int i = lane; i + 1 < n; i += 2 * G) {  // two edges per trip, both requested before either is evaluated
        const int i2 = i + G;
        const bool two = i2 + 1 < n;
        const double2 s = ring[i], e = ring[i + 1], s2 = ring[two ? i2 : i], e2 = ring[two ? i2 + 1 : i + 1];
        on |= dev::ring_edge(s.x, s.y, e.x, e.y, cx, cy, wn) ? 1 : 0;
        if (two) on |= dev::ring_edge(s2.x, s2.y, e2.x, e2.y, cx, cy, wn) ? 1 : 0;
    }
    {  // one packed reduction: winding sum in the high half, on-boundary count in the low half
        const int packed = dev::group_sum<G>(wn * 65536 + on);
        wn = packed >> 16;
        on = packed & 0xFFFF;
    }
    if (on) return dev::POS_BOUNDARY;
    return wn == 0 ? dev::POS_OUTSIDE : dev::POS_INSIDE;
}
// Polygon::coordinate_position, cooperatively (dev::polygon_pos is the one-lane form)
template <int G>
__device__ inline int polygon_pos_group(const DevGeo& a, int r0, int r1, double cx, double cy, int lane) {
    if (r1 <= r0) return dev::POS_OUTSIDE;
    int c0 = a.ring_off[r0], c1 = a.ring_off[r0 + 1];
    if (c1 == c0) return dev::POS_OUTSIDE;
    const int pe = coord_pos_ring_group<G>(a.xy + c0, c1 - c0, cx, cy, lane);
    if (pe != dev::POS_INSIDE) return pe;
    for (int r = r0 + 1; r < r1; ++r) {
        c0 = a.ring_off[r];
        c1 = a.ring_off[r + 1];
        const int ph = coord_pos_ring_group<G>(a.xy + c0, c1 - c0, cx, cy, lane);
        if (ph == dev::POS_BOUNDARY) return dev::POS_BOUNDARY;
        if (ph == dev::POS_INSIDE) return dev::POS_OUTSIDE;
    }
    return dev::POS_INSIDE;
}

// bboxes of a polygon's exterior (.x.. of `ext`) and of all its rings (`all`), G lanes, same value on every lane
// `known` (when have_known): the exterior's box the caller already has (bounds of a Polygon row).  By VALUE: a pointer to the
// caller's local put that local into scratch memory — 64 bytes stored per lane and candidate pair, 4 GB of HBM writes per
// 1M x 1M join (profiles/r02e_pmc_traffic_configs.json: WRITE_SIZE of gpk_pair_refine against 4 MB of results).
template <int G>
__device__ inline void polygon_bboxes_group(const DevGeo& a, int r0, int r1, int lane, bool have_known, double4 known, double4& ext, double4& all) {
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    const int c0 = a.ring_off[r0], ce = a.ring_off[r0 + 1], c1 = a.ring_off[r1];
    if (have_known) {
        mnx = known.x; mny = known.y; mxx = known.z; mxy = known.w;
    } else {
        for (int i = c0 + lane; i < ce; i += G) {
            const double2 p = a.xy[i];
            mnx = fmin(mnx, p.x); mny = fmin(mny, p.y); mxx = fmax(mxx, p.x); mxy = fmax(mxy, p.y);
        }
        mnx = dev::group_min<G>(mnx); mny = dev::group_min<G>(mny); mxx = dev::group_max<G>(mxx); mxy = dev::group_max<G>(mxy);
    }
    ext = make_double4(mnx, mny, mxx, mxy);
    if (ce < c1) {  // holes: invalid input may have them poke out of the exterior, and pruning must never change the answer
        for (int i = ce + lane; i < c1; i += G) {
            const double2 p = a.xy[i];
            mnx = fmin(mnx, p.x); mny = fmin(mny, p.y); mxx = fmax(mxx, p.x); mxy = fmax(mxy, p.y);
        }
        mnx = dev::group_min<G>(mnx); mny = dev::group_min<G>(mny); mxx = dev::group_max<G>(mxx); mxy = dev::group_max<G>(mxy);
    }
    all = make_double4(mnx, mny, mxx, mxy);
}

// Same boolean as polygon_intersects_polygon, G lanes per pair, with WINDOW CLIPPING: two segments can only meet
// inside the intersection of the two polygons' boxes, so
//   1. B's segments whose box misses A's box are dropped; the survivors are compacted (ballot within the group)
//      into a per-group LDS list of PP_LIST entries (endpoints, 32 B each); A's segments likewise into a second list;
//   2. list x list: lane k owns entries k, k+G, ... of A's list and walks B's list (LDS broadcasts), with a
//      group-wide early exit; lists that fill up are processed in chunks;
//   3. if no boundary pair touches, a ring never changes side of the other polygon's boundary, so ONE vertex per ring
//      decides containment (upstream tests every endpoint; the outcome is identical) — evaluated cooperatively.
// Neighbouring polygons overlap in a small window, which takes the O(n*m) pruning loop down to the few segments
// that can matter.  seg_list: 2 * PP_LIST double4 owned by this group (all lanes of a group sit in one wave).
constexpr int PP_VOTE = 4;   // list entries between two group votes in the cross test
#ifndef GPK_PP_LIST
#define GPK_PP_LIST 32
#endif
constexpr int PP_LIST = GPK_PP_LIST;  // per list; a group owns two lists (2 * PP_LIST double4 = 2 KB)
template <int G>
__device__ inline bool polygon_intersects_polygon_group(const DevGeo& a, int ar0, int ar1, const DevGeo& b, int br0, int br1, int lane,
                                                        double4* __restrict__ seg_list, bool have_a_box = false,
                                                        double4 a_box = double4{0, 0, 0, 0}, bool have_b_box = false,
                                                        double4 b_box = double4{0, 0, 0, 0}) {
    static_assert(G <= 64 && (G & (G - 1)) == 0 && PP_LIST >= G, "group size");
    if (ar1 <= ar0 || br1 <= br0) return false;
    const int a_c0 = a.ring_off[ar0], a_c1 = a.ring_off[ar1];
    const int b_c0 = b.ring_off[br0], b_c1 = b.ring_off[br1];
    if (a.ring_off[ar0 + 1] == a_c0 || b.ring_off[br0 + 1] == b_c0) return false;  // empty exterior
    double4 ea, fa, eb, fb;
    polygon_bboxes_group<G>(a, ar0, ar1, lane, have_a_box, a_box, ea, fa);
    polygon_bboxes_group<G>(b, br0, br1, lane, have_b_box, b_box, eb, fb);
    if (ea.z < eb.x || ea.w < eb.y || eb.z < ea.x || eb.w < ea.y) return false;  // has_disjoint_bboxes (exteriors)

    const int gbase = (int)(threadIdx.x & 63) & ~(G - 1);  // first lane of this group within its wave
    const unsigned long long gmask_all = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    double4* __restrict__ list_b = seg_list;            // in-window segments of B
    double4* __restrict__ list_a = seg_list + PP_LIST;  // in-window segments of A
    auto lds_sync = [] {  // the lanes of a group sit in one wave: a compiler-level fence orders the LDS traffic
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // one round of compaction: lane's segment (c, c + 1) of polygon `g` is appended to `list` when its box meets `box`
    auto append_round = [&](const DevGeo& g, int r0, int r1, int c_end, int c, const double4& box, double4* list, int& m) {
        bool keep = false;
        double2 q0 = make_double2(0.0, 0.0), q1 = q0;
        if (c + 1 < c_end) {
            int r = r0;  // ring of c: few rings; find by scan
            while (r + 1 < r1 && g.ring_off[r + 1] <= c) ++r;
            if (c + 1 < g.ring_off[r + 1]) {
                q0 = g.xy[c];
                q1 = g.xy[c + 1];
                keep = !(fmax(q0.x, q1.x) < box.x || fmin(q0.x, q1.x) > box.z || fmax(q0.y, q1.y) < box.y || fmin(q0.y, q1.y) > box.w);
            }
        }
        const unsigned long long mine = (__ballot(keep) >> gbase) & gmask_all;
        if (keep) list[m + __popcll(mine & ((1ull << lane) - 1ull))] = make_double4(q0.x, q0.y, q1.x, q1.y);
        m += __popcll(mine);
    };
    // every segment of list_a[0..ma) against every segment of list_b[0..mb): lane k owns entries k, k + G, ... of A
    auto cross = [&](int ma, int mb) -> bool {
        lds_sync();
        bool hit = false;
        for (int i0 = 0; i0 < ma && !hit; i0 += G) {
            const bool active = i0 + lane < ma;
            const double4 pa = active ? list_a[i0 + lane] : make_double4(0.0, 0.0, 0.0, 0.0);
            const double2 p0 = make_double2(pa.x, pa.y), p1 = make_double2(pa.z, pa.w);
            const double plx = fmin(p0.x, p1.x), phx = fmax(p0.x, p1.x), ply = fmin(p0.y, p1.y), phy = fmax(p0.y, p1.y);
            // B's list in steps of PP_VOTE entries with a group vote in between: most candidate pairs do intersect, and
            // the first crossing found ends the whole pair
            for (int e0 = 0; e0 < mb && !hit; e0 += PP_VOTE) {
                bool found = false;
                if (active) {
                    const int e1 = e0 + PP_VOTE < mb ? e0 + PP_VOTE : mb;
                    for (int e = e0; e < e1; ++e) {
                        const double4 q = list_b[e];
                        if (fmax(q.x, q.z) < plx || fmin(q.x, q.z) > phx || fmax(q.y, q.w) < ply || fmin(q.y, q.w) > phy) continue;
                        if (line_intersects_line(p0, p1, make_double2(q.x, q.y), make_double2(q.z, q.w))) {
                            found = true;
                            break;
                        }
                    }
                }
                hit = group_any<G>(found);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the lists may be overwritten after this point
        return hit;
    };
    // all of A (compacted in chunks of at most PP_LIST) against the current chunk of B
    auto against_a = [&](int mb) -> bool {
        int ma = 0;
        for (int i0 = a_c0; i0 < a_c1; i0 += G) {
            append_round(a, ar0, ar1, a_c1, i0 + lane, fb, list_a, ma);
            if (ma + G > PP_LIST) {
                if (cross(ma, mb)) return true;
                ma = 0;
            }
        }
        return ma > 0 && cross(ma, mb);
    };

    int mb = 0;  // entries in list_b (uniform within the group)
    for (int j0 = b_c0; j0 < b_c1; j0 += G) {
        append_round(b, br0, br1, b_c1, j0 + lane, fa, list_b, mb);
        if (mb + G > PP_LIST) {  // the next round might not fit
            if (against_a(mb)) return true;
            mb = 0;
        }
    }
    if (mb > 0 && against_a(mb)) return true;

    // containment: one vertex per ring of B against A, then A's exterior against B (a vertex outside the other polygon's exterior box
    // is outside the polygon: no walk)
    for (int rb = br0; rb < br1; ++rb) {
        const int c = b.ring_off[rb];
        if (b.ring_off[rb + 1] > c) {
            const double2 q = b.xy[c];
            if (GPK_PP_BOX_SKIP && !(q.x >= ea.x && q.x <= ea.z && q.y >= ea.y && q.y <= ea.w)) continue;
            if (polygon_pos_group<G>(a, ar0, ar1, q.x, q.y, lane) != dev::POS_OUTSIDE) return true;
        }
    }
    const double2 p = a.xy[a_c0];
    if (GPK_PP_BOX_SKIP && !(p.x >= eb.x && p.x <= eb.z && p.y >= eb.y && p.y <= eb.w)) return false;
    return polygon_pos_group<G>(b, br0, br1, p.x, p.y, lane) != dev::POS_OUTSIDE;
}

// ---- the common case of a polygon x polygon join: two single-ring polygons of at most PP_SMALL coordinates -------------------
// Same boolean as polygon_intersects_polygon_group, but both rings are staged ONCE in the group's LDS slice (coalesced requests,
// the two rings' requests in flight together) and everything after that — window clipping, the segment cross test, the two
// containment tests — reads LDS.  The general routine walks global memory at every step (ring offsets per lane and round, the
// coordinates again for the containment test): eight dependent round trips per candidate pair against two here, and the refine
// of a 1M x 1M join is a latency chain, not arithmetic.
constexpr int PP_SMALL = 66;  // coordinates per ring (closed: 65 edges) the small-pair path stages: 36 KB per work-group, four per CU
struct PairSmallLds {
    double2 a[PP_SMALL], b[PP_SMALL];
    uint8_t la[PP_SMALL], lb[PP_SMALL];  // in-window segments (their first coordinate's index)
};
template <int G>
// a_staged: ring A is in t->a already (the join's refine walks the candidates of one left row one after the other: its ring is staged
// once for all of them — 4.5 candidates a row on C4, so 40 % of the staging requests and their round trip go)
__device__ inline bool polygon_pair_small(const double2* __restrict__ axy, int na, const double2* __restrict__ bxy, int nb, double4 ea, double4 eb, int lane,
                                          PairSmallLds* __restrict__ t, bool a_staged = false) {
    // (the caller has checked: 1 <= na, nb <= PP_SMALL; ea / eb = the rings' boxes)
    if (ea.z < eb.x || ea.w < eb.y || eb.z < ea.x || eb.w < ea.y) {  // has_disjoint_bboxes
        if (!a_staged) {
            for (int i = lane; i < na; i += G) t->a[i] = axy[i];  // (the caller counts on A being there afterwards)
        }
        return false;
    }
    // (all rounds of both rings requested up front — five fixed rounds of clamped loads — was slower: 3.74 against 3.67 ms)
    if (!a_staged)
        for (int i = lane; i < na; i += G) t->a[i] = axy[i];
    for (int i = lane; i < nb; i += G) t->b[i] = bxy[i];
    const int gbase = (int)(threadIdx.x & 63) & ~(G - 1);
    const unsigned long long gmask_all = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    lds_sync();
    // window clipping: a segment of one ring can only meet the other ring inside the other ring's box
    auto clip = [&](const double2* ring, int n, const double4& box, uint8_t* list) {
        int m = 0;
        for (int i0 = 0; i0 + 1 < n; i0 += 2 * G) {  // two rounds per trip: four LDS reads in flight instead of two
            const int i = i0 + lane, i2 = i + G;
            const bool in1 = i + 1 < n, in2 = i2 + 1 < n;
            const double2 q0 = ring[in1 ? i : 0], q1 = ring[in1 ? i + 1 : 0], r0 = ring[in2 ? i2 : 0], r1 = ring[in2 ? i2 + 1 : 0];
            const bool keep = in1 && !(fmax(q0.x, q1.x) < box.x || fmin(q0.x, q1.x) > box.z || fmax(q0.y, q1.y) < box.y || fmin(q0.y, q1.y) > box.w);
            const bool keep2 = in2 && !(fmax(r0.x, r1.x) < box.x || fmin(r0.x, r1.x) > box.z || fmax(r0.y, r1.y) < box.y || fmin(r0.y, r1.y) > box.w);
            const unsigned long long mine = (__ballot(keep) >> gbase) & gmask_all;
            if (keep) list[m + __popcll(mine & ((1ull << lane) - 1ull))] = (uint8_t)i;
            m += __popcll(mine);
            const unsigned long long mine2 = (__ballot(keep2) >> gbase) & gmask_all;
            if (keep2) list[m + __popcll(mine2 & ((1ull << lane) - 1ull))] = (uint8_t)i2;
            m += __popcll(mine2);
        }
        return m;
    };
    const int ma = clip(t->a, na, eb, t->la), mb = clip(t->b, nb, ea, t->lb);
    lds_sync();
    bool hit = false;
    for (int i0 = 0; i0 < ma && !hit; i0 += G) {
        const bool active = i0 + lane < ma;
        const int ia = active ? (int)t->la[i0 + lane] : 0;
        const double2 p0 = t->a[ia], p1 = t->a[active ? ia + 1 : 0];
        const double plx = fmin(p0.x, p1.x), phx = fmax(p0.x, p1.x), ply = fmin(p0.y, p1.y), phy = fmax(p0.y, p1.y);
        for (int e0 = 0; e0 < mb && !hit; e0 += PP_VOTE) {
            bool found = false;
            if (active) {
                const int e1 = e0 + PP_VOTE < mb ? e0 + PP_VOTE : mb;
                // the PP_VOTE entries of a trip are requested together before any is tested (an entry is two dependent LDS reads — index,
                // then coordinates — and three waves per SIMD do not hide them: one entry per trip 4.07 ms, two 3.76 ms)
                static_assert(PP_VOTE == 4, "the cross test is written for four entries per vote");
                const int last = e1 - 1;
                const int i0 = (int)t->lb[e0], i1 = (int)t->lb[e0 + 1 < last ? e0 + 1 : last], i2 = (int)t->lb[e0 + 2 < last ? e0 + 2 : last], i3 = (int)t->lb[last];
                const double2 qa0 = t->b[i0], qa1 = t->b[i0 + 1], qb0 = t->b[i1], qb1 = t->b[i1 + 1];
                const double2 qc0 = t->b[i2], qc1 = t->b[i2 + 1], qd0 = t->b[i3], qd1 = t->b[i3 + 1];
                auto meets = [&](const double2 q0, const double2 q1) {
                    if (fmax(q0.x, q1.x) < plx || fmin(q0.x, q1.x) > phx || fmax(q0.y, q1.y) < ply || fmin(q0.y, q1.y) > phy) return false;
                    return line_intersects_line(p0, p1, q0, q1);
                };
                // (entries past the list's end repeat its last one: the same answer again)
                found = meets(qa0, qa1) || meets(qb0, qb1) || meets(qc0, qc1) || meets(qd0, qd1);
            }
            hit = group_any<G>(found);
        }
    }
    if (!hit) {
        // no boundary pair touches: one vertex per ring decides containment (see polygon_intersects_polygon_group).  A vertex outside the
        // other ring's BOX is outside the ring: neighbours whose boxes merely overlap — most candidates that do not intersect — skip both
        // ring walks (round 6: the walks were a third of the refine)
        const double2 q = t->b[0];
        if (!GPK_PP_BOX_SKIP || (q.x >= ea.x && q.x <= ea.z && q.y >= ea.y && q.y <= ea.w)) hit = coord_pos_ring_group<G>(t->a, na, q.x, q.y, lane) != dev::POS_OUTSIDE;
        if (!hit) {
            const double2 p = t->a[0];
            if (!GPK_PP_BOX_SKIP || (p.x >= eb.x && p.x <= eb.z && p.y >= eb.y && p.y <= eb.w)) hit = coord_pos_ring_group<G>(t->b, nb, p.x, p.y, lane) != dev::POS_OUTSIDE;
        }
    }
    __builtin_amdgcn_wave_barrier();  // the slice may be overwritten after this point
    return hit;
}

template <int G>
// a_boxes / b_boxes (optional): per-geometry exterior bounds (gpk_bounds layout); used for Polygon rows, where the
// geometry's bounds ARE its one exterior's box
__device__ inline bool polygonal_intersects_polygonal_group(const DevGeo& a, int64_t ia, const DevGeo& b, int64_t ib, int lane,
                                                            double4* __restrict__ seg_list, const double4* a_boxes = nullptr,
                                                            const double4* b_boxes = nullptr) {
    const double4 abox = a_boxes ? a_boxes[ia] : make_double4(0, 0, 0, 0), bbox = b_boxes ? b_boxes[ib] : make_double4(0, 0, 0, 0);
    const bool ah = a_boxes && a.type == GPK_GEOM_POLYGON, bh = b_boxes && b.type == GPK_GEOM_POLYGON;
    int a0, a1, b0, b1;
    dev::geom_parts(a, ia, a0, a1);
    dev::geom_parts(b, ib, b0, b1);
    for (int p = a0; p < a1; ++p) {
        int ar0, ar1;
        dev::part_rings(a, p, ar0, ar1);
        for (int q = b0; q < b1; ++q) {
            int br0, br1;
            dev::part_rings(b, q, br0, br1);
            if (polygon_intersects_polygon_group<G>(a, ar0, ar1, b, br0, br1, lane, seg_list, ah, abox, bh, bbox)) return true;
        }
    }
    return false;
}

}  // namespace gpk
