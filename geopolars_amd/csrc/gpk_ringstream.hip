// gpk_ringstream.hip — area / signed_area / euclidean_length / bounds of a POLYGON or MULTIPOLYGON column as ONE pass over the
// coordinate buffer in storage order (geoseries.rs:14-16,28-41,188-190,200-202; geo 0.27 area.rs / bounding_rect.rs /
// euclidean_length.rs semantics, the same arithmetic per edge and per geometry as gpk_unary.hip's two-stage form).
//
// Why a second form: gpk_unary.hip gives every RING a lane group sized by the ring's length (rings bucketed into classes once per
// column) and folds the rings of a geometry in a second launch.  On a ragged column (ring lengths Pareto-distributed: most rings a
// handful of coordinates, a few of 100 000) the groups read short, scattered runs, the classes interleave in memory, and the second
// launch re-reads a table the first one wrote: 0.34 - 0.41 of 8 TB/s where a column of equal rings reaches 0.66.  Here the unit of work
// is a STRIP of RS_STRIP consecutive coordinates, whatever rings they belong to:
//   * a wave takes one strip; a lane takes RS_CPL = 4 CONSECUTIVE coordinates of a 256-coordinate block (loaded coalesced, 16 bytes a lane and
//     1 KB a wave per instruction, the next block requested before this one is worked on, and turned through a padded LDS buffer);
//   * where rings begin is a bit mask of the strip built from the ring offsets (the column's strip table — first ring / first geometry
//     of every strip, built once per handle — says which offsets); a lane walks its coordinates with the mask's bits for them (+ the next lane's first) in a register:
//     rings that begin and end inside the lane are finished there, the ring that crosses lanes is finished by a SEGMENTED SCAN over
//     the lanes' open partial sums, the ring that crosses blocks rides in wave-uniform registers;
//   * ring values land in an LDS table indexed by the ring's number within the strip; at the end of the strip a lane per GEOMETRY that
//     began in the strip folds its rings from that table with the polygon / multipolygon rules and writes the result row: no table in
//     global memory, no second launch;
//   * what crosses a strip boundary — a ring (its partial sums before and after the boundary) or a geometry (the values of its rings
//     that are complete in some strip) — is left in two small global arrays, and a second, tiny launch (one thread per strip boundary)
//     finishes those geometries: one in ~30 on the power-law column.
// Work is balanced by construction (every wave streams the same number of bytes), every load is coalesced, and the sums have a fixed
// order given the column's layout (bit-reproducible run to run; the order differs from the two-stage form's pairing tree, both are
// within the tests' tolerance of the oracle's sequential sums; min / max are exact).
//
// Eligible columns (checked once per handle, gpk_seq_classes::strips_ok): ring offsets start at 0 and end at n_coords, no zero-length
// ring (a strip's ring numbers are then consecutive: ring = first ring of the strip + heads seen), at most RS_CAP rings beginning in
// any strip (rings of 4 coordinates — triangles — are exactly at the cap).  Everything else takes the two-stage form.
#include "gpk_device.h"
#include "gpk_ringstream.h"
#include "gpk_scan.h"

namespace gpk {

namespace {

template <int OP>
constexpr int rs_k() { return OP == RS_BOUNDS ? 4 : 1; }
template <int OP>
struct RsVal {
    double v[rs_k<OP>()];
};
template <int OP>
__device__ __forceinline__ RsVal<OP> rs_identity() {
    RsVal<OP> r;
    if constexpr (OP == RS_BOUNDS) {
        r.v[0] = INFINITY;
        r.v[1] = INFINITY;
        r.v[2] = -INFINITY;
        r.v[3] = -INFINITY;
    } else {
        r.v[0] = 0.0;
    }
    return r;
}
// `left` covers coordinates before `right`'s
template <int OP>
__device__ __forceinline__ RsVal<OP> rs_combine(const RsVal<OP>& left, const RsVal<OP>& right) {
    RsVal<OP> r;
    if constexpr (OP == RS_BOUNDS) {
        r.v[0] = right.v[0] < left.v[0] ? right.v[0] : left.v[0];
        r.v[1] = right.v[1] < left.v[1] ? right.v[1] : left.v[1];
        r.v[2] = right.v[2] > left.v[2] ? right.v[2] : left.v[2];
        r.v[3] = right.v[3] > left.v[3] ? right.v[3] : left.v[3];
    } else {
        r.v[0] = left.v[0] + right.v[0];
    }
    return r;
}
// coordinate p of a ring whose first coordinate is `first`; q = the coordinate stored after p; `tail`: p is the ring's last coordinate
// (no edge leaves it).  The terms of seq_accumulate (gpk_unary.hip): the area term assumes a closed ring — rs_ring_value drops the sum
// of a ring that turns out open, which added nothing in the two-stage form either.
template <int OP>
__device__ __forceinline__ void rs_accumulate(RsVal<OP>& a, double2 p, double2 q, double2 first, bool tail) {
    if constexpr (OP == RS_BOUNDS) {
        a.v[0] = p.x < a.v[0] ? p.x : a.v[0];
        a.v[1] = p.y < a.v[1] ? p.y : a.v[1];
        a.v[2] = p.x > a.v[2] ? p.x : a.v[2];
        a.v[3] = p.y > a.v[3] ? p.y : a.v[3];
    } else if constexpr (OP == RS_LENGTH) {
        const double l = hypot(q.x - p.x, q.y - p.y);
        a.v[0] += tail ? 0.0 : l;
    } else {
        const double sx = p.x - first.x, sy = p.y - first.y;
        const double ex = q.x - first.x, ey = q.y - first.y;
        const double cr = sx * ey - sy * ex;
        if (!tail) a.v[0] += cr;
    }
}
// a ring's value from the sum over its coordinates: twice_signed_ring_area is 0 for a ring that is not closed (area.rs); a ring of one
// or two coordinates that IS closed has only zero terms
template <int OP>
__device__ __forceinline__ RsVal<OP> rs_ring_value(const RsVal<OP>& sum, double2 first, double2 last) {
    if constexpr (OP == RS_AREA || OP == RS_SIGNED_AREA) {
        RsVal<OP> r;
        r.v[0] = (first.x == last.x && first.y == last.y) ? sum.v[0] : 0.0;
        return r;
    } else {
        return sum;
    }
}

// ---- the per-geometry rules (area_combine_kernel / length_combine_kernel / bounds_combine_kernel of gpk_unary.hip, the ring values
// read through `val`) ----
template <int OP, typename F>
__device__ __forceinline__ void rs_geometry(const DevGeo& a, int64_t g, F val, double* __restrict__ out) {
    const bool valid = dev::valid_row(a.validity, g);
    int p0, p1;
    dev::geom_parts(a, g, p0, p1);
    if constexpr (OP == RS_BOUNDS) {
        RsVal<OP> b = rs_identity<OP>();
        bool have = false;
        if (valid)
            for (int p = p0; p < p1; ++p) {
                int r0, r1;
                dev::part_rings(a, p, r0, r1);
                if (r1 <= r0) continue;  // (no zero-length rings in an eligible column)
                have = true;
                const RsVal<OP> e = val(r0);  // Polygon::bounding_rect scans the exterior only
                b.v[0] = fmin(b.v[0], e.v[0]);
                b.v[1] = fmin(b.v[1], e.v[1]);
                b.v[2] = fmax(b.v[2], e.v[2]);
                b.v[3] = fmax(b.v[3], e.v[3]);
            }
        reinterpret_cast<double4*>(out)[g] = have ? make_double4(b.v[0], b.v[1], b.v[2], b.v[3]) : make_double4(NAN, NAN, NAN, NAN);
    } else {
        if (!valid) {
            out[g] = NAN;
            return;
        }
        double v = 0.0;
        for (int p = p0; p < p1; ++p) {
            int r0, r1;
            dev::part_rings(a, p, r0, r1);
            if (r1 <= r0) continue;
            if constexpr (OP == RS_LENGTH) {
                v += val(r0).v[0];  // exterior rings only
            } else {
                double area = val(r0).v[0] / 2.0;
                const bool neg = area < 0.0;
                area = fabs(area);
                for (int r = r0 + 1; r < r1; ++r) area -= fabs(val(r).v[0] / 2.0);
                const double sa = neg ? -area : area;
                v += OP == RS_SIGNED_AREA ? sa : fabs(sa);
            }
        }
        out[g] = v;
    }
}
// first ring of geometry g and the ring after its last one (g = n_geoms: the column's ring count twice)
__device__ __forceinline__ void rs_geom_rings(const DevGeo& a, int64_t g, int& r_begin, int& r_end) {
    if (a.type == GPK_GEOM_MULTIPOLYGON) {
        r_begin = a.part_off[a.geom_off[g]];
        r_end = g < a.n_geoms ? a.part_off[a.geom_off[g + 1]] : r_begin;
    } else {
        r_begin = a.geom_off[g];
        r_end = g < a.n_geoms ? a.geom_off[g + 1] : r_begin;
    }
}

__device__ __forceinline__ void rs_wave_fence() {  // this wave's LDS writes so far are seen by its other lanes' reads that follow
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int rs_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double rs_lane_value(double v, int src_lane) {  // (src_lane wave-uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

constexpr int CPL = RS_CPL;  // consecutive coordinates a lane takes of a block
constexpr uint32_t CMASK = (1u << CPL) - 1u;
constexpr int RS_XY_SLOTS = RS_BLOCK + RS_BLOCK / CPL + 2;  // coordinate k of a block sits at slot k + k / CPL: a lane's CPL + 1 reads are 16 (CPL + 1) bytes apart (no bank
                                                          // conflicts for CPL = 4 or 8); slot (CPL + 1) l + CPL — between lane l's coordinates and lane l + 1's — is lane l's scratch
constexpr int RS_MASK_WORDS = RS_STRIP / 32 + 2;
constexpr int RS_OPEN_WORDS = (RS_CAP + 2 + 31) / 32;

// One more term into a ring's slot of the strip's table.  The table belongs to ONE wave and every update is an LDS read-modify-write
// instruction of that wave (ds_add_f64 / ds_min_f64 / ds_max_f64, nothing returned, nothing waited for): the updates of a slot happen in
// program order, lanes of one instruction in the hardware's fixed order — the sums are reproducible run to run.
template <int OP>
__device__ __forceinline__ void rs_flush(double* s_val, int lid, const RsVal<OP>& v) {
    if constexpr (OP == RS_BOUNDS) {
        __hip_atomic_fetch_min(&s_val[lid * 4 + 0], v.v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_min(&s_val[lid * 4 + 1], v.v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_max(&s_val[lid * 4 + 2], v.v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_max(&s_val[lid * 4 + 3], v.v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
        __hip_atomic_fetch_add(&s_val[lid], v.v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

template <int OP>
__global__ __launch_bounds__(64 * RS_WAVES) void ring_stream_kernel(DevGeo a, const int32_t* __restrict__ ring_first, const int32_t* __restrict__ geom_first,
                                                                    int64_t n_strips, double* __restrict__ ring_vals, double* __restrict__ strip_part,
                                                                    double* __restrict__ out) {
    constexpr int K = rs_k<OP>();
    constexpr bool AREA = OP == RS_AREA || OP == RS_SIGNED_AREA;
    __shared__ double2 s_xy_all[RS_WAVES][RS_XY_SLOTS];
    __shared__ uint32_t s_mask_all[RS_WAVES][RS_MASK_WORDS];
    __shared__ uint32_t s_wpre_all[RS_WAVES][RS_MASK_WORDS];
    __shared__ uint32_t s_open_all[RS_WAVES][RS_OPEN_WORDS];
    __shared__ double s_val_all[RS_WAVES][(RS_CAP + 2) * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t strip = (int64_t)blockIdx.x * RS_WAVES + wave;
    if (strip >= n_strips) return;  // (no work-group barrier anywhere: the waves of a work-group only share its launch)
    double2* const s_xy = s_xy_all[wave];
    uint32_t* const s_mask = s_mask_all[wave];
    uint32_t* const s_wpre = s_wpre_all[wave];
    uint32_t* const s_open = s_open_all[wave];
    double* const s_val = s_val_all[wave];
    const double2* __restrict__ xy = a.xy;
    const int32_t* __restrict__ ring_off = a.ring_off;
    const int64_t n_coords = a.n_coords;
    const int64_t base = strip * RS_STRIP;
    const int s_lo = rs_uniform(ring_first[strip]), s_next = rs_uniform(ring_first[strip + 1]);  // rings [s_lo, s_next) begin in this strip
    const int g_lo = rs_uniform(geom_first[strip]), g_hi = rs_uniform(geom_first[strip + 1]);    // geometries [g_lo, g_hi) begin in it

    // the block's coordinates, requested a block ahead: coordinate r * 64 + lane in round r (1 KB a wave per instruction)
    double2 pre[CPL], pre_x = make_double2(0.0, 0.0);
    auto request = [&](int b) {
        const int64_t b0 = base + (int64_t)b * RS_BLOCK;
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int64_t i = b0 + r * 64 + lane;
            pre[r] = i < n_coords ? xy[i] : make_double2(0.0, 0.0);
        }
        if (lane == 0) pre_x = b0 + RS_BLOCK < n_coords ? xy[b0 + RS_BLOCK] : make_double2(0.0, 0.0);  // (the coordinate after the block: the last edge's far end)
    };
    request(0);

    // where rings begin: one bit per coordinate of the strip (+ one for the coordinate after it: is the strip's last coordinate a ring's last?)
    for (int w = lane; w < RS_MASK_WORDS; w += 64) s_mask[w] = 0u;
    if (lane < RS_OPEN_WORDS) s_open[lane] = 0u;
    for (int i = lane; i < (RS_CAP + 2) * K; i += 64) s_val[i] = rs_identity<OP>().v[i % K];
    rs_wave_fence();
    for (int i = lane; i <= s_next - s_lo; i += 64) {  // (ring s_next too: it begins at the strip's end or later; the column's end is offset n_rings)
        const int64_t c = (int64_t)ring_off[s_lo + i] - base;
        if (c <= RS_STRIP) atomicOr(&s_mask[c >> 5], 1u << (c & 31));
    }
    // the ring that entered the strip (if the strip does not begin with a ring): its first coordinate shifts its area terms
    double2 first_carry = make_double2(0.0, 0.0);
    if (AREA && s_lo > 0) first_carry = xy[ring_off[s_lo - 1]];  // (wave-uniform address)
    rs_wave_fence();
    {  // rings begun before each word of the mask
        const uint32_t cnt = lane < RS_MASK_WORDS ? __popc(s_mask[lane]) : 0u;
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane < RS_MASK_WORDS) s_wpre[lane] = incl - cnt;
    }
    rs_wave_fence();

#pragma unroll 1
    for (int b = 0; b < RS_BLOCKS; ++b) {
        const int64_t b0 = base + (int64_t)b * RS_BLOCK;
        if (b0 >= n_coords) break;
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int k = r * 64 + lane;
            s_xy[k + k / CPL] = pre[r];
        }
        if (lane == 0) s_xy[RS_BLOCK + RS_BLOCK / CPL] = pre_x;
        if (b + 1 < RS_BLOCKS) request(b + 1);
        rs_wave_fence();
        const int pos0 = b * RS_BLOCK + CPL * lane;  // my first coordinate within the strip
        const uint32_t m_lo = s_mask[pos0 >> 5], m_hi = s_mask[(pos0 >> 5) + 1];
        const uint32_t hb = (uint32_t)(((((unsigned long long)m_hi) << 32) | m_lo) >> (pos0 & 31)) & ((2u << CPL) - 1u);  // bit j: a ring begins at my coordinate j (j = 8: at the next lane's first)
        int lid = (int)(s_wpre[pos0 >> 5] + __popc(m_lo & ((1u << (pos0 & 31)) - 1u)));  // rings begun before my first coordinate: its ring's number in the strip (0: the ring that entered the strip)
        double2 c[CPL + 1];
#pragma unroll
        for (int j = 0; j < CPL; ++j) c[j] = s_xy[(CPL + 1) * lane + j];
        c[CPL] = s_xy[(CPL + 1) * lane + CPL + 1];  // (the next lane's first coordinate; lane 63: the coordinate after the block)
        const uint32_t heads8 = hb & CMASK, tails8 = (hb >> 1) & CMASK;
        const int64_t left = n_coords - (b0 + CPL * lane);
        const uint32_t live8 = left >= CPL ? CMASK : (left <= 0 ? 0u : (1u << (int)left) - 1u);
        const uint32_t drop8 = tails8 | ~live8;  // no edge leaves a ring's last coordinate (or the column)
        // the first coordinate of the ring my first coordinate belongs to: the last ring begun in a lane before mine, else the block's carry
        double2 first = first_carry;
        if (AREA) {
            const unsigned long long head_lanes = __ballot(heads8 != 0u);
            const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            if (heads8) {
                double2 h = c[0];
#pragma unroll
                for (int j = 1; j < CPL; ++j)
                    if ((heads8 >> j) & 1u) h = c[j];
                s_xy[(CPL + 1) * lane + CPL] = h;  // (my scratch slot)
            }
            rs_wave_fence();
            if (head_lanes & below) first = s_xy[(CPL + 1) * (63 - __clzll(head_lanes & below)) + CPL];
            if (head_lanes) first_carry = s_xy[(CPL + 1) * (63 - __clzll(head_lanes)) + CPL];
        }
        // my 8 coordinates in order.  The area terms are geo's (twice_signed_ring_area shifts every coordinate by the ring's first one): the same
        // products bit for bit, only the order of the sum differs — a ring whose terms are all exactly 0 (collinear) has area exactly 0.
        // e = my coordinate minus ITS ring's first coordinate; an edge's far end belongs to another ring only when the edge is dropped anyway.
        RsVal<OP> acc = rs_identity<OP>();
        auto flush = [&]() { rs_flush<OP>(s_val, lid, acc); };
        double2 e = make_double2(0.0, 0.0);
        if (AREA) {
            if (hb & 1u) first = c[0];
            e = make_double2(c[0].x - first.x, c[0].y - first.y);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            if ((hb >> j) & 1u) ++lid;
            bool closed_j = true;  // (area: my coordinate j is its ring's first coordinate again — e, its offset from it, is exactly 0)
            if constexpr (OP == RS_BOUNDS) {
                if ((live8 >> j) & 1u) {
                    acc.v[0] = c[j].x < acc.v[0] ? c[j].x : acc.v[0];
                    acc.v[1] = c[j].y < acc.v[1] ? c[j].y : acc.v[1];
                    acc.v[2] = c[j].x > acc.v[2] ? c[j].x : acc.v[2];
                    acc.v[3] = c[j].y > acc.v[3] ? c[j].y : acc.v[3];
                }
            } else if constexpr (OP == RS_LENGTH) {
                const double l = hypot(c[j + 1].x - c[j].x, c[j + 1].y - c[j].y);
                acc.v[0] += ((drop8 >> j) & 1u) ? 0.0 : l;
            } else {
                closed_j = e.x == 0.0 && e.y == 0.0;
                if ((hb >> (j + 1)) & 1u) first = c[j + 1];
                const double2 en = make_double2(c[j + 1].x - first.x, c[j + 1].y - first.y);
                const double cr = e.x * en.y - e.y * en.x;
                acc.v[0] += ((drop8 >> j) & 1u) ? 0.0 : cr;
                e = en;
            }
            if ((tails8 >> j) & 1u) {
                flush();
                if (AREA && !closed_j) atomicOr(&s_open[lid >> 5], 1u << (lid & 31));  // not closed: its area is 0 (area.rs)
                acc = rs_identity<OP>();
            }
        }
        if (!((tails8 >> (CPL - 1)) & 1u) && ((live8 >> (CPL - 1)) & 1u)) flush();  // the run still open at my last coordinate: its ring goes on in the next lane
        rs_wave_fence();  // (the next block overwrites s_xy)
    }
    rs_wave_fence();
    const int lid_end = (int)s_wpre[RS_STRIP / 32];  // rings begun in the strip (the bit of the coordinate after the strip is in the next word: not counted)
    // what the strip leaves for rings that cross its boundaries: [0, K) the sum from its start to the first ring end (the whole strip when no
    // ring ends in it), [K, 2K) the sum after its last ring end (the same whole when none)
    if (lane < K) {
        strip_part[(strip * 2) * K + lane] = s_val[lane];
        strip_part[(strip * 2 + 1) * K + lane] = s_val[lid_end * K + lane];
    }

    // ---- the geometries that began in this strip ----
    const int64_t strip_end = base + RS_STRIP < n_coords ? base + RS_STRIP : n_coords;
    const int s_end = (int64_t)ring_off[s_next] == strip_end ? s_next : s_next - 1;  // rings [s_lo, s_end) are complete in this strip
    auto val = [&](int r) {
        const int l = r - s_lo + 1;
        RsVal<OP> v;
#pragma unroll
        for (int k = 0; k < K; ++k) v.v[k] = s_val[l * K + k];
        if (AREA && ((s_open[l >> 5] >> (l & 31)) & 1u)) v.v[0] = 0.0;
        return v;
    };
    auto spill = [&](int r) {
        const RsVal<OP> v = val(r);
#pragma unroll
        for (int k = 0; k < K; ++k) ring_vals[(int64_t)r * K + k] = v.v[k];
    };
    {  // complete rings of the geometry that entered the strip: gpk_ring_stream_fix folds that geometry
        int r_in, r_unused;
        rs_geom_rings(a, g_lo, r_in, r_unused);
        const int upto_r = r_in < s_end ? r_in : s_end;
        for (int r = s_lo + lane; r < upto_r; r += 64) spill(r);
    }
    for (int64_t g = (int64_t)g_lo + lane; g < g_hi; g += 64) {
        int r_begin, r_end;
        rs_geom_rings(a, g, r_begin, r_end);
        if (r_end <= s_end) {
            rs_geometry<OP>(a, g, val, out);
        } else {  // (at most one: the strip's last geometry) its complete rings
            for (int r = r_begin; r < s_end; ++r) spill(r);
        }
    }
}

// the geometry that began in the strip and did not end in it (false: none)
__device__ __forceinline__ bool rs_crossing_geometry(const DevGeo& a, const int32_t* __restrict__ ring_first, const int32_t* __restrict__ geom_first,
                                                     int64_t strip, int64_t& g, int& r_begin, int& r_end) {
    g = (int64_t)geom_first[strip + 1] - 1;
    if (g < (int64_t)geom_first[strip]) return false;  // no geometry begins in this strip
    rs_geom_rings(a, g, r_begin, r_end);
    const int64_t strip_end = (strip + 1) * RS_STRIP < a.n_coords ? (strip + 1) * RS_STRIP : a.n_coords;
    const int s_next = ring_first[strip + 1];
    const int s_end = (int64_t)a.ring_off[s_next] == strip_end ? s_next : s_next - 1;
    return r_end > s_end;  // (else it ended here: the strip's wave writes its row)
}
// What the second launch needs of such a geometry is laid down ONCE per column (gpk_seq_classes): per strip a header {geometry, where its
// ring records begin, how many}, per ring a record {ring, its first and past-the-end coordinate, "first ring of its polygon"}.  Found from
// the offsets, that is a chain of eight dependent reads (strip table -> geometry offsets -> part offsets -> ring offsets -> values): 25 us
// for a launch that moves a few hundred KB.  From the records it is three: header -> ring records -> ring values.
struct RsCross {
    int32_t g, desc_off, n_desc, pad;
};
struct RsDesc {
    int32_t r, c0, c1, part_first;
};
__global__ void ring_cross_count_kernel(DevGeo a, const int32_t* __restrict__ ring_first, const int32_t* __restrict__ geom_first, int64_t n_strips,
                                        int32_t* __restrict__ counts) {
    const int64_t strip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (strip >= n_strips) return;
    int64_t g;
    int r_begin, r_end;
    counts[strip] = rs_crossing_geometry(a, ring_first, geom_first, strip, g, r_begin, r_end) ? r_end - r_begin : 0;
}
__global__ void ring_cross_fill_kernel(DevGeo a, const int32_t* __restrict__ ring_first, const int32_t* __restrict__ geom_first, int64_t n_strips,
                                       const int32_t* __restrict__ desc_off, RsCross* __restrict__ cross, RsDesc* __restrict__ desc) {
    const int64_t strip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (strip >= n_strips) return;
    int64_t g;
    int r_begin, r_end;
    if (!rs_crossing_geometry(a, ring_first, geom_first, strip, g, r_begin, r_end)) {
        cross[strip] = RsCross{-1, 0, 0, 0};
        return;
    }
    cross[strip] = RsCross{(int32_t)g, desc_off[strip], r_end - r_begin, 0};
    RsDesc* d = desc + desc_off[strip];
    int p0, p1;
    dev::geom_parts(a, g, p0, p1);
    for (int p = p0; p < p1; ++p) {
        int r0, r1;
        dev::part_rings(a, p, r0, r1);
        for (int r = r0; r < r1; ++r) *d++ = RsDesc{r, a.ring_off[r], a.ring_off[r + 1], r == r0 ? 1 : 0};
    }
}

// one thread per strip: the geometry that began in the strip and did not end in it, folded from its ring records with the rules of rs_geometry
template <int OP>
__global__ __launch_bounds__(256) void ring_stream_fix_kernel(DevGeo a, const RsCross* __restrict__ cross, const RsDesc* __restrict__ desc, int64_t n_strips,
                                                              const double* __restrict__ ring_vals, const double* __restrict__ strip_part,
                                                              double* __restrict__ out) {
    constexpr int K = rs_k<OP>();
    const int64_t strip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (strip >= n_strips) return;
    const int4 hv = reinterpret_cast<const int4*>(cross)[strip];
    const int64_t g = hv.x;
    if (g < 0) return;
    const bool valid = dev::valid_row(a.validity, g);
    auto val = [&](const int4 d) {  // (r, c0, c1, part_first)
        const int64_t sa = d.y / RS_STRIP, sb = (d.z - 1) / RS_STRIP;
        RsVal<OP> v;
        if (sa == sb) {  // complete in one strip: that strip's wave left its value
#pragma unroll
            for (int k = 0; k < K; ++k) v.v[k] = ring_vals[(int64_t)d.x * K + k];
            return v;
        }
        const double2 first = a.xy[d.y], last = a.xy[d.z - 1];  // (requested before the sums: one more round trip otherwise)
#pragma unroll
        for (int k = 0; k < K; ++k) v.v[k] = strip_part[(sa * 2 + 1) * K + k];
        for (int64_t t = sa + 1; t <= sb; t += 8) {  // (a ring of 100 000 coordinates: a hundred strips — eight reads in flight, folded in strip order)
            RsVal<OP> p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < K; ++k) p[u].v[k] = t + u <= sb ? strip_part[((t + u) * 2) * K + k] : rs_identity<OP>().v[k];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t + u <= sb) v = rs_combine<OP>(v, p[u]);
        }
        return rs_ring_value<OP>(v, first, last);
    };
    // The ring records eight at a time: the records, then the eight rings' values (as if each were complete in one strip: an address that is
    // always valid), are each ONE round trip for the batch; only a ring that crosses a strip boundary (one or two a geometry) then takes its
    // own.  Record by record the fold of a multipolygon of six rings was twelve dependent round trips.
    const int4* dd = reinterpret_cast<const int4*>(desc) + hv.y;
    RsVal<OP> b = rs_identity<OP>();
    bool have = false;
    double v = 0.0, area = 0.0;
    bool open_part = false, neg = false;
    auto close_part = [&]() {
        const double sa = neg ? -area : area;
        v += OP == RS_SIGNED_AREA ? sa : fabs(sa);
    };
    if (OP != RS_BOUNDS && !valid) {
        out[g] = NAN;
        return;
    }
    const int n_desc = (OP == RS_BOUNDS && !valid) ? 0 : hv.z;
    for (int i0 = 0; i0 < n_desc; i0 += 8) {
        int4 d[8];
        RsVal<OP> rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = dd[i0 + u < n_desc ? i0 + u : n_desc - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k) rv[u].v[k] = ring_vals[(int64_t)d[u].x * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u >= n_desc) break;
            const bool crossing = d[u].y / RS_STRIP != (d[u].z - 1) / RS_STRIP;
            if constexpr (OP == RS_BOUNDS) {
                if (!d[u].w) continue;  // Polygon::bounding_rect scans the exterior only
                have = true;
                const RsVal<OP> e = crossing ? val(d[u]) : rv[u];
                b.v[0] = fmin(b.v[0], e.v[0]);
                b.v[1] = fmin(b.v[1], e.v[1]);
                b.v[2] = fmax(b.v[2], e.v[2]);
                b.v[3] = fmax(b.v[3], e.v[3]);
            } else if constexpr (OP == RS_LENGTH) {
                if (d[u].w) v += (crossing ? val(d[u]) : rv[u]).v[0];  // exterior rings only
            } else {
                const double h = (crossing ? val(d[u]) : rv[u]).v[0] / 2.0;
                if (d[u].w) {
                    if (open_part) close_part();
                    open_part = true;
                    neg = h < 0.0;
                    area = fabs(h);
                } else {
                    area -= fabs(h);
                }
            }
        }
    }
    if constexpr (OP == RS_BOUNDS) {
        reinterpret_cast<double4*>(out)[g] = have ? make_double4(b.v[0], b.v[1], b.v[2], b.v[3]) : make_double4(NAN, NAN, NAN, NAN);
    } else {
        if (OP != RS_LENGTH && open_part) close_part();
        out[g] = v;
    }
}

// ---- the strip table of a column (once per handle) ----
// entry t: the first ring (geometry) whose first coordinate is at or after coordinate t * RS_STRIP, the ring (geometry) count when none is
__global__ void ring_strip_table_kernel(DevGeo a, int64_t n_strips, int32_t* __restrict__ ring_first, int32_t* __restrict__ geom_first,
                                        int32_t* __restrict__ flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_strips) return;
    const int64_t target = t * RS_STRIP;
    int64_t lo = 0, hi = a.n_rings;  // first ring in [0, n_rings) with ring_off[ring] >= target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)a.ring_off[mid] >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    ring_first[t] = (int32_t)lo;
    lo = 0;
    hi = a.n_geoms;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        int r_begin, r_end;
        rs_geom_rings(a, mid, r_begin, r_end);
        if ((int64_t)a.ring_off[r_begin] >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    geom_first[t] = (int32_t)lo;
    if (t == 0 && (a.ring_off[0] != 0 || (int64_t)a.ring_off[a.n_rings] != a.n_coords)) atomicOr(flags, 1);
}
// flags |= 2: a zero-length ring; |= 4: more than RS_CAP rings begin in some strip; |= 8: the offsets above the rings do not cover them from 0
__global__ void ring_strip_check_kernel(DevGeo a, int64_t n_strips, const int32_t* __restrict__ ring_first, int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_rings && a.ring_off[i + 1] <= a.ring_off[i]) atomicOr(flags, 2);
    if (i < n_strips && ring_first[i + 1] - ring_first[i] > RS_CAP) atomicOr(flags, 4);
    if (i == 0) {
        int r_begin, r_end;
        rs_geom_rings(a, 0, r_begin, r_end);
        int r_last, r_unused;
        rs_geom_rings(a, a.n_geoms, r_last, r_unused);
        if (r_begin != 0 || r_last != (int)a.n_rings) atomicOr(flags, 8);
    }
}

}  // namespace

int64_t ring_stream_strips(int64_t n_coords) { return n_coords / RS_STRIP + 1; }  // (position n_coords — where an empty last geometry "begins" — lies in a strip too)

int32_t ring_stream_build_table(const DevGeo& a, int32_t* ring_first, int32_t* geom_first, int32_t* flags_dev, hipStream_t s) {
    const int64_t n_strips = ring_stream_strips(a.n_coords);
    GPK_LAUNCH("gpk_ring_strip_table", ring_strip_table_kernel, dim3((unsigned)((n_strips + 1 + 255) / 256)), dim3(256), 0, s, a, n_strips, ring_first, geom_first,
               flags_dev);
    const int64_t n = a.n_rings > n_strips ? a.n_rings : n_strips;
    GPK_LAUNCH("gpk_ring_strip_check", ring_strip_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n_strips, (const int32_t*)ring_first,
               flags_dev);
    return GPK_OK;
}

// the ring records of the strip-crossing geometries (device allocations the caller keeps with the handle)
int32_t ring_stream_build_cross(const DevGeo& a, const int32_t* ring_first, const int32_t* geom_first, hipStream_t s, void** cross_out, void** desc_out) {
    const int64_t n_strips = ring_stream_strips(a.n_coords);
    int32_t *counts = nullptr, *offs = nullptr;
    unsigned long long* btot = nullptr;
    RsCross* cross = nullptr;
    RsDesc* desc = nullptr;
    auto run = [&]() -> int32_t {
        GPK_HIP(device_malloc((void**)&counts, sizeof(int32_t) * (size_t)(n_strips + 1)));
        GPK_HIP(device_malloc((void**)&offs, sizeof(int32_t) * (size_t)(n_strips + 1)));
        GPK_HIP(device_malloc((void**)&btot, sizeof(unsigned long long) * (size_t)((n_strips + 255) / 256 + 2)));
        GPK_HIP(device_malloc((void**)&cross, sizeof(RsCross) * (size_t)n_strips));
        const dim3 grid((unsigned)((n_strips + 255) / 256));
        GPK_LAUNCH("gpk_ring_cross_count", ring_cross_count_kernel, grid, dim3(256), 0, s, a, ring_first, geom_first, n_strips, counts);
        GPK_TRY(exclusive_scan_i32(counts, n_strips, offs, nullptr, btot, s));
        int32_t total = 0;
        GPK_HIP(hipMemcpyAsync(&total, offs + n_strips, sizeof total, hipMemcpyDeviceToHost, s));
        GPK_HIP(hipStreamSynchronize(s));
        GPK_HIP(device_malloc((void**)&desc, sizeof(RsDesc) * (size_t)(total > 0 ? total : 1)));
        GPK_LAUNCH("gpk_ring_cross_fill", ring_cross_fill_kernel, grid, dim3(256), 0, s, a, ring_first, geom_first, n_strips, (const int32_t*)offs, cross, desc);
        GPK_HIP(hipStreamSynchronize(s));
        return GPK_OK;
    };
    const int32_t rc = run();
    if (counts) (void)hipFree(counts);
    if (offs) (void)hipFree(offs);
    if (btot) (void)hipFree(btot);
    if (rc != GPK_OK) {
        if (cross) (void)hipFree(cross);
        if (desc) (void)hipFree(desc);
        return rc;
    }
    *cross_out = cross;
    *desc_out = desc;
    return GPK_OK;
}

template <int OP>
static int32_t ring_stream_launch_op(const DevGeo& a, const int32_t* ring_first, const int32_t* geom_first, const void* cross, const void* desc,
                                     double* ring_vals, double* strip_part, double* out, hipStream_t s, const char* name) {
    const int64_t n_strips = ring_stream_strips(a.n_coords);
    GPK_LAUNCH(name, ring_stream_kernel<OP>, dim3((unsigned)((n_strips + RS_WAVES - 1) / RS_WAVES)), dim3(64 * RS_WAVES), 0, s, a, ring_first, geom_first, n_strips,
               ring_vals, strip_part, out);
    GPK_LAUNCH("gpk_ring_stream_fix", ring_stream_fix_kernel<OP>, dim3((unsigned)((n_strips + 255) / 256)), dim3(256), 0, s, a, (const RsCross*)cross,
               (const RsDesc*)desc, n_strips, (const double*)ring_vals, (const double*)strip_part, out);
    return GPK_OK;
}

int32_t ring_stream_launch(int op, const DevGeo& a, const int32_t* ring_first, const int32_t* geom_first, const void* cross, const void* desc,
                           double* ring_vals, double* strip_part, double* out, hipStream_t s) {
    switch (op) {
    case RS_AREA: return ring_stream_launch_op<RS_AREA>(a, ring_first, geom_first, cross, desc, ring_vals, strip_part, out, s, "gpk_ring_stream_area");
    case RS_SIGNED_AREA: return ring_stream_launch_op<RS_SIGNED_AREA>(a, ring_first, geom_first, cross, desc, ring_vals, strip_part, out, s, "gpk_ring_stream_area");
    case RS_LENGTH: return ring_stream_launch_op<RS_LENGTH>(a, ring_first, geom_first, cross, desc, ring_vals, strip_part, out, s, "gpk_ring_stream_length");
    case RS_BOUNDS: return ring_stream_launch_op<RS_BOUNDS>(a, ring_first, geom_first, cross, desc, ring_vals, strip_part, out, s, "gpk_ring_stream_bounds");
    default: return fail(GPK_ERR_INVALID_ARGUMENT, "ring_stream_launch: op %d", op);
    }
}

}  // namespace gpk
