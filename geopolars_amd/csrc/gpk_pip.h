// gpk_pip.h — exact point-vs-part tests driven by the slab tables of PipView (gpk_index.h).
//
// A slab holds every edge of one ring whose closed y-range meets one slab row of that ring; the point's row is
// computed with the same monotone function, so the slab is a superset of the edges that
// coord_pos_relative_to_ring (geo 0.27) can count or report as boundary for that point.  Walking the
// slab with dev::ring_edge therefore returns exactly the ring position of the full ring walk.
#pragma once

#include "gpk_device.h"
#include "gpk_index.h"

namespace gpk {
namespace pip {

// FINEST slab row of a y coordinate (an exact power-of-two refinement of the raster row function); a ring's own slab
// row is this value shifted right by PIP_FINE_LOG2 - shift[ring]
__device__ __forceinline__ int row_of(const PipView& pv, double py) {
    return dev::cell_of(py, pv.ry0, pv.inv_fh * (double)(PIP_SLAB_MUL << PIP_FINE_LOG2), pv.R * (PIP_SLAB_MUL << PIP_FINE_LOG2));
}
__device__ __forceinline__ int col_of(const PipView& pv, double px) { return dev::cell_of(px, pv.rx0, pv.inv_fw, pv.R); }

// `row` = finest slab row (row_of)
__device__ __forceinline__ bool slab_range(const PipView& pv, int r, int row, int& e0, int& e1) {
    const int base = pv.ring_slab_base[r], ns = pv.ring_slab_base[r + 1] - base;
    const int rr = pv.ring_row0[r];
    const int j = (row >> (PIP_FINE_LOG2 - slab_shift_of(rr))) - slab_row0_of(rr);
    if (j < 0 || j >= ns) return false;  // p.y outside the ring's y-range: Outside, cannot be on it
    e0 = pv.slab_off[base + j];
    e1 = pv.slab_off[base + j + 1];
    return true;
}
// every edge of ring r registered in a slab row that lies in raster row cj: one contiguous range (consecutive rows of a
// ring are adjacent in slab_off; an edge that spans several of them appears once per row)
__device__ __forceinline__ bool slab_span_of_raster_row(const PipView& pv, int r, int cj, int& e0, int& e1) {
    const int base = pv.ring_slab_base[r], ns = pv.ring_slab_base[r + 1] - base;
    const int rr = pv.ring_row0[r];
    const int sh = slab_shift_of(rr);
    int first = cj * (PIP_SLAB_MUL << sh) - slab_row0_of(rr), last = first + (PIP_SLAB_MUL << sh) - 1;
    if (last < 0 || first >= ns) return false;
    first = first < 0 ? 0 : first;
    last = last >= ns ? ns - 1 : last;
    e0 = pv.slab_off[base + first];
    e1 = pv.slab_off[base + last + 1];
    return e1 > e0;
}

__device__ __forceinline__ int slab_vertex(int v) { return v < 0 ? ~v : v; }  // coordinate index a slab_vidx entry names
// slab entry k as an edge (x0, y0, x1, y1): a stored copy, or the two coordinates its index names (PipView::slab_vidx)
__device__ __forceinline__ double4 slab_edge(const PipView& pv, int k) {
    if (pv.slab_edges) return pv.slab_edges[k];
    const int v = pv.slab_vidx[k];
    const double2 a = pv.slab_xy[v < 0 ? ~v : v], b = pv.slab_xy[v < 0 ? ~v : v + 1];
    return make_double4(a.x, a.y, b.x, b.y);
}

// ---- one lane --------------------------------------------------------------------------------
__device__ inline int ring_pos_single(const PipView& pv, int r, double px, double py, int row) {
    int e0, e1;
    if (!slab_range(pv, r, row, e0, e1)) return dev::POS_OUTSIDE;
    int wn = 0;
    bool on = false;
    for (int k = e0; k < e1; ++k) {
        const double4 ed = slab_edge(pv, k);
        on |= dev::ring_edge(ed.x, ed.y, ed.z, ed.w, px, py, wn);
    }
    if (on) return dev::POS_BOUNDARY;
    return wn == 0 ? dev::POS_OUTSIDE : dev::POS_INSIDE;
}

// Polygon::coordinate_position for one part (exterior, then holes)
__device__ inline int part_pos_single(const PipView& pv, const DevGeo& a, int part, double px, double py) {
    int r0, r1;
    dev::part_rings(a, part, r0, r1);
    if (r1 <= r0) return dev::POS_OUTSIDE;
    const int row = row_of(pv, py);
    const int pe = ring_pos_single(pv, r0, px, py, row);
    if (pe != dev::POS_INSIDE) return pe;
    for (int r = r0 + 1; r < r1; ++r) {
        const int ph = ring_pos_single(pv, r, px, py, row);
        if (ph == dev::POS_BOUNDARY) return dev::POS_BOUNDARY;
        if (ph == dev::POS_INSIDE) return dev::POS_OUTSIDE;
    }
    return dev::POS_INSIDE;
}

// ---- GS lanes cooperating on one (point, part) pair --------------------------------------------------
// All GS lanes of the group must call with identical (r / part, point); lane k reads edges k, k+GS, ...
// so one group reads a slab as GS consecutive 32-byte records (two cache lines for GS = 8).
template <int GS>
__device__ __forceinline__ int ring_pos_group(const PipView& pv, int r, double px, double py, int row, int lane) {
    int e0, e1;
    if (!slab_range(pv, r, row, e0, e1)) return dev::POS_OUTSIDE;
    int wn = 0, on = 0;
    for (int k = e0 + lane; k < e1; k += GS) {
        const double4 ed = slab_edge(pv, k);
        on |= (int)dev::ring_edge(ed.x, ed.y, ed.z, ed.w, px, py, wn);
    }
    {  // one packed reduction: winding sum in the high half, on-boundary count in the low half
        const int packed = dev::group_sum<GS>(wn * 65536 + on);
        wn = packed >> 16;
        on = packed & 0xFFFF;
    }
    if (on) return dev::POS_BOUNDARY;
    return wn == 0 ? dev::POS_OUTSIDE : dev::POS_INSIDE;
}

// exterior slab already resolved to the edge range [e0, e0 + cnt): walk it, then the holes if needed
template <int GS>
__device__ __forceinline__ int part_pos_group_from_edges(const PipView& pv, const DevGeo& a, int part, int n_rings, int e0, int cnt,
                                                         double px, double py, int lane) {
    int wn = 0, on = 0;
    for (int k = lane; k < cnt; k += GS) {
        const double4 ed = slab_edge(pv, e0 + k);
        on |= (int)dev::ring_edge(ed.x, ed.y, ed.z, ed.w, px, py, wn);
    }
    {  // one packed reduction: winding sum in the high half, on-boundary count in the low half
        const int packed = dev::group_sum<GS>(wn * 65536 + on);
        wn = packed >> 16;
        on = packed & 0xFFFF;
    }
    if (on) return dev::POS_BOUNDARY;
    if (wn == 0) return dev::POS_OUTSIDE;
    if (n_rings <= 1) return dev::POS_INSIDE;
    int r0, r1;
    dev::part_rings(a, part, r0, r1);
    const int row = row_of(pv, py);
    for (int r = r0 + 1; r < r1; ++r) {
        const int ph = ring_pos_group<GS>(pv, r, px, py, row, lane);
        if (ph == dev::POS_BOUNDARY) return dev::POS_BOUNDARY;
        if (ph == dev::POS_INSIDE) return dev::POS_OUTSIDE;
    }
    return dev::POS_INSIDE;
}

template <int GS>
__device__ __forceinline__ int part_pos_group(const PipView& pv, const DevGeo& a, int part, double px, double py, int lane) {
    int r0, r1;
    dev::part_rings(a, part, r0, r1);
    if (r1 <= r0) return dev::POS_OUTSIDE;
    const int row = row_of(pv, py);
    const int pe = ring_pos_group<GS>(pv, r0, px, py, row, lane);
    if (pe != dev::POS_INSIDE) return pe;
    for (int r = r0 + 1; r < r1; ++r) {
        const int ph = ring_pos_group<GS>(pv, r, px, py, row, lane);
        if (ph == dev::POS_BOUNDARY) return dev::POS_BOUNDARY;
        if (ph == dev::POS_INSIDE) return dev::POS_OUTSIDE;
    }
    return dev::POS_INSIDE;
}

}  // namespace pip
}  // namespace gpk
