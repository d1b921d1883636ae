// gpk_index.h — device-resident spatial index over one geoarray: the GPU replacement for
// `SpatialIndex { r_tree: RTree<TreeNode> }` (geopolars/src/spatial_index.rs:314-350).
//
// rstar answers "which right-side AABBs intersect this left-side AABB" (spatial_index.rs:74-76).
// On the GPU the same candidate set comes from a uniform-grid directory over the right side's
// bounding boxes: cell -> ascending list of geometry ids whose closed bbox overlaps the cell.
// Closed-interval semantics are preserved (KA-2, spatial_index.rs:361-395) because a geometry is
// registered in every cell its bbox touches under the SAME monotone cell function that queries use,
// and each candidate is then re-tested against the exact bbox.
#pragma once

#include "gpk_common.h"

namespace gpk {

struct GridParams {  // lives in device memory; filled by the extent kernel (no host round trip)
    double x0, y0, inv_w, inv_h;
    int32_t gx, gy;
};

struct IndexView {  // passed to kernels by value
    const double4* bbox;      // per geometry: minx, miny, maxx, maxy (NaN = empty)
    const GridParams* grid;   // device
    const int32_t* cell_off;  // gx*gy + 1
    const int32_t* items;     // geometry ids, ascending within a cell
    int32_t gx, gy;
};

// ---- point-in-polygon accelerator (polygonal arrays only) -----------------------------------------
// A fine R x R raster over the (padded) extent plus per-ring edge slabs on the raster rows.
//   cell word  : tag(2) | payload(30).  tag 0 = no polygon can contain a point of this cell;
//                tag 1 = exactly one entry, inline; tag 2 = payload is an offset into `list`
//                (list[off] = n, then n entries); tag 3 = exactly one part's edges cross the cell and
//                payload indexes a 32-byte SubCell record: the cell split 8 x 8 with the same exact
//                labelling, plus the slab location, so most of its points finish with one more gather.
//                (Level 1 is kept coarse — 1 MB at R = 512 — so that its gathers stay L2 hits; the
//                effective resolution is 8R.)
//   entry      : part << 1 | boundary (list cells: see PipView::lrec).  boundary = 0 means EVERY representable point that maps to this
//                cell is strictly inside that part (holes included) — decided once, exactly, at build
//                time; boundary = 1 means some edge of the part may touch the cell: run the exact test.
//   slabs      : for ring r and slab row j in [row0[r], row0[r] + nrows) the list of edges whose closed y-range meets
//                the row (under the same monotone row function the points use), stored as contiguous double4
//                (sx, sy, ex, ey) records -> 8 lanes read one slab with 2 cache lines.  Rows are PER RING:
//                PIP_SLAB_MUL << shift[r] slab rows per raster row, the shift (0..PIP_FINE_LOG2) chosen at build time
//                so that a slab keeps about a dozen edges however many vertices the ring has (a 100k-vertex ring on
//                the base rows holds hundreds of edges per slab, and every boundary-cell point walks them all).
//                Every ring's rows are a right shift of ONE finest row function (exact power-of-two rescales of the
//                raster row function), so edge registration and point lookup agree by monotonicity.  ring_row0[r]
//                and PartInfo::row0 carry `row0 | shift << 24`.
// Exactness does not depend on the raster: it only routes points; every boundary decision is made by
// the exact winding walk over the slab's edges, which are a superset of the edges that can count.
struct PartInfo {  // one 16-byte load tells a lane where the exterior ring's slabs of a part live
    int32_t slab_base, row0 /* | shift << 24 */, nrows, n_rings;
};
constexpr int PIP_SUB = 8;       // level-2 sub-cells per raster cell side
constexpr int PIP_SLAB_MUL = 2;  // base slab rows per raster row (slabs stay ~7 edges while the level-1 table stays small)
constexpr int PIP_FINE_LOG2 = 6;  // finest slab rows: PIP_SLAB_MUL << 6 = 128 per raster row
__host__ __device__ inline int slab_row0_of(int32_t packed) { return packed & 0xFFFFFF; }
__host__ __device__ inline int slab_shift_of(int32_t packed) { return packed >> 24; }
constexpr uint32_t SUB_INDIRECT = 1u << 30;  // in a SubCell part word: slab ranges not inline (refined ring): go through PartInfo
struct SubCell {  // level-2 record of a raster cell crossed by edges of exactly ONE part: 32 bytes, one cache line touch
    uint32_t part_flags;  // part | SUB_INDIRECT | (part has holes) << 31
    uint32_t e0, e1, e2;  // exterior-ring slab of the cell's lower base slab row = [e0, e1), upper = [e1, e2) (shift 0 only)
    uint32_t labels[4];   // 8 x 8 sub-cells, 2 bits each (x fastest): 0 outside, 1 strictly inside, 2 test exactly
};
// level-2 record of a raster cell crossed by exactly TWO parts (and holding nothing else): the shared borders of a
// tessellation — administrative boundaries, census tracts — where every boundary cell is of this kind.
// 64 bytes; the first 32 are laid out like SubCell for slot A, so a kernel loads it the same way.
//   labels: 0 in neither, 1 strictly inside A, 2 strictly inside B, 3 test both exactly
struct SubCell2 {
    SubCell a;              // slot A: part_flags, its slab ranges, the labels
    uint32_t b_part_flags;  // part B | (B has holes) << 31
    uint32_t b_e0, b_e1, b_e2;
    uint32_t pad[4];
};
constexpr uint32_t SUB2_BIT = 1u << 29;         // in a CELL_TAG_SUB payload: the index refers to PipView::sub2
// "test" sub-cells of a lean index (one part per cell), decided in the owning lane: for the padded sub-cell Q the LOCAL CHAIN is
// the run of ring edges that covers every edge meeting Q, grown at both ends while the end vertex's y lies in Q's y-interval;
// then winding(p) = base + sum of the chain edges' contributions for every p of Q (the rule is checked on
// the CPU by tools/proto_local_chain.py / tests/test_local_chain_rule.py).  count == 0: no single short chain (several runs,
// more than CHAIN_MAX edges, a part with holes): the row is decided by the generic walk, by the whole wave at the end of its tile.
constexpr int CHAIN_MAX = 12;
// chain entry i of an index = chain_head[i] (count, base, where vertices 4 .. are) + sub_aux[i] (the first four vertices, one cache
// line): a `test` point reads both with independent requests and needs nothing else — 99.8 % of the chains of the C2 right side
// have at most three edges.
constexpr uint32_t CHAIN_COUNT_MASK = 0xFu;  // bits 0-3: edges in the chain, 1 .. CHAIN_MAX (0 = no chain entry: the generic walk decides)
constexpr int CHAIN_BASE_SHIFT = 4;          // bits 4-11: summed winding contribution of every ring edge outside the chain (signed):
                                             // constant over the padded sub-cell
constexpr int CHAIN_EXT_SHIFT = 12;          // bits 12-31: count > 3: PipView::chain_ext[ext .. ext + count - 3) are vertices 4 .. count
// GPK_HALF_CHAINS (round 4, the default): ONE chain per half-cell record instead of one per `test` sub-cell.  The chain is the arc of
// the ring that covers every edge meeting the padded HALF CELL (grown at both ends while the end vertex's y lies in the half's
// y-interval); the same argument gives winding(p) = base + sum over the chain for every p of the half cell, and an on-boundary
// point lies on a chain edge.  The chain is named by a word IN the record (HalfCell::aux_base) — count, base, and where its
// vertices start in PipView::chain_xy, the right side's coordinates with CHAIN_MAX more per ring (a chain may run over the closing
// vertex: vertex k of the extended ring is coordinate k % edges) — so a `test` point needs no head-word request and reads vertices of
// a table as hot as the coordinates themselves (C2: 1.2 MB) instead of a cold 64-byte line of a per-sub-cell table (C2: 56 MB +
// 3.5 MB of head words).  It walks about two edges where the per-sub-cell chain had 1.26.
#ifndef GPK_HALF_CHAINS
#define GPK_HALF_CHAINS 1
#endif
constexpr uint32_t HCHAIN_COUNT_MASK = 0xFu;  // bits 0-3: edges, 1 .. CHAIN_MAX (0: no chain — several runs, too long, a part with holes,
                                              // an unclosed ring: the generic walk decides the half's `test` points)
constexpr int HCHAIN_BASE_SHIFT = 4;          // bits 4-7: the other edges' summed winding contribution, signed
constexpr int HCHAIN_START_SHIFT = 8;         // bits 8-31: first vertex in PipView::chain_xy
struct ChainAux {
    double2 v[4];  // the chain's first four vertices, copied (a chain may run over the ring's closing vertex)
};
// An index with chains keeps its one-part level-2 records (PipView::sub, 32 bytes per raster cell) as TWO half-cell records: the
// labels of four sub-cell rows, the part, and where the half's chain entries start — everything a point needs, in one 16-byte
// request.  (The queue kernels read the SubCell form; an index has one or the other: pip_join_enqueue picks the kernel.)
struct HalfCell {
    uint32_t lw[2];       // 2 x 16 labels (label words 2 * half and 2 * half + 1 of the cell)
    uint32_t part_flags;  // part | (part has holes) << 31
    uint32_t aux_base;    // chain entry of the half's first `test` label (label order)
};
static_assert(sizeof(HalfCell) * 2 == sizeof(SubCell), "two half-cell records overlay one SubCell");
static_assert(sizeof(ChainAux) == 64, "one chain entry per cache line");
static_assert(CHAIN_MAX <= (int)CHAIN_COUNT_MASK, "the edge count of a chain is a 4-bit field");
// Level-1 routing of a small raster (R <= PIP_ROUTE_RMAX) as an LDS image: one 16-byte word per 32 consecutive cells of a
// raster row.  A persistent work-group keeps the whole image in LDS (128 KB at R = 512) and a point learns from ONE LDS
// read whether its cell is empty (nothing to fetch), carries a one-part record (its index = rec0 + rank of the cell's bit:
// records are numbered in cell order) or needs the level-1 word from memory (interiors, the few list cells).
constexpr int PIP_ROUTE_RMAX = 512;
struct RouteWord {
    uint32_t bmask;  // bit i: cell 32 * w + i carries a one-part level-2 record
    uint32_t gmask;  // bit i: the cell's level-1 word must be read (interior of a part, list cell, anything else non-empty)
    uint32_t rec0;   // record index of the first cell set in bmask
    uint32_t pad;
};
struct PipView {
    int32_t R;  // 0 = accelerator not built (degenerate extent): kernels use the generic walk
    double rx0, ry0, fw, fh, inv_fw, inv_fh;
    const uint32_t* cell;
    const uint32_t* list;
    const SubCell* sub;              // level-2 records (cell tag 3)
    const SubCell2* sub2;            // two-part level-2 records (cell tag 3, payload & SUB2_BIT)
    const ChainAux* sub_aux;         // lean indexes with chains (see ChainAux); else nullptr
    const uint32_t* chain_head;      // count | base | ext per chain entry
    const double2* chain_ext;        // vertices 4 .. of the chains longer than three edges
    const double2* chain_xy;         // GPK_HALF_CHAINS: extended ring coordinates the half-cell chains index (then sub_aux / chain_head / chain_ext are null)
    const RouteWord* route;          // LDS image of the level-1 routing (chains + R <= PIP_ROUTE_RMAX); else nullptr
    const SubCell* lrec;             // level-2 records of the BOUNDARY entries of list cells: when set, such an entry is
                                     // `record index << 1 | 1` (the record names the part), else `part << 1 | 1`
    const uint32_t* part_geom;       // nullptr for POLYGON arrays (part == geometry)
    const PartInfo* part_info;       // n_parts
    const float4* part_box;          // per part: its exterior ring's box rounded OUTWARD to f32 (minx, miny, maxx, maxy), or nullptr.
                                     // Built for indexes whose list cells carry no per-entry records: a point outside the box skips
                                     // the entry before PartInfo, the slab offsets and the exact walk are touched
    const int32_t* ring_row0;        // row0 | shift << 24 per ring
    const int32_t* ring_slab_base;   // n_rings + 1
    const int32_t* slab_off;         // n_slabs + 1
    const double4* slab_edges;       // slab entries as edge copies (32 B each) — small right sides; or nullptr:
    const int32_t* slab_vidx;        // slab entries as coordinate indices (4 B each): the edge is (slab_xy[v], slab_xy[v + 1]); ~v for the
                                     // degenerate edge of a one-coordinate ring.  Large right sides: an eighth of the bytes, and consecutive
                                     // edges of a ring share their vertices' cache lines (pip::slab_edge reads either form)
    const double2* slab_xy;          // the indexed array's coordinates (slab_vidx form)
};
constexpr uint32_t CELL_TAG_EMPTY = 0u, CELL_TAG_SINGLE = 1u, CELL_TAG_LIST = 2u, CELL_TAG_SUB = 3u;

namespace dev {
// Monotone non-decreasing in v (subtract, multiply by a non-negative constant, floor, clamp), so
// minx <= px <= maxx implies cell(minx) <= cell(px) <= cell(maxx): no candidate can be missed.
__device__ __forceinline__ int cell_of(double v, double v0, double inv, int g) {
    // branch-free: negative / NaN products clamp to 0 (fmax drops the NaN), large ones to g - 1; the conversion
    // truncates, which is floor() on the clamped non-negative value
    return (int)fmin(fmax((v - v0) * inv, 0.0), (double)(g - 1));
}
}  // namespace dev

}  // namespace gpk

struct gpk_index {
    gpk::IndexView v;
    gpk::PipView pip;
    gpk::GridParams host_grid;
    int device;
    int64_t n_geoms, n_coords, n_rings;  // of the indexed array: a join checks the array it is handed against all three
    int32_t geom_type;
    void* owned[24];  // bbox, grid, cell_off, items, then the PipView tables
    int64_t nbytes;
    int32_t pip_lean;  // 1: every raster cell names at most one part and boundary cells carry inline records (build_pip_index)
    uint64_t serial;  // unique per built index (never reused): what a cached copy of an index's pointers is keyed with
    int32_t pip_list_heavy;  // 1: more than a list word per four raster cells (overlapping parts, very many small parts): the general tile
                             // kernel runs its one-point-per-lane instance
};

namespace gpk {
// gpk_pipindex.hip: builds ix->pip for a polygonal array (no-op otherwise).  ix->v must be complete.
// list_records_mode: 0 never (GPK_INDEX_PIP_LIGHT), 1 where they pay for themselves (default), 2 always (GPK_INDEX_PIP_FULL)
int32_t build_pip_index(const gpk_geoarray* a, gpk_index* ix, hipStream_t s, int list_records_mode = 1);
// gpk_unary.hip: closed bbox of every coordinate sequence (ring) as AoS double4; NaN for empty ones
int32_t ring_bboxes(const gpk_geoarray* a, double4* out_dev, hipStream_t s);
// gpk_unary.hip: one affine matrix per geometry; matrices in `mat_space`, output in `out_space`
int32_t affine_rows_impl(const gpk_geoarray* a, const double* matrices, int32_t mat_space, double* out_xy, int32_t out_space, hipStream_t s);
}  // namespace gpk
