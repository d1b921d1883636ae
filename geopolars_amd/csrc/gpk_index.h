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

namespace dev {
// Monotone non-decreasing in v (subtract, multiply by a non-negative constant, floor, clamp), so
// minx <= px <= maxx implies cell(minx) <= cell(px) <= cell(maxx): no candidate can be missed.
__device__ __forceinline__ int cell_of(double v, double v0, double inv, int g) {
    const double f = floor((v - v0) * inv);
    if (!(f >= 0.0)) return 0;
    if (f >= (double)g) return g - 1;
    return (int)f;
}
}  // namespace dev

}  // namespace gpk

struct gpk_index {
    gpk::IndexView v;
    int device;
    int64_t n_geoms;
    int32_t geom_type;
    void* owned[4];  // bbox, grid, cell_off, items
    int64_t nbytes;
};
