// gpk_lineal.h — Contains<Coord> for Line / LineString / MultiLineString, shared by the join's refine (gpk_join.hip) and
// the row-wise predicates (gpk_rowwise.hip).
#pragma once

#include "gpk_device.h"

namespace gpk {

// Contains<Coord> for Line / LineString / MultiLineString (geo 0.27 algorithm/contains/{line,line_string}.rs), reached from
// the join dispatch spatial_index.rs:126-135 (`line.contains(point)` whichever side the point is on): the point lies on
// the linestring and is not one of its two end points (unless the linestring is closed).  Exact orientation.
__device__ inline bool line_contains_coord(double2 s, double2 e, double px, double py) {
    if (s.x == e.x && s.y == e.y) return s.x == px && s.y == py;
    if ((px == s.x && py == s.y) || (px == e.x && py == e.y)) return false;
    return dev::orient2d(s.x, s.y, e.x, e.y, px, py) == 0 && dev::value_in_between(px, s.x, e.x) && dev::value_in_between(py, s.y, e.y);
}
__device__ inline bool linestring_contains_coord(const double2* __restrict__ xy, int n, double px, double py) {
    if (n == 0) return false;
    const double2 f = xy[0], l = xy[n - 1];
    if ((px == f.x && py == f.y) || (px == l.x && py == l.y)) return f.x == l.x && f.y == l.y;
    for (int i = 0; i + 1 < n; ++i) {
        const double2 a = xy[i], b = xy[i + 1];
        if (line_contains_coord(a, b, px, py)) return true;
        if (i > 0 && px == a.x && py == a.y) return true;
    }
    return false;
}
__device__ inline bool lineal_contains_point(const DevGeo& a, int64_t g, double px, double py) {
    if (a.type == GPK_GEOM_LINESTRING) return linestring_contains_coord(a.xy + a.geom_off[g], a.geom_off[g + 1] - a.geom_off[g], px, py);
    for (int l = a.geom_off[g]; l < a.geom_off[g + 1]; ++l)  // MULTILINESTRING: any member
        if (linestring_contains_coord(a.xy + a.ring_off[l], a.ring_off[l + 1] - a.ring_off[l], px, py)) return true;
    return false;
}

}  // namespace gpk
