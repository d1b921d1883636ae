// gpk_ringstream.h — the one-pass form of area / signed_area / euclidean_length / bounds over polygonal columns (gpk_ringstream.hip)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gpk_common.h"

namespace gpk {

enum : int { RS_AREA = 0, RS_SIGNED_AREA = 1, RS_LENGTH = 2, RS_BOUNDS = 3 };
#ifndef GPK_RS_CPL
#define GPK_RS_CPL 4
#endif
// consecutive coordinates a lane takes of a block.  4: 102 / 85 registers (area / length) and 30 KB of LDS a work-group — 16 / 20 waves a CU —
// against 156 / 135 and 46 KB — 12 waves — at 8; 19 % more instructions a coordinate and 4 - 9 % less time (the launch is bound by the
// dependent round trips at a strip's ends: more waves hide more of them); 2: as 4 for length, slower for area
constexpr int RS_CPL = GPK_RS_CPL;
constexpr int RS_BLOCK = 64 * RS_CPL;             // coordinates a wave works on at a time
constexpr int RS_BLOCKS = 1024 / RS_BLOCK;        // blocks of a strip
constexpr int RS_STRIP = RS_BLOCK * RS_BLOCKS;   // coordinates of a strip: one wave's job
constexpr int RS_CAP = RS_STRIP / 4;             // rings that may begin in one strip (rings of 4 coordinates — triangles — are at the cap)
constexpr int RS_WAVES = 4;                      // waves of a work-group (they share nothing but the launch)

int64_t ring_stream_strips(int64_t n_coords);
// fills ring_first / geom_first (ring_stream_strips() + 1 entries each) and ORs into *flags_dev what makes the column ineligible
// (0 afterwards: eligible)
int32_t ring_stream_build_table(const DevGeo& a, int32_t* ring_first, int32_t* geom_first, int32_t* flags_dev, hipStream_t s);
// the ring records of the geometries that cross strip boundaries (two device allocations the caller keeps with the handle and hipFree's)
int32_t ring_stream_build_cross(const DevGeo& a, const int32_t* ring_first, const int32_t* geom_first, hipStream_t s, void** cross_out, void** desc_out);
// ring_vals: rs values per ring (4 doubles for RS_BOUNDS, 1 otherwise); strip_part: twice that per strip
int32_t ring_stream_launch(int op, const DevGeo& a, const int32_t* ring_first, const int32_t* geom_first, const void* cross, const void* desc, double* ring_vals,
                           double* strip_part, double* out, hipStream_t s);

}  // namespace gpk
