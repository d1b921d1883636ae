// gpk_common.h — host-side plumbing shared by every translation unit of libgeopolars_hip.so:
// error reporting across the C ABI, the device-resident GeoArrow view, the per-thread scratch
// workspace and the launch wrapper that feeds gpk_profile_*.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/geopolars_hip.h"

namespace gpk {

// ---- errors ------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int32_t fail(int32_t code, const char* fmt, ...);

#define GPK_HIP(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::gpk::fail(_e == hipErrorOutOfMemory ? GPK_ERR_OOM : GPK_ERR_DEVICE,         \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                               __LINE__);                                                        \
    } while (0)

#define GPK_TRY(expr)                   \
    do {                                \
        int32_t _rc = (expr);           \
        if (_rc != GPK_OK) return _rc;  \
    } while (0)

// ---- device-resident GeoArrow view ---------------------------------------------------------
// What every kernel receives by value.  Polygonal arrays are normalised to three levels
// (geom -> part -> ring -> coord); a POLYGON array gets an identity geom->part mapping, signalled by
// part_off == nullptr (geom g owns exactly part g, whose rings are geom_off[g]..geom_off[g+1]).
struct DevGeo {
    int32_t type;
    int64_t n_geoms, n_parts, n_rings, n_coords;
    const double2* xy;        // interleaved coordinates, one 16-byte load per vertex
    const int32_t* geom_off;  // level-1 offsets (nullptr for POINT)
    const int32_t* part_off;  // MULTIPOLYGON only
    const int32_t* ring_off;  // POLYGON / MULTILINESTRING / MULTIPOLYGON
    const uint8_t* validity;  // Arrow bitmap or nullptr
};

}  // namespace gpk

// Coordinate sequences (rings / linestrings) of an array bucketed by length, built on first use by the streaming
// reductions (gpk_unary.hip) and kept with the handle: arrays are immutable, the classification is paid once.
struct gpk_seq_classes {
    int64_t count[4];  // sequences per class: three lane-group classes, then "long" (whole work-group)
    int64_t begin[4];  // start of each class in `lists`
    int32_t* lists;    // device, ids grouped by class; nullptr when a single class holds every sequence
    // the "long" class is cut into chunks of SEQ_CHUNK coordinates, one work-group each (a 100k-vertex ring would
    // otherwise be one work-group's job): chunk_begin[k] = first chunk of the k-th long sequence (count[3] + 1 entries)
    int32_t* chunk_begin;
    int64_t n_chunks;
    bool one_to_one;  // sequence s is exactly geometry s (LINESTRING column; POLYGON column of single-ring polygons)
    // polygonal columns: the strip table of the one-pass reductions (gpk_ringstream.hip): first ring / first geometry of every strip of
    // RS_STRIP coordinates (one allocation, ring entries then geometry entries); strips_ok: the column is eligible for that form
    int32_t* strip_first;
    void *strip_cross, *strip_desc;  // the ring records of the geometries that cross strip boundaries
    bool strips_ok;
};
struct gpk_geoarray {
    gpk::DevGeo d;
    int device;
    void* owned[5];  // hipMalloc'ed copies (xy, geom_off, part_off, ring_off, validity) or nullptr
    int64_t nbytes;
    gpk_seq_classes* classes;  // lazily built (under a lock), freed with the handle
    // gpk_spatial_join called WITHOUT an index (the reference's default: SpatialJoinArgs::default() has r_index: None and builds an R-tree
    // inside every call, spatial_index.rs:24-35,60-71) keeps the index it builds on the right-side handle — handles are immutable after
    // upload, so the index of a handle never goes stale — and the next such call finds it.  [0]: boxes + grid directory, [1]: + the
    // point-in-polygon tables (GPK_INDEX_PIP_LIGHT).  Built under a lock, freed with the handle; only indexes of at most
    // GPK_AUTO_INDEX_MAX_MB (default 256) are kept; GPK_AUTO_INDEX=0 turns the memo off.
    struct gpk_index* auto_index[2];
};

namespace gpk {

// ---- per-thread grow-only scratch ------------------------------------------------------------
// Steady-state calls do no hipMalloc/hipFree.  One arena per calling thread keeps the ABI
// re-entrant; a thread that issues calls on several streams must order them itself.
class Workspace {
  public:
    // carve `bytes` (256-byte aligned) out of the arena; grows (and invalidates earlier carves) only
    // between begin() calls.
    int32_t begin(size_t total_bytes);
    void* take(size_t bytes);
    void trim(size_t keep_max);  // give the arena back when it has grown beyond keep_max bytes (after a device sync)
    ~Workspace();
    // A caller that leaves the same small record at the same place of the arena call after call (the rare arm's arguments of the
    // fused point join) tags it: tag_matches() is true when the PREVIOUS begin() .. begin() span ended with set_tag / a match of
    // these very bytes at this very place and the arena has not been re-allocated since — then the record need not be written again.
    bool tag_matches(const void* where, const void* bytes, size_t n);
    void set_tag(const void* where, const void* bytes, size_t n);

  private:
    char* base_ = nullptr;
    size_t cap_ = 0, used_ = 0;
    int device_ = -1;
    const void* tag_where_ = nullptr;
    unsigned char tag_[192];
    size_t tag_n_ = 0;
    bool tag_prev_ = false, tag_cur_ = false;
};
Workspace& workspace();
// second and third per-thread arenas for calls whose scratch is sized in stages (the polygon x polygon join learns its
// candidate count only after a first pass, and gpk_bounds inside it uses workspace() itself)
Workspace& workspace_aux(int which);

// Small read-backs (a count that sizes the next table): the copy lands in PINNED host memory — into a pageable variable this runtime stages
// it, + 8 us a copy on top of the 11 us a stream sync costs (tools/micro/readback_probe.hip) — and reaches the caller's variable in
// sync_small, which MUST be the sync that follows (per calling thread; at most 16 copies of at most 64 bytes between two syncs).
hipError_t d2h_small(void* host_dst, const void* dev_src, size_t bytes, hipStream_t s);
hipError_t sync_small(hipStream_t s);
// one more arena per (calling thread, stream): scratch of stream-ordered calls that must survive until the stream gets there
Workspace& workspace_for_stream(hipStream_t s);

inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

// ---- cached device blocks ------------------------------------------------------------------------
// The tables and temporaries of an index build come from a small caching layer over hipMalloc / hipFree: a released block is kept
// (GPK_DEVICE_CACHE_MB in total; default a sixteenth of the device's memory, 16 GB at most; 0 switches the cache off) and handed to the next request it fits (at most a quarter
// larger than asked).  On this runtime a hipFree costs 70 us for a small block and — on some boxes — hundreds of milliseconds for
// gigabytes: the second build of a 5M-multipolygon index took 306 ms where its kernels take 95.
// cached_free does NOT wait for the device the way hipFree does: the caller guarantees that nothing still reads the block
// (gpk_index_free synchronises the device once for all of an index's tables).
hipError_t cached_malloc(void** p, size_t bytes);
// hipMalloc for everything that is NOT recycled (uploads, workspaces, result arrays): on failure the cache's idle blocks are handed
// back to the driver and the request is tried once more — up to GPK_DEVICE_CACHE_MB of device memory may sit idle in the cache
hipError_t device_malloc(void** p, size_t bytes);
template <typename T>
inline hipError_t device_malloc(T** p, size_t bytes) {
    return device_malloc(reinterpret_cast<void**>(p), bytes);
}
void cached_free(void* p);
void cached_release_all();  // gpk_device_cache_release

// ---- profiling -----------------------------------------------------------------------------
bool profiling_enabled();
void profile_begin(const char* name, hipStream_t s, void** token);
void profile_end(void* token, hipStream_t s);

// Launch wrapper: every kernel of the library goes through this so bench.py can read per-kernel
// HIP-event durations on the launching stream (gpk_profile_query).
bool debug_sync();  // GPK_DEBUG_SYNC=1: name every launch on stderr and wait for it (locating a faulting kernel)
#define GPK_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                        \
    do {                                                                                 \
        void* _tok = nullptr;                                                            \
        if (::gpk::profiling_enabled()) ::gpk::profile_begin(name, stream, &_tok);       \
        if (::gpk::debug_sync()) fprintf(stderr, "[gpk] launch %s\n", name);              \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);             \
        if (_tok) ::gpk::profile_end(_tok, stream);                                      \
        if (::gpk::debug_sync()) (void)hipStreamSynchronize(stream);                     \
        hipError_t _le = hipGetLastError();                                              \
        if (_le != hipSuccess)                                                           \
            return ::gpk::fail(GPK_ERR_DEVICE, "launch of %s failed: %s", name,          \
                               hipGetErrorString(_le));                                  \
    } while (0)

int32_t require_device();  // GPK_ERR_DEVICE unless the current device is a gfx950
int cu_count();

// copy helpers honouring the ABI's memory-space tags
int32_t copy_out(void* dst, int32_t dst_space, const void* src_dev, size_t bytes, hipStream_t s);

__host__ __device__ inline bool is_polygonal(int32_t t) { return t == GPK_GEOM_POLYGON || t == GPK_GEOM_MULTIPOLYGON; }

}  // namespace gpk
