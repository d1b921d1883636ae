// gpk_runtime.hip — library plumbing behind the C ABI: thread-local error text, the per-thread
// scratch arena, HIP-event profiling of every launch, device checks and the "copy once to HBM"
// upload (include/geopolars_hip.h: gpk_geoarray_upload replaces the per-op row decode of
// geopolars/geopolars-geo/src/util.rs:27-37).
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "gpk_common.h"

namespace gpk {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void drop_small_readbacks();  // (below: read-backs queued by d2h_small and never waited for — an error return in between — must not be delivered later)
int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    drop_small_readbacks();
    return code;
}

// ---- cached device blocks -------------------------------------------------------------------------
namespace {
struct BlockCache {
    std::mutex mu;
    std::unordered_map<void*, std::pair<size_t, int>> live;  // block -> (bytes, device), for blocks handed out by cached_malloc
    std::multimap<size_t, std::pair<void*, int>> idle;       // bytes -> (block, device)
    size_t idle_bytes = 0;
    size_t budget() {
        static const size_t b = [] {
            if (const char* e = getenv("GPK_DEVICE_CACHE_MB")) return (size_t)atoll(e) << 20;
            size_t free_b = 0, total_b = 0;  // default: a sixteenth of the device's memory, 16 GB at most
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = size_t(64) << 30;
            const size_t cap = size_t(16) << 30;
            return total_b / 16 < cap ? total_b / 16 : cap;
        }();
        return b;
    }
    void release_all_locked() {
        for (auto& kv : idle) (void)hipFree(kv.second.first);
        idle.clear();
        idle_bytes = 0;
    }
};
BlockCache& block_cache() {
    static BlockCache* c = new BlockCache;  // (never destroyed: blocks may be released during process teardown)
    return *c;
}
}  // namespace

hipError_t device_malloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cached_release_all();
        e = hipMalloc(p, bytes);
    }
    return e;
}
hipError_t cached_malloc(void** p, size_t bytes) {
    BlockCache& c = block_cache();
    if (bytes == 0) bytes = 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> g(c.mu);
        for (auto it = c.idle.lower_bound(bytes); it != c.idle.end() && it->first <= bytes + bytes / 4 + 4096; ++it) {
            if (it->second.second != dev) continue;
            *p = it->second.first;
            c.live[*p] = {it->first, dev};
            c.idle_bytes -= it->first;
            c.idle.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // give the idle blocks back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> g(c.mu);
            c.release_all_locked();
        }
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(c.mu);
        c.live[*p] = {bytes, dev};
    }
    return e;
}
void cached_free(void* p) {
    if (!p) return;
    BlockCache& c = block_cache();
    std::unique_lock<std::mutex> g(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) {  // not one of ours
        g.unlock();
        (void)hipFree(p);
        return;
    }
    const size_t bytes = it->second.first;
    const int dev = it->second.second;
    c.live.erase(it);
    if (bytes > c.budget()) {
        g.unlock();
        (void)hipFree(p);
        return;
    }
    while (c.idle_bytes + bytes > c.budget() && !c.idle.empty()) {  // make room: the largest idle block goes first
        auto last = std::prev(c.idle.end());
        (void)hipFree(last->second.first);
        c.idle_bytes -= last->first;
        c.idle.erase(last);
    }
    c.idle.emplace(bytes, std::make_pair(p, dev));
    c.idle_bytes += bytes;
}
void cached_release_all() {
    BlockCache& c = block_cache();
    std::lock_guard<std::mutex> g(c.mu);
    c.release_all_locked();
}

// ---- workspace ---------------------------------------------------------------------------------
int32_t Workspace::begin(size_t total) {
    int dev = 0;
    GPK_HIP(hipGetDevice(&dev));
    total = align256(total) + 256;
    if (dev != device_ || total > cap_) {
        if (base_) {
            GPK_HIP(hipDeviceSynchronize());  // earlier launches may still read the old arena
            (void)hipFree(base_);
            base_ = nullptr;
            cap_ = 0;
        }
        size_t want = total + total / 4;
        hipError_t e = device_malloc((void**)&base_, want);
        if (e != hipSuccess) {
            want = total;
            e = device_malloc((void**)&base_, want);
        }
        if (e != hipSuccess) {
            base_ = nullptr;
            return fail(GPK_ERR_OOM, "workspace device_malloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap_ = want;
        device_ = dev;
        tag_cur_ = false;
    }
    used_ = 0;
    tag_prev_ = tag_cur_;
    tag_cur_ = false;
    return GPK_OK;
}
bool Workspace::tag_matches(const void* where, const void* bytes, size_t n) {
    const bool same = tag_prev_ && where == tag_where_ && n == tag_n_ && n <= sizeof tag_ && memcmp(tag_, bytes, n) == 0;
    if (same) tag_cur_ = true;
    return same;
}
void Workspace::set_tag(const void* where, const void* bytes, size_t n) {
    if (n > sizeof tag_) return;
    memcpy(tag_, bytes, n);
    tag_n_ = n;
    tag_where_ = where;
    tag_cur_ = true;
}
void* Workspace::take(size_t bytes) {
    bytes = align256(bytes);
    if (used_ + bytes > cap_) return nullptr;
    void* p = base_ + used_;
    used_ += bytes;
    return p;
}
void Workspace::trim(size_t keep_max) {
    if (!base_ || cap_ <= keep_max) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(base_);
    base_ = nullptr;
    cap_ = used_ = 0;
    tag_cur_ = tag_prev_ = false;
}
Workspace::~Workspace() {
    // process teardown: the HIP runtime may already be gone; leak rather than crash.
}
Workspace& workspace() {
    static thread_local Workspace ws;
    return ws;
}
Workspace& workspace_for_stream(hipStream_t s) {
    // A thread keeps at most STREAM_ARENAS of these (least recently used goes): callers that create and destroy a stream per
    // batch would otherwise leave one join-sized arena (> 12 bytes per left row) behind per dead stream handle.  The evicted
    // arena may still be in use by work queued on its stream, so it is released only after the device has drained.
    constexpr size_t STREAM_ARENAS = 8;
    static thread_local std::vector<std::pair<hipStream_t, Workspace*>> arenas;  // most recently used last; a handful: linear search
    for (size_t i = 0; i < arenas.size(); ++i)
        if (arenas[i].first == s) {
            auto kv = arenas[i];
            arenas.erase(arenas.begin() + (long)i);
            arenas.push_back(kv);
            return *kv.second;
        }
    if (arenas.size() >= STREAM_ARENAS) {
        Workspace* old = arenas.front().second;
        arenas.erase(arenas.begin());
        old->trim(0);  // synchronises the device, then frees
        delete old;
    }
    arenas.emplace_back(s, new Workspace);
    return *arenas.back().second;
}
namespace {
struct SmallReadBacks {
    unsigned char* pinned = nullptr;  // 16 slots of 64 bytes
    struct Pending {
        void* dst;
        size_t bytes;
    } pending[16];
    int n = 0;
};
SmallReadBacks& small_readbacks() {
    static thread_local SmallReadBacks r;
    return r;
}
}  // namespace
void drop_small_readbacks() { small_readbacks().n = 0; }
hipError_t d2h_small(void* host_dst, const void* dev_src, size_t bytes, hipStream_t s) {
    SmallReadBacks& r = small_readbacks();
    if (!r.pinned && hipHostMalloc((void**)&r.pinned, 16 * 64, hipHostMallocDefault) != hipSuccess) r.pinned = nullptr;
    if (!r.pinned || bytes > 64 || r.n >= 16) return hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s);  // (the plain way)
    r.pending[r.n] = {host_dst, bytes};
    return hipMemcpyAsync(r.pinned + 64 * r.n++, dev_src, bytes, hipMemcpyDeviceToHost, s);
}
hipError_t sync_small(hipStream_t s) {
    SmallReadBacks& r = small_readbacks();
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess)
        for (int i = 0; i < r.n; ++i) memcpy(r.pending[i].dst, r.pinned + 64 * i, r.pending[i].bytes);
    r.n = 0;
    return e;
}

Workspace& workspace_aux(int which) {
    static thread_local Workspace aux[2];
    return aux[which & 1];
}

// ---- profiling -----------------------------------------------------------------------------
struct ProfRec {
    std::string name;
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::string g_prof_filter;  // when non-empty only kernels whose name contains it are bracketed
static std::vector<ProfRec*> g_prof;
static std::vector<ProfRec*> g_prof_pool;  // recycled records: hipEventCreate is not free inside a timed region

bool profiling_enabled() { return g_prof_on; }
bool debug_sync() {
    static const bool on = [] {
        const char* e = getenv("GPK_DEBUG_SYNC");
        return e && *e && *e != '0';
    }();
    return on;
}
void profile_begin(const char* name, hipStream_t s, void** token) {
    ProfRec* r = nullptr;
    *token = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_filter.empty() && !strstr(name, g_prof_filter.c_str())) return;
        if (!g_prof_pool.empty()) {
            r = g_prof_pool.back();
            g_prof_pool.pop_back();
        }
    }
    if (!r) {
        r = new ProfRec;
        if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) {
            delete r;
            *token = nullptr;
            return;
        }
    }
    r->name = name;
    (void)hipEventRecord(r->a, s);
    *token = r;
}
void profile_end(void* token, hipStream_t s) {
    ProfRec* r = (ProfRec*)token;
    (void)hipEventRecord(r->b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
}

// ---- device --------------------------------------------------------------------------------
static thread_local int g_checked_dev = -1;
static thread_local int g_cus = 0;

int32_t require_device() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return fail(GPK_ERR_DEVICE, "no HIP device: %s (libgeopolars_hip has no CPU fallback)",
                    hipGetErrorString(e));
    if (dev == g_checked_dev) return GPK_OK;
    hipDeviceProp_t p;
    GPK_HIP(hipGetDeviceProperties(&p, dev));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(GPK_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", dev,
                    p.gcnArchName);
    g_checked_dev = dev;
    g_cus = p.multiProcessorCount;
    return GPK_OK;
}
int cu_count() { return g_cus > 0 ? g_cus : 256; }

int32_t copy_out(void* dst, int32_t dst_space, const void* src_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return GPK_OK;
    if (dst_space == GPK_MEM_DEVICE) {
        if (dst != src_dev) GPK_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToDevice, s));
        return GPK_OK;
    }
    GPK_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

}  // namespace gpk

using namespace gpk;

extern "C" {

const char* gpk_version(void) { return "geopolars_hip 0.1.0 (gfx950)"; }

int32_t gpk_last_error(char* buf, size_t cap) {
    if (!buf || cap == 0) return GPK_ERR_INVALID_ARGUMENT;
    strncpy(buf, g_err, cap - 1);
    buf[cap - 1] = 0;
    return GPK_OK;
}

int32_t gpk_device_cache_release(void) {
    (void)hipDeviceSynchronize();
    gpk::cached_release_all();
    return GPK_OK;
}

int32_t gpk_device_count(int32_t* out_n) {
    if (!out_n) return fail(GPK_ERR_INVALID_ARGUMENT, "out_n is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_n = 0;
        return fail(GPK_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out_n = n;
    return GPK_OK;
}

int32_t gpk_device_info(char* name_buf, size_t cap, int32_t* out_cus) {
    GPK_TRY(require_device());
    int dev = 0;
    GPK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    GPK_HIP(hipGetDeviceProperties(&p, dev));
    if (name_buf && cap) snprintf(name_buf, cap, "%s (%s)", p.name, p.gcnArchName);
    if (out_cus) *out_cus = p.multiProcessorCount;
    return GPK_OK;
}

// ---- upload ----------------------------------------------------------------------------------
static int32_t validate_desc(const gpk_geoarrow_desc* d) {
    if (!d) return fail(GPK_ERR_INVALID_ARGUMENT, "desc is NULL");
    if (d->n_geoms < 0 || d->n_coords < 0)
        return fail(GPK_ERR_INVALID_ARGUMENT, "negative length");
    if (d->n_coords > INT32_MAX || d->n_geoms > INT32_MAX)
        return fail(GPK_ERR_INVALID_OFFSETS, "arrays beyond i32 offsets are not supported (split the chunk)");
    if (d->n_coords > 0 && !d->xy && !(d->x && d->y)) return fail(GPK_ERR_INVALID_ARGUMENT, "xy is NULL (and x / y are not both given)");
    if (d->xy && (d->x || d->y)) return fail(GPK_ERR_INVALID_ARGUMENT, "coordinates given twice: xy and x / y");
    switch (d->geom_type) {
    case GPK_GEOM_POINT:
        if (d->n_coords != d->n_geoms)
            return fail(GPK_ERR_INVALID_OFFSETS, "POINT array: n_coords != n_geoms");
        break;
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT:
        if (!d->geom_offsets) return fail(GPK_ERR_INVALID_OFFSETS, "geom_offsets is NULL");
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING:
        if (!d->geom_offsets || !d->ring_offsets)
            return fail(GPK_ERR_INVALID_OFFSETS, "geom_offsets / ring_offsets is NULL");
        break;
    case GPK_GEOM_MULTIPOLYGON:
        if (!d->geom_offsets || !d->ring_offsets || !d->part_offsets)
            return fail(GPK_ERR_INVALID_OFFSETS, "geom/part/ring offsets must all be given");
        break;
    default:
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "unsupported geometry type id %d", d->geom_type);
    }
    return GPK_OK;
}

// host-side monotonicity / range check of one offsets level (only for host descriptors; device
// descriptors are trusted, as a borrowed Arrow buffer would be)
static int32_t check_offsets(const int32_t* off, int64_t n, int64_t child_len, const char* what) {
    if (n == 0) return GPK_OK;
    if (off[0] != 0) return fail(GPK_ERR_INVALID_OFFSETS, "%s[0] != 0", what);
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return fail(GPK_ERR_INVALID_OFFSETS, "%s not monotone at %lld", what, (long long)i);
    if (off[n] != child_len)
        return fail(GPK_ERR_INVALID_OFFSETS, "%s[last] = %d but child length is %lld", what, off[n],
                    (long long)child_len);
    return GPK_OK;
}

// Struct<x, y> coordinates -> the interleaved form the kernels read: two coalesced 8-byte streams in, one 16-byte stream out
__global__ __launch_bounds__(256) void interleave_xy_kernel(const double* __restrict__ x, const double* __restrict__ y, int64_t n, double2* __restrict__ xy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xy[i] = make_double2(x[i], y[i]);
}

__global__ __launch_bounds__(256) void rebase_offsets_kernel(const int32_t* __restrict__ in, int64_t n, int32_t first, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] - first;
}

int32_t gpk_geoarray_upload(const gpk_geoarrow_desc* d, void* stream, gpk_geoarray** out) {
    if (!out) return fail(GPK_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    GPK_TRY(validate_desc(d));
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;

    const int t = d->geom_type;
    const bool has_ring = t == GPK_GEOM_POLYGON || t == GPK_GEOM_MULTILINESTRING || t == GPK_GEOM_MULTIPOLYGON;
    const bool has_part = t == GPK_GEOM_MULTIPOLYGON;
    const int64_t n_rings = has_ring ? d->n_rings : 0;
    const int64_t n_parts = has_part ? d->n_parts : 0;

    if (d->mem_space == GPK_MEM_HOST) {
        if (t == GPK_GEOM_LINESTRING || t == GPK_GEOM_MULTIPOINT)
            GPK_TRY(check_offsets(d->geom_offsets, d->n_geoms, d->n_coords, "geom_offsets"));
        if (t == GPK_GEOM_POLYGON || t == GPK_GEOM_MULTILINESTRING) {
            GPK_TRY(check_offsets(d->geom_offsets, d->n_geoms, n_rings, "geom_offsets"));
            GPK_TRY(check_offsets(d->ring_offsets, n_rings, d->n_coords, "ring_offsets"));
        }
        if (t == GPK_GEOM_MULTIPOLYGON) {
            GPK_TRY(check_offsets(d->geom_offsets, d->n_geoms, n_parts, "geom_offsets"));
            GPK_TRY(check_offsets(d->part_offsets, n_parts, n_rings, "part_offsets"));
            GPK_TRY(check_offsets(d->ring_offsets, n_rings, d->n_coords, "ring_offsets"));
        }
    }

    int cur_dev = 0;
    GPK_HIP(hipGetDevice(&cur_dev));  // before the handle exists: nothing to release on failure
    gpk_geoarray* a = new gpk_geoarray;
    memset(a, 0, sizeof *a);
    a->device = cur_dev;
    a->d.type = t;
    a->d.n_geoms = d->n_geoms;
    a->d.n_parts = has_part ? n_parts : (is_polygonal(t) ? d->n_geoms : 0);
    a->d.n_rings = n_rings;
    a->d.n_coords = d->n_coords;

    struct Buf {
        const void* src;
        size_t bytes;
        const void** dst;
    } bufs[5] = {
        {d->xy, sizeof(double) * 2 * (size_t)d->n_coords, (const void**)&a->d.xy},
        {t != GPK_GEOM_POINT ? d->geom_offsets : nullptr, sizeof(int32_t) * (size_t)(d->n_geoms + 1),
         (const void**)&a->d.geom_off},
        {has_part ? d->part_offsets : nullptr, sizeof(int32_t) * (size_t)(n_parts + 1),
         (const void**)&a->d.part_off},
        {has_ring ? d->ring_offsets : nullptr, sizeof(int32_t) * (size_t)(n_rings + 1),
         (const void**)&a->d.ring_off},
        {d->validity, (size_t)((d->n_geoms + 7) / 8), (const void**)&a->d.validity},
    };
    if (!d->xy && d->x && d->y && d->n_coords > 0) {
        // separated coordinates: interleaved on the device into a buffer the handle owns.  Host arrays travel as they are (two
        // copies into a scratch block that goes back to the block cache) — the host never builds the interleaved array.
        const size_t half = sizeof(double) * (size_t)d->n_coords;
        void *xy_dev = nullptr, *tmp = nullptr;
        hipError_t e = cached_malloc(&xy_dev, 2 * half);
        const double *xs = d->x, *ys = d->y;
        if (e == hipSuccess && d->mem_space != GPK_MEM_DEVICE) {
            e = cached_malloc(&tmp, 2 * half);
            if (e == hipSuccess) e = hipMemcpyAsync(tmp, d->x, half, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync((char*)tmp + half, d->y, half, hipMemcpyHostToDevice, s);
            xs = (const double*)tmp;
            ys = (const double*)((char*)tmp + half);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(interleave_xy_kernel, dim3((unsigned)((d->n_coords + 255) / 256)), dim3(256), 0, s, xs, ys, d->n_coords, (double2*)xy_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess && tmp) e = hipStreamSynchronize(s);  // (the scratch block is handed back below)
        if (tmp) cached_free(tmp);
        if (e != hipSuccess) {
            if (xy_dev) cached_free(xy_dev);
            gpk_geoarray_free(a);
            return fail(e == hipErrorOutOfMemory ? GPK_ERR_OOM : GPK_ERR_DEVICE, "upload: %s", hipGetErrorString(e));
        }
        a->owned[0] = xy_dev;
        a->d.xy = (const double2*)xy_dev;
        a->nbytes += (int64_t)(2 * half);
        bufs[0].src = nullptr;  // (done)
    }
    for (int i = 0; i < 5; ++i) {
        if (i == 0 && a->owned[0]) continue;
        *bufs[i].dst = nullptr;
        if (!bufs[i].src || bufs[i].bytes == 0) continue;
        a->nbytes += (int64_t)bufs[i].bytes;
        if (d->mem_space == GPK_MEM_DEVICE) {
            *bufs[i].dst = bufs[i].src;  // borrowed: the caller keeps it alive, as with Arrow buffers
            continue;
        }
        void* p = nullptr;
        hipError_t e = device_malloc(&p, bufs[i].bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(p, bufs[i].src, bufs[i].bytes, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) {
            if (p) (void)hipFree(p);
            gpk_geoarray_free(a);
            return fail(e == hipErrorOutOfMemory ? GPK_ERR_OOM : GPK_ERR_DEVICE, "upload: %s",
                        hipGetErrorString(e));
        }
        a->owned[i] = p;
        *bufs[i].dst = p;
    }
    if (d->mem_space == GPK_MEM_HOST) GPK_HIP(hipStreamSynchronize(s));  // host buffers are only borrowed for the call
    if (d->mem_space == GPK_MEM_DEVICE && t != GPK_GEOM_POINT) {
        // A device view whose offsets do not start at 0 — a sliced Arrow list array hands over its offsets unrebased next to the SLICE of
        // the child buffer they index — is normalised here: every kernel of the library indexes children from 0 (ring_of_coord, the
        // chain tables, the unary passes), and only concat / all-gatherv knew how to subtract a first offset.  One small read-back per
        // upload of a device view; a level that starts at 0 stays borrowed, one that does not gets an owned, rebased copy.
        struct Lv {
            const int32_t** p;
            int64_t n;
            int slot;
        } lv[3] = {{&a->d.geom_off, d->n_geoms + 1, 1}, {&a->d.part_off, n_parts + 1, 2}, {&a->d.ring_off, n_rings + 1, 3}};
        int32_t first[3] = {0, 0, 0};
        bool any = false;
        for (int i = 0; i < 3; ++i)
            if (*lv[i].p && lv[i].n > 0) {
                const hipError_t e = hipMemcpyAsync(&first[i], *lv[i].p, sizeof(int32_t), hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) {
                    gpk_geoarray_free(a);
                    return fail(GPK_ERR_DEVICE, "upload: %s", hipGetErrorString(e));
                }
                any = true;
            }
        if (any) {
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                gpk_geoarray_free(a);
                return fail(GPK_ERR_DEVICE, "upload: %s", hipGetErrorString(e));
            }
        }
        for (int i = 0; i < 3; ++i) {
            if (first[i] == 0 || !*lv[i].p) continue;
            void* p = nullptr;
            hipError_t e = cached_malloc(&p, sizeof(int32_t) * (size_t)lv[i].n);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((lv[i].n + 255) / 256)), dim3(256), 0, s, *lv[i].p, lv[i].n, first[i], (int32_t*)p);
                e = hipGetLastError();
            }
            if (e != hipSuccess) {
                if (p) cached_free(p);
                gpk_geoarray_free(a);
                return fail(e == hipErrorOutOfMemory ? GPK_ERR_OOM : GPK_ERR_DEVICE, "upload: %s", hipGetErrorString(e));
            }
            a->owned[lv[i].slot] = p;
            *lv[i].p = (const int32_t*)p;
        }
    }
    *out = a;
    return GPK_OK;
}

int32_t gpk_geoarray_free(gpk_geoarray* a) {
    if (!a) return GPK_OK;
    // (owned buffers may come from the block cache — decoded WKB columns — which does not wait the way hipFree does: wait once, on
    // the owning device, for whatever still reads them)
    bool any = false;
    for (int i = 0; i < 5; ++i) any = any || a->owned[i] != nullptr;
    if (any) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != a->device) (void)hipSetDevice(a->device);
        (void)hipDeviceSynchronize();
        if (cur >= 0 && cur != a->device) (void)hipSetDevice(cur);
    }
    for (int i = 0; i < 5; ++i)
        if (a->owned[i]) cached_free(a->owned[i]);
    for (int k = 0; k < 2; ++k)
        if (a->auto_index[k]) {
            gpk_index_free(a->auto_index[k]);
            a->auto_index[k] = nullptr;
        }
    if (a->classes) {
        if (a->classes->lists) (void)hipFree(a->classes->lists);
        if (a->classes->chunk_begin) (void)hipFree(a->classes->chunk_begin);
        if (a->classes->strip_first) (void)hipFree(a->classes->strip_first);
        if (a->classes->strip_cross) (void)hipFree(a->classes->strip_cross);
        if (a->classes->strip_desc) (void)hipFree(a->classes->strip_desc);
        delete a->classes;
    }
    delete a;
    return GPK_OK;
}

int32_t gpk_geoarray_validity(const gpk_geoarray* a, uint8_t* out_bitmap, int32_t* out_has_validity, void* stream) {
    if (!a || !out_has_validity) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_has_validity = a->d.validity ? 1 : 0;
    if (!a->d.validity || !out_bitmap || a->d.n_geoms == 0) return GPK_OK;
    hipStream_t s = (hipStream_t)stream;
    GPK_HIP(hipMemcpyAsync(out_bitmap, a->d.validity, (size_t)((a->d.n_geoms + 7) / 8), hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

int32_t gpk_geoarray_len(const gpk_geoarray* a, int64_t* out_n) {
    if (!a || !out_n) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_n = a->d.n_geoms;
    return GPK_OK;
}

int32_t gpk_geoarray_nbytes(const gpk_geoarray* a, int64_t* out_bytes) {
    if (!a || !out_bytes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    int64_t total = a->nbytes;
    for (int k = 0; k < 2; ++k)  // (+ the indexes joins without an index have left on the handle)
        if (a->auto_index[k]) {
            int64_t ib = 0;
            if (gpk_index_nbytes(a->auto_index[k], &ib) == GPK_OK) total += ib;
        }
    *out_bytes = total;
    return GPK_OK;
}

// ---- profiling ABI ---------------------------------------------------------------------------
int32_t gpk_profile_enable(int32_t on) {
    g_prof_on = on != 0;
    return GPK_OK;
}
int32_t gpk_profile_filter(const char* substr) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_filter = substr ? substr : "";
    return GPK_OK;
}
int32_t gpk_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec* r : g_prof) g_prof_pool.push_back(r);
    g_prof.clear();
    return GPK_OK;
}
int32_t gpk_profile_query(const char* substr, double* out_ms, int64_t* out_launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0.0;
    int64_t n = 0;
    for (ProfRec* r : g_prof) {
        if (substr && *substr && r->name.find(substr) == std::string::npos) continue;
        if (hipEventSynchronize(r->b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r->a, r->b) != hipSuccess) continue;
        ms += t;
        ++n;
    }
    if (out_ms) *out_ms = ms;
    if (out_launches) *out_launches = n;
    return GPK_OK;
}

}  // extern "C"
