// gpk_comm.hip — the one collective of the path, behind the C ABI (the seam is where
// `spatial_join` takes its right side and its index, geopolars/src/spatial_index.rs:37-76).
//
// The left series is sharded by rows over the GPUs of a node, one process per GPU; a right side that is itself produced
// sharded (C4 / C5) is exchanged ONCE with an all-gatherv of its GeoArrow buffers over RCCL / xGMI, and so are the leaves
// of its index (per-geometry boxes, the NodeEnvelopes of spatial_index.rs:206-312) — `gpk_index_build_ex` assembles the
// gathered index from them.  RCCL has no `v` collective: the lengths travel first (one small all-gather), then every
// buffer is moved with one grouped round of broadcasts, rank r's piece landing at its final offset — no padding, no
// trimming, nothing staged through the host; offsets are rebased and validity bytes packed by two small kernels.
//
// RCCL is opened at run time (dlopen): the library has no link-time dependency on it, loads where it is absent, and uses
// the copy the process already has when there is one (a PyTorch process brings its own librccl.so).
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "gpk_common.h"

namespace gpk {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;
static char g_rccl_err[256] = {0};

static void rccl_open() {
    const char* env = getenv("GPK_RCCL_PATH");
    void* lib = nullptr;
    if (env && *env) lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);    // the copy the process already uses
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "librccl.so cannot be opened (%s); set GPK_RCCL_PATH", dlerror());
        return;
    }
    Rccl r;
    r.lib = lib;
#define GPK_RCCL_SYM(field, name)                                                          \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, name));                       \
    if (!r.field) {                                                                        \
        snprintf(g_rccl_err, sizeof g_rccl_err, "librccl.so has no symbol %s", name);      \
        return;                                                                            \
    }
    GPK_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    GPK_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    GPK_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    GPK_RCCL_SYM(AllGather, "ncclAllGather")
    GPK_RCCL_SYM(Broadcast, "ncclBroadcast")
    GPK_RCCL_SYM(GroupStart, "ncclGroupStart")
    GPK_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    GPK_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GPK_RCCL_SYM
    g_rccl = r;
}
static int32_t rccl(const Rccl** out) {
    std::call_once(g_rccl_once, rccl_open);
    if (!g_rccl.lib) return fail(GPK_ERR_DEVICE, "RCCL: %s", g_rccl_err);
    *out = &g_rccl;
    return GPK_OK;
}
#define GPK_NCCL(r, expr)                                                                                               \
    do {                                                                                                                \
        ncclResult_t _n = (expr);                                                                                       \
        if (_n != ncclSuccess) return ::gpk::fail(GPK_ERR_DEVICE, "%s failed: %s", #expr, (r)->GetErrorString(_n));     \
    } while (0)

// offsets of shard k arrive as [0 .. child_k]; in the gathered buffer every shard but the first drops its leading 0 and all
// of them are shifted by the children of the shards before it.  One launch: element j of the gathered buffer belongs to
// the shard whose range holds it (world is small: linear search over the cut points).
constexpr int COMM_MAX_WORLD = 64;
struct RebaseArgs {
    int32_t world;
    int64_t dst_begin[COMM_MAX_WORLD + 1];  // first element of shard k in the gathered buffer
    int64_t shift[COMM_MAX_WORLD];          // to add to shard k's values
};
__global__ void rebase_offsets_kernel(int32_t* __restrict__ off, RebaseArgs a) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.dst_begin[a.world]) return;
    int k = 0;
    while (k + 1 < a.world && j >= a.dst_begin[k + 1]) ++k;
    off[j] += (int32_t)a.shift[k];
}
__global__ void bitmap_to_bytes_kernel(const uint8_t* __restrict__ bitmap, int64_t n, uint8_t* __restrict__ bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bytes[i] = bitmap ? (uint8_t)((bitmap[i >> 3] >> (i & 7)) & 1) : (uint8_t)1;
}
__global__ void bytes_to_bitmap_kernel(const uint8_t* __restrict__ bytes, int64_t n, uint8_t* __restrict__ bitmap) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (n + 7) / 8) return;
    uint32_t v = 0;
    for (int t = 0; t < 8; ++t)
        if (8 * b + t < n && bytes[8 * b + t]) v |= 1u << t;
    bitmap[b] = (uint8_t)v;
}

}  // namespace gpk

using namespace gpk;

struct gpk_comm {
    ncclComm_t comm;
    int32_t rank, world, device;
};

// every rank's piece of one buffer lands at its final place: a grouped round of `world` broadcasts (elements of `elem` bytes)
static int32_t gather_pieces(const Rccl* r, gpk_comm* c, const void* mine, const int64_t* counts, const int64_t* dst_begin, size_t elem, char* out, hipStream_t s) {
    GPK_NCCL(r, r->GroupStart());
    for (int k = 0; k < c->world; ++k) {
        if (counts[k] == 0) continue;  // (every rank skips the same pieces: the counts come from the header)
        const ncclResult_t n = r->Broadcast(k == c->rank ? mine : nullptr, out + (size_t)dst_begin[k] * elem, (size_t)counts[k] * elem, ncclUint8, k, c->comm, s);
        if (n != ncclSuccess) {
            (void)r->GroupEnd();
            return fail(GPK_ERR_DEVICE, "ncclBroadcast failed: %s", r->GetErrorString(n));
        }
    }
    GPK_NCCL(r, r->GroupEnd());
    return GPK_OK;
}

extern "C" {

int32_t gpk_comm_unique_id(uint8_t out_id[128]) {
    if (!out_id) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    const Rccl* r;
    GPK_TRY(rccl(&r));
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI carries RCCL's unique id as 128 opaque bytes");
    ncclUniqueId id;
    GPK_NCCL(r, r->GetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return GPK_OK;
}

int32_t gpk_comm_init(int32_t rank, int32_t world, const uint8_t id[128], gpk_comm** out) {
    if (!id || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (world < 1 || world > COMM_MAX_WORLD || rank < 0 || rank >= world)
        return fail(GPK_ERR_INVALID_ARGUMENT, "comm: rank %d of %d (1 <= world <= %d)", rank, world, COMM_MAX_WORLD);
    GPK_TRY(require_device());
    const Rccl* r;
    GPK_TRY(rccl(&r));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    gpk_comm* c = new gpk_comm;
    c->rank = rank;
    c->world = world;
    (void)hipGetDevice(&c->device);
    const ncclResult_t n = r->CommInitRank(&c->comm, world, uid, rank);
    if (n != ncclSuccess) {
        delete c;
        return fail(GPK_ERR_DEVICE, "ncclCommInitRank failed: %s", r->GetErrorString(n));
    }
    *out = c;
    return GPK_OK;
}

int32_t gpk_comm_free(gpk_comm* c) {
    if (!c) return GPK_OK;
    const Rccl* r;
    if (rccl(&r) == GPK_OK) (void)r->CommDestroy(c->comm);
    delete c;
    return GPK_OK;
}

int32_t gpk_comm_info(const gpk_comm* c, int32_t* out_rank, int32_t* out_world) {
    if (!c) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (out_rank) *out_rank = c->rank;
    if (out_world) *out_world = c->world;
    return GPK_OK;
}

// n_local rows of `width` doubles per rank -> all rows in rank order (device buffers).  out_counts[world] (host, may be NULL)
// receives every rank's row count; out_capacity_rows guards the caller's buffer: pass 0 with out == NULL to learn the total.
int32_t gpk_allgatherv_rows_f64(gpk_comm* c, const double* local_dev, int64_t n_local, int32_t width, double* out_dev, int64_t out_capacity_rows,
                                int64_t* out_total_rows, int64_t* out_counts, void* stream) {
    if (!c || !out_total_rows || n_local < 0 || width < 1 || (n_local > 0 && !local_dev)) return fail(GPK_ERR_INVALID_ARGUMENT, "bad argument");
    const Rccl* r;
    GPK_TRY(rccl(&r));
    hipStream_t s = (hipStream_t)stream;
    GPK_TRY(workspace_aux(0).begin(sizeof(int64_t) * (size_t)(c->world + 1) + 512));
    int64_t* hdr_dev = (int64_t*)workspace_aux(0).take(sizeof(int64_t) * (size_t)(c->world + 1));
    GPK_HIP(hipMemcpyAsync(hdr_dev + c->world, &n_local, sizeof n_local, hipMemcpyHostToDevice, s));
    GPK_NCCL(r, r->AllGather(hdr_dev + c->world, hdr_dev, 1, ncclInt64, c->comm, s));
    std::vector<int64_t> cnt((size_t)c->world), begin((size_t)c->world + 1, 0);
    GPK_HIP(hipMemcpyAsync(cnt.data(), hdr_dev, sizeof(int64_t) * (size_t)c->world, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < c->world; ++k) begin[(size_t)k + 1] = begin[(size_t)k] + cnt[(size_t)k];
    *out_total_rows = begin[(size_t)c->world];
    if (out_counts) memcpy(out_counts, cnt.data(), sizeof(int64_t) * (size_t)c->world);
    if (!out_dev) return GPK_OK;
    if (out_capacity_rows < *out_total_rows)
        return fail(GPK_ERR_CAPACITY, "allgatherv: %lld rows but capacity %lld", (long long)*out_total_rows, (long long)out_capacity_rows);
    GPK_TRY(gather_pieces(r, c, local_dev, cnt.data(), begin.data(), sizeof(double) * (size_t)width, (char*)out_dev, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

// Every rank contributes its shard of a column and receives the concatenation in rank order as a new array handle (owned
// buffers, free with gpk_geoarray_free); *out_row_base = the first row of this rank's shard in it.  All ranks must call it
// with arrays of the same geometry type.
int32_t gpk_allgatherv_geoarray(gpk_comm* c, const gpk_geoarray* shard, void* stream, gpk_geoarray** out, int64_t* out_row_base, int64_t* out_bytes) {
    if (!c || !shard || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    const Rccl* r;
    GPK_TRY(rccl(&r));
    hipStream_t s = (hipStream_t)stream;
    const DevGeo& d = shard->d;
    const int W = c->world;
    const bool has_geom = d.type != GPK_GEOM_POINT, has_part = d.type == GPK_GEOM_MULTIPOLYGON,
               has_ring = d.type == GPK_GEOM_POLYGON || d.type == GPK_GEOM_MULTILINESTRING || d.type == GPK_GEOM_MULTIPOLYGON;
    // header: n_geoms, n_parts, n_rings, n_coords, has_validity, type
    constexpr int H = 6;
    int64_t mine[H] = {d.n_geoms, has_part ? d.n_parts : 0, has_ring ? d.n_rings : 0, d.n_coords, d.validity ? 1 : 0, d.type};
    GPK_TRY(workspace_aux(0).begin(sizeof(int64_t) * (size_t)(H * (W + 1)) + 512));
    int64_t* hdr_dev = (int64_t*)workspace_aux(0).take(sizeof(int64_t) * (size_t)(H * (W + 1)));
    GPK_HIP(hipMemcpyAsync(hdr_dev + (size_t)H * W, mine, sizeof mine, hipMemcpyHostToDevice, s));
    GPK_NCCL(r, r->AllGather(hdr_dev + (size_t)H * W, hdr_dev, H, ncclInt64, c->comm, s));
    std::vector<int64_t> hdr((size_t)H * W);
    GPK_HIP(hipMemcpyAsync(hdr.data(), hdr_dev, sizeof(int64_t) * (size_t)(H * W), hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));  // the only host read of the exchange: six integers per rank
    bool any_valid = false;
    int64_t tot[4] = {0, 0, 0, 0};
    for (int k = 0; k < W; ++k) {
        if (hdr[(size_t)H * k + 5] != d.type) return fail(GPK_ERR_MISMATCHED_GEOMETRY, "allgatherv: rank %d holds geometry type %lld, this rank %d", k, (long long)hdr[(size_t)H * k + 5], d.type);
        any_valid = any_valid || hdr[(size_t)H * k + 4] != 0;
        for (int q = 0; q < 4; ++q) tot[q] += hdr[(size_t)H * k + q];
    }
    if (tot[0] > INT32_MAX || tot[1] > INT32_MAX || tot[2] > INT32_MAX || tot[3] > INT32_MAX)
        return fail(GPK_ERR_INVALID_ARGUMENT, "allgatherv: the gathered column exceeds 2^31 - 1 rows / rings / coordinates (Arrow i32 offsets)");

    gpk_geoarray* a = new gpk_geoarray;
    memset(a, 0, sizeof *a);
    a->device = shard->device;
    a->d.type = d.type;
    a->d.n_geoms = tot[0];
    a->d.n_parts = has_part ? tot[1] : (is_polygonal(d.type) ? tot[0] : 0);
    a->d.n_rings = has_ring ? tot[2] : 0;
    a->d.n_coords = tot[3];
    auto done = [&](int32_t rc) {
        if (rc != GPK_OK) {
            (void)hipStreamSynchronize(s);
            gpk_geoarray_free(a);
        } else {
            *out = a;
        }
        return rc;
    };
    auto alloc = [&](int slot, size_t bytes, const void** view) -> int32_t {
        void* p = nullptr;
        GPK_HIP(hipMalloc(&p, bytes ? bytes : 8));
        a->owned[slot] = p;
        *view = p;
        a->nbytes += (int64_t)bytes;
        return GPK_OK;
    };
    std::vector<int64_t> cnt((size_t)W), begin((size_t)W + 1);
    int32_t rc;
    // coordinates
    if ((rc = alloc(0, sizeof(double2) * (size_t)tot[3], (const void**)&a->d.xy)) != GPK_OK) return done(rc);
    begin[0] = 0;
    for (int k = 0; k < W; ++k) {
        cnt[(size_t)k] = hdr[(size_t)H * k + 3];
        begin[(size_t)k + 1] = begin[(size_t)k] + cnt[(size_t)k];
    }
    if ((rc = gather_pieces(r, c, d.xy, cnt.data(), begin.data(), sizeof(double2), (char*)a->owned[0], s)) != GPK_OK) return done(rc);
    // offsets, outermost first: level `lvl` has rows[lvl] entries + 1 per shard; its values count the children of the next level
    struct Level {
        bool present;
        int slot, rows_q, child_q;  // header columns: rows of this level, rows of its child level (3 = coordinates)
        const int32_t* src;
        const int32_t** view;
    } levels[3] = {
        {has_geom, 1, 0, has_part ? 1 : (has_ring ? 2 : 3), d.geom_off, &a->d.geom_off},
        {has_part, 2, 1, 2, d.part_off, &a->d.part_off},
        {has_ring, 3, 2, 3, d.ring_off, &a->d.ring_off},
    };
    for (const Level& L : levels) {
        if (!L.present) continue;
        const int64_t total_rows = tot[L.rows_q];
        if ((rc = alloc(L.slot, sizeof(int32_t) * (size_t)(total_rows + 1), (const void**)L.view)) != GPK_OK) return done(rc);
        RebaseArgs ra;
        ra.world = W;
        int64_t at = 0, child = 0;
        const int32_t* my_src = L.src;
        for (int k = 0; k < W; ++k) {
            const int64_t rows = hdr[(size_t)H * k + L.rows_q];
            // shard 0 sends rows + 1 entries (its leading 0 included), the others rows entries (from their entry 1 on)
            cnt[(size_t)k] = k == 0 ? rows + 1 : rows;
            begin[(size_t)k] = at;
            ra.dst_begin[k] = at;
            ra.shift[k] = child;
            at += cnt[(size_t)k];
            child += hdr[(size_t)H * k + L.child_q];
        }
        begin[(size_t)W] = at;
        ra.dst_begin[W] = at;
        if (c->rank != 0 && my_src) my_src += 1;
        if ((rc = gather_pieces(r, c, my_src, cnt.data(), begin.data(), sizeof(int32_t), (char*)a->owned[L.slot], s)) != GPK_OK) return done(rc);
        if (at > 0) hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((at + 255) / 256)), dim3(256), 0, s, (int32_t*)a->owned[L.slot], ra);
    }
    // validity: one byte per row travels (bit offsets of a shard's rows are not byte-aligned in the gathered bitmap)
    if (any_valid && tot[0] > 0) {
        if ((rc = alloc(4, (size_t)((tot[0] + 7) / 8), (const void**)&a->d.validity)) != GPK_OK) return done(rc);
        if ((rc = workspace_aux(1).begin((size_t)tot[0] + (size_t)d.n_geoms + 1024)) != GPK_OK) return done(rc);
        uint8_t* all_bytes = (uint8_t*)workspace_aux(1).take((size_t)tot[0]);
        uint8_t* my_bytes = (uint8_t*)workspace_aux(1).take((size_t)(d.n_geoms ? d.n_geoms : 1));
        if (d.n_geoms > 0)
            hipLaunchKernelGGL(bitmap_to_bytes_kernel, dim3((unsigned)((d.n_geoms + 255) / 256)), dim3(256), 0, s, d.validity, d.n_geoms, my_bytes);
        begin[0] = 0;
        for (int k = 0; k < W; ++k) {
            cnt[(size_t)k] = hdr[(size_t)H * k + 0];
            begin[(size_t)k + 1] = begin[(size_t)k] + cnt[(size_t)k];
        }
        if ((rc = gather_pieces(r, c, my_bytes, cnt.data(), begin.data(), 1, (char*)all_bytes, s)) != GPK_OK) return done(rc);
        hipLaunchKernelGGL(bytes_to_bitmap_kernel, dim3((unsigned)(((tot[0] + 7) / 8 + 255) / 256)), dim3(256), 0, s, (const uint8_t*)all_bytes, tot[0], (uint8_t*)a->owned[4]);
    }
    if (out_row_base) {
        int64_t b = 0;
        for (int k = 0; k < c->rank; ++k) b += hdr[(size_t)H * k + 0];
        *out_row_base = b;
    }
    if (out_bytes) *out_bytes = a->nbytes;
    {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "allgatherv: %s", hipGetErrorString(e)));
    }
    return done(GPK_OK);
}

}  // extern "C"
