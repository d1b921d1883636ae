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

#include <condition_variable>
#include <mutex>
#include <vector>

#include "gpk_common.h"

// The handful of RCCL declarations this file needs, written out: the library is opened with dlopen, and a ROCm install without
// the RCCL development headers must still be able to BUILD this translation unit (the entry points then report GPK_ERR_DEVICE
// at run time when no librccl.so can be opened).  Values as in rccl.h (they are NCCL's ABI).
extern "C" {
typedef struct gpkNcclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
}

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

struct gpk_comm {
    ncclComm_t comm;
    const gpk::Rccl* table;  // the transport: RCCL's entry points (rccl()), or the in-process one of gpk_comm_init_mock
    int32_t rank, world, device;
    // device scratch of the collectives' small words (header rows of every rank + this rank's own row; the agreement words), reserved
    // when the communicator is created: nothing is allocated between the first collective of an exchange and its last, so a rank
    // cannot fail locally — and leave its peers blocked inside a collective — for want of scratch
    int64_t* scratch;
};

namespace gpk {

// ---- an in-process transport (tests): `world` THREADS of one process on one device stand in for the ranks -------------------------
// The collectives keep RCCL's signatures, so gpk_allgatherv_* run unchanged over it — header all-gather, the agreement, the grouped
// broadcasts, placement, rebase and validity repack — with W ranks that really are apart in time (each on its own thread and stream).
// Every call is a rendezvous: the rank makes what it queued visible (a stream sync), posts its pointer, all ranks meet, each copies what
// it is owed (device to device), all ranks meet again.
struct MockWorld {
    int W = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long gen = 0;
    const void* ptr[64];
    void meet() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long g = gen;
        if (++arrived == W) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};
struct MockRank {
    MockWorld* w;
    int rank;
};
static size_t mock_size(ncclDataType_t t) { return t == ncclInt64 || t == ncclUint64 || t == ncclFloat64 ? 8 : (t == ncclInt32 || t == ncclUint32 || t == ncclFloat32 ? 4 : (t == ncclFloat16 ? 2 : 1)); }
static ncclResult_t mock_all_gather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    MockRank* m = reinterpret_cast<MockRank*>(comm);
    const size_t bytes = count * mock_size(t);
    bool ok = hipStreamSynchronize(s) == hipSuccess;
    m->w->ptr[m->rank] = send;
    m->w->meet();
    for (int k = 0; k < m->w->W; ++k) ok = ok && hipMemcpy((char*)recv + (size_t)k * bytes, m->w->ptr[k], bytes, hipMemcpyDeviceToDevice) == hipSuccess;
    m->w->meet();
    return ok ? ncclSuccess : (ncclResult_t)1;
}
static ncclResult_t mock_broadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s) {
    MockRank* m = reinterpret_cast<MockRank*>(comm);
    bool ok = hipStreamSynchronize(s) == hipSuccess;
    if (m->rank == root) m->w->ptr[root] = send;
    m->w->meet();
    if (recv != m->w->ptr[root]) ok = ok && hipMemcpy(recv, m->w->ptr[root], count * mock_size(t), hipMemcpyDeviceToDevice) == hipSuccess;
    m->w->meet();
    return ok ? ncclSuccess : (ncclResult_t)1;
}
static ncclResult_t mock_group() { return ncclSuccess; }
static ncclResult_t mock_destroy(ncclComm_t comm) {
    delete reinterpret_cast<MockRank*>(comm);
    return ncclSuccess;
}
static const char* mock_error(ncclResult_t) { return "in-process transport: a device copy failed"; }
static const Rccl* mock_table() {
    static const Rccl t = [] {
        Rccl r;
        r.lib = (void*)1;
        r.CommDestroy = mock_destroy;
        r.AllGather = mock_all_gather;
        r.Broadcast = mock_broadcast;
        r.GroupStart = mock_group;
        r.GroupEnd = mock_group;
        r.GetErrorString = mock_error;
        return r;
    }();
    return &t;
}

// ---- who holds the shards of a column, and how a piece gets to its place --------------------------------------------------
// The exchange has two halves.  MOVE: the lengths of every shard become known everywhere and shard k's bytes of each buffer land
// at an offset of the gathered buffer — between ranks with one small all-gather + one grouped round of broadcasts per buffer
// (RCCL), or, when one process holds all the shards (a chunked column, a test), with device-to-device copies.  ASSEMBLE: where
// the pieces land (shard k > 0 drops the leading entry of its offsets), what is added to shard k's offsets (the children of the
// shards before it, minus the shard's own first offset: a device view need not start at 0), how validity bits are repacked
// (a shard's rows do not start on a byte of the gathered bitmap).  assemble_column is written once against `Shards`; only
// move_pieces / gather_header know which kind it is — so gpk_geoarray_concat runs, on one GPU with K shards, every line of
// placement arithmetic that gpk_allgatherv_geoarray runs with K ranks.
struct Shards {
    int W = 0;                                   // ranks, or shards held here
    int me = -1;                                 // this rank (RCCL), -1: every shard is local
    const Rccl* r = nullptr;
    gpk_comm* c = nullptr;
    const gpk_geoarray* const* local = nullptr;  // me < 0: the K shards; me >= 0: local[0] = this rank's shard
    const DevGeo& mine_or(int k) const { return me < 0 ? local[k]->d : local[0]->d; }
    bool holds(int k) const { return me < 0 || k == me; }
};
constexpr int HDR = 10;  // n_geoms, n_parts, n_rings, n_coords, has_validity, type, first geom / part / ring offset, the owning rank's local status
// the first offsets of up to COMM_MAX_WORLD shards, read on the device: thread k fills row k's entries 6 .. 8 (the rows' other entries
// were copied in from the host; a shard without rows has no first offsets)
struct ShardOffsets {
    const int32_t* geom_off[COMM_MAX_WORLD];
    const int32_t* part_off[COMM_MAX_WORLD];
    const int32_t* ring_off[COMM_MAX_WORLD];
    int32_t n;
};
__global__ void first_offsets_kernel(ShardOffsets so, int64_t* __restrict__ hdr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= so.n) return;
    int64_t* h = hdr + (size_t)HDR * k;
    if (h[0] <= 0) return;
    h[6] = so.geom_off[k] ? so.geom_off[k][0] : 0;
    h[7] = so.part_off[k] ? so.part_off[k][0] : 0;
    h[8] = so.ring_off[k] ? so.ring_off[k][0] : 0;
}
static void header_of(const DevGeo& d, int64_t* h) {
    const bool has_part = d.type == GPK_GEOM_MULTIPOLYGON,
               has_ring = d.type == GPK_GEOM_POLYGON || d.type == GPK_GEOM_MULTILINESTRING || d.type == GPK_GEOM_MULTIPOLYGON;
    h[0] = d.n_geoms;
    h[1] = has_part ? d.n_parts : 0;
    h[2] = has_ring ? d.n_rings : 0;
    h[3] = d.n_coords;
    h[4] = d.validity ? 1 : 0;
    h[5] = d.type;
    h[6] = h[7] = h[8] = h[9] = 0;
}
// hdr[HDR * W] on the host: every shard's sizes and first offsets (the only host read of the exchange).  Entry 9 of a row is the
// owning rank's LOCAL STATUS (GPK_OK unless something failed before the exchange): it travels inside the all-gather every rank takes
// part in, so a rank that failed locally still makes the collective call and every rank learns of it (*peer_rc) instead of blocking.
// One host -> device copy and one kernel for all the rows this process stages (a chunked column's K rows used to pay K syncs).
static int32_t gather_header(const Shards& S, std::vector<int64_t>& hdr, hipStream_t s, int32_t local_rc, int32_t* peer_rc) {
    const int W = S.W;
    *peer_rc = GPK_OK;
    hdr.assign((size_t)HDR * W, 0);
    int64_t* hdr_dev = nullptr;
    if (S.me >= 0) {
        hdr_dev = S.c->scratch;  // (reserved with the communicator: HDR * (W + 1) words)
    } else {
        GPK_TRY(workspace_aux(0).begin(sizeof(int64_t) * (size_t)(HDR * (W + 1)) + 512));
        hdr_dev = (int64_t*)workspace_aux(0).take(sizeof(int64_t) * (size_t)(HDR * (W + 1)));
    }
    const int n_rows = S.me < 0 ? W : 1;
    std::vector<int64_t> rows((size_t)HDR * n_rows, 0);
    ShardOffsets so;
    memset(&so, 0, sizeof so);
    so.n = n_rows;
    for (int k = 0; k < n_rows; ++k) {
        const DevGeo& d = S.local[k]->d;
        header_of(d, rows.data() + (size_t)HDR * k);
        rows[(size_t)HDR * k + 9] = S.me < 0 ? GPK_OK : local_rc;
        so.geom_off[k] = d.geom_off;
        so.part_off[k] = d.part_off;
        so.ring_off[k] = d.ring_off;
    }
    int64_t* stage_at = S.me < 0 ? hdr_dev : hdr_dev + (size_t)HDR * W;
    hipError_t e = hipMemcpyAsync(stage_at, rows.data(), sizeof(int64_t) * rows.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && local_rc == GPK_OK) {
        hipLaunchKernelGGL(first_offsets_kernel, dim3(1), dim3(COMM_MAX_WORLD), 0, s, so, stage_at);
        e = hipGetLastError();
    }
    if (S.me >= 0) {  // (always: also after a local failure — the row then says so)
        const ncclResult_t n = S.r->AllGather(stage_at, hdr_dev, HDR, ncclInt64, S.c->comm, s);
        if (n != ncclSuccess) return fail(GPK_ERR_DEVICE, "ncclAllGather failed: %s", S.r->GetErrorString(n));
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hdr.data(), hdr_dev, sizeof(int64_t) * (size_t)(HDR * W), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // (`rows` is a local: the copy out of it is done by now)
    if (e != hipSuccess) return fail(GPK_ERR_DEVICE, "allgatherv: %s", hipGetErrorString(e));
    for (int k = 0; k < W; ++k)
        if (hdr[(size_t)HDR * k + 9] != GPK_OK && *peer_rc == GPK_OK) *peer_rc = (int32_t)hdr[(size_t)HDR * k + 9];
    return GPK_OK;
}
// piece k of one buffer (counts[k] elements of `elem` bytes, read at src(k)) -> out + begin[k] * elem.  Every rank issues the same
// sequence of collectives whatever its own shard holds (the counts come from the header).
template <typename SrcOf>
static int32_t move_pieces(const Shards& S, SrcOf src, const int64_t* counts, const int64_t* begin, size_t elem, char* out, hipStream_t s) {
    if (S.me < 0) {
        for (int k = 0; k < S.W; ++k)
            if (counts[k] > 0) GPK_HIP(hipMemcpyAsync(out + (size_t)begin[k] * elem, src(k), (size_t)counts[k] * elem, hipMemcpyDeviceToDevice, s));
        return GPK_OK;
    }
    GPK_NCCL(S.r, S.r->GroupStart());
    for (int k = 0; k < S.W; ++k) {
        if (counts[k] == 0) continue;
        const ncclResult_t n = S.r->Broadcast(k == S.me ? src(k) : nullptr, out + (size_t)begin[k] * elem, (size_t)counts[k] * elem, ncclUint8, k, S.c->comm, s);
        if (n != ncclSuccess) {
            (void)S.r->GroupEnd();
            return fail(GPK_ERR_DEVICE, "ncclBroadcast failed: %s", S.r->GetErrorString(n));
        }
    }
    GPK_NCCL(S.r, S.r->GroupEnd());
    return GPK_OK;
}
// every rank learns whether every rank got this far (its allocations succeeded): a rank that failed locally must not leave
// the others waiting inside the broadcasts that follow
static int32_t agree(const Shards& S, int32_t my_rc, hipStream_t s) {
    if (S.me < 0) return my_rc;
    int64_t* w = S.c->scratch + (size_t)HDR * (S.W + 1);  // (reserved with the communicator: W + 1 words behind the header rows)
    const int64_t mine = my_rc;
    std::vector<int64_t> all((size_t)S.W);
    GPK_HIP(hipMemcpyAsync(w + S.W, &mine, sizeof mine, hipMemcpyHostToDevice, s));
    GPK_NCCL(S.r, S.r->AllGather(w + S.W, w, 1, ncclInt64, S.c->comm, s));
    GPK_HIP(hipMemcpyAsync(all.data(), w, sizeof(int64_t) * (size_t)S.W, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    if (my_rc != GPK_OK) return my_rc;
    for (int k = 0; k < S.W; ++k)
        if (all[(size_t)k] != GPK_OK) return fail((int32_t)all[(size_t)k], "allgatherv: rank %d could not allocate its copy of the column", k);
    return GPK_OK;
}

static int32_t assemble_column(const Shards& S, hipStream_t s, gpk_geoarray** out, int64_t* row_bases /* W + 1, may be NULL */, int64_t* out_bytes) {
    const int W = S.W;
    *out = nullptr;
    const DevGeo& d0 = S.mine_or(0);
    const int32_t type = d0.type;
    const bool has_geom = type != GPK_GEOM_POINT, has_part = type == GPK_GEOM_MULTIPOLYGON,
               has_ring = type == GPK_GEOM_POLYGON || type == GPK_GEOM_MULTILINESTRING || type == GPK_GEOM_MULTIPOLYGON;
    std::vector<int64_t> hdr;
    int32_t peer_rc = GPK_OK;
    GPK_TRY(gather_header(S, hdr, s, GPK_OK, &peer_rc));
    if (peer_rc != GPK_OK) return fail(peer_rc, "allgatherv: a rank reported status %d before the exchange", peer_rc);
    bool any_valid = false;
    int64_t tot[4] = {0, 0, 0, 0};
    for (int k = 0; k < W; ++k) {
        if (hdr[(size_t)HDR * k + 5] != type)
            return fail(GPK_ERR_MISMATCHED_GEOMETRY, "allgatherv: shard %d holds geometry type %lld, this one %d", k, (long long)hdr[(size_t)HDR * k + 5], type);
        any_valid = any_valid || hdr[(size_t)HDR * k + 4] != 0;
        for (int q = 0; q < 4; ++q) tot[q] += hdr[(size_t)HDR * k + q];
    }
    if (tot[0] > INT32_MAX || tot[1] > INT32_MAX || tot[2] > INT32_MAX || tot[3] > INT32_MAX)
        return fail(GPK_ERR_INVALID_ARGUMENT, "allgatherv: the gathered column exceeds 2^31 - 1 rows / rings / coordinates (Arrow i32 offsets)");
    if (row_bases) {
        row_bases[0] = 0;
        for (int k = 0; k < W; ++k) row_bases[k + 1] = row_bases[k] + hdr[(size_t)HDR * k + 0];
    }

    gpk_geoarray* a = new gpk_geoarray;
    memset(a, 0, sizeof *a);
    a->device = S.local[0]->device;
    a->d.type = type;
    a->d.n_geoms = tot[0];
    a->d.n_parts = has_part ? tot[1] : (is_polygonal(type) ? tot[0] : 0);
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
    // ---- every output buffer and all scratch first, then agree that every rank has them: nothing below can fail locally
    struct Level {
        bool present;
        int slot, rows_q, child_q, first_q;  // header columns: rows of this level, rows of its child level (3 = coordinates), first offset
        const int32_t* DevGeo::*src;
        const int32_t** view;
    } levels[3] = {
        {has_geom, 1, 0, has_part ? 1 : (has_ring ? 2 : 3), 6, &DevGeo::geom_off, &a->d.geom_off},
        {has_part, 2, 1, 2, 7, &DevGeo::part_off, &a->d.part_off},
        {has_ring, 3, 2, 3, 8, &DevGeo::ring_off, &a->d.ring_off},
    };
    auto alloc = [&](int slot, size_t bytes, const void** view) -> int32_t {
        void* p = nullptr;
        const hipError_t e = device_malloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) return fail(GPK_ERR_OOM, "allgatherv: device_malloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        a->owned[slot] = p;
        *view = p;
        a->nbytes += (int64_t)bytes;
        return GPK_OK;
    };
    int32_t rc = alloc(0, sizeof(double2) * (size_t)tot[3], (const void**)&a->d.xy);
    for (const Level& L : levels)
        if (rc == GPK_OK && L.present) rc = alloc(L.slot, sizeof(int32_t) * (size_t)(tot[L.rows_q] + 1), (const void**)L.view);
    const bool validity = any_valid && tot[0] > 0;
    uint8_t *all_bytes = nullptr, *my_bytes = nullptr;
    int32_t* zero_word = nullptr;
    if (rc == GPK_OK && validity) rc = alloc(4, (size_t)((tot[0] + 7) / 8), (const void**)&a->d.validity);
    if (rc == GPK_OK) {
        int64_t local_rows = 0;
        for (int k = 0; k < W; ++k)
            if (S.holds(k)) local_rows += hdr[(size_t)HDR * k + 0];
        rc = workspace_aux(1).begin((validity ? (size_t)tot[0] + (size_t)local_rows : 0) + 2048);
        if (rc == GPK_OK) {
            zero_word = (int32_t*)workspace_aux(1).take(256);
            if (validity) {
                all_bytes = (uint8_t*)workspace_aux(1).take((size_t)tot[0]);
                my_bytes = (uint8_t*)workspace_aux(1).take((size_t)(local_rows ? local_rows : 1));
            }
            if (!zero_word || (validity && (!all_bytes || !my_bytes))) rc = fail(GPK_ERR_OOM, "allgatherv: scratch");
        }
    }
    rc = agree(S, rc, s);
    if (rc != GPK_OK) return done(rc);
    {
        const hipError_t e = hipMemsetAsync(zero_word, 0, 256, s);
        if (e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "allgatherv: %s", hipGetErrorString(e)));
    }

    std::vector<int64_t> cnt((size_t)W), begin((size_t)W + 1);
    // ---- coordinates
    begin[0] = 0;
    for (int k = 0; k < W; ++k) {
        cnt[(size_t)k] = hdr[(size_t)HDR * k + 3];
        begin[(size_t)k + 1] = begin[(size_t)k] + cnt[(size_t)k];
    }
    if ((rc = move_pieces(S, [&](int k) { return (const void*)S.mine_or(k).xy; }, cnt.data(), begin.data(), sizeof(double2), (char*)a->owned[0], s)) != GPK_OK) return done(rc);
    // ---- offsets, outermost first: level `lvl` has rows + 1 entries per shard; its values count the children of the next level
    for (const Level& L : levels) {
        if (!L.present) continue;
        RebaseArgs ra;
        ra.world = W;
        int64_t at = 0, child = 0;
        for (int k = 0; k < W; ++k) {
            const int64_t rows = hdr[(size_t)HDR * k + L.rows_q];
            // shard 0 sends rows + 1 entries (its leading entry included), the others rows entries (from their entry 1 on)
            cnt[(size_t)k] = k == 0 ? rows + 1 : rows;
            begin[(size_t)k] = at;
            ra.dst_begin[k] = at;
            ra.shift[k] = child - hdr[(size_t)HDR * k + L.first_q];  // (a shard's offsets need not start at 0)
            at += cnt[(size_t)k];
            child += hdr[(size_t)HDR * k + L.child_q];
        }
        begin[(size_t)W] = at;
        ra.dst_begin[W] = at;
        auto src = [&](int k) -> const void* {
            const int32_t* p = S.mine_or(k).*(L.src);
            if (!p) return (const void*)zero_word;  // an empty shard without an offsets buffer: its one entry (shard 0 only) is 0
            return (const void*)(k == 0 ? p : p + 1);
        };
        if ((rc = move_pieces(S, src, cnt.data(), begin.data(), sizeof(int32_t), (char*)a->owned[L.slot], s)) != GPK_OK) return done(rc);
        if (at > 0) hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((at + 255) / 256)), dim3(256), 0, s, (int32_t*)a->owned[L.slot], ra);
    }
    // ---- validity: one byte per row travels (bit offsets of a shard's rows are not byte-aligned in the gathered bitmap)
    if (validity) {
        std::vector<const uint8_t*> my_at((size_t)W, nullptr);
        int64_t used = 0;
        for (int k = 0; k < W; ++k) {
            if (!S.holds(k)) continue;
            const DevGeo& dk = S.mine_or(k);
            my_at[(size_t)k] = my_bytes + used;
            if (dk.n_geoms > 0)
                hipLaunchKernelGGL(bitmap_to_bytes_kernel, dim3((unsigned)((dk.n_geoms + 255) / 256)), dim3(256), 0, s, dk.validity, dk.n_geoms, my_bytes + used);
            used += dk.n_geoms;
        }
        begin[0] = 0;
        for (int k = 0; k < W; ++k) {
            cnt[(size_t)k] = hdr[(size_t)HDR * k + 0];
            begin[(size_t)k + 1] = begin[(size_t)k] + cnt[(size_t)k];
        }
        if ((rc = move_pieces(S, [&](int k) { return (const void*)my_at[(size_t)k]; }, cnt.data(), begin.data(), 1, (char*)all_bytes, s)) != GPK_OK) return done(rc);
        hipLaunchKernelGGL(bytes_to_bitmap_kernel, dim3((unsigned)(((tot[0] + 7) / 8 + 255) / 256)), dim3(256), 0, s, (const uint8_t*)all_bytes, tot[0], (uint8_t*)a->owned[4]);
    }
    if (out_bytes) *out_bytes = a->nbytes;
    {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(GPK_ERR_DEVICE, "allgatherv: %s", hipGetErrorString(e)));
    }
    return done(GPK_OK);
}

}  // namespace gpk

using namespace gpk;

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
    c->table = r;
    const ncclResult_t n = r->CommInitRank(&c->comm, world, uid, rank);
    if (n != ncclSuccess) {
        delete c;
        return fail(GPK_ERR_DEVICE, "ncclCommInitRank failed: %s", r->GetErrorString(n));
    }
    c->scratch = nullptr;
    const hipError_t he = hipMalloc((void**)&c->scratch, sizeof(int64_t) * (size_t)((HDR + 1) * (world + 1)));
    if (he != hipSuccess) {
        (void)r->CommDestroy(c->comm);
        delete c;
        return fail(GPK_ERR_OOM, "comm: scratch: %s", hipGetErrorString(he));
    }
    *out = c;
    return GPK_OK;
}

int32_t gpk_comm_mock_world(int32_t world, void** out_world) {
    if (!out_world || world < 1 || world > COMM_MAX_WORLD) return fail(GPK_ERR_INVALID_ARGUMENT, "mock world of %d ranks", world);
    MockWorld* w = new MockWorld;
    w->W = world;
    *out_world = w;
    return GPK_OK;
}
int32_t gpk_comm_mock_world_free(void* world) {
    delete static_cast<MockWorld*>(world);
    return GPK_OK;
}
int32_t gpk_comm_init_mock(int32_t rank, void* world, gpk_comm** out) {
    if (!world || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    MockWorld* w = static_cast<MockWorld*>(world);
    *out = nullptr;
    if (rank < 0 || rank >= w->W) return fail(GPK_ERR_INVALID_ARGUMENT, "comm: rank %d of %d", rank, w->W);
    GPK_TRY(require_device());
    gpk_comm* c = new gpk_comm;
    c->rank = rank;
    c->world = w->W;
    c->table = mock_table();
    c->comm = reinterpret_cast<ncclComm_t>(new MockRank{w, rank});
    (void)hipGetDevice(&c->device);
    c->scratch = nullptr;
    const hipError_t he = hipMalloc((void**)&c->scratch, sizeof(int64_t) * (size_t)((HDR + 1) * (w->W + 1)));
    if (he != hipSuccess) {
        (void)mock_destroy(c->comm);
        delete c;
        return fail(GPK_ERR_OOM, "comm: scratch: %s", hipGetErrorString(he));
    }
    *out = c;
    return GPK_OK;
}

int32_t gpk_comm_free(gpk_comm* c) {
    if (!c) return GPK_OK;
    if (c->table) (void)c->table->CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
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
    const Rccl* r = c->table;
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
    Shards S;
    S.W = c->world;
    S.me = c->rank;
    S.r = r;
    S.c = c;
    GPK_TRY(move_pieces(S, [&](int) { return (const void*)local_dev; }, cnt.data(), begin.data(), sizeof(double) * (size_t)width, (char*)out_dev, s));
    GPK_HIP(hipStreamSynchronize(s));
    return GPK_OK;
}

// Every rank contributes its shard of a column and receives the concatenation in rank order as a new array handle (owned
// buffers, free with gpk_geoarray_free); *out_row_base = the first row of this rank's shard in it.  All ranks must call it
// with arrays of the same geometry type.
int32_t gpk_allgatherv_geoarray(gpk_comm* c, const gpk_geoarray* shard, void* stream, gpk_geoarray** out, int64_t* out_row_base, int64_t* out_bytes) {
    if (!c || !shard || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    const Rccl* r = c->table;
    Shards S;
    S.W = c->world;
    S.me = c->rank;
    S.r = r;
    S.c = c;
    const gpk_geoarray* one[1] = {shard};
    S.local = one;
    std::vector<int64_t> bases((size_t)c->world + 1);
    GPK_TRY(assemble_column(S, (hipStream_t)stream, out, bases.data(), out_bytes));
    if (out_row_base) *out_row_base = bases[(size_t)c->rank];
    return GPK_OK;
}

// K chunks of one column held by THIS process -> one array (Arrow's rechunk, py-geopolars/src/ffi.rs:56,73,93: the reference
// turns every Series into a single chunk before it looks at it).  The assembly is the all-gatherv's own (assemble_column): the
// pieces are placed with device copies instead of broadcasts.  out_row_bases[n_shards + 1] (host, may be NULL): first row of
// every chunk in the result.
int32_t gpk_geoarray_concat(const gpk_geoarray* const* shards, int32_t n_shards, void* stream, gpk_geoarray** out, int64_t* out_row_bases, int64_t* out_bytes) {
    if (!shards || !out || n_shards < 1 || n_shards > COMM_MAX_WORLD) return fail(GPK_ERR_INVALID_ARGUMENT, "concat: 1 .. %d chunks", COMM_MAX_WORLD);
    *out = nullptr;
    for (int k = 0; k < n_shards; ++k)
        if (!shards[k]) return fail(GPK_ERR_INVALID_ARGUMENT, "concat: chunk %d is NULL", k);
    GPK_TRY(require_device());
    Shards S;
    S.W = n_shards;
    S.me = -1;
    S.local = shards;
    return assemble_column(S, (hipStream_t)stream, out, out_row_bases, out_bytes);
}

}  // extern "C"
