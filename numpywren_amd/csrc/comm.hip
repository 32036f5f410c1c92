// comm.hip -- tile transport between the GPUs of one node, directly on RCCL (xGMI), behind the C-ABI.
//
// The reference has no worker-to-worker path: every tile a task needs is an S3 GET of the object another worker PUT
// (reference numpywren/matrix.py:508, 527) and every dependency a Redis counter.  Here a tile stays in the HBM of the
// GPU that produced it and moves only to the GPUs that own a consumer task, as grouped ncclSend / ncclRecv on a
// dedicated high-priority HIP stream per rank.
//
// librccl.so is ~570 MB, so it is NOT a link-time dependency of libnpw_hip.so: npw_comm_init() dlopen()s it and binds
// the dozen entry points it needs.  A single-GPU process never loads it.
//
// Ordering contract (the caller's side of NCCL's rule): all ranks issue their point-to-point operations in an order
// that is consistent per pair of ranks.  numpywren_amd/dist.py walks ONE global task sequence on every rank and posts
// the sends / receives of a task's outputs at the same point of it, which gives a total order.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "npw_internal.h"

namespace npw {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

template <typename F>
bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle != nullptr) return NPW_OK;
    const char* override_path = getenv("NPW_RCCL_LIB");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        if (n == nullptr || *n == 0) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h != nullptr) break;
    }
    if (h == nullptr)
        return set_error(NPW_ERR_UNSUPPORTED, "npw_comm: cannot load librccl (%s); set NPW_RCCL_LIB to its path", dlerror());
    Rccl r;
    r.handle = h;
    const bool ok = bind(h, "ncclGetUniqueId", r.GetUniqueId) && bind(h, "ncclCommInitRank", r.CommInitRank) &&
                    bind(h, "ncclCommDestroy", r.CommDestroy) && bind(h, "ncclCommAbort", r.CommAbort) &&
                    bind(h, "ncclSend", r.Send) && bind(h, "ncclRecv", r.Recv) && bind(h, "ncclGroupStart", r.GroupStart) &&
                    bind(h, "ncclGroupEnd", r.GroupEnd) && bind(h, "ncclGetErrorString", r.GetErrorString);
    if (!ok) return set_error(NPW_ERR_UNSUPPORTED, "npw_comm: librccl lacks a required entry point (%s)", dlerror());
    (void)bind(h, "ncclCommCount", r.CommCount);        // optional: npw_comm_info then reports RCCL's own answer
    (void)bind(h, "ncclCommUserRank", r.CommUserRank);
    g_rccl = r;
    return NPW_OK;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;  // the rank's transport stream (high priority: its small kernels must not queue
                                   // behind chip-filling trailing updates)
};

#define NPW_NCCL_CHECK(expr)                                                                           \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess)                                                                         \
            return ::npw::set_error(NPW_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), \
                                    __FILE__, __LINE__);                                               \
    } while (0)

inline Comm* as_comm(npw_comm_t c) { return static_cast<Comm*>(c); }

}  // namespace
}  // namespace npw

using namespace npw;

extern "C" {

int npw_comm_unique_id(void* id_out, size_t id_bytes) {
    NPW_REQUIRE(id_out != nullptr && id_bytes >= NPW_COMM_ID_BYTES, "npw_comm_unique_id: need a %d-byte buffer",
                NPW_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) <= NPW_COMM_ID_BYTES, "NPW_COMM_ID_BYTES too small");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NPW_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memset(id_out, 0, id_bytes);
    memcpy(id_out, &id, sizeof(id));
    return NPW_OK;
}

int npw_comm_init(npw_comm_t* comm_out, int rank, int world, const void* unique_id) {
    NPW_REQUIRE(comm_out != nullptr && unique_id != nullptr, "npw_comm_init: NULL argument");
    NPW_REQUIRE(world >= 1 && rank >= 0 && rank < world, "npw_comm_init: bad rank %d of %d", rank, world);
    int rc = load_rccl();
    if (rc) return rc;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) {
        delete c;
        return set_error(NPW_ERR_HIP, "npw_comm_init: hipGetDevice failed: %s", hipGetErrorString(e));
    }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // numerically lowest = highest priority
    e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    if (e != hipSuccess) {
        delete c;
        return set_error(NPW_ERR_HIP, "npw_comm_init: stream creation failed: %s", hipGetErrorString(e));
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return set_error(NPW_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    comm_live_changed(+1);   // resident-grid kernels now leave CUs to the transfer kernels (resident_cu_count)
    *comm_out = c;
    return NPW_OK;
}

int npw_comm_destroy(npw_comm_t comm) {
    if (comm == nullptr) return NPW_OK;
    Comm* c = as_comm(comm);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm != nullptr) {
        (void)g_rccl.CommDestroy(c->comm);
        comm_live_changed(-1);
    }
    // The transport stream is NOT destroyed: tiles that travelled on it remember its handle and record their release
    // events on it long after the communicator is gone (one idle stream per communicator for the life of the process).
    delete c;
    return NPW_OK;
}

int npw_comm_abort(npw_comm_t comm) {
    // ncclCommAbort: drops everything the communicator still has posted or in flight (an open group included) and frees
    // it without waiting for the peers -- what a rank does when it fails in the middle of an exchange, instead of
    // launching half a group whose transfers no longer match what its peers posted.  The handle stays valid for
    // npw_comm_destroy; every other call on it fails.
    NPW_REQUIRE(comm != nullptr, "npw_comm_abort: NULL communicator");
    Comm* c = as_comm(comm);
    if (c->comm != nullptr) {
        ncclComm_t dead = c->comm;
        c->comm = nullptr;
        comm_live_changed(-1);
        NPW_NCCL_CHECK(g_rccl.CommAbort(dead));
    }
    return NPW_OK;
}

int npw_comm_info(npw_comm_t comm, int* rank, int* world, npw_stream_t* stream) {
    NPW_REQUIRE(comm != nullptr, "npw_comm_info: NULL communicator");
    Comm* c = as_comm(comm);
    // rank / world as RCCL itself reports them for the live communicator (what the ranks that really joined add up to),
    // not what npw_comm_init was told; an aborted communicator answers with the values it was created with
    int r = c->rank, w = c->world;
    if (c->comm != nullptr && g_rccl.CommCount != nullptr) NPW_NCCL_CHECK(g_rccl.CommCount(c->comm, &w));
    if (c->comm != nullptr && g_rccl.CommUserRank != nullptr) NPW_NCCL_CHECK(g_rccl.CommUserRank(c->comm, &r));
    if (rank) *rank = r;
    if (world) *world = w;
    if (stream) *stream = c->stream;
    return NPW_OK;
}

int npw_comm_group_start(npw_comm_t comm) {
    NPW_REQUIRE(comm != nullptr, "npw_comm_group_start: NULL communicator");
    NPW_NCCL_CHECK(g_rccl.GroupStart());
    return NPW_OK;
}

int npw_comm_group_end(npw_comm_t comm) {
    NPW_REQUIRE(comm != nullptr, "npw_comm_group_end: NULL communicator");
    NPW_NCCL_CHECK(g_rccl.GroupEnd());
    return NPW_OK;
}

int npw_send_tile(npw_comm_t comm, const void* tile, size_t bytes, int dst, npw_stream_t stream) {
    NPW_REQUIRE(comm != nullptr && (tile != nullptr || bytes == 0), "npw_send_tile: NULL argument");
    Comm* c = as_comm(comm);
    NPW_REQUIRE(c->comm != nullptr, "npw_send_tile: the communicator has been aborted");
    NPW_REQUIRE(dst >= 0 && dst < c->world, "npw_send_tile: bad destination rank %d", dst);
    if (bytes == 0) return NPW_OK;
    NPW_NCCL_CHECK(g_rccl.Send(tile, bytes, ncclUint8, dst, c->comm, stream ? as_stream(stream) : c->stream));
    return NPW_OK;
}

int npw_recv_tile(npw_comm_t comm, void* tile, size_t bytes, int src, npw_stream_t stream) {
    NPW_REQUIRE(comm != nullptr && (tile != nullptr || bytes == 0), "npw_recv_tile: NULL argument");
    Comm* c = as_comm(comm);
    NPW_REQUIRE(c->comm != nullptr, "npw_recv_tile: the communicator has been aborted");
    NPW_REQUIRE(src >= 0 && src < c->world, "npw_recv_tile: bad source rank %d", src);
    if (bytes == 0) return NPW_OK;
    NPW_NCCL_CHECK(g_rccl.Recv(tile, bytes, ncclUint8, src, c->comm, stream ? as_stream(stream) : c->stream));
    return NPW_OK;
}

int npw_bcast_tile(npw_comm_t comm, void* tile, size_t bytes, int root, const int* members, int nmembers,
                   npw_stream_t stream) {
    NPW_REQUIRE(comm != nullptr && (tile != nullptr || bytes == 0), "npw_bcast_tile: NULL argument");
    NPW_REQUIRE(nmembers >= 0 && (members != nullptr || nmembers == 0), "npw_bcast_tile: bad member list");
    Comm* c = as_comm(comm);
    NPW_REQUIRE(c->comm != nullptr, "npw_bcast_tile: the communicator has been aborted");
    NPW_REQUIRE(root >= 0 && root < c->world, "npw_bcast_tile: bad root %d", root);
    if (bytes == 0) return NPW_OK;
    hipStream_t s = stream ? as_stream(stream) : c->stream;
    // xGMI is point-to-point: a panel tile needed by k GPUs is k sends on k links inside ONE group (one fused kernel,
    // the links run in parallel) -- not a ring, which would push every byte over every hop.
    if (c->rank == root) {
        NPW_NCCL_CHECK(g_rccl.GroupStart());
        for (int i = 0; i < nmembers; ++i) {
            if (members[i] == root) continue;
            NPW_REQUIRE(members[i] >= 0 && members[i] < c->world, "npw_bcast_tile: bad member rank %d", members[i]);
            ncclResult_t r = g_rccl.Send(tile, bytes, ncclUint8, members[i], c->comm, s);
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                return set_error(NPW_ERR_HIP, "ncclSend failed: %s", g_rccl.GetErrorString(r));
            }
        }
        NPW_NCCL_CHECK(g_rccl.GroupEnd());
        return NPW_OK;
    }
    for (int i = 0; i < nmembers; ++i)
        if (members[i] == c->rank) {
            NPW_NCCL_CHECK(g_rccl.Recv(tile, bytes, ncclUint8, root, c->comm, s));
            return NPW_OK;
        }
    return NPW_OK;  // not a member: nothing to do
}

}  // extern "C"
