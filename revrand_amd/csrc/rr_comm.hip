// The one exchange step of the path: every rank holds the sufficient statistics of ITS rows --
// what the reference's revrand/slm.py:145-157 computes from Phi (Phi^T Phi, Phi^T y) plus y^T y and N -- and ONE
// all-reduce over RCCL / xGMI sums them before the Cholesky.  RCCL is bound directly (dlopen of librccl.so.1, no
// PyTorch): ncclGetUniqueId / ncclCommInitRank / ncclAllReduce on the context's stream.  One process per GPU; the
// 128-byte id travels between the processes by whatever the host side has (a file, a socket, MPI, torchrun's
// store) -- revrand_amd/parallel.py uses a file or a TCP socket.
#include "rr_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;
};

RcclApi g_rccl;

int rccl_load(const char *path) {
    if (g_rccl.handle) return RR_OK;
    std::vector<std::string> cands;
    if (path && *path) cands.push_back(path);
    if (const char *e = getenv("RR_RCCL_LIB")) cands.push_back(e);
    cands.push_back("librccl.so.1");  // an already loaded copy (e.g. torch's) or the runpath / ld cache
    cands.push_back("/opt/rocm/lib/librccl.so.1");
    std::string tried;
    void *h = nullptr;
    for (const std::string &p : cands) {
        h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) {
            g_rccl.path = p;
            break;
        }
        tried += p + " (" + (dlerror() ? "not loadable" : "?") + "); ";
    }
    if (!h) {
        rr_set_error("rr_comm: librccl.so.1 could not be loaded: %s", tried.c_str());
        return RR_ERR_UNSUPPORTED;
    }
#define RR_SYM(field, name)                                                    \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));   \
    if (!g_rccl.field) {                                                       \
        rr_set_error("rr_comm: %s has no symbol %s", g_rccl.path.c_str(), name); \
        dlclose(h);                                                            \
        return RR_ERR_UNSUPPORTED;                                             \
    }
    RR_SYM(GetVersion, "ncclGetVersion")
    RR_SYM(GetUniqueId, "ncclGetUniqueId")
    RR_SYM(CommInitRank, "ncclCommInitRank")
    RR_SYM(CommInitAll, "ncclCommInitAll")
    RR_SYM(GroupStart, "ncclGroupStart")
    RR_SYM(GroupEnd, "ncclGroupEnd")
    RR_SYM(CommDestroy, "ncclCommDestroy")
    RR_SYM(CommAbort, "ncclCommAbort")
    RR_SYM(CommCount, "ncclCommCount")
    RR_SYM(CommUserRank, "ncclCommUserRank")
    RR_SYM(AllReduce, "ncclAllReduce")
    RR_SYM(Broadcast, "ncclBroadcast")
    RR_SYM(GetErrorString, "ncclGetErrorString")
#undef RR_SYM
    g_rccl.handle = h;
    return RR_OK;
}

#define RR_CHECK_NCCL(expr)                                                                      \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess) {                                                                 \
            rr_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return RR_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

// Row i of the packed upper triangle starts at i F - i (i - 1) / 2 and holds G[i][i .. F).
__device__ __forceinline__ int64_t tri_row_offset(int64_t i, int64_t F) { return i * F - (i * (i - 1)) / 2; }

// msg = [ upper triangle of G, row-major | b (F) | yty | nrows ]: F (F + 1) / 2 + F + 2 doubles -- half of the
// [G | b | yty] square a full-matrix exchange would move (67 MB instead of 134 MB at F = 4096).
__global__ void __launch_bounds__(256) rr_stats_pack_kernel(const double *__restrict__ G, const double *__restrict__ b,
                                                            const double *__restrict__ yty, double nrows, int64_t F,
                                                            double *__restrict__ msg) {
    const int64_t i = blockIdx.y;
    const int64_t j = i + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < F) msg[tri_row_offset(i, F) + (j - i)] = G[i * F + j];
    if (blockIdx.y == 0) {
        const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
        double *tail = msg + F * (F + 1) / 2;
        if (t < F) tail[t] = b ? b[t] : 0.0;
        if (t == 0) {
            tail[F] = yty ? yty[0] : 0.0;
            tail[F + 1] = nrows;
        }
    }
}

__global__ void __launch_bounds__(256) rr_stats_unpack_kernel(const double *__restrict__ msg, int64_t F,
                                                              double *__restrict__ G, double *__restrict__ b,
                                                              double *__restrict__ yty) {
    const int64_t i = blockIdx.y;
    const int64_t j = i + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < F) G[i * F + j] = msg[tri_row_offset(i, F) + (j - i)];
    if (blockIdx.y == 0) {
        const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const double *tail = msg + F * (F + 1) / 2;
        if (b && t < F) b[t] = tail[t];
        if (yty && t == 0) yty[0] = tail[F];
    }
}

// ---- the in-process transport (RR_TRANSPORT_PEER) ---------------------------------------------------------------
// The contexts of ONE process see each other's HBM through one address space: after hipDeviceEnablePeerAccess a kernel on
// device i loads device j's buffer directly over the xGMI link between them (the mesh is fully connected: 7 links per
// GPU).  The all-reduce is the two-step direct algorithm that topology wants, no ring:
//   step 1 (reduce-scatter)  member i owns slice i of the message: it reads slice i of EVERY member's buffer (its own
//                            from HBM, the others over their links, all links busy at once), combines them in member
//                            order 0, 1, ... and stores the result into slice i of its own buffer;
//   step 2 (all-gather)      member i copies slice j from its owner j, for every j != i.
// Each link carries count / n elements per step and direction; every element is combined ONCE, in one fixed order, and
// then copied, so all members end with bit-identical buffers and the same bits every run.  Members that share a device
// (the 1-GPU test box, `devices=[0, 0]`) take exactly the same code -- their "peer" loads stay in local HBM.
constexpr int RR_GROUP_MAX = 16;
struct PeerBufs {
    double *p[RR_GROUP_MAX];
};

template <int OP>
__device__ __forceinline__ double rr_peer_combine(double a, double b) {
    if (OP == RR_COMM_SUM) return a + b;
    if (OP == RR_COMM_MAX) return a > b ? a : b;
    return a < b ? a : b;
}

template <int OP>
__global__ void __launch_bounds__(256) rr_peer_reduce_kernel(PeerBufs bufs, int n, int self, int64_t off, int64_t len) {
    // four independent element streams per thread: n x 4 loads in flight cover the link latency
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < len; i0 += 4 * stride) {
        double acc[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            live[u] = i < len;
            acc[u] = live[u] ? bufs.p[0][off + i] : 0.0;
        }
        for (int j = 1; j < n; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (live[u]) acc[u] = rr_peer_combine<OP>(acc[u], bufs.p[j][off + i0 + u * stride]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (live[u]) bufs.p[self][off + i0 + u * stride] = acc[u];
    }
}

// blockIdx.y = owner j of the slice copied (j == self: nothing to do)
__global__ void __launch_bounds__(256) rr_peer_gather_kernel(PeerBufs bufs, int self, int64_t slice, int64_t count) {
    const int j = blockIdx.y;
    if (j == self) return;
    const int64_t off = (int64_t)j * slice;
    const int64_t len = count - off < slice ? count - off : slice;
    const double *__restrict__ src = bufs.p[j] + off;
    double *__restrict__ dst = bufs.p[self] + off;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < len; i0 += 4 * stride) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            v[u] = i < len ? src[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < len) dst[i] = v[u];
        }
    }
}

// dst (this member) = src (the root's buffer), bytes a multiple of 8: the broadcast of the peer transport
__global__ void __launch_bounds__(256) rr_peer_copy_kernel(const double *__restrict__ src, double *__restrict__ dst, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) dst[i] = src[i];
}

template <int OP>
static void peer_launch_reduce(rr_ctx *c, const PeerBufs &pb, int n, int self, int64_t off, int64_t len) {
    const int64_t want = (len + 4 * 256 - 1) / (4 * 256);
    const unsigned blocks = (unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want);
    hipLaunchKernelGGL(rr_peer_reduce_kernel<OP>, dim3(blocks), dim3(256), 0, c->stream, pb, n, self, off, len);
}

}  // namespace

// The members of one in-process group (rr_comm_init_all) share this record.
struct rr_group {
    int n = 0;
    int transport = RR_TRANSPORT_RCCL;
    std::vector<rr_comm *> members;
    // peer transport: per member, "my input is complete", "my slice is reduced", "my copies are done"
    std::vector<hipEvent_t> ready, reduced, gathered;
    hipEvent_t hub[3] = {nullptr, nullptr, nullptr};  // on member 0's device: "all are ready / reduced / done" (peer_wait_all)
    int alive = 0;  // members not yet destroyed
};

struct rr_comm {
    rr_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    rr_group *group = nullptr;  // in-process group (rr_comm_init_all); null for a one-process-per-GPU rank
    int rank = 0, world = 1;
    double *scratch = nullptr;  // staging of rr_comm_allreduce_host, grow-only
    size_t scratch_count = 0;
};

rr_ctx *rr_comm_ctx(rr_comm *comm) { return comm ? comm->ctx : nullptr; }

extern "C" {

int rr_comm_load(const char *path) { return rccl_load(path); }

int rr_comm_version(int *version, char *path, size_t path_len) {
    int rc = rccl_load(nullptr);
    if (rc != RR_OK) return rc;
    if (version) RR_CHECK_NCCL(g_rccl.GetVersion(version));
    if (path && path_len) snprintf(path, path_len, "%s", g_rccl.path.c_str());
    return RR_OK;
}

int rr_comm_unique_id(void *id) {
    RR_REQUIRE(id != nullptr, "rr_comm_unique_id: null output");
    int rc = rccl_load(nullptr);
    if (rc != RR_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == RR_COMM_ID_BYTES, "RR_COMM_ID_BYTES must be NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId u;
    RR_CHECK_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return RR_OK;
}

int rr_comm_init_rank(rr_ctx *ctx, int rank, int world, const void *id, rr_comm **out) {
    RR_REQUIRE(ctx != nullptr && id != nullptr && out != nullptr, "rr_comm_init_rank: null argument");
    RR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rr_comm_init_rank: rank %d not in [0, %d)", rank, world);
    *out = nullptr;
    int rc = rccl_load(nullptr);
    if (rc != RR_OK) return rc;
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    (void)hipGetLastError();  // (see rr_comm_init_all)
    RR_CHECK_NCCL(g_rccl.CommInitRank(&comm, world, u, rank));
    rr_comm *c = new rr_comm();
    c->ctx = ctx;
    c->comm = comm;
    // what RCCL itself reports, not what the caller claimed
    if (g_rccl.CommUserRank(comm, &c->rank) != ncclSuccess || g_rccl.CommCount(comm, &c->world) != ncclSuccess ||
        c->rank != rank || c->world != world) {
        rr_set_error("rr_comm_init_rank: RCCL reports rank %d of %d, asked for %d of %d", c->rank, c->world, rank, world);
        g_rccl.CommAbort(comm);
        delete c;
        return RR_ERR_HIP;
    }
    *out = c;
    return RR_OK;
}

void rr_comm_destroy(rr_comm *comm) {
    if (!comm) return;
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if (comm->comm) g_rccl.CommDestroy(comm->comm);
    if (comm->scratch) (void)hipFree(comm->scratch);
    if (rr_group *g = comm->group) {
        g->members[comm->rank] = nullptr;
        if (--g->alive == 0) {
            for (auto *evs : {&g->ready, &g->reduced, &g->gathered})
                for (hipEvent_t e : *evs)
                    if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : g->hub)
                if (e) (void)hipEventDestroy(e);
            delete g;
        }
    }
    delete comm;
}

int rr_comm_info(rr_comm *comm, int *rank, int *world) {
    RR_REQUIRE(comm != nullptr, "rr_comm_info: null communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return RR_OK;
}

static int comm_op(int op, ncclRedOp_t *out) {
    switch (op) {
    case RR_COMM_SUM: *out = ncclSum; return RR_OK;
    case RR_COMM_MAX: *out = ncclMax; return RR_OK;
    case RR_COMM_MIN: *out = ncclMin; return RR_OK;
    }
    rr_set_error("rr_comm: unknown reduction %d", op);
    return RR_ERR_INVALID;
}

int rr_comm_allreduce_dev(rr_comm *comm, double *dbuf, int64_t count, int op) {
    RR_REQUIRE(comm != nullptr && (dbuf != nullptr || count == 0) && count >= 0, "rr_comm_allreduce_dev: bad argument");
    ncclRedOp_t rop;
    int rc = comm_op(op, &rop);
    if (rc != RR_OK || count == 0) return rc;
    if (comm->group) {
        // one host thread drives all members of an in-process group: a collective entered member by member would wait
        // for its peers forever
        if (comm->world == 1) return RR_OK;
        rr_set_error("rr_comm_allreduce_dev: a member of an in-process group (rr_comm_init_all) -- use rr_comm_group_allreduce_dev");
        return RR_ERR_UNSUPPORTED;
    }
    RR_CHECK_HIP(hipSetDevice(comm->ctx->device));
    RR_CHECK_NCCL(g_rccl.AllReduce(dbuf, dbuf, (size_t)count, ncclDouble, rop, comm->comm, comm->ctx->stream));
    return RR_OK;
}

int rr_comm_allreduce_host(rr_comm *comm, double *hbuf, int64_t count, int op) {
    RR_REQUIRE(comm != nullptr && (hbuf != nullptr || count == 0) && count >= 0, "rr_comm_allreduce_host: bad argument");
    if (count == 0) return RR_OK;
    rr_ctx *c = comm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    if (comm->scratch_count < (size_t)count) {
        if (comm->scratch) {
            RR_CHECK_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(comm->scratch);
            comm->scratch = nullptr;
            comm->scratch_count = 0;
        }
        size_t want = (size_t)count < 4096 ? 4096 : (size_t)count;
        RR_CHECK_HIP(hipMalloc(&comm->scratch, want * sizeof(double)));
        comm->scratch_count = want;
    }
    RR_CHECK_HIP(hipMemcpyAsync(comm->scratch, hbuf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = rr_comm_allreduce_dev(comm, comm->scratch, count, op);
    if (rc != RR_OK) return rc;
    RR_CHECK_HIP(hipMemcpyAsync(hbuf, comm->scratch, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    return RR_OK;
}

int rr_comm_broadcast_host(rr_comm *comm, void *hbuf, int64_t bytes, int root) {
    RR_REQUIRE(comm != nullptr && (hbuf != nullptr || bytes == 0) && bytes >= 0 && root >= 0 && root < comm->world,
               "rr_comm_broadcast_host: bad argument");
    if (bytes == 0) return RR_OK;
    if (comm->group) {
        if (comm->world == 1) return RR_OK;
        rr_set_error("rr_comm_broadcast_host: a member of an in-process group (rr_comm_init_all) holds the host bytes already");
        return RR_ERR_UNSUPPORTED;
    }
    rr_ctx *c = comm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t words = ((size_t)bytes + 7) / 8;
    if (comm->scratch_count < words) {
        if (comm->scratch) {
            RR_CHECK_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(comm->scratch);
            comm->scratch = nullptr;
            comm->scratch_count = 0;
        }
        size_t want = words < 4096 ? 4096 : words;
        RR_CHECK_HIP(hipMalloc(&comm->scratch, want * sizeof(double)));
        comm->scratch_count = want;
    }
    if (comm->rank == root)
        RR_CHECK_HIP(hipMemcpyAsync(comm->scratch, hbuf, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    RR_CHECK_NCCL(g_rccl.Broadcast(comm->scratch, comm->scratch, (size_t)bytes, ncclChar, root, comm->comm, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(hbuf, comm->scratch, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    return RR_OK;
}

int rr_comm_barrier(rr_comm *comm) {
    double one = 1.0;
    int rc = rr_comm_allreduce_host(comm, &one, 1, RR_COMM_SUM);
    if (rc != RR_OK) return rc;
    RR_REQUIRE((int)(one + 0.5) == comm->world, "rr_comm_barrier: %g ranks answered, %d expected", one, comm->world);
    return RR_OK;
}

int64_t rr_stats_msg_count(int64_t F) { return F < 0 ? -1 : F * (F + 1) / 2 + F + 2; }

int rr_stats_pack_dev(rr_ctx *c, int64_t F, const double *dG, const double *db, const double *dyty, double nrows,
                      double *dmsg) {
    RR_REQUIRE(c != nullptr && dG != nullptr && dmsg != nullptr && F > 0 && F < 65536, "rr_stats_pack_dev: bad argument");
    RR_CHECK_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(rr_stats_pack_kernel, dim3((unsigned)((F + 255) / 256), (unsigned)F), dim3(256), 0, c->stream, dG,
                       db, dyty, nrows, F, dmsg);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_stats_unpack_dev(rr_ctx *c, int64_t F, const double *dmsg, double *dG, double *db, double *dyty) {
    RR_REQUIRE(c != nullptr && dG != nullptr && dmsg != nullptr && F > 0 && F < 65536, "rr_stats_unpack_dev: bad argument");
    RR_CHECK_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(rr_stats_unpack_kernel, dim3((unsigned)((F + 255) / 256), (unsigned)F), dim3(256), 0, c->stream,
                       dmsg, F, dG, db, dyty);
    RR_CHECK_HIP(hipGetLastError());
    return rr_symmetrize_dev(c, dG, F);
}

int rr_comm_reduce_stats_dev(rr_comm *comm, int64_t F, double *dG, double *db, double *dyty, double nrows, double *dmsg,
                             double *total_rows) {
    RR_REQUIRE(comm != nullptr, "rr_comm_reduce_stats_dev: null communicator");
    int rc = rr_stats_pack_dev(comm->ctx, F, dG, db, dyty, nrows, dmsg);
    if (rc == RR_OK) rc = rr_comm_allreduce_dev(comm, dmsg, rr_stats_msg_count(F), RR_COMM_SUM);
    if (rc == RR_OK) rc = rr_stats_unpack_dev(comm->ctx, F, dmsg, dG, db, dyty);
    if (rc == RR_OK && total_rows) {
        RR_CHECK_HIP(hipMemcpyAsync(total_rows, dmsg + rr_stats_msg_count(F) - 1, sizeof(double), hipMemcpyDeviceToHost,
                                    comm->ctx->stream));
        RR_CHECK_HIP(hipStreamSynchronize(comm->ctx->stream));
    }
    return rc;
}

// ---- ONE process, several GPUs (rr_init(n_devices, device_ids, ...)) -----------------------------------

static int group_check(rr_comm *const *comms, int n, const char *who) {
    RR_REQUIRE(comms != nullptr && n >= 1 && comms[0] != nullptr && comms[0]->group != nullptr, "%s: not the members of an in-process group", who);
    rr_group *g = comms[0]->group;
    RR_REQUIRE(g->n == n, "%s: the group has %d members, %d given", who, g->n, n);
    for (int i = 0; i < n; ++i)
        RR_REQUIRE(comms[i] != nullptr && comms[i]->group == g && comms[i]->rank == i, "%s: member %d is not member %d of this group", who, i, i);
    return RR_OK;
}

static int peer_enable(int n, rr_ctx *const *ctxs) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int a = ctxs[i]->device, b = ctxs[j]->device;
            if (a == b) continue;
            int can = 0;
            RR_CHECK_HIP(hipDeviceCanAccessPeer(&can, a, b));
            if (!can) {
                rr_set_error("rr_comm_init_all: device %d cannot access device %d's memory (no peer link): peer transport unavailable", a, b);
                return RR_ERR_UNSUPPORTED;
            }
            RR_CHECK_HIP(hipSetDevice(a));
            hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                rr_set_error("hipDeviceEnablePeerAccess(%d -> %d) failed: %s", a, b, hipGetErrorString(e));
                return RR_ERR_HIP;
            }
            (void)hipGetLastError();
        }
    return RR_OK;
}

// ncclGroupStart -> every member's ncclAllReduce on its own stream -> ncclGroupEnd (one thread drives all members)
static int group_rccl_allreduce(rr_group *g, double *const *dbufs, int64_t count, ncclRedOp_t rop, const char *who) {
    RR_CHECK_NCCL(g_rccl.GroupStart());
    for (int i = 0; i < g->n; ++i) {
        rr_comm *m = g->members[(size_t)i];
        rr_ctx *c = m->ctx;
        hipError_t e = hipSetDevice(c->device);
        ncclResult_t r = e == hipSuccess ? g_rccl.AllReduce(dbufs[i], dbufs[i], (size_t)count, ncclDouble, rop, m->comm, c->stream)
                                         : ncclUnhandledCudaError;
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            rr_set_error("%s: member %d: %s", who, i, e != hipSuccess ? hipGetErrorString(e) : g_rccl.GetErrorString(r));
            return RR_ERR_HIP;
        }
    }
    RR_CHECK_NCCL(g_rccl.GroupEnd());
    return RR_OK;
}

// One 64-number all-reduce over the group's RCCL communicators, checked on every member: member i contributes i + 1 + j / 64.
static int group_rccl_probe(rr_group *g) {
    const int n = g->n, cnt = 64;
    std::vector<double *> bufs((size_t)n, nullptr);
    std::vector<double> h((size_t)cnt);
    int rc = RR_OK;
    for (int i = 0; i < n && rc == RR_OK; ++i) {
        rr_ctx *c = g->members[(size_t)i]->ctx;
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipMalloc((void **)&bufs[(size_t)i], cnt * sizeof(double));
        for (int j = 0; j < cnt; ++j) h[(size_t)j] = (double)(i + 1) + j / 64.0;
        if (e == hipSuccess) e = hipMemcpy(bufs[(size_t)i], h.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rr_set_error("probe: member %d: %s", i, hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    if (rc == RR_OK) rc = group_rccl_allreduce(g, bufs.data(), cnt, ncclSum, "probe");
    if (rc == RR_OK && getenv("RR_COMM_PROBE_FAIL")) {  // test switch: take the fallback as if the collective had failed
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(g->members[(size_t)i]->ctx->device);
            (void)hipStreamSynchronize(g->members[(size_t)i]->ctx->stream);
        }
        rr_set_error("probe: failure forced by RR_COMM_PROBE_FAIL");
        rc = RR_ERR_HIP;
    }
    for (int i = 0; i < n && rc == RR_OK; ++i) {
        rr_ctx *c = g->members[(size_t)i]->ctx;
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(h.data(), bufs[(size_t)i], cnt * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            rr_set_error("probe: member %d: %s", i, hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        for (int j = 0; j < cnt; ++j) {
            const double want = n * (n + 1) / 2.0 + n * (j / 64.0);
            if (h[(size_t)j] != want) {
                rr_set_error("probe: member %d holds %.17g where the sum is %.17g", i, h[(size_t)j], want);
                rc = RR_ERR_HIP;
                break;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        if (bufs[(size_t)i]) {
            (void)hipSetDevice(g->members[(size_t)i]->ctx->device);
            (void)hipFree(bufs[(size_t)i]);
        }
    return rc;
}

int rr_comm_init_all(int n, rr_ctx *const *ctxs, int transport, rr_comm **out) {
    RR_REQUIRE(ctxs != nullptr && out != nullptr, "rr_comm_init_all: null argument");
    RR_REQUIRE(n >= 1 && n <= RR_GROUP_MAX, "rr_comm_init_all: %d members (1..%d supported)", n, RR_GROUP_MAX);
    RR_REQUIRE(transport == RR_TRANSPORT_AUTO || transport == RR_TRANSPORT_RCCL || transport == RR_TRANSPORT_PEER,
               "rr_comm_init_all: unknown transport %d", transport);
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        RR_REQUIRE(ctxs[i] != nullptr, "rr_comm_init_all: context %d is null", i);
        out[i] = nullptr;
        for (int j = 0; j < i; ++j) {
            RR_REQUIRE(ctxs[j] != ctxs[i], "rr_comm_init_all: contexts %d and %d are the same (one context per member)", j, i);
            if (ctxs[j]->device == ctxs[i]->device) distinct = false;
        }
    }
    const bool was_auto = transport == RR_TRANSPORT_AUTO;  // (a preference from the environment keeps AUTO's fallback to peer)
    if (const char *e = getenv("RR_COMM_TRANSPORT")) {  // measurement / fallback switch for RR_TRANSPORT_AUTO
        if (transport == RR_TRANSPORT_AUTO && !strcmp(e, "peer")) transport = RR_TRANSPORT_PEER;
        if (transport == RR_TRANSPORT_AUTO && !strcmp(e, "rccl")) transport = RR_TRANSPORT_RCCL;
    }
    // RCCL wants one communicator rank per DEVICE (ncclCommInitAll refuses a device listed twice); members that share a
    // device -- or a box without a loadable librccl -- take the in-process peer transport
    if (transport == RR_TRANSPORT_RCCL)
        RR_REQUIRE(distinct || n == 1, "rr_comm_init_all: RCCL needs one member per device; members share a device -- use RR_TRANSPORT_PEER");
    if (transport == RR_TRANSPORT_AUTO) transport = (distinct && n > 1 && rccl_load(nullptr) == RR_OK) ? RR_TRANSPORT_RCCL : RR_TRANSPORT_PEER;
    std::vector<ncclComm_t> nc((size_t)n, nullptr);
    if (transport == RR_TRANSPORT_RCCL) {
        int rc = rccl_load(nullptr);
        if (rc != RR_OK) return rc;
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
        (void)hipGetLastError();  // (RCCL reads the runtime's last-error record: one left by an earlier, handled failure of
                                  // this process -- a refused allocation, a peer link already enabled -- is not its business)
        const ncclResult_t r = g_rccl.CommInitAll(nc.data(), n, devs.data());
        if (r != ncclSuccess) {
            if (!was_auto) {
                rr_set_error("ncclCommInitAll over %d devices failed: %s", n, g_rccl.GetErrorString(r));
                return RR_ERR_HIP;
            }
            // RR_TRANSPORT_AUTO: a node whose RCCL cannot make this communicator still has its peer links
            fprintf(stderr, "librevrand_hip: ncclCommInitAll over %d devices failed (%s): the device group uses the peer transport\n", n,
                    g_rccl.GetErrorString(r));
            (void)hipGetLastError();
            for (auto &x : nc) x = nullptr;
            transport = RR_TRANSPORT_PEER;
        }
    }
    if (transport == RR_TRANSPORT_PEER) {
        int rc = peer_enable(n, ctxs);
        if (rc != RR_OK) return rc;
    }
    rr_group *g = new rr_group();
    g->n = g->alive = n;
    g->transport = transport;
    g->members.resize((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {
        rr_comm *c = new rr_comm();
        c->ctx = ctxs[i];
        c->comm = nc[(size_t)i];
        c->group = g;
        c->rank = i;
        c->world = n;
        g->members[(size_t)i] = c;
        out[i] = c;
    }
    if (transport == RR_TRANSPORT_RCCL && was_auto) {
        // RR_TRANSPORT_AUTO: the communicator exists, but whether a grouped collective WORKS on this node is only known once
        // one has run (no collective of this library had crossed two physical GPUs when this was written).  One tiny
        // all-reduce through exactly the path the statistics take; if it errors or sums wrong, the group takes the peer
        // transport instead and says so.
        const int prc = group_rccl_probe(g);
        if (prc != RR_OK) {
            fprintf(stderr, "librevrand_hip: the first grouped RCCL all-reduce over %d devices failed (%s): the device group uses the "
                            "peer transport\n", n, rr_last_error());
            (void)hipGetLastError();
            for (int i = 0; i < n; ++i) {
                if (out[i]->comm) (void)g_rccl.CommAbort(out[i]->comm);
                out[i]->comm = nullptr;
            }
            int rc = peer_enable(n, ctxs);
            if (rc != RR_OK) {
                for (int k = 0; k < n; ++k) {
                    rr_comm_destroy(out[k]);
                    out[k] = nullptr;
                }
                return rc;
            }
            transport = g->transport = RR_TRANSPORT_PEER;
        }
    }
    if (transport == RR_TRANSPORT_PEER) {
        g->ready.resize((size_t)n, nullptr);
        g->reduced.resize((size_t)n, nullptr);
        g->gathered.resize((size_t)n, nullptr);
        for (int i = 0; i < n; ++i) {
            hipError_t e = hipSetDevice(ctxs[i]->device);
            for (auto *evs : {&g->ready, &g->reduced, &g->gathered})
                if (e == hipSuccess) e = hipEventCreateWithFlags(&(*evs)[(size_t)i], hipEventDisableTiming);
            for (int k = 0; k < 3 && i == 0 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&g->hub[k], hipEventDisableTiming);
            if (e != hipSuccess) {
                rr_set_error("rr_comm_init_all: event creation failed: %s", hipGetErrorString(e));
                for (int k = 0; k < n; ++k) {
                    rr_comm_destroy(out[k]);
                    out[k] = nullptr;
                }
                return RR_ERR_HIP;
            }
        }
    }
    return RR_OK;
}

int rr_comm_transport(rr_comm *comm) {
    if (!comm) return -1;
    return comm->group ? comm->group->transport : RR_TRANSPORT_RCCL;
}

// Every member's stream waits for the events `evs` of ALL members (recorded by the caller just before).  Pairwise that is
// n (n - 1) hipStreamWaitEvent calls -- three times per all-reduce: 168 of its ~210 runtime calls at n = 8, ~0.3 ms of host
// time, which a caller with small messages (the GLM step's two all-reduces of a few hundred KB: rr_glm_sgd_group_step) pays
// per step.  From three members on the waits go THROUGH member 0: its stream waits for the other n - 1 events and records
// `hub`, the other streams wait for `hub` -- 2 (n - 1) calls, one more hop of event latency.
static int peer_wait_all(rr_group *g, const std::vector<hipEvent_t> &evs, hipEvent_t hub) {
    const int n = g->n;
    static const bool pairwise = getenv("RR_COMM_PAIRWISE_WAITS") != nullptr;  // (A/B runs)
    if (n <= 2 || pairwise) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (j != i) RR_CHECK_HIP(hipStreamWaitEvent(g->members[(size_t)i]->ctx->stream, evs[(size_t)j], 0));
        return RR_OK;
    }
    rr_ctx *c0 = g->members[0]->ctx;
    RR_CHECK_HIP(hipSetDevice(c0->device));
    for (int j = 1; j < n; ++j) RR_CHECK_HIP(hipStreamWaitEvent(c0->stream, evs[(size_t)j], 0));
    RR_CHECK_HIP(hipEventRecord(hub, c0->stream));  // (behind evs[0], recorded on this stream before)
    for (int i = 1; i < n; ++i) RR_CHECK_HIP(hipStreamWaitEvent(g->members[(size_t)i]->ctx->stream, hub, 0));
    return RR_OK;
}

static int peer_allreduce(rr_group *g, double *const *dbufs, int64_t count, int op) {
    const int n = g->n;
    PeerBufs pb;
    for (int i = 0; i < RR_GROUP_MAX; ++i) pb.p[i] = i < n ? dbufs[i] : nullptr;
    // slices of whole 32-element (256-byte) runs, so that neighbouring owners never share a cache line
    int64_t slice = (count + n - 1) / n;
    slice = (slice + 31) / 32 * 32;
    for (int i = 0; i < n; ++i) {
        rr_ctx *c = g->members[(size_t)i]->ctx;
        RR_CHECK_HIP(hipSetDevice(c->device));
        RR_CHECK_HIP(hipEventRecord(g->ready[(size_t)i], c->stream));
    }
    int rc = peer_wait_all(g, g->ready, g->hub[0]);
    if (rc != RR_OK) return rc;
    for (int i = 0; i < n; ++i) {
        rr_ctx *c = g->members[(size_t)i]->ctx;
        RR_CHECK_HIP(hipSetDevice(c->device));
        const int64_t off = (int64_t)i * slice;
        const int64_t len = off >= count ? 0 : (count - off < slice ? count - off : slice);
        if (len > 0) {
            if (op == RR_COMM_SUM) peer_launch_reduce<RR_COMM_SUM>(c, pb, n, i, off, len);
            else if (op == RR_COMM_MAX) peer_launch_reduce<RR_COMM_MAX>(c, pb, n, i, off, len);
            else peer_launch_reduce<RR_COMM_MIN>(c, pb, n, i, off, len);
            RR_CHECK_HIP(hipGetLastError());
        }
        RR_CHECK_HIP(hipEventRecord(g->reduced[(size_t)i], c->stream));
    }
    rc = peer_wait_all(g, g->reduced, g->hub[1]);
    if (rc != RR_OK) return rc;
    for (int i = 0; i < n; ++i) {
        rr_ctx *c = g->members[(size_t)i]->ctx;
        RR_CHECK_HIP(hipSetDevice(c->device));
        const int64_t want = (slice + 4 * 256 - 1) / (4 * 256);
        const unsigned blocks = (unsigned)(want < 1 ? 1 : want > 512 ? 512 : want);
        hipLaunchKernelGGL(rr_peer_gather_kernel, dim3(blocks, (unsigned)n), dim3(256), 0, c->stream, pb, i, slice, count);
        RR_CHECK_HIP(hipGetLastError());
        RR_CHECK_HIP(hipEventRecord(g->gathered[(size_t)i], c->stream));
    }
    // a member's buffer is read by its peers until THEIR copies are done: what the caller queues next on any member's
    // stream (a memset of the accumulators, the next pack) must come after all of them
    return peer_wait_all(g, g->gathered, g->hub[2]);
}

int rr_comm_group_allreduce_dev(rr_comm *const *comms, int n, double *const *dbufs, int64_t count, int op) {
    int rc = group_check(comms, n, "rr_comm_group_allreduce_dev");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(dbufs != nullptr && count >= 0, "rr_comm_group_allreduce_dev: bad argument");
    ncclRedOp_t rop;
    rc = comm_op(op, &rop);
    rr_group *g = comms[0]->group;
    // (a one-member group has nothing to exchange -- except under RR_TRANSPORT_RCCL, where the one member still goes through
    // ncclGroupStart / ncclAllReduce / ncclGroupEnd: the symbols and the call order of the N-member case, on any box)
    if (rc != RR_OK || count == 0 || (n == 1 && g->transport != RR_TRANSPORT_RCCL)) return rc;
    for (int i = 0; i < n; ++i) {
        RR_REQUIRE(dbufs[i] != nullptr, "rr_comm_group_allreduce_dev: member %d has no buffer", i);
        for (int j = 0; j < i; ++j) RR_REQUIRE(dbufs[j] != dbufs[i], "rr_comm_group_allreduce_dev: members %d and %d share a buffer", j, i);
    }
    if (g->transport == RR_TRANSPORT_PEER) return peer_allreduce(g, dbufs, count, op);
    return group_rccl_allreduce(g, dbufs, count, rop, "rr_comm_group_allreduce_dev");
}

int rr_comm_group_broadcast_dev(rr_comm *const *comms, int n, void *const *dbufs, int64_t bytes, int root) {
    int rc = group_check(comms, n, "rr_comm_group_broadcast_dev");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(dbufs != nullptr && bytes >= 0 && bytes % 8 == 0 && root >= 0 && root < n,
               "rr_comm_group_broadcast_dev: bad argument (bytes must be a multiple of 8)");
    rr_group *g = comms[0]->group;
    if (bytes == 0 || (n == 1 && g->transport != RR_TRANSPORT_RCCL)) return RR_OK;
    for (int i = 0; i < n; ++i) RR_REQUIRE(dbufs[i] != nullptr, "rr_comm_group_broadcast_dev: member %d has no buffer", i);
    if (g->transport == RR_TRANSPORT_RCCL) {
        RR_CHECK_NCCL(g_rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            rr_ctx *c = comms[i]->ctx;
            hipError_t e = hipSetDevice(c->device);
            ncclResult_t r = e == hipSuccess ? g_rccl.Broadcast(dbufs[root], dbufs[i], (size_t)bytes, ncclChar, root, comms[i]->comm, c->stream)
                                             : ncclUnhandledCudaError;
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                rr_set_error("rr_comm_group_broadcast_dev: member %d: %s", i, e != hipSuccess ? hipGetErrorString(e) : g_rccl.GetErrorString(r));
                return RR_ERR_HIP;
            }
        }
        RR_CHECK_NCCL(g_rccl.GroupEnd());
        return RR_OK;
    }
    // peer transport: every other member pulls the root's buffer over its own link to the root
    rr_ctx *rc0 = comms[root]->ctx;
    RR_CHECK_HIP(hipSetDevice(rc0->device));
    RR_CHECK_HIP(hipEventRecord(g->ready[(size_t)root], rc0->stream));
    const int64_t count = bytes / 8;
    for (int i = 0; i < n; ++i) {
        if (i == root) continue;
        rr_ctx *c = comms[i]->ctx;
        RR_CHECK_HIP(hipSetDevice(c->device));
        RR_CHECK_HIP(hipStreamWaitEvent(c->stream, g->ready[(size_t)root], 0));
        const int64_t want = (count + 1023) / 1024;
        hipLaunchKernelGGL(rr_peer_copy_kernel, dim3((unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want)), dim3(256), 0, c->stream,
                           (const double *)dbufs[root], (double *)dbufs[i], count);
        RR_CHECK_HIP(hipGetLastError());
        RR_CHECK_HIP(hipEventRecord(g->gathered[(size_t)i], c->stream));
    }
    RR_CHECK_HIP(hipSetDevice(rc0->device));
    for (int i = 0; i < n; ++i)  // the root must not overwrite its buffer before the copies are done
        if (i != root) RR_CHECK_HIP(hipStreamWaitEvent(rc0->stream, g->gathered[(size_t)i], 0));
    return RR_OK;
}

int rr_comm_group_reduce_stats_dev(rr_comm *const *comms, int n, int64_t F, double *const *dG, double *const *db,
                                   double *const *dyty, const double *nrows, double *const *dmsg, double *total_rows) {
    int rc = group_check(comms, n, "rr_comm_group_reduce_stats_dev");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(dG != nullptr && dmsg != nullptr && nrows != nullptr, "rr_comm_group_reduce_stats_dev: null argument");
    for (int i = 0; i < n && rc == RR_OK; ++i)
        rc = rr_stats_pack_dev(comms[i]->ctx, F, dG[i], db ? db[i] : nullptr, dyty ? dyty[i] : nullptr, nrows[i], dmsg[i]);
    if (rc == RR_OK) rc = rr_comm_group_allreduce_dev(comms, n, dmsg, rr_stats_msg_count(F), RR_COMM_SUM);
    for (int i = 0; i < n && rc == RR_OK; ++i)
        rc = rr_stats_unpack_dev(comms[i]->ctx, F, dmsg[i], dG[i], db ? db[i] : nullptr, dyty ? dyty[i] : nullptr);
    if (rc == RR_OK && total_rows) {
        rr_ctx *c = comms[0]->ctx;
        RR_CHECK_HIP(hipSetDevice(c->device));
        RR_CHECK_HIP(hipMemcpyAsync(total_rows, dmsg[0] + rr_stats_msg_count(F) - 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    return rc;
}

}  // extern "C"
