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

}  // namespace

struct rr_comm {
    rr_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *scratch = nullptr;  // staging of rr_comm_allreduce_host, grow-only
    size_t scratch_count = 0;
};

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

}  // extern "C"
