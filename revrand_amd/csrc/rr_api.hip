// Host side of the C ABI: contexts, device memory, basis objects.
// See include/revrand_hip.h for the contract of every entry point.
#include "rr_internal.h"
#include <cstring>

#include <cmath>

static thread_local std::string g_last_error;

void rr_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

#ifdef RR_BOUNDS
#include <map>
#include <mutex>
namespace {
constexpr size_t RR_GUARD = 4096;
constexpr unsigned char RR_GUARD_BYTE = 0xA5;
struct GuardRec { size_t bytes; int device; };
std::map<void *, GuardRec> g_guarded;  // user pointer -> record
std::mutex g_guard_mu;
}  // namespace

hipError_t rr_guard_malloc(void **p, size_t bytes) {
    char *base = nullptr;
    hipError_t e = (hipMalloc)((void **)&base, bytes + 2 * RR_GUARD);
    if (e != hipSuccess) return e;
    (void)hipMemset(base, RR_GUARD_BYTE, RR_GUARD);
    (void)hipMemset(base + RR_GUARD + bytes, RR_GUARD_BYTE, RR_GUARD);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guarded[base + RR_GUARD] = GuardRec{bytes, dev};
    *p = base + RR_GUARD;
    return hipSuccess;
}

static int guard_check_one(void *user, const GuardRec &r, const char *where) {
    unsigned char h[2 * RR_GUARD];
    char *base = (char *)user - RR_GUARD;
    if (hipMemcpy(h, base, RR_GUARD, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h + RR_GUARD, (char *)user + r.bytes, RR_GUARD, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return RR_OK;  // the device is already in an error state: the caller reports that
    }
    for (size_t i = 0; i < 2 * RR_GUARD; ++i)
        if (h[i] != RR_GUARD_BYTE) {
            const long off = i < RR_GUARD ? (long)i - (long)RR_GUARD : (long)(r.bytes + (i - RR_GUARD));
            rr_set_error("RR_BOUNDS: out-of-bounds write at byte offset %ld of a %zu-byte device allocation (%p), found at %s",
                         off, r.bytes, user, where);
            return RR_ERR_HIP;
        }
    return RR_OK;
}

static std::string g_launch_violation;  // first launch made with another device current than its stream's (under g_guard_mu)
static long g_launch_checks = 0;

void rr_launch_device_check(hipStream_t s, const char *file, int line) {
    int cur = -1;
    hipDevice_t sd = -1;
    if (hipGetDevice(&cur) != hipSuccess || hipStreamGetDevice(s, &sd) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    std::lock_guard<std::mutex> lk(g_guard_mu);
    ++g_launch_checks;
    if (cur != (int)sd && g_launch_violation.empty()) {
        char buf[256];
        snprintf(buf, sizeof buf, "RR_BOUNDS: kernel launched at %s:%d with device %d current on a stream of device %d", file, line, cur, (int)sd);
        g_launch_violation = buf;
        fprintf(stderr, "%s\n", buf);
    }
}

long rr_launch_checks_done(void) {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    return g_launch_checks;
}

int rr_guard_check(const char *where) {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    if (!g_launch_violation.empty()) {
        rr_set_error("%s (reported by %s)", g_launch_violation.c_str(), where);
        return RR_ERR_HIP;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (auto &kv : g_guarded)
        if (kv.second.device == dev) {
            int rc = guard_check_one(kv.first, kv.second, where);
            if (rc != RR_OK) return rc;
        }
    return RR_OK;
}

hipError_t rr_guard_free(void *p) {
    if (!p) return hipSuccess;
    GuardRec r{0, 0};
    bool mine = false;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guarded.find(p);
        if (it != g_guarded.end()) {
            r = it->second;
            mine = true;
            g_guarded.erase(it);
        }
    }
    if (!mine) return (hipFree)(p);  // memory the caller allocated elsewhere
    (void)hipDeviceSynchronize();
    if (guard_check_one(p, r, "free") != RR_OK) fprintf(stderr, "%s\n", rr_last_error());
    return (hipFree)((char *)p - RR_GUARD);
}
#endif

extern "C" {

int rr_abi_version(void) { return RR_ABI_VERSION; }

int64_t rr_debug_launch_checks(void) {
#ifdef RR_BOUNDS
    return (int64_t)rr_launch_checks_done();
#else
    return 0;
#endif
}

int rr_build_flags(void) {
#ifdef RR_BOUNDS
    return RR_BUILD_BOUNDS;
#else
    return 0;
#endif
}

const char *rr_last_error(void) { return g_last_error.c_str(); }

int rr_device_count(int *count) {
    RR_REQUIRE(count != nullptr, "rr_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        rr_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return RR_ERR_NO_DEVICE;
    }
    *count = n;
    return RR_OK;
}

int rr_ctx_create(int device, rr_ctx **out) {
    RR_REQUIRE(out != nullptr, "rr_ctx_create: null output");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        rr_set_error("rr_ctx_create: no HIP device is visible (the hot path has no CPU fallback)");
        return RR_ERR_NO_DEVICE;
    }
    RR_REQUIRE(device >= 0 && device < n, "rr_ctx_create: device %d out of range [0,%d)", device, n);
    RR_CHECK_HIP(hipSetDevice(device));
    rr_ctx *c = new rr_ctx();
    c->device = device;
    hipError_t e = hipGetDeviceProperties(&c->prop, device);
    if (e != hipSuccess) {
        delete c;
        rr_set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return RR_ERR_HIP;
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        rr_set_error("rr_ctx_create: device %d is %s; this library carries gfx950 code only",
                     device, c->prop.gcnArchName);
        delete c;
        return RR_ERR_NO_DEVICE;
    }
    c->num_cu = c->prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        rr_set_error("rr_ctx_create: stream/event creation failed");
        delete c;
        return RR_ERR_HIP;
    }
    if (const char *det = getenv("RR_DETERMINISTIC")) c->deterministic = (atoi(det) != 0) ? 1 : 0;
    if (const char *eng = getenv("RR_SYRK_ENGINE")) c->gram_engine = !strcmp(eng, "bf16x3") ? 3 : !strcmp(eng, "bf16x4") ? 4 : !strcmp(eng, "fp16x3") ? 5 : 0;
    *out = c;
    return RR_OK;
}

int rr_set_gram_engine(rr_ctx *ctx, int engine) {
    RR_REQUIRE(ctx != nullptr, "rr_set_gram_engine: null context");
    RR_REQUIRE(engine == RR_GRAM_F32 || engine == RR_GRAM_BF16X3 || engine == RR_GRAM_BF16X4 || engine == RR_GRAM_FP16X3,
               "rr_set_gram_engine: unknown engine %d", engine);
    ctx->gram_engine = engine;
    return RR_OK;
}

int rr_get_gram_engine(rr_ctx *ctx) { return ctx ? ctx->gram_engine : -1; }

int rr_set_deterministic(rr_ctx *ctx, int on) {
    RR_REQUIRE(ctx != nullptr, "rr_set_deterministic: null context");
    ctx->deterministic = on ? 1 : 0;
    return RR_OK;
}

int rr_get_deterministic(rr_ctx *ctx) { return ctx ? ctx->deterministic : -1; }

void rr_ctx_destroy(rr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    if (ctx->stream3) {
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamDestroy(ctx->stream3);
    }
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->tile_map) (void)hipFree(ctx->tile_map);
    if (ctx->det) (void)hipFree(ctx->det);
    if (ctx->det2) (void)hipFree(ctx->det2);
    if (ctx->pb) (void)hipFree(ctx->pb);
    if (ctx->gsa) (void)hipFree(ctx->gsa);
    if (ctx->gsb) (void)hipFree(ctx->gsb);
    rr_posdef_scratch_free(ctx->posdef);
    for (int i = 0; i < 2; ++i) {
        if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]);
        if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

int rr_ctx_sync(rr_ctx *ctx) {
    RR_REQUIRE(ctx != nullptr, "rr_ctx_sync: null context");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    RR_CHECK_HIP(hipStreamSynchronize(ctx->stream));
#ifdef RR_BOUNDS
    return rr_guard_check("rr_ctx_sync");
#else
    return RR_OK;
#endif
}

int rr_ctx_info(rr_ctx *ctx, char name[64], int *compute_units, uint64_t *hbm_bytes) {
    RR_REQUIRE(ctx != nullptr, "rr_ctx_info: null context");
    if (name) {
        snprintf(name, 64, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    if (compute_units) *compute_units = ctx->num_cu;
    if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->prop.totalGlobalMem;
    return RR_OK;
}

void *rr_ctx_stream(rr_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int rr_ctx_pci_bus_id(rr_ctx *ctx, char id[32], int *device) {
    RR_REQUIRE(ctx != nullptr && id != nullptr, "rr_ctx_pci_bus_id: null argument");
    RR_CHECK_HIP(hipDeviceGetPCIBusId(id, 32, ctx->device));
    if (device) *device = ctx->device;
    return RR_OK;
}

int rr_peer_access(int device, int peer, int *can_access) {
    RR_REQUIRE(can_access != nullptr, "rr_peer_access: null output");
    int n = 0;
    RR_CHECK_HIP(hipGetDeviceCount(&n));
    RR_REQUIRE(device >= 0 && device < n && peer >= 0 && peer < n, "rr_peer_access: devices %d, %d out of range [0,%d)", device, peer, n);
    *can_access = 1;
    if (device != peer) RR_CHECK_HIP(hipDeviceCanAccessPeer(can_access, device, peer));
    return RR_OK;
}

int rr_malloc(rr_ctx *ctx, size_t bytes, void **dptr) {
    RR_REQUIRE(ctx != nullptr && dptr != nullptr, "rr_malloc: null argument");
    *dptr = nullptr;
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_malloc: hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return RR_ERR_OOM;
    }
    return RR_OK;
}

int rr_free(rr_ctx *ctx, void *dptr) {
    RR_REQUIRE(ctx != nullptr, "rr_free: null context");
    if (!dptr) return RR_OK;
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    RR_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    RR_CHECK_HIP(hipFree(dptr));
    return RR_OK;
}

int rr_memset(rr_ctx *ctx, void *dptr, int value, size_t bytes) {
    RR_REQUIRE(ctx != nullptr && (dptr != nullptr || bytes == 0), "rr_memset: null argument");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    if (bytes) RR_CHECK_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return RR_OK;
}

int rr_memcpy_h2d(rr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    RR_REQUIRE(ctx != nullptr && (bytes == 0 || (dst && src)), "rr_memcpy_h2d: null argument");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    if (bytes) {
        RR_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        RR_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return RR_OK;
}

int rr_memcpy_d2h(rr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    RR_REQUIRE(ctx != nullptr && (bytes == 0 || (dst && src)), "rr_memcpy_d2h: null argument");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    if (bytes) {
        RR_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        RR_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return RR_OK;
}

int rr_timer_start(rr_ctx *ctx) {
    RR_REQUIRE(ctx != nullptr, "rr_timer_start: null context");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    RR_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return RR_OK;
}

int rr_timer_stop(rr_ctx *ctx, float *ms) {
    RR_REQUIRE(ctx != nullptr && ms != nullptr, "rr_timer_stop: null argument");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    RR_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RR_CHECK_HIP(hipEventSynchronize(ctx->ev1));
    RR_CHECK_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RR_OK;
}

int rr_rff_create(rr_ctx *ctx, int compute, int d, int n, const double *W, rr_basis **out) {
    RR_REQUIRE(ctx != nullptr && out != nullptr && W != nullptr, "rr_rff_create: null argument");
    *out = nullptr;
    RR_REQUIRE(d >= 1 && n >= 1, "rr_rff_create: need d >= 1 and n >= 1 (got d=%d n=%d)", d, n);
    RR_REQUIRE(compute == RR_F32 || compute == RR_F64 || compute == RR_F32P64, "rr_rff_create: bad compute dtype %d", compute);
    if (compute == RR_F32P64 && d > 128) {
        rr_set_error("rr_rff_create: RR_F32P64 (float64 phases) needs Xdim <= 128, got %d: use RR_F64", d);
        return RR_ERR_UNSUPPORTED;
    }
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    rr_basis *b = new rr_basis();
    b->ctx = ctx;
    b->kind = RR_KIND_RFF;
    b->compute = compute == RR_F32P64 ? RR_F32 : compute;
    b->phase64 = compute == RR_F32P64;
    b->d = d;
    b->n = n;
    b->npad = ((n + 127) / 128) * 128;
    b->dpad = rr_pick_dmax(d);
    if (b->dpad == 0) {  // d > 128: W columns no longer fit in registers, see "Xdim > 128" in rr_rff.hip
        if (d > RR_MAX_XDIM) {
            delete b;
            rr_set_error("rr_rff_create: Xdim=%d > %d is not supported", d, RR_MAX_XDIM);
            return RR_ERR_UNSUPPORTED;
        }
        b->large = true;
        b->dpad = ((d + 127) / 128) * 128;
        b->npad = ((n + 255) / 256) * 256;
    }
    b->W.assign(W, W + (size_t)d * n);
    size_t elems = (size_t)b->dpad * b->npad;
    hipError_t e1 = hipMalloc((void **)&b->dWs32, elems * sizeof(float));
    hipError_t e2 = hipMalloc((void **)&b->dWs64, elems * sizeof(double));
    hipError_t e5 = hipMalloc((void **)&b->dWt32, elems * sizeof(float));
    hipError_t e3 = hipMalloc((void **)&b->dgfac32, (size_t)b->dpad * sizeof(float));
    hipError_t e4 = hipMalloc((void **)&b->dgfac64, (size_t)b->dpad * sizeof(double));
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_rff_create: device allocation failed");
        rr_basis_destroy(b);
        return RR_ERR_OOM;
    }
    *out = b;
    return RR_OK;
}

void rr_basis_destroy(rr_basis *b) {
    if (!b) return;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
    }
    if (b->dWs32) (void)hipFree(b->dWs32);
    if (b->dWs64) (void)hipFree(b->dWs64);
    if (b->dWraw) (void)hipFree(b->dWraw);
    if (b->dWt32) (void)hipFree(b->dWt32);
    if (b->dgfac32) (void)hipFree(b->dgfac32);
    if (b->dgfac64) (void)hipFree(b->dgfac64);
    if (b->dmu32) (void)hipFree(b->dmu32);
    if (b->dmu64) (void)hipFree(b->dmu64);
    if (b->zbuf) (void)hipFree(b->zbuf);
    if (b->lg_xt) (void)hipFree(b->lg_xt);
    if (b->lg_z) (void)hipFree(b->lg_z);
    rr_pass2_scratch_free(b->pass2);
    rr_pass2d_scratch_free(b->pass2d);
    for (hipEvent_t ev : b->events) (void)hipEventDestroy(ev);
    void *ff[] = {b->ffB32, b->ffG32, b->ffSrad32, b->ffSrev32, b->ffL32, b->ffB64, b->ffG64,
                  b->ffSrad64, b->ffSrev64, b->ffL64, b->ffPI};
    for (void *q : ff)
        if (q) (void)hipFree(q);
    delete b;
}

const char *rr_rff_gram_kernel_name(rr_basis *basis) { return basis ? basis->gram_kernel : ""; }

}  // extern "C"

// ---- deterministic mode: scratch of the ordered sums and their reduction (rr_internal.h) ----------------------
int rr_det_scratch(rr_ctx *c, size_t bytes, void **out) {
    if (c->det_bytes < bytes) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));  // an earlier reduction may still read the old scratch
        if (c->det) (void)hipFree(c->det);
        c->det = nullptr;
        c->det_bytes = 0;
        hipError_t e = hipMalloc(&c->det, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            rr_set_error("deterministic mode: could not allocate %zu bytes for the ordered partial sums", bytes);
            return RR_ERR_OOM;
        }
        c->det_bytes = bytes;
    }
    *out = c->det;
    return RR_OK;
}

__global__ void __launch_bounds__(256)
rr_det_reduce_kernel(const double *__restrict__ part, int64_t nslots, int64_t stride, int64_t count, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int64_t k = 0; k < nslots; ++k) s += part[k * stride + i];
    out[i] += s;
}

// first stage for many slots: group g sums its contiguous range of slots (fixed by nslots alone) into grp[g * count + i]
__global__ void __launch_bounds__(256)
rr_det_group_kernel(const double *__restrict__ part, int64_t nslots, int64_t stride, int64_t count, int64_t per_group,
                    double *__restrict__ grp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int64_t k0 = (int64_t)blockIdx.y * per_group;
    int64_t k1 = k0 + per_group;
    if (k1 > nslots) k1 = nslots;
    double s = 0.0;
    for (int64_t k = k0; k < k1; ++k) s += part[k * stride + i];
    grp[(int64_t)blockIdx.y * count + i] = s;
}

int rr_det_reduce(rr_ctx *c, const double *part, int64_t nslots, int64_t stride, int64_t count, double *out) {
    if (count <= 0) return RR_OK;
    const unsigned gx = (unsigned)((count + 255) / 256);
    constexpr int64_t GROUPS = 64;
    if (nslots >= 8 * GROUPS && count >= 256) {
        // one thread per element walking thousands of slots is slow (Phi^T y of a 2M-row chunk: 7816 slots, 2.7 ms):
        // 64 groups of consecutive slots first, then the 64 group sums in order -- the grouping depends on nslots only,
        // so the order of the additions is still fixed
        if (c->det2_count < (size_t)(GROUPS * count)) {
            RR_CHECK_HIP(hipStreamSynchronize(c->stream));
            if (c->det2) (void)hipFree(c->det2);
            c->det2 = nullptr;
            c->det2_count = 0;
            RR_CHECK_HIP(hipMalloc((void **)&c->det2, (size_t)(GROUPS * count) * sizeof(double)));
            c->det2_count = (size_t)(GROUPS * count);
        }
        const int64_t per_group = (nslots + GROUPS - 1) / GROUPS;
        const int64_t ngroups = (nslots + per_group - 1) / per_group;
        hipLaunchKernelGGL(rr_det_group_kernel, dim3(gx, (unsigned)ngroups), dim3(256), 0, c->stream, part, nslots, stride, count,
                           per_group, c->det2);
        hipLaunchKernelGGL(rr_det_reduce_kernel, dim3(gx), dim3(256), 0, c->stream, (const double *)c->det2, ngroups, count, count,
                           out);
    } else {
        hipLaunchKernelGGL(rr_det_reduce_kernel, dim3(gx), dim3(256), 0, c->stream, part, nslots, stride, count, out);
    }
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// Scale W by 1/(l_i * 2pi) in f64 on the host (d*n elements: tiny) and upload.  The kernels
// then obtain the phase directly in revolutions, which is what v_sin_f32/v_cos_f32 consume.
// The device copies of a random Fourier basis for new length scales, made ON the device (d <= 128): W stays resident as
// given, the d length scales travel as kernel arguments, one launch rewrites Ws (f32, f64), its transpose and the gradient
// factors -- the same float64 products and the same clamped float32 casts as the host loop below, without its five
// synchronous copies (an SVI step changes the length scales every time: ~0.15 ms of a 5 ms step).
struct LsArgs {
    double ls[128];
    int n_ls;
};
__global__ void __launch_bounds__(256)
rr_scale_w_kernel(const double *__restrict__ W, const LsArgs a, int d, int n, int npad, int dpad, float *__restrict__ w32,
                  double *__restrict__ w64, float *__restrict__ wt32, float *__restrict__ g32, double *__restrict__ g64) {
    const double inv2pi = 0.15915494309189533576888, twopi = 6.283185307179586476925, wmax = 4611686018427387904.0;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)d * n) return;
    const int i = (int)(t / n), f = (int)(t % n);
    const double l = a.ls[a.n_ls == 1 ? 0 : i];
    const double v = W[t] * (inv2pi / l);
    const float vf = (float)(v > wmax ? wmax : (v < -wmax ? -wmax : v));
    w64[(size_t)i * npad + f] = v;
    w32[(size_t)i * npad + f] = vf;
    wt32[(size_t)f * dpad + i] = vf;
    if (f == 0) {
        const double gf = twopi / l;
        g64[i] = gf;
        g32[i] = (float)(gf > 3.0e38 ? 3.0e38 : (gf < -3.0e38 ? -3.0e38 : gf));
    }
}

// The same rescaling with the length scales in DEVICE memory (the resident SVI loop: they are the optimiser's own
// coordinates and never visit the host between steps).
// `shift` (a spectral-mixture component, basis_functions.py:1443-1475: phases VX +- X . mean): every frequency of input dimension
// i is moved by sgn * shift[i] -- x . (W[:, f] / l + sgn mean) = (VX + sgn mX)[f].
__global__ void __launch_bounds__(256)
rr_scale_w_dev_kernel(const double *__restrict__ W, const double *__restrict__ ls, int n_ls, int d, int n, int npad, int dpad,
                      float *__restrict__ w32, double *__restrict__ w64, float *__restrict__ wt32, float *__restrict__ g32,
                      double *__restrict__ g64, const double *__restrict__ shift = nullptr, double sgn = 0.0) {
    const double inv2pi = 0.15915494309189533576888, twopi = 6.283185307179586476925, wmax = 4611686018427387904.0;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)d * n) return;
    const int i = (int)(t / n), f = (int)(t % n);
    const double l = ls[n_ls == 1 ? 0 : i];
    const double v = shift ? (W[t] / l + sgn * shift[i]) * inv2pi : W[t] * (inv2pi / l);
    const float vf = (float)(v > wmax ? wmax : (v < -wmax ? -wmax : v));
    w64[(size_t)i * npad + f] = v;
    w32[(size_t)i * npad + f] = vf;
    wt32[(size_t)f * dpad + i] = vf;
    if (f == 0) {
        const double gf = twopi / l;
        g64[i] = gf;
        g32[i] = (float)(gf > 3.0e38 ? 3.0e38 : (gf < -3.0e38 ? -3.0e38 : gf));
    }
}

static int basis_raw_w(rr_basis *b) {  // first use: W up once, pad rows / columns of the derived copies zero for good
    if (b->dWraw) return RR_OK;
    rr_ctx *c = b->ctx;
    const int d = b->d, n = b->n;
    const size_t elems = (size_t)b->dpad * b->npad;
    RR_CHECK_HIP(hipMalloc((void **)&b->dWraw, (size_t)d * n * sizeof(double)));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(b->dWraw, b->W.data(), (size_t)d * n * sizeof(double), hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemset(b->dWs32, 0, elems * sizeof(float)));
    RR_CHECK_HIP(hipMemset(b->dWs64, 0, elems * sizeof(double)));
    RR_CHECK_HIP(hipMemset(b->dWt32, 0, elems * sizeof(float)));
    RR_CHECK_HIP(hipMemset(b->dgfac32, 0, (size_t)b->dpad * sizeof(float)));
    RR_CHECK_HIP(hipMemset(b->dgfac64, 0, (size_t)b->dpad * sizeof(double)));
    RR_CHECK_HIP(hipDeviceSynchronize());
    return RR_OK;
}

int rr_basis_raw_w(rr_basis *b) {  // (rr_elbo.hip: the resident SVI loop contracts T with W as given)
    RR_REQUIRE(b != nullptr && !b->large && b->d <= 128, "resident W needs Xdim <= 128");
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    return basis_raw_w(b);
}

int rr_basis_prepare_dev(rr_basis *b, const double *dls, int n_ls, const double *dshift, double sgn) {
    RR_REQUIRE(b != nullptr && dls != nullptr, "lenscale: null argument");
    RR_REQUIRE(n_ls == 1 || n_ls == b->d, "Dimension of input parameter is inconsistent! (n_ls=%d, d=%d)", n_ls, b->d);
    RR_REQUIRE(!b->large && b->d <= 128, "device-resident length scales need Xdim <= 128");
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = basis_raw_w(b);
    if (rc != RR_OK) return rc;
    hipLaunchKernelGGL(rr_scale_w_dev_kernel, dim3((unsigned)(((int64_t)b->d * b->n + 255) / 256)), dim3(256), 0, c->stream,
                       b->dWraw, dls, n_ls, b->d, b->n, b->npad, b->dpad, b->dWs32, b->dWs64, b->dWt32, b->dgfac32, b->dgfac64, dshift, sgn);
    RR_CHECK_HIP(hipGetLastError());
    b->ls_cache.clear();  // the host does not know these values: the next rr_basis_prepare rescales
    return RR_OK;
}

int rr_basis_prepare(rr_basis *b, const double *lenscale, int n_ls) {
    RR_REQUIRE(b != nullptr && lenscale != nullptr, "lenscale: null argument");
    RR_REQUIRE(n_ls == 1 || n_ls == b->d,
               "Dimension of input parameter is inconsistent! (n_ls=%d, d=%d)", n_ls, b->d);
    for (int i = 0; i < n_ls; ++i)
        RR_REQUIRE(std::isfinite(lenscale[i]) && lenscale[i] != 0.0, "lenscale[%d] is %g", i, lenscale[i]);
    bool same = (int)b->ls_cache.size() == n_ls;
    for (int i = 0; same && i < n_ls; ++i) same = (b->ls_cache[i] == lenscale[i]);
    if (same) return RR_OK;

    const double inv2pi = 0.15915494309189533576888;
    const double twopi = 6.283185307179586476925;
    const int d = b->d, n = b->n, npad = b->npad;
    static const bool host_scaling = getenv("RR_PREPARE_ON_HOST") != nullptr;
    if (!b->large && d <= 128 && !host_scaling) {
        rr_ctx *c = b->ctx;
        RR_CHECK_HIP(hipSetDevice(c->device));
        {
            const int rcw = basis_raw_w(b);
            if (rcw != RR_OK) return rcw;
        }
        LsArgs a;
        for (int i = 0; i < 128; ++i) a.ls[i] = i < n_ls ? lenscale[i] : 1.0;
        a.n_ls = n_ls;
        hipLaunchKernelGGL(rr_scale_w_kernel, dim3((unsigned)(((int64_t)d * n + 255) / 256)), dim3(256), 0, c->stream, b->dWraw, a, d,
                           n, npad, b->dpad, b->dWs32, b->dWs64, b->dWt32, b->dgfac32, b->dgfac64);
        RR_CHECK_HIP(hipGetLastError());
        b->ls_cache.assign(lenscale, lenscale + n_ls);
        return RR_OK;
    }
    std::vector<double> w64((size_t)b->dpad * npad, 0.0);
    std::vector<float> w32((size_t)b->dpad * npad, 0.0f);
    std::vector<float> wt32((size_t)npad * b->dpad, 0.0f);
    std::vector<double> g64(b->dpad, 0.0);
    std::vector<float> g32(b->dpad, 0.0f);
    // f32 copies are clamped to a finite range: the optimiser's log-space bounds let a length scale reach
    // 1e-100 (optimize/decorators.py:18), where the reference's float64 phases are finite noise; an f32
    // phase that large has no fractional part left either way, but it must not become inf - inf = NaN.
    const double wmax = 4611686018427387904.0;  // 2^62
    auto clampf = [](double v, double lim) { return (float)(v > lim ? lim : (v < -lim ? -lim : v)); };
    for (int i = 0; i < d; ++i) {
        const double l = lenscale[n_ls == 1 ? 0 : i];
        const double s = inv2pi / l;
        for (int f = 0; f < n; ++f) {
            const double v = b->W[(size_t)i * n + f] * s;
            w64[(size_t)i * npad + f] = v;
            w32[(size_t)i * npad + f] = clampf(v, wmax);
            wt32[(size_t)f * b->dpad + i] = clampf(v, wmax);
        }
        g64[i] = twopi / l;
        g32[i] = clampf(g64[i], 3.0e38);
    }
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    // synchronous copies from short-lived host vectors
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(b->dWs32, w32.data(), w32.size() * sizeof(float), hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->dWs64, w64.data(), w64.size() * sizeof(double), hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->dWt32, wt32.data(), wt32.size() * sizeof(float), hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->dgfac32, g32.data(), g32.size() * sizeof(float), hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->dgfac64, g64.data(), g64.size() * sizeof(double), hipMemcpyHostToDevice));
    b->ls_cache.assign(lenscale, lenscale + n_ls);
    return RR_OK;
}

// ---------------------------------------------------------------------------------------------
// rr_host_sink (rr_internal.h)
// ---------------------------------------------------------------------------------------------
#include <thread>

static void sink_spread(char *dst, size_t dst_ld, const char *src, size_t rows, size_t row_bytes) {
    const size_t total = rows * row_bytes;
    unsigned T = std::thread::hardware_concurrency();
    if (T > 8) T = 8;
    if (T < 1 || total < ((size_t)8 << 20)) T = 1;
    auto work = [=](unsigned t) {
        if (dst_ld == row_bytes) {  // contiguous destination: split the bytes
            const size_t a = total * t / T, b = total * (t + 1) / T;
            memcpy(dst + a, src + a, b - a);
        } else {
            const size_t a = rows * t / T, b = rows * (t + 1) / T;
            for (size_t r = a; r < b; ++r) memcpy(dst + r * dst_ld, src + r * row_bytes, row_bytes);
        }
    };
    if (T == 1) {
        work(0);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

int rr_sink_open(rr_ctx *c, size_t chunk_bytes, rr_host_sink *s) {
    *s = rr_host_sink();
    s->c = c;
    const char *env = getenv("RR_HOST_SINK");
    if (env && atoi(env) == 0) {
        s->direct = true;
        return RR_OK;
    }
    if (c->pin_cap < chunk_bytes) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 2; ++i) {
            if (c->pin[i]) (void)hipHostFree(c->pin[i]);
            c->pin[i] = nullptr;
        }
        c->pin_cap = 0;
        if (hipHostMalloc(&c->pin[0], chunk_bytes, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&c->pin[1], chunk_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            for (int i = 0; i < 2; ++i) {
                if (c->pin[i]) (void)hipHostFree(c->pin[i]);
                c->pin[i] = nullptr;
            }
            s->direct = true;  // no pinned memory to be had: the plain route still works
            return RR_OK;
        }
        c->pin_cap = chunk_bytes;
    }
    for (int i = 0; i < 2; ++i)
        if (!c->pin_ev[i]) RR_CHECK_HIP(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
    return RR_OK;
}

static int sink_flush(rr_host_sink *s) {
    if (!s->pending) return RR_OK;
    const int buf = (s->k - 1) & 1;
    RR_CHECK_HIP(hipEventSynchronize(s->c->pin_ev[buf]));
    sink_spread(s->dst, s->dst_ld, (const char *)s->c->pin[buf], s->rows, s->row_bytes);
    s->pending = false;
    return RR_OK;
}

int rr_sink_push(rr_host_sink *s, const void *dsrc, void *dst, size_t rows, size_t row_bytes, size_t dst_ld_bytes) {
    rr_ctx *c = s->c;
    if (s->direct || rows * row_bytes > c->pin_cap) {
        int rc = sink_flush(s);
        if (rc != RR_OK) return rc;
        RR_CHECK_HIP(hipMemcpy2DAsync(dst, dst_ld_bytes, dsrc, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        return RR_OK;
    }
    const int buf = s->k & 1;  // free: chunk k-2 left it during push(k-1)
    RR_CHECK_HIP(hipMemcpyAsync(c->pin[buf], dsrc, rows * row_bytes, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipEventRecord(c->pin_ev[buf], c->stream));
    ++s->k;
    // while that transfer runs, spread the previous chunk
    if (s->pending) {
        const int pb = (s->k - 2) & 1;
        RR_CHECK_HIP(hipEventSynchronize(c->pin_ev[pb]));
        sink_spread(s->dst, s->dst_ld, (const char *)c->pin[pb], s->rows, s->row_bytes);
    }
    s->pending = true;
    s->dst = (char *)dst;
    s->rows = rows;
    s->row_bytes = row_bytes;
    s->dst_ld = dst_ld_bytes;
    return RR_OK;
}

int rr_sink_close(rr_host_sink *s) { return sink_flush(s); }
