// Internal declarations shared by the translation units of librevrand_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/revrand_hip.h"

struct rr_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // second stream (created on first use): chunk k+1's feature kernel under chunk k's SYRK
    hipStream_t stream3 = nullptr;  // third stream (created on first use): the look-ahead Cholesky's bulk updates (rr_posdef.hip)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_fork = nullptr;   // orders stream2 behind the context's stream at the start of an overlapped Gram call
    hipDeviceProp_t prop;
    int num_cu = 0;
    int *tile_map = nullptr;  // XCD-aware tile order of the SYRK kernel for tile_map_nb column blocks
    int tile_map_nb = 0;
    int gram_engine = 0;      // f32 Gram: 0 = f32 MFMA, 3 / 4 = split-bf16 with 3 / 4 products (rr_set_gram_engine)
    int deterministic = 0;    // rr_set_deterministic: ordered partial sums instead of floating-point atomics
    void *det = nullptr;      // scratch of the ordered sums (partials of the kernel in flight), grow-only
    size_t det_bytes = 0;
    double *det2 = nullptr;   // second stage of rr_det_reduce when there are many slots (64 group sums per element), grow-only
    size_t det2_count = 0;
    void *gsa = nullptr, *gsb = nullptr;  // K-blocked operand copies of the GLM step's GEMMs on the split engines, grow-only
    size_t gsa_bytes = 0, gsb_bytes = 0;
    void *pb = nullptr;       // split-bf16 copy of the feature chunk (rr_syrk_bf16x3_kernel), grow-only
    size_t pb_bytes = 0;
    void *posdef = nullptr;  // PosdefScratch (rr_posdef.hip): work matrices + small device vectors of the blocked Cholesky
    void *pin[2] = {nullptr, nullptr};  // pinned host double buffer of rr_host_sink (grow-only)
    size_t pin_cap = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
};

enum rr_kind { RR_KIND_RFF = 0, RR_KIND_FASTFOOD = 1 };
constexpr int RR_MAX_XDIM = 4096;  // random Fourier bases: largest input dimension

struct rr_basis {
    rr_ctx *ctx = nullptr;
    int kind = RR_KIND_RFF;
    int compute = RR_F32;
    bool phase64 = false;         // RR_F32P64: compute == RR_F32 for every product, phases through the f64 MFMA feature kernel
    int d = 0, n = 0;
    int dpad = 0;                 // d rounded up to 8/16/32/64/128 (large: a multiple of 128): rows of Ws, row length kernels read
    int npad = 0;                 // n rounded up to a multiple of 128 (large: 256) (device Ws row length)
    bool large = false;           // d > 128: phases through a GEMM / phase kernel + trig kernel (rr_rff.hip)
    float *lg_xt = nullptr;       // large: X^T scratch (dpad, rows), f32
    void *lg_z = nullptr;         // large: phase scratch (rows, npad), compute dtype
    size_t lg_xt_bytes = 0, lg_z_bytes = 0;
    std::vector<double> W;        // host copy (d, n) row-major, as given
    std::vector<double> ls_cache; // lenscale the device copy was scaled with
    float *dWs32 = nullptr;       // (dpad, npad), zero padded: W[i][f] / (l_i * 2pi)   -> phase in revolutions
    double *dWs64 = nullptr;      // same in f64
    double *dWraw = nullptr;      // (d, n) row-major: W as given, for the device-side rescaling of rr_basis_prepare (d <= 128)
    float *dWt32 = nullptr;       // (npad, dpad): the same weights transposed (feature-major kernels)
    float *dgfac32 = nullptr;     // (d,): 2pi / l_i  (grad kernels)
    float *dmu32 = nullptr;       // (dpad,): mean / 2pi of a spectral-mixture component (rr_gm_*)
    double *dmu64 = nullptr;
    double *dgfac64 = nullptr;
    void *zbuf = nullptr;         // feature scratch of the Gram path (f32 or f64), grow-only
    size_t zbuf_bytes = 0;
    std::vector<hipEvent_t> events;  // 5 per row chunk of the last Gram call: features begin / end, SYRK begin / mid / end
    size_t events_used = 0;
    const char *gram_kernel = "";
    void *pass2 = nullptr;        // scratch of the second _elbo pass (rr_elbo.hip), grow-only
    void *pass2d = nullptr;       // the same for f64 arithmetic
    // FastFood (kind == RR_KIND_FASTFOOD): (k, d2) diagonals / permutation, n = d2 * k
    int ff_d2 = 0, ff_k = 0;
    float *ffB32 = nullptr, *ffG32 = nullptr, *ffSrad32 = nullptr, *ffSrev32 = nullptr, *ffL32 = nullptr;
    double *ffB64 = nullptr, *ffG64 = nullptr, *ffSrad64 = nullptr, *ffSrev64 = nullptr, *ffL64 = nullptr;
    int *ffPI = nullptr;
};

void rr_set_error(const char *fmt, ...);

// ---- deterministic mode (rr_set_deterministic) ------------------------------------------------------------
// Every cross-workgroup sum of the library is "each workgroup (or thread) adds ITS partial to an f64 accumulator with
// unsafeAtomicAdd" -- the order of those additions changes from run to run, and with it the last bits of the result.
// In deterministic mode a kernel instead STORES its partial into slot s of a scratch array (s = a function of its
// block / wave index only) and a second kernel adds the slots of each accumulator element in ascending s -- one thread
// per element, plain read-modify-write, launches ordered by the stream: the same bits every run.
//   rr_acc_out(dst, det_stride, slot, idx, v): the kernels' side (det_stride == 0: the atomic path, dst = accumulator;
//                                              else dst = scratch of nslots * det_stride doubles)
//   rr_det_scratch / rr_det_reduce:            the launchers' side
#ifdef __HIPCC__
__device__ __forceinline__ void rr_acc_out(double *dst, int64_t det_stride, int64_t slot, int64_t idx, double v) {
    if (det_stride) dst[slot * det_stride + idx] = v;
    else unsafeAtomicAdd(&dst[idx], v);
}
#endif
int rr_det_scratch(rr_ctx *c, size_t bytes, void **out);
// out[i] += sum_{s < nslots} part[s * stride + i], i < count, in ascending s (on the context's stream)
int rr_det_reduce(rr_ctx *c, const double *part, int64_t nslots, int64_t stride, int64_t count, double *out);

// sin(2 pi t), cos(2 pi t) in float64 for a phase t in REVOLUTIONS: the fraction f = t - rint(t) is exact, k = rint(4 f)
// picks the quarter turn and the remainder |theta| <= pi / 4 goes through the classic minimax kernels (fdlibm's
// coefficients: < 1 ulp each on that interval); branch-free, ~25 float64 operations instead of the generic sincospi's
// reduction and special cases (the f64 feature kernels are bound by this arithmetic, not by HBM).
#ifdef __HIPCC__
__device__ __forceinline__ void rr_sincos_rev_f64(double t, double &s, double &c) {
    const double f = t - rint(t);            // [-0.5, 0.5]
    const double kq = rint(4.0 * f);         // -2 .. 2
    const double th = (f - 0.25 * kq) * 6.283185307179586476925286766559;  // [-pi/4, pi/4]; f - kq/4 is exact
    const double z = th * th;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sn = fma(th * z, ps, th);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)kq & 3;               // quarter turns: (s, c) -> (c, -s) per +1
    const double s1 = (q & 1) ? cs : sn, c1 = (q & 1) ? -sn : cs;
    s = (q & 2) ? -s1 : s1;
    c = (q & 2) ? -c1 : c1;
}
#endif

// Stores of write-once streams (feature matrices: written by one kernel, read later by another).  RR_NT_STORES=1 marks
// them non-temporal (A/B switch of the Makefile: `make NT=1`); RR_NT_ASM is the matching modifier of the asm stores.
#if defined(RR_NT_STORES) && RR_NT_STORES
#define RR_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define RR_NT_ASM " nt"
#else
#define RR_STREAM_STORE(ptr, val) (*(ptr) = (val))
#define RR_NT_ASM ""
#endif

// ---- debug build (make debug: -DRR_BOUNDS) --------------------------------------------------------
// (1) every device allocation of the library sits between two 4 KiB guard bands filled with a pattern; rr_ctx_sync,
//     rr_free and the library's own frees verify them, so an out-of-bounds WRITE of any kernel (padded-row stores,
//     tile edges, 32-bit offset overflow) fails the next synchronisation with the allocation named;
// (2) RR_DEV_ASSERT traps a kernel whose index arithmetic leaves the extents it was given (out-of-bounds READS of
//     padded rows / tile edges in the feature, SYRK, feature-matrix and FastFood kernels).
// The release build compiles both away.
#ifdef RR_BOUNDS
hipError_t rr_guard_malloc(void **p, size_t bytes);
hipError_t rr_guard_free(void *p);
int rr_guard_check(const char *where);  // RR_OK, or RR_ERR_HIP with the message set
#define hipMalloc(p, n) rr_guard_malloc((void **)(p), (n))
#define hipFree(p) rr_guard_free((void *)(p))
// (3) every kernel launch checks that the calling thread's current device IS the device of the stream it launches on -- the
//     invariant of the in-process device group (one context per member, hipSetDevice at the top of every entry point) that a
//     one-GPU box cannot show broken.  A violation is remembered and fails the next rr_guard_check (rr_ctx_sync) with the
//     launch site named.
void rr_launch_device_check(hipStream_t s, const char *file, int line);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                              \
    do {                                                                                                               \
        rr_launch_device_check((streamId), __FILE__, __LINE__);                                                        \
        hipLaunchKernelGGLInternal((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), __VA_ARGS__);   \
    } while (0)
#define RR_DEV_ASSERT(cond)                                                                         \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            printf("RR_BOUNDS: %s failed at %s:%d (block %u thread %u)\n", #cond, __FILE__, __LINE__, \
                   (unsigned)blockIdx.x, (unsigned)threadIdx.x);                                    \
            __builtin_trap();                                                                       \
        }                                                                                           \
    } while (0)
#else
#define RR_DEV_ASSERT(cond) ((void)0)
#endif

#define RR_CHECK_HIP(expr)                                                            \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            rr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                         __FILE__, __LINE__);                                         \
            return (_e == hipErrorOutOfMemory) ? RR_ERR_OOM : RR_ERR_HIP;             \
        }                                                                             \
    } while (0)

#define RR_REQUIRE(cond, ...)                                                         \
    do {                                                                              \
        if (!(cond)) {                                                                \
            rr_set_error(__VA_ARGS__);                                                \
            return RR_ERR_INVALID;                                                    \
        }                                                                             \
    } while (0)

// Upload W scaled by the given lenscale (cached); implemented in rr_api.hip.
int rr_basis_prepare(rr_basis *b, const double *lenscale, int n_ls);
// length scales in device memory (d <= 128); dshift (d, device) with sgn: the frequencies moved by sgn * shift (a spectral-mixture
// component's phases VX +- X . mean on the random Fourier kernels)
int rr_basis_prepare_dev(rr_basis *b, const double *dls, int n_ls, const double *dshift = nullptr, double sgn = 0.0);
int rr_pick_dmax(int d);
void rr_pass2_scratch_free(void *p);
void rr_pass2d_scratch_free(void *p);
void rr_posdef_scratch_free(void *p);

// Device feature matrix of a concatenated basis (rr_featmat.hip; second pass in rr_elbo.hip).
struct rr_featmat {
    rr_ctx *ctx = nullptr;
    float *P = nullptr;
    int64_t max_rows = 0, ld = 0, rows = 0, rows_pad = 0;
    int F = 0;
    int64_t covered = 0;    // columns written by put_* since rr_featmat_begin (begin zeroes the padding only)
    void *pass2 = nullptr;  // FmPass2 scratch, grow-never (sized by max_rows, ld)
    // P^T next to P (FmPass2::Pt, (ld, max_rows)): pt_rows = the row count whose padding a transposing pass last wrote
    // there (-1: never); random Fourier children then write their blocks of P^T themselves while they write P
    // (pt_covered columns since begin), and a consumer whose pt_covered == F skips its transposing pass
    int64_t pt_rows = -1, pt_covered = 0;
    // the column intervals [c0, c1) put since rr_featmat_begin, kept sorted: a put that overlaps an earlier one is refused
    // (rr_fm_claim), so `covered == F` means every column was written exactly once -- not merely that widths add up
    std::vector<std::pair<int64_t, int64_t>> spans;
};
int rr_fm_claim(rr_featmat *fm, int64_t col0, int64_t width, const char *who);  // rr_featmat.hip
// rr_featmat_put_rff with the length scales in device memory (rr_featmat.hip; the resident SVI loop of rr_elbo.hip)
int rr_fm_put_rff_dev(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *dls, int n_ls,
                      int64_t col0, const double *dshift = nullptr, double sgn = 0.0);
void rr_fm_pass2_free(void *p);
float *rr_fm_pass2_pt(void *p);  // FmPass2::Pt or null
rr_ctx *rr_comm_ctx(rr_comm *comm);  // the context a communicator was bound to (rr_comm.hip)
// Consumers of the feature matrix call this first: every column of [0, F) must have been put since rr_featmat_begin.
#define RR_FM_REQUIRE_FILLED(fm, who)                                                                              \
    RR_REQUIRE((fm)->rows == 0 || (fm)->covered == (fm)->F,                                                        \
               who ": only %lld of the %d columns were written since rr_featmat_begin", (long long)(fm)->covered, (fm)->F)

// Device -> caller's (pageable) host memory for the host-buffer entry points: chunk k is copied into one half of a
// pinned double buffer (full PCIe rate, asynchronous) while chunk k-1 is spread into the caller's array by a few
// host threads -- hipMemcpy to pageable memory does the same two steps serially on one thread (10-17 GB/s).
struct rr_host_sink {
    rr_ctx *c = nullptr;
    bool direct = false;   // no pinned memory (or RR_HOST_SINK=0): plain hipMemcpy2DAsync
    int k = 0;             // chunks pushed
    bool pending = false;  // chunk k-1 is in pin[(k-1)&1], not yet in the caller's array
    char *dst = nullptr;
    size_t rows = 0, row_bytes = 0, dst_ld = 0;
};
int rr_sink_open(rr_ctx *c, size_t chunk_bytes, rr_host_sink *s);
int rr_sink_push(rr_host_sink *s, const void *dsrc, void *dst, size_t rows, size_t row_bytes, size_t dst_ld_bytes);
int rr_sink_close(rr_host_sink *s);
