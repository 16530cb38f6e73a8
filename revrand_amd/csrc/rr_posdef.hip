// Posterior of the Bayesian linear model on the device: iC = diag(1/L) + G / var,
// C = iC^-1 and the O(F^2) statistics of StandardLinearModel._elbo (slm.py:150-171), so that an L-BFGS
// evaluation moves O(F) numbers over PCIe instead of two F x F matrices.
//
// The factorisation and the inverse are rocSOLVER's dpotrf / dpotri (plain LAPACK routines; bound at run
// time with dlopen so the library itself links against nothing but the HIP runtime).  Everything around them
// -- assembling iC, the CHOLTHRESH test of mathfun/linalg.py:31,113, log-determinant, m = C b / var,
// sum(G o C), diag(C) -- are kernels here.  If the matrix is not safely positive definite the call reports
// RR_ERR_NOT_POSDEF and the caller takes the reference's SVD route on the host (linalg.py:128-179).
#include <dlfcn.h>

#include <cmath>
#include <mutex>

#include "rr_internal.h"

namespace {

typedef void *rb_handle;
typedef int (*fn_create)(rb_handle *);
typedef int (*fn_destroy)(rb_handle);
typedef int (*fn_set_stream)(rb_handle, hipStream_t);
typedef int (*fn_potr)(rb_handle, int /*rocblas_fill*/, int, double *, int, int *);

struct Solver {
    void *lib_blas = nullptr, *lib_solver = nullptr;
    fn_create create = nullptr;
    fn_destroy destroy = nullptr;
    fn_set_stream set_stream = nullptr;
    fn_potr potrf = nullptr, potri = nullptr;
    bool tried = false, ok = false;
};
Solver g_solver;

const int RB_FILL_LOWER = 122;  // rocblas_fill_lower (rocblas-types.h)

std::once_flag g_solver_once;

void solver_load_once() {
    Solver &s = g_solver;
    s.tried = true;
    const char *blas_names[] = {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"};
    const char *solver_names[] = {"librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so"};
    for (const char *nm : blas_names)
        if ((s.lib_blas = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    for (const char *nm : solver_names)
        if ((s.lib_solver = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!s.lib_blas || !s.lib_solver) return;
    s.create = (fn_create)dlsym(s.lib_blas, "rocblas_create_handle");
    s.destroy = (fn_destroy)dlsym(s.lib_blas, "rocblas_destroy_handle");
    s.set_stream = (fn_set_stream)dlsym(s.lib_blas, "rocblas_set_stream");
    s.potrf = (fn_potr)dlsym(s.lib_solver, "rocsolver_dpotrf");
    s.potri = (fn_potr)dlsym(s.lib_solver, "rocsolver_dpotri");
    s.ok = s.create && s.destroy && s.set_stream && s.potrf && s.potri;
}

bool solver_load() {
    std::call_once(g_solver_once, solver_load_once);
    return g_solver.ok;
}

}  // namespace

// A = G / var + diag(iL)
__global__ void __launch_bounds__(256)
rr_assemble_ic_kernel(const double *__restrict__ G, const double *__restrict__ iL, double ivar, int64_t F,
                      double *__restrict__ A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= F * F) return;
    const int64_t r = i / F, c = i % F;
    A[i] = G[i] * ivar + (r == c ? iL[r] : 0.0);
}

__global__ void __launch_bounds__(256) rr_get_diag_kernel(const double *__restrict__ A, int64_t F, double *__restrict__ d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < F) d[i] = A[i * F + i];
}

// one wave per row r:  m[r] = (C[r,:] . b) / var,  tr += C[r,:] . G[r,:],  dg[r] = C[r][r]
__global__ void __launch_bounds__(256)
rr_posterior_rows_kernel(const double *__restrict__ C, const double *__restrict__ G, const double *__restrict__ b,
                         double ivar, int64_t F, double *__restrict__ m, double *__restrict__ dg, double *__restrict__ tr) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= F) return;
    const double *cr = C + r * F, *gr = G + r * F;
    double am = 0.0, at = 0.0;
    for (int64_t j = lane; j < F; j += 64) {
        const double cv = cr[j];
        am = fma(cv, b[j], am);
        at = fma(cv, gr[j], at);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        am += __shfl_down(am, o, 64);
        at += __shfl_down(at, o, 64);
    }
    if (lane == 0) {
        m[r] = am * ivar;
        dg[r] = cr[r];
        unsafeAtomicAdd(tr, at);
    }
}

struct PosdefScratch {
    rb_handle handle = nullptr;
    double *diL = nullptr, *dvec = nullptr;  // dvec: [chol diag (F) | m (F) | diagC (F) | tr (1)]
    int *dinfo = nullptr;
    int64_t F = 0;
};

void rr_posdef_scratch_free(void *p) {
    if (!p) return;
    PosdefScratch *s = (PosdefScratch *)p;
    if (s->handle && g_solver.ok) g_solver.destroy(s->handle);
    if (s->diL) (void)hipFree(s->diL);
    if (s->dvec) (void)hipFree(s->dvec);
    if (s->dinfo) (void)hipFree(s->dinfo);
    delete s;
}

extern "C" {

int rr_posterior_available(void) { return solver_load() ? 1 : 0; }

int rr_posterior_dev(rr_ctx *c, int64_t F, const double *dG, const double *db, const double *iL, double var, double *dC,
                     double *m, double *diagC, double *scal) {
    RR_REQUIRE(c != nullptr && dG != nullptr && db != nullptr && iL != nullptr && dC != nullptr && m != nullptr &&
                   diagC != nullptr && scal != nullptr,
               "rr_posterior_dev: null argument");
    RR_REQUIRE(F >= 1 && F < 46340 && var > 0.0 && std::isfinite(var), "rr_posterior_dev: bad F or var");
    if (!solver_load()) {
        rr_set_error("rr_posterior_dev: rocSOLVER (librocsolver.so / librocblas.so) could not be loaded: %s", dlerror());
        return RR_ERR_UNSUPPORTED;
    }
    RR_CHECK_HIP(hipSetDevice(c->device));
    if (!c->posdef) c->posdef = new PosdefScratch();
    PosdefScratch &s = *(PosdefScratch *)c->posdef;
    if (!s.handle) {
        if (g_solver.create(&s.handle) != 0 || g_solver.set_stream(s.handle, c->stream) != 0) {
            rr_set_error("rr_posterior_dev: rocblas_create_handle failed");
            s.handle = nullptr;
            return RR_ERR_HIP;
        }
    }
    if (s.F < F) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (s.diL) (void)hipFree(s.diL);
        if (s.dvec) (void)hipFree(s.dvec);
        s.diL = s.dvec = nullptr;
        s.F = 0;
        RR_CHECK_HIP(hipMalloc((void **)&s.diL, (size_t)F * 8));
        RR_CHECK_HIP(hipMalloc((void **)&s.dvec, (size_t)(3 * F + 1) * 8));
        if (!s.dinfo) RR_CHECK_HIP(hipMalloc((void **)&s.dinfo, sizeof(int)));
        s.F = F;
    }
    const double ivar = 1.0 / var;
    RR_CHECK_HIP(hipMemcpyAsync(s.diL, iL, (size_t)F * 8, hipMemcpyHostToDevice, c->stream));
    const unsigned eb = (unsigned)((F * F + 255) / 256), fb = (unsigned)((F + 255) / 256);
    hipLaunchKernelGGL(rr_assemble_ic_kernel, dim3(eb), dim3(256), 0, c->stream, dG, s.diL, ivar, F, dC);
    RR_CHECK_HIP(hipGetLastError());
    // column-major "lower" == the upper triangle of our row-major symmetric matrix: iC = U^T U as in linalg.py:109
    if (g_solver.potrf(s.handle, RB_FILL_LOWER, (int)F, dC, (int)F, s.dinfo) != 0) {
        rr_set_error("rr_posterior_dev: rocsolver_dpotrf failed");
        return RR_ERR_HIP;
    }
    hipLaunchKernelGGL(rr_get_diag_kernel, dim3(fb), dim3(256), 0, c->stream, dC, F, s.dvec);
    int info = 0;
    std::vector<double> h((size_t)3 * F + 1);
    RR_CHECK_HIP(hipMemcpyAsync(&info, s.dinfo, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(h.data(), s.dvec, (size_t)F * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    double logdet = 0.0, mind = INFINITY;
    for (int64_t i = 0; i < F; ++i) {
        const double dval = h[i];
        if (!(dval > 0.0) || !std::isfinite(dval)) {
            mind = -1.0;
            break;
        }
        logdet += 2.0 * std::log(dval);
        if (dval < mind) mind = dval;
    }
    scal[0] = logdet;
    scal[2] = mind;
    if (info != 0 || mind < 1e-5) {  // CHOLTHRESH, mathfun/linalg.py:31
        rr_set_error("rr_posterior_dev: matrix is not safely positive definite (info %d, min diag %g)", info, mind);
        return RR_ERR_NOT_POSDEF;
    }
    if (g_solver.potri(s.handle, RB_FILL_LOWER, (int)F, dC, (int)F, s.dinfo) != 0) {
        rr_set_error("rr_posterior_dev: rocsolver_dpotri failed");
        return RR_ERR_HIP;
    }
    // column-major lower == row-major upper: mirror it into the lower triangle (rr_symmetrize_kernel)
    int rc = rr_symmetrize_dev(c, dC, F);
    if (rc != RR_OK) return rc;
    double *dm = s.dvec + F, *ddg = s.dvec + 2 * F, *dtr = s.dvec + 3 * F;
    RR_CHECK_HIP(hipMemsetAsync(dtr, 0, 8, c->stream));
    hipLaunchKernelGGL(rr_posterior_rows_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, c->stream, dC, dG, db, ivar, F,
                       dm, ddg, dtr);
    RR_CHECK_HIP(hipGetLastError());
    RR_CHECK_HIP(hipMemcpyAsync(h.data() + F, dm, (size_t)(2 * F + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(&info, s.dinfo, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (info != 0) {
        rr_set_error("rr_posterior_dev: rocsolver_dpotri reported info %d", info);
        return RR_ERR_NOT_POSDEF;
    }
    memcpy(m, h.data() + F, (size_t)F * 8);
    memcpy(diagC, h.data() + 2 * F, (size_t)F * 8);
    scal[1] = h[(size_t)3 * F];
    return RR_OK;
}

}  // extern "C"
