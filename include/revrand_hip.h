/*
 * revrand_hip.h -- C ABI of librevrand_hip.so: the MI355X (gfx950) hot path of
 * NICTA/revrand's random-feature basis expansion + Gram assembly.
 *
 * The reference has no FFI layer (it is pure NumPy); the boundary its estimators
 * program against is the duck-typed Basis protocol.  Each entry point below
 * replaces the arithmetic of one reference method; the Python classes in
 * revrand_amd/ keep the reference's names/arguments/errors and forward here via
 * ctypes (INTEGRATION.md shows the binding).  Citations are file:line in the
 * reference tree (revrand v1.0.0).
 *
 * Conventions
 *  - C linkage, no exceptions cross the ABI.  Every call returns rr_status
 *    (0 = OK, negative = error); rr_last_error() gives the message of the last
 *    failing call made by the calling thread.
 *  - Host buffers are caller-owned, row-major, with 64-bit row counts and explicit
 *    leading dimensions (in ELEMENTS).  The library owns device memory behind the
 *    opaque handles.  A handle is used by one host thread at a time.
 *  - Calls taking host buffers are synchronous: buffers are valid on return.
 *    *_dev calls take DEVICE pointers (from rr_malloc, or any hipMalloc'd memory
 *    of the same device, e.g. a torch tensor's data_ptr) and are asynchronous on
 *    the context's stream unless stated; rr_ctx_sync() waits for them.
 *  - Two scaling models over the row shards, the same exchange message in both: one process per GPU (one rr_ctx each),
 *    the ranks' partial statistics summed with rr_comm_init_rank + rr_comm_* (RCCL, bound directly); or ONE process
 *    holding a context per GPU and summing them with rr_comm_init_all + rr_comm_group_* (the estimator's own call).
 *  - There is no CPU fallback: without a usable gfx950 device rr_ctx_create fails
 *    with RR_ERR_NO_DEVICE.
 */
#ifndef REVRAND_HIP_H
#define REVRAND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 1

typedef struct rr_ctx rr_ctx;     /* device context: device id, stream, scratch  */
typedef struct rr_basis rr_basis; /* device-resident constants of one basis      */

typedef enum rr_status {
    RR_OK = 0,
    RR_ERR_INVALID = -1,     /* bad argument (shape, dtype, null pointer)          */
    RR_ERR_NO_DEVICE = -2,   /* no HIP device / not gfx950                         */
    RR_ERR_HIP = -3,         /* a HIP runtime call failed (message has the detail) */
    RR_ERR_OOM = -4,         /* device allocation failed                           */
    RR_ERR_UNSUPPORTED = -5, /* valid request this build cannot serve              */
    RR_ERR_NOT_POSDEF = -6   /* rr_posterior_dev: take the host SVD route          */
} rr_status;

typedef enum rr_dtype { RR_F32 = 0, RR_F64 = 1 } rr_dtype;
/* A third ARITHMETIC mode of a random Fourier basis (rr_rff_create's `compute` only; never a buffer dtype):
 * float32 features, Gram, products and outputs exactly as RR_F32, but the phase x . w / l is accumulated in float64 (on
 * the f64 matrix cores) and reduced modulo one revolution in float64 before the float32 sin / cos.  For heavy-tailed
 * frequencies -- RandomLaplace draws W from a Cauchy distribution (basis_functions.py:993-995): |x . w| reaches 1e5
 * revolutions, of which a float32 accumulator keeps ~1e-2 rad -- this holds the 1e-3 tolerance of the f32 path where
 * RR_F32 is 6e-2 off, and everything behind the feature kernel (SYRK, second pass, feature matrix, GLM step) is the
 * f32 pipeline.  Pass X in float64 (device X resident in float64): rounding x to float32 alone would move such a phase
 * by |x . w| 2^-24 revolutions.  transform / grad of such a basis run the RR_F64 kernels.  Xdim <= 128. */
#define RR_F32P64 2

/* ---- library / context ------------------------------------------------- */

int rr_abi_version(void);
/* Bit mask describing the build: RR_BUILD_BOUNDS = `make debug` (-DRR_BOUNDS: guard bands around every device
 * allocation, verified at rr_ctx_sync / rr_free, and index assertions inside the kernels). */
#define RR_BUILD_BOUNDS 1
int rr_build_flags(void);
/* RR_BUILD_BOUNDS builds: the number of kernel launches checked so far for "the calling thread's current device is the
 * device of the stream launched on" (the invariant of the in-process device group; a violation fails the next rr_ctx_sync
 * with the launch site named).  0 in a release build, which makes no such check. */
int64_t rr_debug_launch_checks(void);
const char *rr_last_error(void);
int rr_device_count(int *count);

/* Bind a context to HIP device `device` (creates a stream, queries the arch). */
int rr_ctx_create(int device, rr_ctx **out);
void rr_ctx_destroy(rr_ctx *ctx);
int rr_ctx_sync(rr_ctx *ctx);
/* Device facts: name (<= 63 chars), number of CUs, total HBM bytes. */
int rr_ctx_info(rr_ctx *ctx, char name[64], int *compute_units, uint64_t *hbm_bytes);
/* The context's hipStream_t as an opaque pointer (for event timing by callers). */
void *rr_ctx_stream(rr_ctx *ctx);
/* Where the context's GPU sits: its PCI bus id ("0000:c1:00.0"; /sys/bus/pci/devices/<id>/numa_node names the host
 * memory node next to it, which is where a rank's threads and staging buffers belong) and its HIP device index;
 * rr_peer_access: can kernels on `device` load `peer`'s memory directly (an xGMI / PCIe peer link)? */
int rr_ctx_pci_bus_id(rr_ctx *ctx, char id[32], int *device);
int rr_peer_access(int device, int peer, int *can_access);

/* ---- device memory (keeps X, y resident across optimiser iterations) ---- */

int rr_malloc(rr_ctx *ctx, size_t bytes, void **dptr);
int rr_free(rr_ctx *ctx, void *dptr);
int rr_memset(rr_ctx *ctx, void *dptr, int value, size_t bytes);
int rr_memcpy_h2d(rr_ctx *ctx, void *dst, const void *src, size_t bytes);
int rr_memcpy_d2h(rr_ctx *ctx, void *dst, const void *src, size_t bytes);

/* Timing helper: run between two events on the context's stream.
 * rr_timer_start records an event; rr_timer_stop records a second one, waits for
 * it and returns the elapsed milliseconds of everything enqueued in between. */
int rr_timer_start(rr_ctx *ctx);
int rr_timer_stop(rr_ctx *ctx, float *ms);

/* ---- random Fourier feature bases --------------------------------------
 * One object serves RandomRBF / RandomLaplace / RandomCauchy / RandomMatern32 /
 * RandomMatern52 / OrthogonalRBF: they differ only in how the host samples W
 * (basis_functions.py:952-954, 993-995, 1034-1045, 1051-1065, 1198-1208), which
 * stays host-side NumPy.
 *
 * W: host, row-major (d, n) float64 -- the `self.W` of _RandomKernelBasis
 * (basis_functions.py:830-834).  compute: RR_F32 or RR_F64 arithmetic, or RR_F32P64 (f32 pipeline, f64 phases). */
int rr_rff_create(rr_ctx *ctx, int compute, int d, int n, const double *W, rr_basis **out);
void rr_basis_destroy(rr_basis *basis);

/* Row length the kernels read from a DEVICE X: Xdim rounded up to 8/16/32/64/128, above
 * that to a multiple of 128 (Xdim <= 4096).  Every
 * *_dev entry point requires ldx >= this and elements [d, padded) of each row to be zero
 * (they meet zero weights; NaN/Inf there would poison the row).  rr_upload_matrix with
 * ld_dev = rr_rff_padded_dim() produces exactly this layout; the host-buffer entry points
 * do it internally. */
int rr_rff_padded_dim(rr_basis *basis);

/* Allocate a device (N, ld_dev) matrix of `dtype`, zero the pad columns and copy the host
 * (N, d) matrix with leading dimension ldx into it.  Free with rr_free. */
int rr_upload_matrix(rr_ctx *ctx, const void *X, int dtype, int64_t N, int64_t d, int64_t ldx,
                     int64_t ld_dev, void **dptr);

/* Copy host rows (N, d; leading dimension ldx) into rows [row0, row0 + N) of an existing
 * device matrix with leading dimension ld_dev (pad columns are left untouched: allocate
 * with rr_malloc + rr_memset(0) first).  Lets callers stream a large X up in chunks. */
int rr_upload_rows(rr_ctx *ctx, void *dptr, int64_t ld_dev, int64_t row0, const void *X, int dtype,
                   int64_t N, int64_t d, int64_t ldx);

/* lenscale: host float64, n_ls == 1 (isotropic) or n_ls == d (ARD), as validated
 * by _LengthScaleBasis._check_dim (basis_functions.py:590-613). */

/* Phi = [cos(X W/l), sin(X W/l)] / sqrt(n)            (N, 2n)
 * replaces _RandomKernelBasis.transform  basis_functions.py:838-864.
 * X host (N, d) of x_dtype, leading dimension ldx; Phi host (N, 2n) of out_dtype,
 * leading dimension ldphi. */
int rr_rff_transform(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx,
                     const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi);

/* dPhi/dl  replaces _RandomKernelBasis.grad  basis_functions.py:866-901.
 * n_ls == 1: out is (N, 2n) and -- as in the reference -- holds ONLY input
 * dimension 0's contribution (the loop at :896 runs once with i = 0).
 * n_ls == d: out is (N, 2n, d), C-order (the np.dstack of :901). */
int rr_rff_grad(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx,
                const double *lenscale, int n_ls, void *dPhi, int out_dtype);

/* Device-resident forms (X on the device, x_dtype, row-major, ldx). */
int rr_rff_transform_dev(rr_basis *basis, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                         const double *lenscale, int n_ls, void *dPhi_out, int out_dtype,
                         int64_t ldphi);

/* The metric's unit of work: for every row, project, cos/sin, scale and
 * accumulate   G += phi phi^T (F x F, F = 2n),  b += phi y,  yty += y^2.
 * Two kernels per row chunk: the feature kernel writes the chunk's Phi into an f32 SCRATCH the basis owns
 * (4 * Fp bytes per row, Fp = F rounded up to 256; never returned, never crossing PCIe) and the MFMA SYRK
 * kernels reduce it (DESIGN.md 3.1 has the measurement behind that split).  Replaces, for a random Fourier basis,
 *   Phi = basis.transform(X, l)      slm.py:145 (basis_functions.py:859-864)
 *   PhiPhi = Phi.T.dot(Phi)          slm.py:146
 *   Phi.T.dot(y)                     slm.py:157
 * dX (N, d) x_dtype and dy (N,) x_dtype are DEVICE pointers.  dG (F*F), db (F),
 * dyty (1) are DEVICE float64 buffers that are ACCUMULATED INTO (zero them with
 * rr_memset for a fresh sum) so that row shards / chunks can be summed; only the
 * upper triangle (row <= col) of G is written.  dy/db/dyty may be NULL together.
 * Asynchronous on the context stream. */
int rr_rff_gram_dev(rr_basis *basis, const void *dX, const void *dy, int x_dtype, int64_t N,
                    int64_t ldx, const double *lenscale, int n_ls, double *dG, double *db,
                    double *dyty);

/* HIP-event times of the kernels the LAST rr_rff_gram_dev call launched on this basis, summed
 * over its row chunks: the feature (projection + cos/sin) kernel, the off-diagonal-tile SYRK
 * kernel (the dominant one) and the diagonal-tile SYRK kernel (0 in f64 mode, where one kernel
 * does all tiles and is reported as syrk_ms).  Waits for the stream.  launches = number of row
 * chunks (= launches of each kernel). */
int rr_rff_gram_timings(rr_basis *basis, float *features_ms, float *syrk_ms, float *diag_ms, int *launches);

/* Mirror the upper triangle of a device (F, F) float64 matrix into the lower. */
int rr_symmetrize_dev(rr_ctx *ctx, double *dG, int64_t F);

/* Host-buffer convenience: uploads X, y in row chunks, runs rr_rff_gram_dev,
 * symmetrises and downloads.  G host (F, F) float64 full symmetric; b (F,); yty (1). */
int rr_rff_gram(rr_basis *basis, const void *X, const void *y, int x_dtype, int64_t N,
                int64_t ldx, const double *lenscale, int n_ls, double *G, double *b,
                double *yty);

/* G = Phi^T Phi (full symmetric, F x F), b = Phi^T y, yty = y^T y for an ARBITRARY host feature
 * matrix Phi (N, F) of `dtype` with leading dimension ldphi -- the Gram of concatenated or
 * non-random bases (BasisCat.transform output, LinearBasis, ...), i.e. `Phi.T.dot(Phi)` and
 * `Phi.T.dot(y)` of slm.py:146,157 for any basis.  The arithmetic follows `dtype`: float32 rows go through the f32
 * MFMA SYRK kernels of the fused path, float64 rows (the reference's dtype) through the f64 MFMA SYRK kernel.
 * y/b/yty may be NULL together. */
int rr_dense_gram(rr_ctx *ctx, const void *Phi, int dtype, int64_t N, int64_t F, int64_t ldphi,
                  const void *y, double *G, double *b, double *yty);

/* predict_moments (slm.py:240-244) for an ARBITRARY host feature matrix Phi (N, F) of `dtype` -- bases without a fused
 * device route (LinearBasis alone, concatenations with float64 children, ...):  Ey = Phi m,  Vf = rowsum((Phi C) o Phi)
 * with m (F) and C (F, F) host float64; float64 arithmetic (f64 MFMA GEMM) whatever `dtype`.  Host outputs of length N;
 * the caller adds var to Vf. */
int rr_dense_predict(rr_ctx *ctx, const void *Phi, int dtype, int64_t N, int64_t F, int64_t ldphi, const double *m,
                     const double *C, double *Ey, double *Vf);

/* Minibatch gather from resident data: ddst[r][0:ld_words] = dsrc[didx[r]][0:ld_words] for rows of ld_words 4-byte
 * words (float32 rows; float64 rows with ld_words = 2 * columns); didx is a DEVICE int32 vector.  Asynchronous. */
int rr_gather_rows(rr_ctx *ctx, const void *dsrc, const int *didx, int64_t rows, int64_t ld_words, void *ddst);

/* ---- concatenated bases: one device feature matrix, one Gram ---------------------------------
 * BasisCat.transform hstacks the children's Phi (basis_functions.py:1599-1627) before slm.py:146,157
 * contract it.  Here the children write their column blocks into ONE zero-padded f32 device matrix
 * (rows <= max_rows, F columns) and the MFMA SYRK kernel reduces it; nothing crosses PCIe.
 *   rr_featmat_begin(rows)      start a row block (zeroes the matrix)
 *   rr_featmat_put_rff          [cos | sin] / sqrt(n) of a random Fourier basis at columns [col0, col0 + 2n)
 *   rr_featmat_put_linear       LinearBasis.transform ([1, X] or X, basis_functions.py:468-485) at col0
 *   rr_featmat_put_fastfood     FastFoodRBF.transform (the chain kernel, basis_functions.py:1263-1289) at col0
 *   rr_featmat_put_host         any other basis: a host (rows, ncols) block at col0
 *   rr_featmat_gram             G(upper) += P^T P, b += P^T y, yty += y^T y into DEVICE f64 buffers
 * dX / dy are device pointers (dX in the padded layout for put_rff); all calls asynchronous on the
 * context stream except put_host. */
typedef struct rr_featmat rr_featmat;
int rr_featmat_create(rr_ctx *ctx, int64_t max_rows, int64_t F, rr_featmat **out);
void rr_featmat_destroy(rr_featmat *fm);
int rr_featmat_begin(rr_featmat *fm, int64_t rows);
int rr_featmat_put_rff(rr_featmat *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx,
                       const double *lenscale, int n_ls, int64_t col0);
int rr_featmat_put_linear(rr_featmat *fm, const void *dX, int x_dtype, int64_t ldx, int d, int onescol,
                          int64_t col0);
/* FastFoodRBF.transform (basis_functions.py:1263-1289) of the rows of the current rr_featmat_begin at columns
 * [col0, col0 + 2 d2 k): the Hadamard / permute / diagonal chain writes Phi straight into the feature matrix (f32). */
int rr_featmat_put_fastfood(rr_featmat *fm, rr_basis *fastfood, const void *dX, int x_dtype, int64_t ldx,
                            const double *lenscale, int n_ls, int64_t col0);
int rr_featmat_put_host(rr_featmat *fm, const void *Phi, int dtype, int64_t ncols, int64_t ldphi, int64_t col0);
int rr_featmat_gram(rr_featmat *fm, const void *dy, int y_dtype, double *dG, double *db, double *dyty);

/* ---- the same in FLOAT64: the feature matrix of a concatenation with a dtype="f64" child -------------------
 * BasisCat.transform's hstack (basis_functions.py:1599-1627) in float64 in HBM, reduced by the f64 MFMA SYRK
 * (slm.py:146,157) and consumed by the second pass of _elbo (slm.py:160-197) and predict_moments (slm.py:240-244) in
 * float64 -- the reference's arithmetic end to end (north star: 1e-5 relative in fp64) for a resident concatenated fit.
 * Same protocol as rr_featmat_*: begin(rows); every child puts its column block (random Fourier children are evaluated in
 * float64 whatever their own arithmetic); gram accumulates into DEVICE f64 buffers; pass2_begin(m, C) -> pass2_rows(dy) ->
 * pass2_rff per random Fourier child -> pass2_end(&sqErr), or pass2_begin -> predict_rows(Ey, Vf).  C: host (F, F) or,
 * with c_on_device != 0, a device pointer (rr_posterior_dev's output).  No GLM step, no split engines. */
typedef struct rr_featmat64 rr_featmat64;
int rr_featmat64_create(rr_ctx *ctx, int64_t max_rows, int64_t F, rr_featmat64 **out);
void rr_featmat64_destroy(rr_featmat64 *fm);
int rr_featmat64_begin(rr_featmat64 *fm, int64_t rows);
int rr_featmat64_put_rff(rr_featmat64 *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx,
                         const double *lenscale, int n_ls, int64_t col0);
int rr_featmat64_put_linear(rr_featmat64 *fm, const void *dX, int x_dtype, int64_t ldx, int d, int onescol, int64_t col0);
int rr_featmat64_put_host(rr_featmat64 *fm, const void *Phi, int dtype, int64_t ncols, int64_t ldphi, int64_t col0);
int rr_featmat64_gram(rr_featmat64 *fm, const void *dy, int y_dtype, double *dG, double *db, double *dyty);
int rr_featmat64_pass2_begin(rr_featmat64 *fm, const double *m, const double *C, int c_on_device);
int rr_featmat64_pass2_rows(rr_featmat64 *fm, const void *dy, int y_dtype);
int rr_featmat64_pass2_rff(rr_featmat64 *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx, int64_t col0,
                           double *dT);
int rr_featmat64_pass2_end(rr_featmat64 *fm, double *sqErr);
int rr_featmat64_predict_rows(rr_featmat64 *fm, double *Ey, double *Vf);

/* Second data pass of _elbo / predict_moments for a concatenated basis (slm.py:160-162,193-197,240-244),
 * over the rows currently in the matrix (after rr_featmat_begin + put_*):
 *   pass2_begin(m, C)      posterior (host float64, (F) and (F, F) row-major) to the device; sqErr = 0
 *   pass2_rows(dy)         dot = P m, Err = y - dot, sqErr += |Err|^2, U = P C          (dy may be NULL)
 *   pass2_rff(b, dX, col0, dT)   dT (d, n) float64 DEVICE buffer  +=  X^T A  for the random Fourier child
 *                          at columns [col0, col0 + 2n), A as in rr_rff_elbo_pass2_dev
 *   pass2_end(&sqErr)      wait, fetch the accumulated sqErr
 *   predict_rows(Ey, Vf)   Ey = P m, Vf = rowsum((P C) o P) for the current rows (host float64) */
int rr_featmat_pass2_begin(rr_featmat *fm, const double *m, const double *C);
int rr_featmat_pass2_begin_devc(rr_featmat *fm, const double *m, const double *dC); /* C on the DEVICE */
/* The same for rr_featmat_predict_rows only: C is kept in its upper-triangular form with doubled off-diagonal entries
 * (phi^T C phi is all the variance needs) and the Phi C GEMM stops at the diagonal -- half the product.  c_on_device:
 * C is a device pointer.  rr_featmat_pass2_rows / _rff must not follow this call. */
int rr_featmat_predict_begin(rr_featmat *fm, const double *m, const double *C, int c_on_device);
int rr_featmat_pass2_rows(rr_featmat *fm, const void *dy, int y_dtype);
int rr_featmat_pass2_rff(rr_featmat *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx, int64_t col0,
                         double *dT);
/* The gradient contraction without U = Phi C in memory (slm.py:193-195 for a concatenation): announce every
 * rr_featmat_pass2_rff call that will follow (same arguments), then call rr_featmat_pass2_rows_planned instead of
 * rr_featmat_pass2_rows -- the caller's promise that the planned children are the ONLY consumers of U.  When every plan sits
 * in whole 256-column tiles (col0 % 256 == 0, n % 256 == 0, d <= 128, float32 X, f32 engine, not deterministic mode) each
 * child's columns of U are contracted with Phi, Err m^T and X block by block in registers (rr_gemm_gradt_f32_kernel) and
 * added to its dT; columns nobody consumes (a linear child's) are not computed; the rr_featmat_pass2_rff calls that follow
 * return at once.  Otherwise this is rr_featmat_pass2_rows.  Plans hold for ONE rows call.  RR_PASS2_NO_FUSE=1: never. */
int rr_featmat_pass2_plan_rff(rr_featmat *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx, int64_t col0,
                              double *dT);
int rr_featmat_pass2_rows_planned(rr_featmat *fm, const void *dy, int y_dtype);
int rr_featmat_pass2_end(rr_featmat *fm, double *sqErr);
int rr_featmat_predict_rows(rr_featmat *fm, double *Ey, double *Vf);

/* One SVI minibatch step of the generalised linear model (glm.py:205-322) over the rows currently in the matrix.
 * Likelihood ids (likelihoods.py): */
#define RR_LIK_BERNOULLI 0        /* logistic link                      :18-150  */
#define RR_LIK_BINOMIAL 1         /* per-row n in drowarg               :153-258 */
#define RR_LIK_GAUSSIAN 2         /* lik_param = variance               :261-423 */
#define RR_LIK_POISSON_EXP 3      /* exp link                           :426-545 */
#define RR_LIK_POISSON_SOFTPLUS 4 /* softplus link                               */
/* WS: host (K*L, F) float64 weight samples, component-major (rows [k*L, (k+1)*L) belong to component k).
 * dy, drowarg: DEVICE vectors of fm rows (dtype RR_F32 / RR_F64), drowarg NULL unless binomial.
 * Outputs (host float64): Edws (K*L, F) = dfs Phi (glm.py:308); per component k: llsum[k] = sum over its L samples
 * and the rows of loglike WITHOUT its f-independent constant (-gammaln(y+1), log C(n,y), -log(2 pi var)/2), and
 * aux[k] = sum (y - f)^2 (Gaussian; 0 otherwise).  EdPhi = dfs^T ws / (K L) (glm.py:311,229) stays on the device
 * for rr_featmat_glm_rff / rr_featmat_glm_edphi. */
int rr_featmat_glm_step(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param,
                        const double *WS, int K, int L, double *Edws, double *llsum, double *aux);
/* The same step with the reparameterisation draws made ON THE DEVICE (opt-in fast route; rr_featmat_glm_step with the
 * caller's own draws is the parity route): e ~ N(0, 1) per (sample, feature) from a counter-based generator keyed by
 * (seed, step), ws = m_k + sqrt(C_k) e.  m, C: host (F, K) float64 (the mixture means and diagonal covariances).
 * Outputs (host float64, (K, F) row-major): Edm = sum_l Edws / L (glm.py:309), EdC = sum_l Edws e / (L sqrt C) (:310);
 * llsum, aux as above.  EdPhi stays on the device as above.
 * Edm == EdC == NULL (this entry point and rr_featmat_glm_step_draws): objective-only evaluation -- fs, the likelihood
 * kernel and its sums (llsum, aux), none of the gradient GEMMs; what the random starts of the SGD front-end rank by. */
int rr_featmat_glm_step_sampled(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik,
                                double lik_param, const double *m, const double *C, int K, int L, uint64_t seed,
                                uint64_t step, double *Edm, double *EdC, double *llsum, double *aux);
/* The parity route without the 2 x (K L, F) float64 traffic of rr_featmat_glm_step: the CALLER's standard-normal draws
 * E (host float32 (K*L, F), component-major, e.g. the reference's random stream) go up once, ws = m_k + sqrt(C_k) e is
 * formed on the device and only Edm, EdC (K, F) come back, as in rr_featmat_glm_step_sampled. */
int rr_featmat_glm_step_draws(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik,
                              double lik_param, const double *m, const double *C, int K, int L, const float *E,
                              double *Edm, double *EdC, double *llsum, double *aux);
/* The same with the draws already in DEVICE memory (float32 (K*L, F), contiguous): the caller uploaded them while the
 * previous step ran (glm.py: the minibatch worker thread, on a context of its own), so the step starts without the
 * 4 MB host-to-device copy of config 5. */
int rr_featmat_glm_step_draws_dev(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik,
                                  double lik_param, const double *m, const double *C, int K, int L, const float *dE,
                                  double *Edm, double *EdC, double *llsum, double *aux);
/* dT (d, n) float64 DEVICE buffer += X^T (E_s o P_c - E_c o P_s) for the random Fourier child at columns
 * [col0, col0 + 2n):  sum(EdPhi o dPhi_i) = -(1/l_i^2) W[i,:].T[i,:]  (glm.py:274-275 without basis.grad). */
int rr_featmat_glm_rff(rr_featmat *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx, int64_t col0,
                       double *dT);
/* Announce, BEFORE a step, the rr_featmat_glm_rff call that will follow it (same arguments; dT zeroed by the caller).
 * When the whole matrix is this child's [cos | sin] block (col0 == 0, 2n == F, n % 256 == 0, d <= 128, float32 X, f32
 * engine, not deterministic mode) the step contracts every 256x256 block of EdPhi = dfs^T ws / (K L) (glm.py:311) with
 * P and X while it is still in registers and adds the result to dT (rr_gemm_gradt_f32_kernel): EdPhi is neither written
 * nor read back, and the rr_featmat_glm_rff call that follows returns at once.  Otherwise the plan is dropped and
 * nothing changes.  A plan holds for ONE step.  RR_GLM_NO_FUSE=1 drops every plan (A/B runs). */
int rr_featmat_glm_plan_rff(rr_featmat *fm, rr_basis *basis, const void *dX, int x_dtype, int64_t ldx, int64_t col0,
                            double *dT);
/* EdPhi[:, col0:col0+ncols] to the host (rows, ncols) float64, for bases whose gradient is formed on the host. */
int rr_featmat_glm_edphi(rr_featmat *fm, int64_t col0, int64_t ncols, double *E);
/* out (rows, S) = P W for a host (F, S) float64 matrix: the latent function samples of glm.py:572-620. */
int rr_featmat_project(rr_featmat *fm, const double *W, int S, double *out);

/* ---- the SVI loop of GeneralizedLinearModel.fit with its parameters resident (round 5) ----------------------------
 * Replaces, for a model whose basis is a random Fourier basis, a linear basis or a concatenation of such children (Xdim <= 128
 * each) and whose minibatches are gathered on the device, the host loop  sgd (optimize/sgd.py:337-425)  o  logtrick_sgd
 * (optimize/decorators.py:329-408)  o  structured_sgd (:133-252)  around  GeneralizedLinearModel._elbo (glm.py:205-294):  the flat
 * parameter vector
 *     z = [ m (F, K) | C (F, K) | regularisers (one per child) | likelihood parameter (n_lik = 1: the Gaussian variance) |
 *           basis parameters, child by child: length scales (a spectral-mixture component: its means, then its length scales) ]
 * (row-major blocks in the order of glm.py:170-176 / BasisCat's parameter lists, basis_functions.py:1750-1763; coordinates with
 * is_log != 0 are log(x), the log trick's Positive coordinates), the updater's state, the gradient and every intermediate stay
 * in HBM.  One rr_glm_sgd_step call queues a whole step -- x = from_log(z), the bases rescaled by the length scales in x, Phi of
 * the minibatch rows, the step of rr_featmat_glm_step_draws_dev / _sampled with (m, C) read from x, the length-scale
 * contractions, the mixture terms (glm.py:238-262), the gradient (:255-283) with every child's own regulariser over its column
 * slice (basis_functions.py:1712-1748), the chain rule of the log trick, bound truncation + updater + clip (sgd.py:404-420), the
 * step's -ELBO (glm.py:285-292) and gradient norm -- and returns without waiting for it; at most two steps are in flight (the
 * call waits for step t - 2), so a caller cycling through >= 3 minibatch buffers may refill the oldest.  Float64 arithmetic of
 * the NumPy expressions, the feature / GEMM kernels in float32 as in rr_featmat_glm_step. */
typedef struct rr_glm_sgd rr_glm_sgd;
#define RR_UPD_SGD 0      /* upd_par = (eta)                          sgd.py:14-68   */
#define RR_UPD_ADADELTA 1 /* (rho, epsilon)                                  :71-133  */
#define RR_UPD_ADAGRAD 2  /* (eta, epsilon)                                  :136-196 */
#define RR_UPD_MOMENTUM 3 /* (rho, eta)                                      :199-256 */
#define RR_UPD_ADAM 4     /* (alpha, beta1, beta2, epsilon)                  :259-330 */
#define RR_SGD_CHILD_RFF 0    /* [cos | sin] of a random Fourier basis: 2 n columns, n_ls = 1 (isotropic) or Xdim length scales */
#define RR_SGD_CHILD_LINEAR 1 /* LinearBasis (basis_functions.py:468-485): d columns of X (+ a leading column of ones)           */
#define RR_SGD_CHILD_GM 2     /* FastFoodGM (basis_functions.py:1386-1562), one spectral-mixture component through the dense equivalent \
                                 of its chain: basis = that random Fourier basis (n frequencies), 4 n columns                        \
                                 [cos | sin](VX + mX) | [cos | sin](VX - mX) (the reference's column order is cos+, sin+, cos-, sin-:  \
                                 the same blocks), n_ls = 2 Xdim coordinates [mean | length scales]                                   */
typedef struct rr_glm_sgd_child {
    int kind;        /* RR_SGD_CHILD_* */
    rr_basis *basis; /* RFF, GM: the basis (same context as the feature matrix); else NULL */
    int d;           /* LINEAR: columns of X */
    int onescol;     /* LINEAR: 1 = a column of ones first */
    int n_ls;        /* RFF: 1 or Xdim; LINEAR: 0; GM: 2 Xdim ([mean | length scales]) */
} rr_glm_sgd_child;
/* fm: an (empty) feature matrix whose F columns the children fill in order; z0, lower, upper (float64) and is_log (bytes):
 * host vectors of 2 F K + n_children + n_lik + (all length scales) entries; upd_par: 4 doubles (unused ones ignored);
 * maxiter: steps at most. */
int rr_glm_sgd_create(rr_featmat *fm, int n_children, const rr_glm_sgd_child *children, int K, int n_lik, const double *z0,
                      const double *lower, const double *upper, const unsigned char *is_log, int updater, const double *upd_par,
                      int64_t maxiter, rr_glm_sgd **out);
/* One step on the minibatch: dX[c], x_dtype[c], ldx[c] -- every child's rows of ITS columns of X (device; random Fourier: the
 * rr_rff_padded layout), targets dy / per-row argument drowarg (device, dtype).  llconst: the f-independent constant of
 * sum(loglike) per latent sample (ignored for the Gaussian, whose constant follows the variance in z); bmag = N / minibatch
 * size (glm.py:158).  dE: the caller's standard normals (device float32 (K L, F), the reference's stream) or NULL:
 * counter-based device draws keyed by (seed, key).
 * Stream order: dX, dy, drowarg and dE are read by kernels on the context's stream AND (large steps) on a second stream of the
 * loop's own; the call orders both behind everything queued on the context's stream before it, so a caller that fills these
 * buffers with asynchronous work on that stream (row gathers, rr_memcpy_h2d) need not synchronise.  Work on OTHER streams
 * (an upload context's) must be complete -- or waited for on the context's stream -- before the call.  The context serves this
 * call's thread alone while it runs. */
int rr_glm_sgd_step(rr_glm_sgd *s, const void *const *dX, const int *x_dtype, const int64_t *ldx, int64_t rows, const void *dy,
                    const void *drowarg, int dtype, int lik, double llconst, double bmag, int L, const float *dE, uint64_t seed,
                    uint64_t key);
/* Waits for the queued steps; z (all coordinates, log space where is_log), objs / norms: -ELBO and |gradient| of every
 * step done so far (*steps of them); any pointer may be NULL. */
int rr_glm_sgd_read(rr_glm_sgd *s, double *z, double *objs, double *norms, int64_t *steps);
/* -ELBO of one finished step (waits for the queue): the `Iter n: ELBO = ...` log line of glm.py:287-290. */
int rr_glm_sgd_objective(rr_glm_sgd *s, int64_t step, double *obj);
void rr_glm_sgd_destroy(rr_glm_sgd *s);

/* ---- the SAME loop for small minibatches: many SGD steps per launch (rr_svi.hip) ----------------------------------------
 * The reference's own defaults -- GeneralizedLinearModel(K=10, maxiter=3000, batch_size=10, nsamples=50, nstarts=500),
 * glm.py:120-124 -- make a step ~1 MFLOP: rr_glm_sgd_step's ~30 dependent launches are then all of its time.  rr_glm_svi
 * keeps X, y, the per-row argument, the flat vector z of rr_glm_sgd and the updater's state in HBM and runs `steps` steps of
 *     sgd (optimize/sgd.py:337-425)  o  logtrick_sgd (decorators.py:329-408)  o  structured_sgd (:133-252)  around  _elbo (glm.py:205-294)
 * inside ONE kernel launch: K cooperating workgroups (workgroup k owns mixture component k), the minibatch rows gathered by
 * index, Phi, the three products, the likelihood terms, the mixture terms, the gradient, the log trick's chain rule, bounds,
 * updater, -ELBO and |gradient| per step -- float64 throughout (draws are float32 values).  Same children, z layout, updaters
 * and likelihoods as rr_glm_sgd; rr_glm_svi_supported says whether a shape is in range (F K and minibatch x F small enough for
 * one CU's LDS).
 * dX[c], x_dtype[c], ldx[c]: child c's RESIDENT rows of its columns of X (all N rows, device); dy / drowarg: targets and the
 * binomial's n for all N rows (device, dtype); dlconst: per row, the part of loglike that does not depend on f (device float64:
 * Poisson -lgamma(y + 1), binomial lgamma(n + 1) - lgamma(y + 1) - lgamma(n - y + 1); NULL: zero -- Bernoulli, Gaussian);
 * M: minibatch rows; bmag = N / M (glm.py:158). */
typedef struct rr_glm_svi rr_glm_svi;
int rr_glm_svi_supported(int F, int K, int L, int M, int n_children, int dsum, int n_ls);
int rr_glm_svi_create(rr_ctx *ctx, int n_children, const rr_glm_sgd_child *children, const void *const *dX, const int *x_dtype,
                      const int64_t *ldx, int64_t N, const void *dy, const void *drowarg, const double *dlconst, int dtype, int K,
                      int L, int M, int lik, int n_lik, const double *z0, const double *lower, const double *upper,
                      const unsigned char *is_log, int updater, const double *upd_par, int64_t maxiter, double bmag,
                      rr_glm_svi **out);
/* The start point (structured_sgd picks it after the random starts, decorators.py:223-234), the bounds and the log-space
 * flags as logtrick_sgd leaves them (decorators.py:586-616), before the first step; NULL: unchanged. */
int rr_glm_svi_set_start(rr_glm_svi *s, const double *z0, const double *lower, const double *upper, const unsigned char *is_log);
/* `steps` SGD steps in one launch, asynchronous on the context's stream.  d_idx: device int32 (steps, M), the minibatches' row
 * indices in gen_batch's order (sgd.py:428-470); dE: device float32 (steps, K L, F), the caller's standard normals in the
 * reference's order (glm.py:300: randn(L, D) per component, component by component), or NULL: the counter-based device
 * generator of rr_featmat_glm_step_sampled keyed by (seed, key0 + step). */
int rr_glm_svi_run(rr_glm_svi *s, int64_t steps, const int *d_idx, const float *dE, uint64_t seed, uint64_t key0);
/* The random starts of structured_sgd (decorators.py:541-583) as ONE launch: candidate c (cand_host: (ncand, np) float64 in x
 * space, i.e. the Parameters' draws themselves) is scored on minibatch d_idx[c] (device int32 (ncand, M)) with draws dE[c]
 * (device float32 (ncand, K L, F)) or the device generator keyed by (seed, key0 + c); objs_host[c] = -ELBO.  Synchronous. */
int rr_glm_svi_starts(rr_glm_svi *s, int ncand, const int *d_idx, const double *cand_host, const float *dE, uint64_t seed,
                      uint64_t key0, double *objs_host);
/* As rr_glm_sgd_read. */
int rr_glm_svi_read(rr_glm_svi *s, double *z, double *objs, double *norms, int64_t *steps);
void rr_glm_svi_destroy(rr_glm_svi *s);

/* ---- arithmetic of the f32 Gram (Phi^T Phi of slm.py:145-150 for "f32" bases) -------------------
 * RR_GRAM_F32 (default): f32 features, v_mfma_f32_32x32x2_f32, f32 accumulation per K-split -- bitwise an fmaf chain.
 * RR_GRAM_BF16X3 / RR_GRAM_BF16X4: each f32 feature value is split into bf16 hi + lo and the Gram accumulates
 * hi.hi + hi.lo + lo.hi (+ lo.lo) in f32 on the bf16 matrix pipe (16x the f32 MFMA rate).  Results agree with the
 * f32 engine to ~4e-6 of max|G|, well inside the 1e-3 tolerance of the f32 path; the f64 Gram is
 * unaffected.  The environment variable RR_SYRK_ENGINE = bf16x3 | bf16x4 | fp16x3 sets the default of new contexts. */
#define RR_GRAM_F32 0
#define RR_GRAM_BF16X3 3
#define RR_GRAM_BF16X4 4
#define RR_GRAM_FP16X3 5 /* random Fourier features (bounded by 1/sqrt(n)) scaled into [-1, 1] and split into fp16 hi + lo:
                          * 22 mantissa bits, 3 fp16 products; other feature matrices (concatenations, Xdim > 128) use
                          * the bf16x3 split under this setting */
int rr_set_gram_engine(rr_ctx *ctx, int engine);
int rr_get_gram_engine(rr_ctx *ctx);

/* ---- run-to-run reproducibility (opt-in) -----------------------------------------------------
 * The reference's `Phi.T.dot(Phi)` (slm.py:146) gives the same bits every run.  By default the kernels here sum their
 * workgroups' partial results into float64 accumulators with floating-point atomics, whose order -- and so the last
 * bits of G, b, the ELBO and its gradients -- changes from run to run.  With rr_set_deterministic(ctx, 1) (or
 * RR_DETERMINISTIC=1 in the environment of a new context) every such sum is made in a fixed order instead: workgroups
 * store their partials (per K-split tiles of the SYRK kernels, per row-block vectors of Phi^T y / y^T y / sqErr / the
 * hyper-gradient contraction, per row traces of the posterior), and a second kernel adds them in ascending order.
 * Covered: rr_rff_gram[_dev], rr_dense_gram, rr_featmat_gram (f32 MFMA and f64 engines), rr_posterior_dev, the second
 * pass (rr_rff_elbo_pass2_dev[c], rr_featmat_pass2_*), rr_rff_grad_contract -- i.e. `basis.gram` and a whole
 * StandardLinearModel._elbo; two ranks holding the same reduced statistics then compute bit-identical objectives and
 * gradients.  Not covered (still atomics): the split 16-bit Gram engines (refused in this mode), the GLM minibatch
 * step, Xdim > 128.  Cost: scratch for the partials (nsplits x Fp^2 x 4 bytes for the f32 SYRK: 4.3 GB at F = 4096,
 * 2M rows per chunk) and < 2 % time. */
int rr_set_deterministic(rr_ctx *ctx, int on);
int rr_get_deterministic(rr_ctx *ctx);

/* ---- posterior of the standard linear model on the device -------------------
 * iC = diag(iL) + G / var,  C = iC^-1 by Cholesky (solve_posdef(iC, I), mathfun/linalg.py:84-125, as called at
 * slm.py:155),  m = C b / var (slm.py:157), and the O(F^2) reductions of slm.py:160,165-171.
 *   dG (F, F) float64 DEVICE, symmetric (after rr_symmetrize_dev), unchanged;  db (F) float64 DEVICE;
 *   iL (F) host float64 = 1 / regularizer_diagonal;  dC (F, F) float64 DEVICE output (full symmetric C).
 *   Host outputs: m (F), diagC (F), scal[3] = { log|iC|, sum(G o C), smallest diagonal entry of the factor }.
 * Blocked float64 Cholesky and inverse in hand-written kernels (rr_posdef.hip); rr_posterior_available() is 1.
 * RR_ERR_NOT_POSDEF when the factorisation meets a non-positive pivot or its smallest diagonal entry is below
 * 1e-5 (CHOLTHRESH, linalg.py:31,113): the caller then takes the reference's SVD route on the host. */
int rr_posterior_available(void);
int rr_posterior_dev(rr_ctx *ctx, int64_t F, const double *dG, const double *db, const double *iL, double var,
                     double *dC, double *m, double *diagC, double *scal);

/* ---- second data pass of the standard linear model (posterior known) -----------------------
 * With m (F,) and C (F, F) from the host Cholesky (slm.py:154-157), for a random Fourier basis and
 * DEVICE-resident X (padded layout, see rr_rff_padded_dim) and y:
 *   *sqerr = sum_r (y_r - Phi_r . m)^2                                     (slm.py:161-162)
 *   T (d, n) row-major = X^T A,  A = Err (Phi_c m_s - Phi_s m_c) - (Phi_c U_s - Phi_s U_c),  U = Phi C,
 * from which the hyper-gradient of slm.py:193-197 follows WITHOUT the (N, 2n, d) dPhi tensor:
 *   dhyps(dPhi_i) = -( m.(Err.dPhi_i) - sum(dPhi_i^T Phi o C) ) / var  =  sum_f T[i][f] W[i][f] / (var l_i^2)
 * (isotropic length scale: the reference uses i = 0 only).  f32 arithmetic; U is an MFMA GEMM.
 * m, C, sqerr, T are host buffers; the call is synchronous. */
int rr_rff_elbo_pass2_dev(rr_basis *basis, const void *dX, const void *dy, int x_dtype, int64_t N,
                          int64_t ldx, const double *lenscale, int n_ls, const double *m, const double *C,
                          double *sqerr, double *T);
/* The same with C (F, F) float64 resident on the DEVICE (rr_posterior_dev's dC); m stays a host vector. */
int rr_rff_elbo_pass2_devc(rr_basis *basis, const void *dX, const void *dy, int x_dtype, int64_t N,
                           int64_t ldx, const double *lenscale, int n_ls, const double *m, const double *dC,
                           double *sqerr, double *T);

/* Basis-gradient contraction for ANY consumer of basis.grad: given a host matrix E (N, 2n) -- the factor
 * the caller would multiply element-wise with each dPhi_i and sum (slm.py:193-195: E = Err m^T - Phi C;
 * glm.py:274-275: E = EdPhi) -- return T (d, n) row-major float64 with
 *     sum_{r,j} E[r][j] dPhi_i[r][j]  ==  -(1 / l_i^2) * sum_f W[i][f] * T[i][f]
 * (i = 0 only for an isotropic length scale, as in the reference).  The (N, 2n, d) tensor of
 * basis_functions.py:866-901 is never formed.  f32 arithmetic; X, E are host buffers. */
int rr_rff_grad_contract(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx,
                         const double *lenscale, int n_ls, const void *E, int e_dtype, int64_t lde, double *T);

/* predict_moments (slm.py:240-244) for DEVICE-resident query rows: Ey = Phi m, Vf = rowsum((Phi C) o Phi)
 * (host outputs, length N; the caller adds var to Vf). */
int rr_rff_predict_dev(rr_basis *basis, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                       const double *lenscale, int n_ls, const double *m, const double *C, double *Ey,
                       double *Vf);
/* The same with C (F, F) float64 resident on the DEVICE: a fitted estimator uploads its covariance once and
 * serves many predict_moments calls (the host conversion + upload of C is 9 ms per call at F = 4096). */
int rr_rff_predict_devc(rr_basis *basis, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                        const double *lenscale, int n_ls, const double *m, const double *dC, double *Ey,
                        double *Vf);

/* `predict` (slm.py:201-217) asks for the mean only -- the reference forms it through predict_moments, variance and all;
 * here Ey = Phi m comes from the feature kernel alone (no N x F x F product, no feature matrix in HBM).  float32 bases
 * of Xdim <= 128 (RR_ERR_UNSUPPORTED otherwise: use rr_rff_predict_dev and drop Vf). */
int rr_rff_predict_mean_dev(rr_basis *basis, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                            const double *lenscale, int n_ls, const double *m, double *Ey);

/* The variance of predict_moments without cancellation.  phi^T C phi in float32 loses digits when C is badly scaled
 * (error ~ eps * |phi|^T |C| |phi|, which can exceed the result).  With C = M M^T (M upper triangular: the "UL" Cholesky
 * factor, float64 on the device, hand-written blocked kernels of rr_posdef.hip) it is the sum of squares || phi^T M ||^2:
 *   rr_variance_factor_dev   dC (F, F) float64 DEVICE  ->  dB (Fp, Fp) float32 DEVICE, Fp = F rounded up to 256.
 *                            *form = 1: dB = M, Vf = rowsum((Phi dB)^2).  *form = 0 (C not safely positive definite, e.g.
 *                            from the SVD route): dB = the triangular form of C, Vf = rowsum((Phi dB) o Phi) as before.
 *                            Either way the product stops at the diagonal (half of Phi C).  Compute once per covariance.
 *   rr_rff_predict_devb / rr_featmat_predict_begin_b   the prediction entry points taking that factor ("f32" bases). */
int rr_variance_factor_dev(rr_ctx *ctx, int64_t F, const double *dC, float *dB, int *form);
int rr_rff_predict_devb(rr_basis *basis, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                        int n_ls, const double *m, const float *dB, int form, double *Ey, double *Vf);
int rr_featmat_predict_begin_b(rr_featmat *fm, const double *m, const float *dB, int form);

/* ---- FastFood -------------------------------------------------------------------------
 * FastFoodRBF (basis_functions.py:1211-1383).  B (+-1, int64), G, PI (int64 permutations) and S are
 * the host-sampled (k, d2) matrices of _init_matrices / _weightsamples (:1342-1354), row-major;
 * d2 = 2^ceil(log2 d), n = d2 * k (:1331-1340).  The handle serves rr_fastfood_transform /
 * rr_fastfood_vx; gradients and the Gram use the dense equivalent W = _makeVX(I_d) (obtained with
 * rr_fastfood_vx on the identity) through the rr_rff_* entry points -- the chain is linear in x. */
int rr_fastfood_create(rr_ctx *ctx, int compute, int d, int d2, int k, const int64_t *B, const double *G,
                       const int64_t *PI, const double *S, rr_basis **out);

/* Phi = [cos VX, sin VX] / sqrt(n), (N, 2n): FastFoodRBF.transform (basis_functions.py:1263-1289)
 * by the Hadamard / permute / diagonal chain (in-wave butterflies, LDS permutation gather). */
int rr_fastfood_transform(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx,
                          const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi);
/* The same for DEVICE-resident X (N, ldx >= d; no padding needed) and DEVICE output (N, ldphi >= 2n); asynchronous on
 * the context's stream. */
int rr_fastfood_transform_dev(rr_basis *basis, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                              const double *lenscale, int n_ls, void *dPhi, int out_dtype, int64_t ldphi);

/* VX = _makeVX(X / lenscale), (N, n) in radians (basis_functions.py:1356-1371). */
int rr_fastfood_vx(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx,
                   const double *lenscale, int n_ls, void *VX, int out_dtype, int64_t ldvx);

/* FastFoodGM (basis_functions.py:1386-1562), a spectral-mixture component, on an rr_rff_create handle
 * holding the dense equivalent W = _makeVX(I_d) of the FastFood chain:
 *   Phi (N, 4n) = [cos(VX + mX), sin(VX + mX), cos(VX - mX), sin(VX - mX)] / sqrt(2n),  mX = X . mean   (:1443-1475)
 *   rr_gm_grad: dPhi/dmean and dPhi/dlenscale, each (N, 4n, d) C-order ((N, 4n) when d == 1)          (:1477-1537)
 * mean (d) and lenscale (n_ls == d) are host float64. */
int rr_gm_transform(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                    const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi);
int rr_gm_grad(rr_basis *basis, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
               const double *lenscale, int n_ls, void *dmean, void *dlen, int out_dtype);

/* The same features by the Hadamard / permute / diagonal CHAIN itself, on an rr_fastfood_create handle (16 <= d2 <= 256;
 * RR_ERR_UNSUPPORTED otherwise -- the dense route above serves those):  VX from the chain kernel, mX = x . mean reduced in
 * registers next to it, four trig blocks of width n = d2 k scaled by 1 / sqrt(2 n) -- no dense (d, n) equivalent, no
 * (N, n) intermediate.  Host buffers (N, 4n) / device buffers / straight into a device feature matrix at columns
 * [col0, col0 + 4n), where the block pairs [cos | sin](VX + mX) at col0 and [cos | sin](VX - mX) at col0 + 2n look like
 * two random Fourier children to the second pass: with T+ / T- from rr_featmat_pass2_rff at the two offsets (on the dense
 * handle, for its n and padded X layout),
 *     sum(E o dPhi/dmean_i) = sum_f (T+ - T-)[i][f],     sum(E o dPhi/dl_i) = -(1 / l_i^2) sum_f V[i][f] (T+ + T-)[i][f]
 * (basis_functions.py:1477-1537 without the two (N, 4n, d) tensors; V = _makeVX(I_d)). */
int rr_fastfood_gm_transform(rr_basis *fastfood, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                             const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi);
int rr_fastfood_gm_transform_dev(rr_basis *fastfood, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                                 const double *lenscale, int n_ls, void *dPhi, int out_dtype, int64_t ldphi);
int rr_featmat_put_fastfood_gm(rr_featmat *fm, rr_basis *fastfood, const void *dX, int x_dtype, int64_t ldx,
                               const double *mean, const double *lenscale, int n_ls, int64_t col0);

/* mathfun.linalg.hadamard (mathfun/linalg.py:182-236): natural-order Walsh-Hadamard transform of each
 * row of host Y (rows, n), n = 2^p <= 4096, normalised by 1/n; ordering != 0 applies the sequency
 * permutation.  out has Y's dtype and shape. */
int rr_hadamard(rr_ctx *ctx, const void *Y, int dtype, int64_t rows, int64_t n, int ordering, void *out);

/* Name of the dominant kernel the last rr_rff_gram_dev launched (for profiles). */
const char *rr_rff_gram_kernel_name(rr_basis *basis);

/* ---- multi-GPU: the one exchange step of the path --------------------------------
 * Rows shard across GPUs, one process (and one rr_ctx) per GPU.  What is summed over the ranks is what
 * slm.py:145-157 computes from ALL rows: PhiPhi = Phi.T.dot(Phi) (:146), Phi.T.dot(y) (:157), plus y^T y and N
 * (the ELBO of :165-171 needs them), and after the replicated Cholesky the 1 + d numbers [sqErr | dhyp] of
 * :161-162,193-197.  RCCL (librccl.so.1) is bound directly with dlopen -- no PyTorch in the process -- and the
 * collectives run on the context's stream, ordered after the Gram kernels that produce their input.
 *
 *   rr_comm_unique_id      ncclGetUniqueId: 128 bytes that rank 0 hands to every other rank (file, socket, ...)
 *   rr_comm_init_rank      ncclCommInitRank on ctx's device; collective over all `world` ranks
 *   rr_comm_info           rank / world AS RCCL REPORTS THEM (ncclCommUserRank / ncclCommCount)
 *   rr_comm_allreduce_dev  in-place ncclAllReduce(ncclDouble) of a DEVICE float64 buffer, asynchronous on the stream
 *   rr_comm_allreduce_host the same for a small HOST vector (staged through HBM), synchronous
 *   rr_comm_broadcast_host root's host bytes to every rank (random start points etc.), synchronous
 *   rr_comm_barrier        all ranks arrived and their streams are idle
 *   rr_comm_load           optional: path of the librccl to bind (default: $RR_RCCL_LIB, an already loaded
 *                          librccl.so.1, /opt/rocm/lib/librccl.so.1); rr_comm_version reports version and path. */
typedef struct rr_comm rr_comm;
#define RR_COMM_ID_BYTES 128
#define RR_COMM_SUM 0
#define RR_COMM_MAX 1
#define RR_COMM_MIN 2
int rr_comm_load(const char *path);
int rr_comm_version(int *version, char *path, size_t path_len);
int rr_comm_unique_id(void *id /* RR_COMM_ID_BYTES */);
int rr_comm_init_rank(rr_ctx *ctx, int rank, int world, const void *id, rr_comm **out);
void rr_comm_destroy(rr_comm *comm);
int rr_comm_info(rr_comm *comm, int *rank, int *world);
int rr_comm_allreduce_dev(rr_comm *comm, double *dbuf, int64_t count, int op);
int rr_comm_allreduce_host(rr_comm *comm, double *hbuf, int64_t count, int op);
int rr_comm_broadcast_host(rr_comm *comm, void *hbuf, int64_t bytes, int root);
int rr_comm_barrier(rr_comm *comm);

/* The message of the exchange: [ upper triangle of G, row-major, row i = G[i][i..F) | b (F) | yty | nrows ],
 * rr_stats_msg_count(F) = F (F + 1) / 2 + F + 2 float64 -- 67 MB at F = 4096 where the full square is 134 MB.
 *   rr_stats_pack_dev    from the accumulators rr_rff_gram_dev / rr_featmat_gram wrote (upper triangle of dG valid);
 *                        db / dyty may be NULL (zeros are sent)
 *   rr_stats_unpack_dev  back into dG as the FULL symmetric matrix (mirrors like rr_symmetrize_dev), db, dyty
 *   rr_comm_reduce_stats_dev   pack -> ncclAllReduce(sum) -> unpack on the context's stream; dmsg is a DEVICE scratch of
 *                        rr_stats_msg_count(F) float64.  total_rows != NULL: waits and returns the summed N.
 * All DEVICE pointers, asynchronous on the context's stream unless stated. */
int64_t rr_stats_msg_count(int64_t F);
int rr_stats_pack_dev(rr_ctx *ctx, int64_t F, const double *dG, const double *db, const double *dyty, double nrows,
                      double *dmsg);
int rr_stats_unpack_dev(rr_ctx *ctx, int64_t F, const double *dmsg, double *dG, double *db, double *dyty);
int rr_comm_reduce_stats_dev(rr_comm *comm, int64_t F, double *dG, double *db, double *dyty, double nrows, double *dmsg,
                             double *total_rows);

/* ---- ONE process, several GPUs: the estimator's own call (rr_init(n_devices, device_ids, ...)) ----------
 * The reference's fit is one call in one process (slm.py:74-140), driven by sklearn Pipelines / GridSearchCV
 * (tests/test_models.py:39-80) that cannot be wrapped in a per-GPU launcher.  For that caller the row shards live in n
 * contexts of the SAME process (rr_ctx_create once per member; two members may share a device), every member accumulates
 * the statistics of ITS rows on its own stream, and one host thread sums them over the members:
 *   rr_comm_init_all            n communicators, member i bound to ctxs[i] (rank i of n).  transport:
 *                               RR_TRANSPORT_RCCL  ncclCommInitAll over the members' devices (all distinct); collectives are
 *                                                  ncclGroupStart / per-member call on its stream / ncclGroupEnd
 *                               RR_TRANSPORT_PEER  no library: after hipDeviceEnablePeerAccess member i's kernels load the
 *                                                  other members' buffers directly over the xGMI mesh -- reduce-scatter
 *                                                  (member i combines slice i of all buffers, in member order) then
 *                                                  all-gather; bit-identical on every member, the same bits every run;
 *                                                  also the transport of members that SHARE a device
 *                               RR_TRANSPORT_AUTO  RCCL when every member has its own device and librccl loads, else PEER
 *                                                  ($RR_COMM_TRANSPORT = rccl | peer overrides)
 *   rr_comm_transport           the transport chosen
 *   rr_comm_group_allreduce_dev in-place reduction of the members' DEVICE float64 buffers dbufs[i] (count each),
 *                               asynchronous: ordered on every member's stream behind what produced its buffer; work queued
 *                               afterwards on any member's stream sees the result
 *   rr_comm_group_broadcast_dev dbufs[root]'s bytes (a multiple of 8) into every member's buffer, same ordering
 *   rr_comm_group_reduce_stats_dev   rr_comm_reduce_stats_dev for the group: pack -> all-reduce -> unpack + mirror on every
 *                               member; nrows[i] = member i's row count; total_rows != NULL: waits for member 0 and returns
 *                               the summed N
 * The per-rank collectives (rr_comm_allreduce_dev / _host, rr_comm_broadcast_host, rr_comm_barrier) refuse a member of a
 * group of more than one: entered member by member from one thread they would wait for their peers forever.
 * rr_comm_destroy frees a member; the group's shared record goes with its last member. */
#define RR_TRANSPORT_AUTO 0
#define RR_TRANSPORT_RCCL 1
#define RR_TRANSPORT_PEER 2
int rr_comm_init_all(int n, rr_ctx *const *ctxs, int transport, rr_comm **out /* n */);
int rr_comm_transport(rr_comm *comm);
int rr_comm_group_allreduce_dev(rr_comm *const *comms, int n, double *const *dbufs, int64_t count, int op);
int rr_comm_group_broadcast_dev(rr_comm *const *comms, int n, void *const *dbufs, int64_t bytes, int root);
int rr_comm_group_reduce_stats_dev(rr_comm *const *comms, int n, int64_t F, double *const *dG, double *const *db,
                                   double *const *dyty, const double *nrows, double *const *dmsg, double *total_rows);

/* The same step with the minibatch's rows spread over the n members of an in-process device group (rr_comm_init_all; the
 * reference has no counterpart: GeneralizedLinearModel(devices=[...]).fit, the estimator's one call over several GPUs).
 * loops[i]: member i's loop, created on the context comms[i] is bound to from the SAME z0 / bounds / updater; batches[i]: ITS
 * rows of the minibatch (rows == 0 allowed: a member none of the minibatch's indices fell to), draws dE on ITS device (the
 * same values for every member) or NULL with the shared (seed, key).  llconst and bmag are the whole minibatch's.  Member i
 * forms the products on its rows; what a step sums over rows -- the length-scale contractions, [Edm | EdC], the likelihood
 * sums -- is added over the members in HBM (two stream-ordered all-reduces: 8 dT and 8 (2 F K + 2 K) bytes), and every member
 * makes the same update of its copy of z: the copies stay bit-identical, rr_glm_sgd_read of any member returns the fit.
 * Queued from ONE host thread; returns without waiting, like rr_glm_sgd_step. */
typedef struct rr_glm_sgd_batch {
    const void *const *dX; /* [children] */
    const int *x_dtype;
    const int64_t *ldx;
    int64_t rows;
    const void *dy, *drowarg;
    const float *dE;
} rr_glm_sgd_batch;
int rr_glm_sgd_group_step(int n, rr_glm_sgd *const *loops, rr_comm *const *comms, const rr_glm_sgd_batch *batches, int dtype,
                          int lik, double llconst, double bmag, int L, uint64_t seed, uint64_t key);

/* The same step on ONE rank of a one-process-per-GPU job (comm from rr_comm_init_rank, bound to the loop's context):
 * GeneralizedLinearModel(distributed=True).fit -- every rank calls it with a minibatch of ITS rows (rows == 0 allowed), its
 * llconst and the whole job's bmag; dT and [Edm | EdC | llsum | aux | llconst | rows] are all-reduced over the ranks in HBM
 * (ncclAllReduce on the context's stream) and every rank makes the same update of its copy of z.  Collective: all ranks, the
 * same number of times.  Returns without waiting, like rr_glm_sgd_step. */
int rr_glm_sgd_dist_step(rr_glm_sgd *s, rr_comm *comm, const void *const *dX, const int *x_dtype, const int64_t *ldx, int64_t rows,
                         const void *dy, const void *drowarg, int dtype, int lik, double llconst, double bmag, int L, const float *dE,
                         uint64_t seed, uint64_t key);

/* ---- host-side random stream of the GLM step ------------------------------ */

/* Advance a NumPy legacy RandomState (MT19937 + polar Box-Muller with one cached value) by n standard normals and write
 * them to `out` (RR_F32 or RR_F64; float32 is the round-to-nearest cast of the float64 value), bit for bit what
 * `RandomState.randn(n)` returns and leaves behind.  Replaces `self.random_.randn(self.nsamples, D)` of
 * revrand/glm.py:300 (`_reparam_k`), called K times per SVI step.  key: the 624 state words; pos: 0..624; has_gauss / gauss:
 * the cached second value of the last pair (`RandomState.get_state()[3:5]`).  The MT19937 words and the accept / reject
 * loop run on the calling thread, sqrt(-2 log(r2) / r2) of the accepted pairs on up to `threads` worker threads.  Host
 * only: needs no device. */
int rr_legacy_randn(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, void *out, int out_dtype, int64_t n,
                    int threads);
/* random_state.permutation(n) of the same generator (the minibatch index stream of optimize/sgd.py:428-470 through
 * utils/rand.py:7-31), bit for bit and leaving the same state: out = n int64 values.  NumPy's own loop holds the GIL (18 ms at
 * n = 2M, once per epoch) -- every Python thread of the fit stalls with it; this one does not.  n <= 2^32. */
int rr_legacy_permutation(uint32_t *key, int32_t *pos, int64_t n, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* REVRAND_HIP_H */
