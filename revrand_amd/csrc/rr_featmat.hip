// Device feature matrix: the zero-padded f32 scratch P (rows % 32 == 0, ld % 256 == 0) that the MFMA
// SYRK kernel consumes, filled column block by column block by the children of a concatenated basis
// (BasisCat.transform's hstack, basis_functions.py:1599-1627, without leaving the GPU), then reduced
// to Phi^T Phi / Phi^T y (slm.py:146,157).
#include "rr_internal.h"

int rr_features_rowmajor_f32(rr_basis *b, const void *dX, int x_dtype, int64_t m, int64_t mpad, int64_t ldx,
                             float *P, int64_t ldp, bool zero_pad_cols, float *Pt = nullptr, int64_t ldt = 0,
                             bool *pt_written = nullptr, double scale_mult = 1.0);
int rr_launch_syrk_f32(rr_ctx *c, const float *P, int64_t rows, int64_t ldp, int F, double *dG, hipEvent_t mid,
                       double *bcol = nullptr);


// [1, X] or X (LinearBasis.transform, basis_functions.py:468-485) into columns [col0, col0 + d + onescol)
template <typename TX>
__global__ void __launch_bounds__(256)
rr_linear_features_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, int d, int onescol, float *__restrict__ P,
                          int64_t ldp) {
    const int w = d + onescol;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * w) return;
    const int64_t r = i / w;
    const int c = (int)(i % w);
    RR_DEV_ASSERT(w <= ldp && d <= ldx);
    P[r * ldp + c] = (onescol && c == 0) ? 1.f : (float)X[r * ldx + (c - onescol)];
}

// the same block feature-major: Pt[c][r] for r < Npad (zero for r >= N), coalesced along r
template <typename TX>
__global__ void __launch_bounds__(256)
rr_linear_features_t_kernel(const TX *__restrict__ X, int64_t N, int64_t Npad, int64_t ldx, int d, int onescol,
                            float *__restrict__ Pt, int64_t ldt) {
    const int w = d + onescol;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad * w) return;
    const int c = (int)(i / Npad);
    const int64_t r = i % Npad;
    RR_DEV_ASSERT(Npad <= ldt && d <= ldx);
    Pt[c * ldt + r] = r < N ? ((onescol && c == 0) ? 1.f : (float)X[r * ldx + (c - onescol)]) : 0.f;
}

template <typename TS>
__global__ void __launch_bounds__(256)
rr_copy_cols_kernel(const TS *__restrict__ src, int64_t N, int64_t lds_, int ncols, float *__restrict__ P, int64_t ldp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * ncols) return;
    const int64_t r = i / ncols;
    const int c = (int)(i % ncols);
    RR_DEV_ASSERT(ncols <= ldp && ncols <= lds_);
    P[r * ldp + c] = (float)src[r * lds_ + c];
}

// zero the pad columns [F, ld) of rows [0, rows)
__global__ void __launch_bounds__(256) rr_fm_zero_padcols_kernel(float *P, int64_t rows, int64_t ld, int F) {
    const int64_t w = ld - F;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows * w) P[(i / w) * ld + F + (i % w)] = 0.f;
}

template <typename TY>
__global__ void __launch_bounds__(256) rr_fm_gemv_t_kernel(const float *__restrict__ P, const TY *__restrict__ y,
                                                           int64_t rows, int F, int64_t ldp, double *__restrict__ bvec,
                                                           int rows_per_block, int64_t bdet = 0) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (c >= F) return;
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc = fmaf(P[r * ldp + c], (float)y[r], acc);
    rr_acc_out(bvec, bdet, blockIdx.y, c, (double)acc);
}

// P[r][col] = y[r] (y == nullptr: 0) for r < rows
template <typename TY>
__global__ void __launch_bounds__(256) rr_fm_set_column_kernel(float *__restrict__ P, int64_t ld, int64_t col, const TY *__restrict__ y,
                                                               int64_t rows) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) P[r * ld + col] = y ? (float)y[r] : 0.f;
}

template <typename TY>
__global__ void __launch_bounds__(256) rr_fm_yty_kernel(const TY *__restrict__ y, int64_t N, double *out, int64_t det = 0) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)y[i];
        acc += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rr_acc_out(out, det, blockIdx.x, 0, part[0] + part[1] + part[2] + part[3]);
}

// dst[r][:] = src[idx[r]][:] for rows of ld 4-byte elements (minibatch gather from resident data)
__global__ void __launch_bounds__(256)
rr_gather_rows_kernel(const uint32_t *__restrict__ src, const int *__restrict__ idx, int64_t rows, int64_t ld,
                      uint32_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld) return;
    const int64_t r = i / ld, c = i % ld;
    RR_DEV_ASSERT(idx[r] >= 0);
    dst[i] = src[(int64_t)idx[r] * ld + c];
}

// the same in 16-byte pieces (ld % 4 == 0, 16-byte aligned buffers): a 32-word row is 8 lanes, one wave gathers 8 rows
__global__ void __launch_bounds__(256)
rr_gather_rows16_kernel(const uint4 *__restrict__ src, const int *__restrict__ idx, unsigned rows, unsigned ld4,
                        uint4 *__restrict__ dst) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const unsigned r = i / ld4, c = i - r * ld4;
    if (r >= rows) return;
    RR_DEV_ASSERT(idx[r] >= 0);
    dst[i] = src[(size_t)idx[r] * ld4 + c];
}

extern "C" {

int rr_gather_rows(rr_ctx *c, const void *dsrc, const int *didx, int64_t rows, int64_t ld_words, void *ddst) {
    RR_REQUIRE(c != nullptr && rows >= 0 && ld_words >= 1, "rr_gather_rows: bad argument");
    if (rows == 0) return RR_OK;
    RR_REQUIRE(dsrc != nullptr && didx != nullptr && ddst != nullptr, "rr_gather_rows: null buffer");
    RR_CHECK_HIP(hipSetDevice(c->device));
    if (ld_words % 4 == 0 && ((uintptr_t)dsrc & 15) == 0 && ((uintptr_t)ddst & 15) == 0 && rows * (ld_words / 4) < (1ll << 31))
        hipLaunchKernelGGL(rr_gather_rows16_kernel, dim3((unsigned)((rows * (ld_words / 4) + 255) / 256)), dim3(256), 0, c->stream,
                           (const uint4 *)dsrc, didx, (unsigned)rows, (unsigned)(ld_words / 4), (uint4 *)ddst);
    else
        hipLaunchKernelGGL(rr_gather_rows_kernel, dim3((unsigned)((rows * ld_words + 255) / 256)), dim3(256), 0, c->stream,
                           (const uint32_t *)dsrc, didx, rows, ld_words, (uint32_t *)ddst);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_featmat_create(rr_ctx *ctx, int64_t max_rows, int64_t F, rr_featmat **out) {
    RR_REQUIRE(ctx != nullptr && out != nullptr, "rr_featmat_create: null argument");
    *out = nullptr;
    RR_REQUIRE(max_rows >= 1 && F >= 1 && F < (1 << 30), "rr_featmat_create: bad shape");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    rr_featmat *fm = new rr_featmat();
    fm->ctx = ctx;
    fm->F = (int)F;
    fm->ld = (F + 255) / 256 * 256;
    fm->max_rows = (max_rows + 255) / 256 * 256;  // the second pass' GEMM tiles rows by 256
    hipError_t e = hipMalloc((void **)&fm->P, (size_t)fm->max_rows * fm->ld * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_featmat_create: hipMalloc(%zu bytes) failed", (size_t)fm->max_rows * fm->ld * 4);
        delete fm;
        return RR_ERR_OOM;
    }
    *out = fm;
    return RR_OK;
}

void rr_featmat_destroy(rr_featmat *fm) {
    if (!fm) return;
    (void)hipSetDevice(fm->ctx->device);
    (void)hipStreamSynchronize(fm->ctx->stream);
    if (fm->P) (void)hipFree(fm->P);
    rr_fm_pass2_free(fm->pass2);
    delete fm;
}

int rr_featmat_begin(rr_featmat *fm, int64_t rows) {
    RR_REQUIRE(fm != nullptr && rows >= 0 && rows <= fm->max_rows, "rr_featmat_begin: rows out of range");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    fm->rows = rows;
    fm->rows_pad = (rows + 31) / 32 * 32;
    // Every column of [0, F) is overwritten by a put_* call for the rows [0, rows) (checked by the consumers:
    // RR_FM_REQUIRE_FILLED), so only the PADDING is zeroed here: the pad columns [F, ld) of every row and the pad rows up
    // to the next multiple of 256 (the second pass' and the GLM step's GEMMs read whole 256-row tiles).  Zeroing the
    // whole matrix cost 1.5 ms per 254 200 x 8448 chunk of config 3 (1 % of its Gram pass).
    const int64_t rows256 = (rows + 255) / 256 * 256;
    fm->covered = 0;
    fm->pt_covered = 0;
    fm->spans.clear();
    if (rows256 > rows)
        RR_CHECK_HIP(hipMemsetAsync(fm->P + rows * fm->ld, 0, (size_t)(rows256 - rows) * fm->ld * sizeof(float), fm->ctx->stream));
    const int64_t w = fm->ld - fm->F;
    if (w > 0 && rows > 0) {
        hipLaunchKernelGGL(rr_fm_zero_padcols_kernel, dim3((unsigned)((rows * w + 255) / 256)), dim3(256), 0, fm->ctx->stream,
                           fm->P, rows, fm->ld, fm->F);
        RR_CHECK_HIP(hipGetLastError());
    }
    return RR_OK;
}

}  // extern "C"

// Record that columns [col0, col0 + width) are written; refuses a block that overlaps one put since rr_featmat_begin (a
// child put twice, or wrong offsets, would otherwise satisfy a width count while another block keeps stale data).
int rr_fm_claim(rr_featmat *fm, int64_t col0, int64_t width, const char *who) {
    const int64_t c1 = col0 + width;
    size_t pos = 0;
    while (pos < fm->spans.size() && fm->spans[pos].first < col0) ++pos;
    const bool clash = (pos > 0 && fm->spans[pos - 1].second > col0) || (pos < fm->spans.size() && fm->spans[pos].first < c1);
    RR_REQUIRE(!clash, "%s: columns [%lld, %lld) overlap a block already written since rr_featmat_begin", who,
               (long long)col0, (long long)c1);
    fm->spans.insert(fm->spans.begin() + (std::ptrdiff_t)pos, std::make_pair(col0, c1));
    fm->covered += width;
    return RR_OK;
}

// lenscale: host values, or (ls_on_device) the same in device memory -- rr_basis_prepare_dev, the resident SVI loop
static int fm_put_rff(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *lenscale, int n_ls,
                      int64_t col0, bool ls_on_device, const double *dshift = nullptr, double sgn = 0.0) {
    RR_REQUIRE(fm != nullptr && b != nullptr && b->kind == RR_KIND_RFF, "rr_featmat_put_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_put_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat_put_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad, "rr_featmat_put_rff: device X needs ldx >= rr_rff_padded_dim() = %d", b->dpad);
    int rc = ls_on_device ? rr_basis_prepare_dev(b, lenscale, n_ls, dshift, sgn) : rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK || fm->rows == 0) return rc;
    RR_REQUIRE(dX != nullptr, "rr_featmat_put_rff: null X");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    // the feature kernel writes columns [0, 2n) relative to its base; pad handling is ours (begin())
    rc = rr_fm_claim(fm, col0, 2 * (int64_t)b->n, "rr_featmat_put_rff");
    if (rc != RR_OK) return rc;
    // the same block of P^T, if a transposing pass has laid out P^T's padding for this row count before
    static const bool no_pt = getenv("RR_FM_NO_DIRECT_PT") != nullptr;
    float *Pt = (fm->pt_rows == fm->rows && !no_pt) ? rr_fm_pass2_pt(fm->pass2) : nullptr;
    bool wrote = false;
    rc = rr_features_rowmajor_f32(b, dX, x_dtype, fm->rows, fm->rows, ldx, fm->P + col0, fm->ld, false,
                                  Pt ? Pt + col0 * fm->max_rows : nullptr, fm->max_rows, &wrote, dshift ? 0.70710678118654752440 : 1.0);
    if (wrote) fm->pt_covered += 2 * (int64_t)b->n;
    return rc;
}

int rr_fm_put_rff_dev(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *dls, int n_ls,
                      int64_t col0, const double *dshift, double sgn) {
    return fm_put_rff(fm, b, dX, x_dtype, ldx, dls, n_ls, col0, true, dshift, sgn);
}

extern "C" {

int rr_featmat_put_rff(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *lenscale,
                       int n_ls, int64_t col0) {
    return fm_put_rff(fm, b, dX, x_dtype, ldx, lenscale, n_ls, col0, false);
}

int rr_featmat_put_linear(rr_featmat *fm, const void *dX, int x_dtype, int64_t ldx, int d, int onescol, int64_t col0) {
    RR_REQUIRE(fm != nullptr && d >= 1 && ldx >= d, "rr_featmat_put_linear: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_put_linear: bad dtype");
    const int w = d + (onescol ? 1 : 0);
    RR_REQUIRE(col0 >= 0 && col0 + w <= fm->F, "rr_featmat_put_linear: columns out of range");
    if (fm->rows == 0) return RR_OK;
    RR_REQUIRE(dX != nullptr, "rr_featmat_put_linear: null X");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    const int64_t cnt = fm->rows * w;
    {
        const int rcc = rr_fm_claim(fm, col0, w, "rr_featmat_put_linear");
        if (rcc != RR_OK) return rcc;
    }
    const dim3 grid((unsigned)((cnt + 255) / 256));
    if (x_dtype == RR_F32)
        hipLaunchKernelGGL(rr_linear_features_kernel<float>, grid, dim3(256), 0, fm->ctx->stream, (const float *)dX,
                           fm->rows, ldx, d, onescol ? 1 : 0, fm->P + col0, fm->ld);
    else
        hipLaunchKernelGGL(rr_linear_features_kernel<double>, grid, dim3(256), 0, fm->ctx->stream, (const double *)dX,
                           fm->rows, ldx, d, onescol ? 1 : 0, fm->P + col0, fm->ld);
    // the same block of P^T (see rr_featmat_put_rff)
    static const bool no_pt = getenv("RR_FM_NO_DIRECT_PT") != nullptr;
    float *Pt = (fm->pt_rows == fm->rows && !no_pt) ? rr_fm_pass2_pt(fm->pass2) : nullptr;
    if (Pt) {
        const dim3 gt((unsigned)((fm->rows_pad * w + 255) / 256));
        if (x_dtype == RR_F32)
            hipLaunchKernelGGL(rr_linear_features_t_kernel<float>, gt, dim3(256), 0, fm->ctx->stream, (const float *)dX, fm->rows,
                               fm->rows_pad, ldx, d, onescol ? 1 : 0, Pt + col0 * fm->max_rows, fm->max_rows);
        else
            hipLaunchKernelGGL(rr_linear_features_t_kernel<double>, gt, dim3(256), 0, fm->ctx->stream, (const double *)dX, fm->rows,
                               fm->rows_pad, ldx, d, onescol ? 1 : 0, Pt + col0 * fm->max_rows, fm->max_rows);
        fm->pt_covered += w;
    }
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_featmat_put_host(rr_featmat *fm, const void *Phi, int dtype, int64_t ncols, int64_t ldphi, int64_t col0) {
    RR_REQUIRE(fm != nullptr && ncols >= 1 && ldphi >= ncols, "rr_featmat_put_host: bad argument");
    RR_REQUIRE(dtype == RR_F32 || dtype == RR_F64, "rr_featmat_put_host: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + ncols <= fm->F, "rr_featmat_put_host: columns out of range");
    if (fm->rows == 0) return RR_OK;
    RR_REQUIRE(Phi != nullptr, "rr_featmat_put_host: null Phi");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = dtype == RR_F32 ? 4 : 8;
    void *raw = nullptr;
    RR_CHECK_HIP(hipMalloc(&raw, (size_t)fm->rows * ncols * es));
    hipError_t e = hipMemcpy2DAsync(raw, (size_t)ncols * es, Phi, (size_t)ldphi * es, (size_t)ncols * es, (size_t)fm->rows,
                                    hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int64_t cnt = fm->rows * ncols;
        const dim3 grid((unsigned)((cnt + 255) / 256));
        if (dtype == RR_F32)
            hipLaunchKernelGGL(rr_copy_cols_kernel<float>, grid, dim3(256), 0, c->stream, (const float *)raw, fm->rows, ncols,
                               (int)ncols, fm->P + col0, fm->ld);
        else
            hipLaunchKernelGGL(rr_copy_cols_kernel<double>, grid, dim3(256), 0, c->stream, (const double *)raw, fm->rows,
                               ncols, (int)ncols, fm->P + col0, fm->ld);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(raw);
    if (e != hipSuccess) {
        rr_set_error("rr_featmat_put_host: copy failed: %s", hipGetErrorString(e));
        return RR_ERR_HIP;
    }
    return rr_fm_claim(fm, col0, ncols, "rr_featmat_put_host");
}

// y^T y into *dyty: atomics, or (deterministic mode) per-block partials added in order
static int fm_yty(rr_ctx *c, const void *dy, int y_dtype, int64_t rows, double *dyty) {
    int yb = (int)((rows + 255) / 256);
    if (yb > c->num_cu * 8) yb = c->num_cu * 8;
    double *dst = dyty;
    const int64_t det = c->deterministic ? 1 : 0;
    if (det) {
        void *part = nullptr;
        int rc = rr_det_scratch(c, (size_t)yb * 8, &part);
        if (rc != RR_OK) return rc;
        dst = (double *)part;
    }
    if (y_dtype == RR_F32) hipLaunchKernelGGL(rr_fm_yty_kernel<float>, dim3(yb), dim3(256), 0, c->stream, (const float *)dy, rows, dst, det);
    else hipLaunchKernelGGL(rr_fm_yty_kernel<double>, dim3(yb), dim3(256), 0, c->stream, (const double *)dy, rows, dst, det);
    RR_CHECK_HIP(hipGetLastError());
    return det ? rr_det_reduce(c, dst, yb, 1, 1, dyty) : RR_OK;
}

int rr_featmat_gram(rr_featmat *fm, const void *dy, int y_dtype, double *dG, double *db, double *dyty) {
    RR_REQUIRE(fm != nullptr && dG != nullptr, "rr_featmat_gram: null argument");
    RR_REQUIRE((dy == nullptr) == (db == nullptr) && (dy == nullptr) == (dyty == nullptr),
               "rr_featmat_gram: y, b and yty must be given together");
    RR_REQUIRE(dy == nullptr || y_dtype == RR_F32 || y_dtype == RR_F64, "rr_featmat_gram: bad dtype");
    RR_FM_REQUIRE_FILLED(fm, "rr_featmat_gram");
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    // Phi^T y: as a rider of the SYRK -- y written into the first pad column of P for the duration of the product (one
    // pass over P instead of two; the f32 MFMA engine, a pad column to spare) -- or by its own kernel
    static const bool no_rider = getenv("RR_FM_NO_RIDER") != nullptr;
    const bool rider = dy != nullptr && c->gram_engine == 0 && fm->F < fm->ld && !no_rider;
    if (rider) {
        const unsigned cb = (unsigned)((fm->rows + 255) / 256);
        if (y_dtype == RR_F32)
            hipLaunchKernelGGL(rr_fm_set_column_kernel<float>, dim3(cb), dim3(256), 0, c->stream, fm->P, fm->ld, fm->F, (const float *)dy, fm->rows);
        else
            hipLaunchKernelGGL(rr_fm_set_column_kernel<double>, dim3(cb), dim3(256), 0, c->stream, fm->P, fm->ld, fm->F, (const double *)dy, fm->rows);
        RR_CHECK_HIP(hipGetLastError());
        int rc = fm_yty(c, dy, y_dtype, fm->rows, dyty);
        if (rc != RR_OK) return rc;
        rc = rr_launch_syrk_f32(c, fm->P, fm->rows_pad, fm->ld, fm->F, dG, nullptr, db);
        // the pad column is zero again for every other consumer of P
        hipLaunchKernelGGL(rr_fm_set_column_kernel<float>, dim3(cb), dim3(256), 0, c->stream, fm->P, fm->ld, fm->F, (const float *)nullptr, fm->rows);
        RR_CHECK_HIP(hipGetLastError());
        return rc;
    }
    if (dy) {
        const int rpb = 512;
        const dim3 gg((unsigned)((fm->F + 255) / 256), (unsigned)((fm->rows + rpb - 1) / rpb));
        double *bdst = db;
        const int64_t bdet = c->deterministic ? fm->F : 0;
        if (bdet) {
            void *part = nullptr;
            int rc = rr_det_scratch(c, (size_t)gg.y * (size_t)fm->F * 8, &part);
            if (rc != RR_OK) return rc;
            bdst = (double *)part;
        }
        if (y_dtype == RR_F32)
            hipLaunchKernelGGL(rr_fm_gemv_t_kernel<float>, gg, dim3(256), 0, c->stream, fm->P, (const float *)dy, fm->rows,
                               fm->F, fm->ld, bdst, rpb, bdet);
        else
            hipLaunchKernelGGL(rr_fm_gemv_t_kernel<double>, gg, dim3(256), 0, c->stream, fm->P, (const double *)dy, fm->rows,
                               fm->F, fm->ld, bdst, rpb, bdet);
        RR_CHECK_HIP(hipGetLastError());
        int rc = bdet ? rr_det_reduce(c, bdst, gg.y, fm->F, fm->F, db) : RR_OK;
        if (rc == RR_OK) rc = fm_yty(c, dy, y_dtype, fm->rows, dyty);
        if (rc != RR_OK) return rc;
    }
    return rr_launch_syrk_f32(c, fm->P, fm->rows_pad, fm->ld, fm->F, dG, nullptr);
}

}  // extern "C"
