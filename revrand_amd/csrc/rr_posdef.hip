// Posterior of the Bayesian linear model on the device: iC = diag(1/L) + G / var,
// C = iC^-1 and the O(F^2) statistics of StandardLinearModel._elbo (slm.py:150-171), so that an L-BFGS
// evaluation moves O(F) numbers over PCIe instead of two F x F matrices.
//
// Blocked right-looking Cholesky iC = U^T U (upper, as mathfun/linalg.py:109) and the inverse, all in float64:
//   per 128-column panel j:  rr_chol_diag_kernel   U_jj = chol(A_jj), U_jj^-1  one workgroup, the block in registers
//                            rr_gemm_tn_f64_kernel U_j,> = U_jj^-T A_j,>       f64 MFMA tiles
//                            rr_gemm_tn_f64_kernel A_>,> -= U_j,>^T U_j,>      upper tiles only
//   inverse:  Y = U^-T (block forward substitution on the identity: the same two GEMMs, Y lower triangular),
//             C = Y^T Y  (rr_syrk_f64_kernel + mirror).
// Everything is asynchronous on the context's stream; the host reads the factor's diagonal once for log|iC| and
// the CHOLTHRESH test of linalg.py:31,113.  If the matrix is not safely positive definite the call reports
// RR_ERR_NOT_POSDEF and the caller takes the reference's SVD route on the host (linalg.py:128-179).
#include <cmath>

#include "rr_internal.h"

int rr_launch_gemm_tn_f64(rr_ctx *c, const double *A, int64_t lda, const double *B, int64_t ldb, double *D, int64_t ldd,
                          int64_t K, int64_t M, int64_t N, int subtract, int upper_only);             // rr_rff.hip
int rr_launch_syrk_f64(rr_ctx *c, const double *P, int64_t rows, int64_t ldp, int F, double *dG, int lower_tri = 0);  // rr_rff.hip

constexpr int PB = 128;  // panel width = the f64 GEMM tile

// W (Fp, Fp) = G / var + diag(iL) in the top-left F x F, identity on the pad diagonal
__global__ void __launch_bounds__(256)
rr_assemble_ic_kernel(const double *__restrict__ G, const double *__restrict__ iL, double ivar, int64_t F, int64_t Fp,
                      double *__restrict__ W) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Fp * Fp) return;
    const int64_t r = i / Fp, c = i % Fp;
    double v = 0.0;
    if (r < F && c < F)
        v = G[r * F + c] * ivar + (r == c ? iL[r] : 0.0);
    else if (r == c)
        v = 1.0;
    W[i] = v;
}

__global__ void __launch_bounds__(256) rr_set_identity_kernel(double *Y, int64_t Fp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < Fp * Fp) Y[i] = (i / Fp == i % Fp) ? 1.0 : 0.0;
}

__global__ void __launch_bounds__(256) rr_get_diag_kernel(const double *__restrict__ A, int64_t n, int64_t ld, double *__restrict__ d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = A[i * ld + i];
}

// C (F, F) <- top-left of Cp (Fp, Fp)
__global__ void __launch_bounds__(256)
rr_extract_kernel(const double *__restrict__ Cp, int64_t Fp, double *__restrict__ C, int64_t F) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < F * F) C[i] = Cp[(i / F) * Fp + (i % F)];
}

// In-place upper Cholesky of one 128 x 128 diagonal block (row-major, leading dimension ld), A = U^T U, AND the
// inverse of the factor, Uinv = U^-1 (128 x 128, dense leading dimension 128), which turns the two triangular
// solves of a panel step into MFMA GEMMs.  The strict lower part of the block is zeroed.
//
// Right-looking and unblocked, with the block in REGISTERS: the 256 threads form a 16 x 16 grid, thread (ty, tx)
// owns the 8 x 8 elements (16 i + ty, 16 j + tx); per step p only row p travels through LDS (written by its 16
// owners, read by everybody) and the rank-1 update is 64 predicated FMAs on registers.  The upper-triangle slots
// hold the Cholesky work matrix; the otherwise unused LOWER-triangle slots hold W of the forward substitution
// U^T T = I done in the same right-looking form (T[p,:] = W[p,:] / U[p][p], then W[k,:] -= U[p][k] T[p,:] for
// k > p): element (r, c), r > p, is updated by -U[p][r] * v[c] with v = U[p][c] in the upper triangle (c >= r) and
// v = T[p][c] in the lower one (c <= p) -- one loop, one row buffer.  T = U^-T comes out in the lower triangle and
// is written transposed.  (A first version kept the block in LDS and updated it in place: 300 us per block of LDS
// read-modify-write latency; per-column triangular solves instead of the inverse: 145 us per call, each column a
// serial chain of 8192 f64 FMAs.)  A non-positive pivot is replaced by 1 and flagged through negative diagonal
// entries so that the host sees it in the diagonal it reads anyway.
__global__ void __launch_bounds__(256) rr_chol_diag_kernel(double *__restrict__ A, int64_t ld, double *__restrict__ Uinv) {
    __shared__ double rowbuf[2][PB];
    __shared__ int bad;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    if (tid == 0) bad = 0;
    double a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * i + ty, c = 16 * j + tx;
            a[i][j] = c >= r ? A[(int64_t)r * ld + c] : 0.0;  // lower slots: W = strictly lower part of I
        }
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) {
        for (int pk = 0; pk < 16; ++pk) {
            const int p = 16 * pi + pk;
            double *rb = rowbuf[p & 1];  // double-buffered: one barrier per step is enough
            if (ty == pk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) rb[16 * j + tx] = a[pi][j];  // U part (c > p), pivot, W part (c < p)
            }
            __syncthreads();
            double app = rb[p];
            if (!(app > 0.0) || !isfinite(app)) {  // uniform
                if (tid == 0) bad = 1;
                app = 1.0;
            }
            const double dp = sqrt(app), inv = 1.0 / dp;
            double ur[8], vc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ur[i] = rb[16 * i + ty] * inv;  // U[p][r] for this thread's rows r > p
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * j + tx;
                vc[j] = c == p ? inv : rb[c] * inv;  // U[p][c] (c > p), T[p][p], T[p][c] (c < p)
            }
            if (ty == pk) {  // row p becomes final: U[p][c], and T[p][c] in the lower slots
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * j + tx;
                    a[pi][j] = c == p ? dp : vc[j];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < pi) continue;  // rows 16 i + ty <= p
                const int r = 16 * i + ty;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * j + tx;
                    if (r > p && (c >= r || c <= p)) a[i][j] = fma(-ur[i], vc[j], a[i][j]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * i + ty, c = 16 * j + tx;
            double u = c >= r ? a[i][j] : 0.0;
            if (bad && r == c) u = -1.0;
            A[(int64_t)r * ld + c] = u;
            // Uinv = T^T: T[r][c] sits in the lower slots (c < r), T[r][r] = 1 / U[r][r]
            if (c < r) Uinv[c * PB + r] = a[i][j];
            else Uinv[c * PB + r] = (c == r) ? 1.0 / a[i][j] : 0.0;
        }
}

// sqrt(x) and 1 / sqrt(x) of a positive, normal x from ONE v_rsq_f64 and two coupled Newton (Goldschmidt) steps, each
// result with a final correction (both within an ulp or two): half the dependent chain of sqrt() followed by a division.
__device__ __forceinline__ void rr_sqrt_and_rsqrt(double x, double &s, double &inv) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    double t = h + h;
    t = fma(t, fma(-g, t, 1.0), t);
    s = g;
    inv = t;
}

// One step q = 16 QI + qk of rr_chol_diag_pipe_kernel (below).  LAST: qk == 15, the next pivot row lives in slot QI + 1.
template <int QI, bool LAST>
__device__ __forceinline__ void rr_chol_pipe_step(double (&a)[8][8], double (&ur)[8], double (&vc)[8], double (*rowbuf)[PB],
                                                  const int qk, const int ty, const int tx, int &badflag) {
    const int q = 16 * QI + qk;
    __syncthreads();  // row q is in rowbuf[q & 1]; everybody is done with rowbuf[(q + 1) & 1]
    const double *rb = rowbuf[q & 1];
    double app = rb[q];
    double rr[8], rc[8];
#pragma unroll
    for (int i = QI; i < 8; ++i) rr[i] = rb[16 * i + ty];
#pragma unroll
    for (int j = 0; j < 8; ++j) rc[j] = rb[16 * j + tx];
    // ---- the rest of step q - 1: slots below the pivot row's (every row of them is > q - 1); its own slot was updated
    // ---- before row q was published.  Columns: c <= q - 1 (W part) or c >= r (U part), nothing in between.
    {
        const double vq = tx < qk ? vc[QI] : 0.0;
#pragma unroll
        for (int i = QI + 1; i < 8; ++i) {
            const double vd = tx >= ty ? vc[i] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j > QI && j < i) continue;
                const double v = j == QI ? vq : (j == i ? vd : vc[j]);
                a[i][j] = fma(-ur[i], v, a[i][j]);
            }
        }
    }
    // ---- pivot q
    const bool ok = app > 0.0 && app < INFINITY;  // uniform (every thread reads the same value)
    badflag |= ok ? 0 : 1;
    app = ok ? app : 1.0;
    double dp, inv;
    rr_sqrt_and_rsqrt(app, dp, inv);
    double urn[8], vcn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) urn[i] = i >= QI ? rr[i] * inv : 0.0;  // U[q][r] for this thread's rows
#pragma unroll
    for (int j = 0; j < 8; ++j) vcn[j] = 16 * j + tx == q ? inv : rc[j] * inv;  // U[q][c] (c > q), T[q][q], T[q][c] (c < q)
    if (ty == qk) {  // row q becomes final
#pragma unroll
        for (int j = 0; j < 8; ++j) a[QI][j] = 16 * j + tx == q ? dp : vcn[j];
    }
    // ---- step q on the slot that holds row q + 1, and row q + 1 on its way to the others
    if constexpr (!LAST) {
        const double ue = ty > qk ? urn[QI] : 0.0;
        const double vm = (tx <= qk || tx >= ty) ? vcn[QI] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[QI][j] = fma(-ue, j == QI ? vm : vcn[j], a[QI][j]);
        if (ty == qk + 1) {
            double *wb = rowbuf[(q + 1) & 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) wb[16 * j + tx] = a[QI][j];
        }
    } else if constexpr (QI < 7) {
        const double vd = tx >= ty ? vcn[QI + 1] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[QI + 1][j] = fma(-urn[QI + 1], j == QI + 1 ? vd : vcn[j], a[QI + 1][j]);
        if (ty == 0) {
            double *wb = rowbuf[(q + 1) & 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) wb[16 * j + tx] = a[QI + 1][j];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ur[i] = urn[i];
        vc[i] = vcn[i];
    }
}

// rr_chol_diag_kernel with its steps software-pipelined (round 3, last session).  The kernel above is bound by the
// LATENCY of a step's dependent chain -- row p to LDS, barrier, LDS reads, sqrt and division, scaling -- with one wave
// per SIMD and nothing to overlap it: ~1650 cycles per step for ~500 cycles of issue.  Here step q's rank-1 update is
// split: the slot (8 register elements per thread) that holds row q + 1 is updated FIRST and row q + 1 published, and
// the update of the other slots runs in step q + 1 under its LDS reads and pivot chain.  The per-element predicates
// become compile-time column ranges plus three selected row vectors per step (no v_cndmask pair per FMA, no FMAs on
// the columns strictly between the pivot and the row), and sqrt + division one v_rsq_f64 with coupled Newton steps.
// Same layout, same outputs (the factor to an ulp or two of the kernel above).  tools/chol_diag_emu.py is the NumPy
// model of this schedule.
__global__ void __launch_bounds__(256) rr_chol_diag_pipe_kernel(double *__restrict__ A, int64_t ld, double *__restrict__ Uinv) {
    __shared__ double rowbuf[2][PB];
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    double a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * i + ty, c = 16 * j + tx;
            a[i][j] = c >= r ? A[(int64_t)r * ld + c] : 0.0;  // lower slots: W = strictly lower part of I
        }
    if (ty == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rowbuf[0][16 * j + tx] = a[0][j];
    }
    double ur[8], vc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ur[i] = vc[i] = 0.0;
    int badflag = 0;
#define RR_CHOL_BLOCK(QI)                                                                              \
    for (int qk = 0; qk < 15; ++qk) rr_chol_pipe_step<QI, false>(a, ur, vc, rowbuf, qk, ty, tx, badflag); \
    rr_chol_pipe_step<QI, true>(a, ur, vc, rowbuf, 15, ty, tx, badflag);
    RR_CHOL_BLOCK(0) RR_CHOL_BLOCK(1) RR_CHOL_BLOCK(2) RR_CHOL_BLOCK(3)
    RR_CHOL_BLOCK(4) RR_CHOL_BLOCK(5) RR_CHOL_BLOCK(6) RR_CHOL_BLOCK(7)
#undef RR_CHOL_BLOCK
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * i + ty, c = 16 * j + tx;
            double u = c >= r ? a[i][j] : 0.0;
            if (badflag && r == c) u = -1.0;
            A[(int64_t)r * ld + c] = u;
            if (c < r) Uinv[c * PB + r] = a[i][j];
            else Uinv[c * PB + r] = (c == r) ? 1.0 / a[i][j] : 0.0;
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same block, factored in 16-column SUB-PANELS with the f64 matrix cores (round 5).  The two kernels above spend ~1000
// cycles per column: every one of the 128 pivots is a workgroup-wide round trip (row to LDS, barrier, LDS reads, rsqrt,
// up to 64 FMAs per thread).  Here the work matrix lives in LDS (128 x 130 f64) and, per sub-panel kb of 16 columns:
//   P1  wave 0      R = chol(S[kb][kb]) and T_kb = R^-T, pivot by pivot, the 16 x 16 block in the MFMA accumulator layout
//                   (chol16_pivots: ~340 cycles per pivot, the only serial part)
//   P2  all waves   S[kb][jb] <- R^-T S[kb][jb], jb > kb                    4 v_mfma_f64_16x16x4 per 16 x 16 block
//   P3  wave 0      S[kb+1][kb+1] -= S[kb][kb+1]^T S[kb][kb+1], then straight on to P1 of sub-panel kb + 1 -- while
//       waves 1-3   the other trailing blocks S[ib][jb] -= S[kb][ib]^T S[kb][jb] and block row kb of T = U^-T by forward
//                   substitution, T[kb][jb] = -T_kb sum_{i = jb .. kb-1} U[i][kb]^T T[i][jb] (T goes into the unused lower blocks)
// so that the MFMA work of a sub-panel hides under the next sub-panel's pivots.  Outputs as rr_chol_diag_kernel's (factor to
// rounding; a non-positive pivot is replaced by 1 and flagged through -1 on the diagonal; Uinv = T^T).
// Measured (box K, 128 launches per F = 4096 posterior): 61 us (pipelined kernel above) -> 40 us with P1 - P3 in sequence
// -> see docs/KERNELS.md 3.24 for the overlapped form.  RR_CHOL_DIAG=1: the pipelined kernel above, =0: the plain one.
// ---------------------------------------------------------------------------------------------------------------------
typedef double doublex4_ __attribute__((ext_vector_type(4)));
constexpr int CB = 16;    // sub-panel width
constexpr int SLD = 130;  // LDS row stride of the work matrix (f64): column-direction operand reads stay off one bank
constexpr int RLD = 18;   // row stride of a 16 x 16 diagonal T block / of a wave's scratch block

__device__ __forceinline__ double rr_readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// P1 (one wave): the 16 x 16 diagonal sub-block kb in the MFMA accumulator layout -- lane (g, c) holds rows g, g + 4, g + 8,
// g + 12 of column c; U: the Cholesky work block (upper part meaningful), Wb: W of the substitution R^T T = I (lower part; T =
// R^-T comes out in it).  Four pivots at a time: pivot p = 4 s + q sits in register s of lane group q.  Per pivot only the rest
// of its own 4-row strip is updated (one FMA per block; the row's values travel by ds_bpermute, the multipliers by
// v_readlane); the rows below get all four pivots at once: the registers that hold the strip ARE the A and B operands of a
// 16x16x4 MFMA -- a rank-4 update without moving anything.  (First version: lane = column, 16 rows in registers, every
// multiplier by v_readlane: 590 cycles per pivot.)
__device__ __forceinline__ void chol16_pivots(double *S, double *Rv, int *flag, const int kb, const int lane) {
    const int c = lane & 15, g = lane >> 4, k0 = kb * CB;
    doublex4_ U, Wb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < 4; ++e) U[e] = S[(k0 + g + 4 * e) * SLD + k0 + c];
    int bad = 0;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_) {
            const int p = 4 * s_ + q_;
            double app = rr_readlane_f64(U[s_], 16 * q_ + p);
            const bool ok = app > 0.0 && app < INFINITY;
            bad |= ok ? 0 : 1;
            app = ok ? app : 1.0;
            double dp, inv;
            rr_sqrt_and_rsqrt(app, dp, inv);
            const bool mine = g == q_;
            const double us = U[s_] * inv, ws = Wb[s_] * inv;
            U[s_] = mine ? (c == p ? dp : us) : U[s_];       // row p: U[p][c] (c > p), the diagonal entry
            Wb[s_] = mine ? (c == p ? inv : ws) : Wb[s_];    // T[p][c] (c < p), T[p][p]; zero beyond (W is lower triangular)
            if (q_ < 3) {  // the rest of the strip: rows 4 s + q' (q' > q), one per lane group
                const double urow = __shfl(U[s_], 16 * q_ + c, 64), wrow = __shfl(Wb[s_], 16 * q_ + c, 64);
                double ur = 0.0;
#pragma unroll
                for (int gr = q_ + 1; gr < 4; ++gr) {
                    const double u1 = rr_readlane_f64(U[s_], 16 * q_ + 4 * s_ + gr);  // U[p][4 s + gr]
                    ur = g == gr ? u1 : ur;
                }
                U[s_] = fma(-ur, urow, U[s_]);    // (groups <= q: ur = 0)
                Wb[s_] = fma(-ur, wrow, Wb[s_]);
            }
        }
        if (s_ < 3) {  // rows 4 s + 4 .. 15: all four pivots of the strip at once
            const double an = c >= 4 * s_ + 4 ? -U[s_] : 0.0;  // A[i = c][k = g] = -U[4 s + g][c], rows above untouched
            U = __builtin_amdgcn_mfma_f64_16x16x4f64(an, U[s_], U, 0, 0, 0);
            Wb = __builtin_amdgcn_mfma_f64_16x16x4f64(an, Wb[s_], Wb, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = g + 4 * e;
        S[(k0 + r) * SLD + k0 + c] = r <= c ? U[e] : 0.0;
        Rv[(k0 + r) * RLD + c] = r >= c ? Wb[e] : 0.0;  // T_kb = R^-T (lower triangular), row-major
    }
    if (bad && lane == 0) *flag = 1;
}

// S[ib][jb] -= S[kb][ib]^T S[kb][jb]  (one wave)
__device__ __forceinline__ void chol16_update(double *S, const int kb, const int ib, const int jb, const int lane) {
    const int c = lane & 15, g = lane >> 4, k0 = kb * CB;
    doublex4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double a = S[(k0 + g + 4 * t) * SLD + ib * CB + c];
        const double b = S[(k0 + g + 4 * t) * SLD + jb * CB + c];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) S[(ib * CB + g + 4 * e) * SLD + jb * CB + c] -= acc[e];
}

// T[kb][jb] = -T_kb sum_{i = jb .. kb-1} U[i][kb]^T T[i][jb]  into the lower block (kb, jb) of S  (one wave; T[jb][jb] = T_jb in Rv)
__device__ __forceinline__ void chol16_trow(double *S, const double *Rv, double *T, const int kb, const int jb, const int lane) {
    const int c = lane & 15, g = lane >> 4;
    doublex4_ acc = {0.0, 0.0, 0.0, 0.0};
    for (int i = jb; i < kb; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double a = S[(i * CB + g + 4 * t) * SLD + kb * CB + c];                          // U[i][kb][k][a = c]
            const double b = i == jb ? Rv[(jb * CB + g + 4 * t) * RLD + c] : S[(i * CB + g + 4 * t) * SLD + jb * CB + c];  // T[i][jb][k][b = c]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) T[(g + 4 * e) * RLD + c] = acc[e];
    doublex4_ acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double a = Rv[(kb * CB + c) * RLD + g + 4 * t];  // T_kb[a = c][k]
        const double b = T[(g + 4 * t) * RLD + c];
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) S[(kb * CB + g + 4 * e) * SLD + jb * CB + c] = -acc2[e];
}

// the kernel's body on LDS the caller provides (S: PB * SLD doubles, Rv: 8 * CB * RLD, Tw: 4 * CB * RLD, flag): a workgroup of
// 256 threads; also workgroup 0's part of a panel step inside rr_posterior_coop_kernel
__device__ __forceinline__ void chol_diag_mfma_body(double *__restrict__ A, int64_t ld, double *__restrict__ Uinv, double *S, double *Rv,
                                                    double *Tw, int *flagp) {
    int &flag = *flagp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    {   // thread -> column tid & 127, rows (tid >> 7) + 2 it: all 64 loads of a thread in flight together (as a rolled loop
        // every iteration waits a global-memory round trip: the first version of this kernel spent 50 us here)
        const int cc = tid & 127, r0 = tid >> 7;
        double v[64];
#pragma unroll
        for (int it = 0; it < 64; ++it) v[it] = A[(int64_t)(r0 + 2 * it) * ld + cc];
#pragma unroll
        for (int it = 0; it < 64; ++it) S[(r0 + 2 * it) * SLD + cc] = cc >= r0 + 2 * it ? v[it] : 0.0;
    }
    if (tid == 0) flag = 0;
    __syncthreads();
    for (int kb = -1; kb < 8; ++kb) {
        if (kb >= 0) {
            // ---- P2: block row kb right of the diagonal: S[kb][jb] <- R^-T S[kb][jb] = T_kb S[kb][jb]
            const int k0 = kb * CB;
            for (int jb = kb + 1 + wave; jb < 8; jb += 4) {
                doublex4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double a = Rv[(k0 + c) * RLD + g + 4 * t];              // (R^-1)[k][i = c] = T_kb[i][k]
                    const double b = S[(k0 + g + 4 * t) * SLD + jb * CB + c];     // S[kb][jb][k][j = c]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) S[(k0 + g + 4 * e) * SLD + jb * CB + c] = acc[e];
            }
            __syncthreads();
        }
        // ---- P3 of sub-panel kb next to P1 of sub-panel kb + 1
        const bool chain = wave == 0 && kb < 7;  // wave 0: the next diagonal block first, then its pivots
        if (chain) {
            if (kb >= 0) chol16_update(S, kb, kb + 1, kb + 1, lane);
            chol16_pivots(S, Rv, &flag, kb + 1, lane);
        }
        if (kb >= 0 && !chain) {
            // the other waves (all four behind the last sub-panel): block row kb of T, longest sums first, then the trailing blocks
            const int nw = kb < 7 ? 3 : 4, me = kb < 7 ? wave - 1 : wave;
            int q = 0;
            for (int jb = 0; jb < kb; ++jb, ++q)
                if (q % nw == me) chol16_trow(S, Rv, Tw + wave * CB * RLD, kb, jb, lane);
            for (int ib = kb + 1; ib < 8; ++ib)
                for (int jb = ib; jb < 8; ++jb) {
                    if (ib == kb + 1 && jb == kb + 1) continue;  // (wave 0's)
                    if (q++ % nw == me) chol16_update(S, kb, ib, jb, lane);
                }
        }
        __syncthreads();
    }
    const int isbad = flag;
    {   // U (upper) back to A; Uinv = T^T: Uinv[r][cc] = T[cc][r]
        const int cc = tid & 127, r0 = tid >> 7, bj = cc >> 4;
#pragma unroll 16
        for (int it = 0; it < 64; ++it) {
            const int r = r0 + 2 * it, bi = r >> 4;
            double u = cc >= r ? S[r * SLD + cc] : 0.0;
            if (isbad && r == cc) u = -1.0;
            A[(int64_t)r * ld + cc] = u;
            double xi = 0.0;
            if (bi < bj) xi = S[cc * SLD + r];
            else if (bi == bj) xi = Rv[(bi * CB + (cc & 15)) * RLD + (r & 15)];
            Uinv[r * PB + cc] = xi;
        }
    }
}

__global__ void __launch_bounds__(256) rr_chol_diag_mfma_kernel(double *__restrict__ A, int64_t ld, double *__restrict__ Uinv) {
    __shared__ double S[PB * SLD];          // 133 120 B
    __shared__ double Rv[8 * CB * RLD];     //  18 432 B: T_kb = R_kb^-T, row-major 16 x 16 each
    __shared__ double Tw[4 * CB * RLD];     //   9 216 B: one scratch block per wave
    __shared__ int flag;
    chol_diag_mfma_body(A, ld, Uinv, S, Rv, Tw, &flag);
}

// RR_CHOL_DIAG=0: the unpipelined kernel, =1: the pipelined one (A/B runs); default: the sub-panel / MFMA kernel
static void launch_chol_diag(hipStream_t stream, double *Ujj, int64_t ld, double *Uij) {
    static const int which = getenv("RR_CHOL_DIAG") != nullptr ? atoi(getenv("RR_CHOL_DIAG")) : 2;
    if (which == 0)
        hipLaunchKernelGGL(rr_chol_diag_kernel, dim3(1), dim3(256), 0, stream, Ujj, ld, Uij);
    else if (which == 1)
        hipLaunchKernelGGL(rr_chol_diag_pipe_kernel, dim3(1), dim3(256), 0, stream, Ujj, ld, Uij);
    else
        hipLaunchKernelGGL(rr_chol_diag_mfma_kernel, dim3(1), dim3(256), 0, stream, Ujj, ld, Uij);
}

// W (Fp, Fp) = J C J in the top-left F x F (both index orders reversed), identity on the pad diagonal
__global__ void __launch_bounds__(256)
rr_reverse_pad_kernel(const double *__restrict__ C, int64_t F, int64_t Fp, double *__restrict__ W) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Fp * Fp) return;
    const int64_t r = i / Fp, c = i % Fp;
    W[i] = (r < F && c < F) ? C[(F - 1 - r) * F + (F - 1 - c)] : (r == c ? 1.0 : 0.0);
}

// B (Fb, Fb) f32 = M in the top-left F x F, zero elsewhere, where J C J = R^T R (R upper, in W) and
// M[k][j] = R[F-1-j][F-1-k] (upper triangular): C = M M^T, so phi^T C phi = || phi^T M ||^2.
__global__ void __launch_bounds__(256)
rr_ul_factor_f32_kernel(const double *__restrict__ W, int64_t F, int64_t Fp, float *__restrict__ B, int64_t Fb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Fb * Fb) return;
    const int64_t k = i / Fb, j = i % Fb;
    B[i] = (k <= j && j < F) ? (float)W[(F - 1 - j) * Fp + (F - 1 - k)] : 0.f;
}

// one wave per row r:  m[r] = (C[r,:] . b) / var,  tr += C[r,:] . G[r,:],  dg[r] = C[r][r]
__global__ void __launch_bounds__(256)
rr_posterior_rows_kernel(const double *__restrict__ C, const double *__restrict__ G, const double *__restrict__ b,
                         double ivar, int64_t F, double *__restrict__ m, double *__restrict__ dg, double *__restrict__ tr,
                         int64_t det = 0) {
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
        rr_acc_out(tr, det, r, 0, at);  // deterministic mode: one slot per row, added in row order afterwards
    }
}

// =============================================================================================
// The posterior at SMALL feature counts (F <= 1024; round 6) as ONE cooperative kernel -- opt-in, see rr_posterior_dev.
// BASELINE config 1 (F = 512), the reference's SARCOS model (nbases = 512) and everything below sit where the panel
// pipeline above is nothing but dependent launches: 8 diagonal-block kernels + ~35 one-tile f64 products + their gaps
// = 1.2 ms of kernel time for 0.13 GFLOP (profiles/r06_c1_latency) -- `_elbo`'s critical path at config 1.  Here 32
// workgroups walk through the same algebra (W = U^T U, upper; 128-column panels) inside one launch, separated by
// device-scope barriers (as rr_svi.hip):
//   assemble  W = G / var + diag(iL) (identity on the padding)
//   per panel p:  S1 workgroup 0: the diagonal block on the matrix cores -- chol_diag_mfma_body, the SAME code as the
//                    pipeline's diagonal-block kernel (0.3 us per pivot) -> U_pp and U_pp^-1
//                 S2 block row p right of it: U_pj = U_pp^-T W_pj, a 64-column strip per workgroup
//                 S3 trailing 64 x 64 tiles: W_ij -= U_pi^T U_pj
//   Y = U^-T in 16-column slabs (no barrier inside: Y_ip = -U_ii^-T sum_{k=p}^{i-1} U_ki^T Y_kp), C = Y^T Y tile by tile,
//   written to both triangles.  3 P + 3 barriers for P = F / 128 panels; the products are plain float64 FMAs out of LDS
//   (64 x 64 x 64 per workgroup and turn) -- the flops (F^3) are nothing here.
// m, diag C, sum(G o C) follow in rr_posterior_rows_kernel as before; a pivot that is not positive leaves -1 in the
// factor's diagonal and the host reports RR_ERR_NOT_POSDEF exactly as the pipeline does.
// (First version, 64-column panels factored by an unblocked loop out of LDS: 0.8 us per pivot, 1.35 ms at F = 512.)
// =============================================================================================
constexpr int PS_NB = 64, PS_WG = 32, PS_T = 256, PS_LD = PS_NB + 1;

struct PsArgs {
    const double *G, *iL;
    double ivar;
    int F, Fp, P;      // P panels of PB = 128 columns
    double *A, *M, *C, *dvec, *Uinv;
    unsigned int *bar;
    long long *prof;  // RR_PS_PROF=1: workgroup 0's 100 MHz ticks per phase
};

__device__ __forceinline__ void ps_barrier(unsigned int *ctr, unsigned int target) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();
}

typedef double (*ps_tile)[PS_LD];

// LDS block loads of a 64 x 64 block of a row-major matrix: dst[r][c] = src[r][c] or dst[c][r] = src[r][c]
__device__ __forceinline__ void ps_load(ps_tile dst, const double *src, int64_t ld, bool transpose) {
    for (int e = threadIdx.x; e < PS_NB * PS_NB; e += PS_T) {
        const int r = e >> 6, c = e & 63;
        const double v = src[(int64_t)r * ld + c];
        if (transpose) dst[c][r] = v;
        else dst[r][c] = v;
    }
}

// acc[4][4] += sum_t As[t][4 tm + a] Bs[t][4 tn + b]  (thread (tm, tn) of a 16 x 16 grid: a 64 x 64 x 64 product per workgroup)
__device__ __forceinline__ void ps_mma(const double (*As)[PS_LD], const double (*Bs)[PS_LD], double (&acc)[4][4]) {
    const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
#pragma unroll 4
    for (int t = 0; t < PS_NB; ++t) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = As[t][4 * tm + u];
            b[u] = Bs[t][4 * tn + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
    }
}

// out (row tm, columns 4 tn .. 4 tn + 3 of a 64 x 16 slab) += sum_t As[t][tm] Bs[t][4 tn + v]   (thread = (tm = tid / 4, tn = tid % 4))
__device__ __forceinline__ void ps_mma16(const double (*As)[PS_LD], const double (*Bs)[PS_LD], int bcol0, double (&acc)[4]) {
    const int tm = threadIdx.x >> 2, tn = threadIdx.x & 3;
#pragma unroll 8
    for (int t = 0; t < PS_NB; ++t) {
        const double a = As[t][tm];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = fma(a, Bs[t][bcol0 + 4 * tn + v], acc[v]);
    }
}

__global__ void __launch_bounds__(PS_T) rr_posterior_coop_kernel(const PsArgs p) {
    // one LDS block: the diagonal-block body's (S | Rv | Tw | flag) or four 64 x 65 tiles of the products
    __shared__ __attribute__((aligned(16))) double lds[PB * SLD + 8 * CB * RLD + 4 * CB * RLD + 2];
    ps_tile L0 = (ps_tile)lds, L1 = (ps_tile)(lds + PS_NB * PS_LD), L2 = (ps_tile)(lds + 2 * PS_NB * PS_LD),
            L3 = (ps_tile)(lds + 3 * PS_NB * PS_LD);
    const int tid = threadIdx.x, wg = blockIdx.x, F = p.F, P = p.P;
    const int64_t ld = p.Fp;
    const int n64 = p.Fp / PS_NB;   // 64-column tiles per row
    long long tprev = p.prof ? wall_clock64() : 0, pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PS_MARK(i)                                                  \
    do {                                                            \
        if (p.prof && wg == 0 && tid == 0) {                        \
            const long long now_ = wall_clock64();                  \
            pacc[i] += now_ - tprev;                                \
            tprev = now_;                                           \
        }                                                           \
    } while (0)
    unsigned int nbar = 0;
    // ---- assemble (identity on the padding); Y = 0
    for (int64_t e = (int64_t)wg * PS_T + tid; e < ld * ld; e += (int64_t)PS_WG * PS_T) {
        const int64_t r = e / ld, c = e % ld;
        double v = 0.0;
        if (r < F && c < F) v = p.G[r * F + c] * p.ivar + (r == c ? p.iL[r] : 0.0);
        else if (r == c) v = 1.0;
        p.A[e] = v;
        p.M[e] = 0.0;
    }
    ps_barrier(p.bar, ++nbar * PS_WG);
    PS_MARK(0);
    for (int pp = 0; pp < P; ++pp) {
        double *App = p.A + (int64_t)pp * PB * (ld + 1);
        double *Ui = p.Uinv + (int64_t)pp * PB * PB;
        // ---- S1: the diagonal block, by workgroup 0, on the matrix cores
        if (wg == 0) {
            chol_diag_mfma_body(App, ld, Ui, lds, lds + PB * SLD, lds + PB * SLD + 8 * CB * RLD, (int *)(lds + PB * SLD + 12 * CB * RLD));
            __threadfence_block();
            __syncthreads();
            if (tid < PB) p.dvec[pp * PB + tid] = App[(int64_t)tid * (ld + 1)];
        }
        PS_MARK(1);
        ps_barrier(p.bar, ++nbar * PS_WG);
        PS_MARK(2);
        // ---- S2: U_p,strip = U_pp^-T W_p,strip for the 64-column strips right of the diagonal block: rows 0..63 need the
        // top-left quarter of U_pp^-1 only (it is upper triangular), rows 64..127 the two right quarters
        for (int ct = (pp + 1) * 2 + wg; ct < n64; ct += PS_WG) {
            double *strip = p.A + (int64_t)pp * PB * ld + (int64_t)ct * PS_NB;
            __syncthreads();
            ps_load(L0, strip, ld, false);                          // W rows 0..63   [t][n]
            ps_load(L1, strip + (int64_t)PS_NB * ld, ld, false);    // W rows 64..127 [t][n]
            ps_load(L2, Ui, PB, false);                             // Uinv[0:64][0:64]    [t][m]
            __syncthreads();
            double top[4][4] = {}, bot[4][4] = {};
            ps_mma(L2, L0, top);
            __syncthreads();
            ps_load(L2, Ui + PS_NB, PB, false);                     // Uinv[0:64][64:128]  [t][m]
            ps_load(L3, Ui + (int64_t)PS_NB * PB + PS_NB, PB, false);   // Uinv[64:128][64:128]
            __syncthreads();
            ps_mma(L2, L0, bot);
            ps_mma(L3, L1, bot);
            const int tm = tid >> 4, tn = tid & 15;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    strip[(int64_t)(4 * tm + u) * ld + 4 * tn + v] = top[u][v];
                    strip[(int64_t)(PS_NB + 4 * tm + u) * ld + 4 * tn + v] = bot[u][v];
                }
        }
        PS_MARK(3);
        ps_barrier(p.bar, ++nbar * PS_WG);
        PS_MARK(4);
        // ---- S3: trailing 64 x 64 tiles (i, j), (pp + 1) 2 <= i <= j: W_ij -= U_p,i^T U_p,j (128 rows of block row pp)
        {
            const int i0 = (pp + 1) * 2, n = n64 - i0, ntile = n * (n + 1) / 2;
            for (int t = wg; t < ntile; t += PS_WG) {
                int jr = 0, rem = t;
                while (rem > jr) {  // column jr of the triangle holds jr + 1 tiles (i <= j)
                    rem -= jr + 1;
                    ++jr;
                }
                const int bi = i0 + rem, bj = i0 + jr;
                const double *Ui_ = p.A + (int64_t)pp * PB * ld + (int64_t)bi * PS_NB, *Uj_ = p.A + (int64_t)pp * PB * ld + (int64_t)bj * PS_NB;
                __syncthreads();
                ps_load(L0, Ui_, ld, false);                          // [t][m], rows 0..63
                ps_load(L1, Uj_, ld, false);                          // [t][n]
                ps_load(L2, Ui_ + (int64_t)PS_NB * ld, ld, false);    // rows 64..127
                ps_load(L3, Uj_ + (int64_t)PS_NB * ld, ld, false);
                __syncthreads();
                double acc[4][4] = {};
                ps_mma(L0, L1, acc);
                ps_mma(L2, L3, acc);
                double *Wij = p.A + (int64_t)bi * PS_NB * ld + (int64_t)bj * PS_NB;
                const int tm = tid >> 4, tn = tid & 15;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) Wij[(int64_t)(4 * tm + u) * ld + 4 * tn + v] -= acc[u][v];
            }
        }
        PS_MARK(5);
        ps_barrier(p.bar, ++nbar * PS_WG);
        PS_MARK(6);
    }
    // ---- not safely positive definite? (every workgroup reads the same diagonal: the same decision everywhere)
    {
        double mn = INFINITY;
        for (int i = tid; i < F; i += PS_T) {
            const double d = p.dvec[i];
            mn = fmin(mn, (d > 0.0 && d == d) ? d : -1.0);
        }
        __syncthreads();
        L0[0][tid & 63] = INFINITY;
        __syncthreads();
        for (int w = 0; w < 4; ++w) {   // (four waves, one after the other: a min over 256 values without atomics)
            if ((tid >> 6) == w) L0[0][tid & 63] = fmin(L0[0][tid & 63], mn);
            __syncthreads();
        }
        if (tid == 0) {
            double m2 = INFINITY;
            for (int i = 0; i < 64; ++i) m2 = fmin(m2, L0[0][i]);
            L0[1][0] = m2;
        }
        __syncthreads();
        if (L0[1][0] < 1e-5) return;   // CHOLTHRESH, mathfun/linalg.py:31 (no barrier follows for anybody)
        __syncthreads();
    }
    // ---- Y = U^-T (lower): 16-column slabs of block column pb walk down the 64-row blocks below; Y_pp = (U_pp^-1)^T
    {
        const int tm = tid >> 2, tn = tid & 3;
        for (int item = wg; item < P * 8; item += PS_WG) {
            const int pb = item >> 3, q = item & 7, c0 = pb * PB + 16 * q;   // global columns c0 .. c0 + 15
            // rows of block pb: Y[pb PB + r][c0 + c] = Uinv_pb[16 q + c][r]  (transposed read; zero above the diagonal by itself)
            {
                const double *Up = p.Uinv + (int64_t)pb * PB * PB;
                for (int e = tid; e < PB * 16; e += PS_T) {
                    const int r = e >> 4, c = e & 15;
                    p.M[((int64_t)pb * PB + r) * ld + c0 + c] = Up[(int64_t)(16 * q + c) * PB + r];
                }
                __threadfence_block();
            }
            for (int i = pb + 1; i < P; ++i) {        // 128-row block i of the slab
                double acc0[4] = {0.0, 0.0, 0.0, 0.0}, acc1[4] = {0.0, 0.0, 0.0, 0.0};   // rows 0..63 / 64..127 of R = sum_k U_ki^T Y_k
                for (int kh = pb * 2; kh < i * 2; ++kh) {   // 64-row blocks kh of U's block column i and of the slab
                    __syncthreads();
                    ps_load(L0, p.A + (int64_t)kh * PS_NB * ld + (int64_t)i * PB, ld, false);            // U[kh][i, left half]  [t][m]
                    ps_load(L1, p.A + (int64_t)kh * PS_NB * ld + (int64_t)i * PB + PS_NB, ld, false);    // right half
                    for (int e = tid; e < PS_NB * 16; e += PS_T)
                        L2[e >> 4][e & 15] = p.M[((int64_t)kh * PS_NB + (e >> 4)) * ld + c0 + (e & 15)];  // Y rows of block kh  [t][c]
                    __syncthreads();
                    ps_mma16(L0, L2, 0, acc0);
                    ps_mma16(L1, L2, 0, acc1);
                }
                __syncthreads();
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    L2[tm][32 + 4 * tn + v] = acc0[v];          // R rows 0..63   in columns [32, 48)
                    L3[tm][32 + 4 * tn + v] = acc1[v];          // R rows 64..127
                }
                // Y_i = -U_ii^-T R = -Uinv_i^T R:  out[m][c] = -sum_t Uinv_i[t][m] R[t][c], t <= m (upper triangular)
                const double *Uii = p.Uinv + (int64_t)i * PB * PB;
                ps_load(L0, Uii, PB, false);                         // Uinv[0:64][0:64]
                ps_load(L1, Uii + PS_NB, PB, false);                 // Uinv[0:64][64:128]
                __syncthreads();
                double o0[4] = {0.0, 0.0, 0.0, 0.0}, o1[4] = {0.0, 0.0, 0.0, 0.0};
                ps_mma16(L0, L2, 32, o0);                            // rows 0..63:   t in [0, 64)
                ps_mma16(L1, L2, 32, o1);                            // rows 64..127: t in [0, 64) ...
                __syncthreads();
                ps_load(L0, Uii + (int64_t)PS_NB * PB + PS_NB, PB, false);   // Uinv[64:128][64:128]
                __syncthreads();
                ps_mma16(L0, L3, 32, o1);                            // ... and t in [64, 128)
                double *Yi = p.M + ((int64_t)i * PB) * ld + c0;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    Yi[(int64_t)tm * ld + 4 * tn + v] = -o0[v];
                    Yi[(int64_t)(PS_NB + tm) * ld + 4 * tn + v] = -o1[v];
                }
                __threadfence_block();   // (this workgroup reads the slab back in the next block's sums)
            }
        }
    }
    PS_MARK(7);
    ps_barrier(p.bar, ++nbar * PS_WG);
    PS_MARK(8);
    // ---- C = Y^T Y: 64 x 64 tile (a, b), b <= a: sum over 64-row blocks r >= a of Y_r,a^T Y_r,b; both triangles of the output
    {
        const int ntile = n64 * (n64 + 1) / 2;
        for (int t = wg; t < ntile; t += PS_WG) {
            int a = 0, rem = t;   // a ascending: the tiles with the longest sums first
            while (rem > a) {
                rem -= a + 1;
                ++a;
            }
            const int b = rem;
            double acc[4][4] = {};
            for (int r = a; r < n64; ++r) {
                __syncthreads();
                ps_load(L0, p.M + (int64_t)r * PS_NB * ld + (int64_t)a * PS_NB, ld, false);   // [t][m]
                ps_load(L1, p.M + (int64_t)r * PS_NB * ld + (int64_t)b * PS_NB, ld, false);   // [t][n]
                __syncthreads();
                ps_mma(L0, L1, acc);
            }
            const int tm = tid >> 4, tn = tid & 15;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t gr = (int64_t)a * PS_NB + 4 * tm + u, gc = (int64_t)b * PS_NB + 4 * tn + v;
                    if (gr < F && gc < F) {
                        p.C[gr * F + gc] = acc[u][v];
                        p.C[gc * F + gr] = acc[u][v];
                    }
                }
        }
    }
    PS_MARK(9);
    if (p.prof && wg == 0 && tid == 0)
        for (int i = 0; i < 10; ++i) p.prof[i] += pacc[i];
#undef PS_MARK
}

struct PosdefScratch {
    double *W = nullptr, *Y = nullptr, *Cp = nullptr;  // (Fp, Fp) each
    double *diL = nullptr, *dvec = nullptr;            // dvec: [chol diag (Fp) | m (F) | diagC (F) | tr (1)]
    double *Uinv = nullptr;                            // (Fp / 128) blocks of 128 x 128: U_jj^-1
    unsigned int *bar = nullptr;                       // barrier counter of rr_posterior_small_kernel
    int64_t Fp = 0;
    std::vector<hipEvent_t> ev;                        // "block row j of the factor is final" (+ one for the join), grow-only
    void release() {
        void *q[] = {W, Y, Cp, diL, dvec, Uinv, bar};
        for (void *x : q)
            if (x) (void)hipFree(x);
        W = Y = Cp = diL = dvec = Uinv = nullptr;
        bar = nullptr;
        Fp = 0;
    }
};

void rr_posdef_scratch_free(void *p) {
    if (!p) return;
    PosdefScratch *s = (PosdefScratch *)p;
    s->release();
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    delete s;
}

static int posdef_scratch(rr_ctx *c, int64_t Fp, int64_t F, const char *who) {
    if (!c->posdef) c->posdef = new PosdefScratch();
    PosdefScratch &s = *(PosdefScratch *)c->posdef;
    if (s.Fp != Fp) {  // sized exactly: the matrices are dense (Fp, Fp)
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        s.release();
        hipError_t ea = hipMalloc((void **)&s.W, (size_t)Fp * Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Y, (size_t)Fp * Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Cp, (size_t)Fp * Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.diL, (size_t)Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.dvec, (size_t)(3 * Fp + 1) * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Uinv, (size_t)Fp * PB * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.bar, 64);
        if (ea != hipSuccess) {
            (void)hipGetLastError();
            s.release();
            rr_set_error("%s: device allocation failed (F = %lld)", who, (long long)F);
            return RR_ERR_OOM;
        }
        s.Fp = Fp;
    }
    return RR_OK;
}

// W = U^T U in place (upper triangle of the (Fp, Fp) matrix s.W); Uinv_j = U_jj^-1 on the side
static int chol_upper_blocked(rr_ctx *c, PosdefScratch &s, int64_t Fp) {
    const int64_t ld = Fp, nblk = Fp / PB;
    int rc = RR_OK;
    for (int64_t j = 0; j < nblk && rc == RR_OK; ++j) {
        double *Ujj = s.W + j * PB * (ld + 1);
        double *Uij = s.Uinv + j * PB * PB;
        launch_chol_diag(c->stream, Ujj, ld, Uij);
        const int64_t rest = Fp - (j + 1) * PB;
        if (rest > 0) {
            double *panel = Ujj + PB;  // block row j, columns right of the diagonal block
            rc = rr_launch_gemm_tn_f64(c, Uij, PB, panel, ld, panel, ld, PB, PB, rest, 0, 0);  // panel <- U_jj^-T panel
            if (rc == RR_OK) rc = rr_launch_gemm_tn_f64(c, panel, ld, panel, ld, Ujj + PB * (ld + 1), ld, PB, rest, rest, 1, 1);
        }
    }
    return rc;
}

void rr_launch_c64_to_c32_tri(rr_ctx *c, const double *dC, int64_t F, float *dB, int64_t Fb);  // rr_elbo.hip

extern "C" {

int rr_posterior_available(void) { return 1; }

int rr_variance_factor_dev(rr_ctx *c, int64_t F, const double *dC, float *dB, int *form) {
    RR_REQUIRE(c != nullptr && dC != nullptr && dB != nullptr && form != nullptr, "rr_variance_factor_dev: null argument");
    RR_REQUIRE(F >= 1 && F < 46340, "rr_variance_factor_dev: bad F");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int64_t Fp = (F + PB - 1) / PB * PB, Fb = (F + 255) / 256 * 256;
    int rc = posdef_scratch(c, Fp, F, "rr_variance_factor_dev");
    if (rc != RR_OK) return rc;
    PosdefScratch &s = *(PosdefScratch *)c->posdef;
    hipLaunchKernelGGL(rr_reverse_pad_kernel, dim3((unsigned)((Fp * Fp + 255) / 256)), dim3(256), 0, c->stream, dC, F, Fp, s.W);
    RR_CHECK_HIP(hipGetLastError());
    rc = chol_upper_blocked(c, s, Fp);
    if (rc != RR_OK) return rc;
    hipLaunchKernelGGL(rr_get_diag_kernel, dim3((unsigned)((Fp + 255) / 256)), dim3(256), 0, c->stream, s.W, Fp, Fp, s.dvec);
    RR_CHECK_HIP(hipGetLastError());
    std::vector<double> h((size_t)Fp);
    RR_CHECK_HIP(hipMemcpyAsync(h.data(), s.dvec, (size_t)Fp * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    double lo = INFINITY, hi = 0.0;
    for (int64_t i = 0; i < F; ++i) {
        const double d = h[i];
        if (!(d > 0.0) || !std::isfinite(d)) {
            lo = -1.0;
            break;
        }
        lo = d < lo ? d : lo;
        hi = d > hi ? d : hi;
    }
    // a factor whose diagonal spans more than 1e7 is not worth its float32 copy: keep the quadratic form then
    if (lo > 0.0 && lo > 1e-7 * hi) {
        hipLaunchKernelGGL(rr_ul_factor_f32_kernel, dim3((unsigned)((Fb * Fb + 255) / 256)), dim3(256), 0, c->stream, s.W, F, Fp,
                           dB, Fb);
        RR_CHECK_HIP(hipGetLastError());
        *form = 1;
    } else {
        rr_launch_c64_to_c32_tri(c, dC, F, dB, Fb);
        RR_CHECK_HIP(hipGetLastError());
        *form = 0;
    }
    return RR_OK;
}

int rr_posterior_dev(rr_ctx *c, int64_t F, const double *dG, const double *db, const double *iL, double var, double *dC,
                     double *m, double *diagC, double *scal) {
    RR_REQUIRE(c != nullptr && dG != nullptr && db != nullptr && iL != nullptr && dC != nullptr && m != nullptr &&
                   diagC != nullptr && scal != nullptr,
               "rr_posterior_dev: null argument");
    RR_REQUIRE(F >= 1 && F < 46340 && var > 0.0 && std::isfinite(var), "rr_posterior_dev: bad F or var");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int64_t Fp = (F + PB - 1) / PB * PB, nblk = Fp / PB;
    {
        int rc0 = posdef_scratch(c, Fp, F, "rr_posterior_dev");
        if (rc0 != RR_OK) return rc0;
    }
    PosdefScratch &s = *(PosdefScratch *)c->posdef;
    const int64_t ld = Fp;
    const double ivar = 1.0 / var;
    RR_CHECK_HIP(hipMemcpyAsync(s.diL, iL, (size_t)F * 8, hipMemcpyHostToDevice, c->stream));
    // small feature counts: the whole factorisation, inverse and C in ONE cooperative launch -- OPT-IN (RR_POSDEF_SMALL=1).
    // Measured at config 1 (F = 512) and not adopted: 0.82 ms per call (S1 174 us, S2 138, S3 83, Y 192, C 117, barriers
    // 100) against ~0.46 ms for the panel pipeline's share of `_elbo`: the pipeline's three streams overlap what this kernel
    // runs in sequence.  Kept for the next attempt and held to the oracle by tests/test_gpu_posterior.py.
    const char *small_env = getenv("RR_POSDEF_SMALL");
    if (F <= 1024 && small_env != nullptr && atoi(small_env) != 0 && c->num_cu >= PS_WG) {
        PsArgs a;
        a.G = dG; a.iL = s.diL; a.ivar = ivar; a.F = (int)F; a.Fp = (int)Fp; a.P = (int)(Fp / PB);
        a.A = s.W; a.M = s.Y; a.C = dC; a.dvec = s.dvec; a.bar = s.bar; a.Uinv = s.Uinv;
        static long long *dprof = nullptr;
        static int prof_calls = 0;
        if (getenv("RR_PS_PROF") && !dprof && hipMalloc((void **)&dprof, 80) == hipSuccess) (void)hipMemset(dprof, 0, 80);
        a.prof = dprof;
        if (dprof && ++prof_calls % 50 == 0) {
            long long h[10];
            (void)hipMemcpy(h, dprof, 80, hipMemcpyDeviceToHost);
            static const char *nm[] = {"assemble", "S1", "bar1", "S2", "bar2", "S3", "bar3", "Y", "bar", "C"};
            fprintf(stderr, "rr_posterior_coop (us per call over %d calls):", prof_calls - 1);
            for (int i = 0; i < 10; ++i) fprintf(stderr, " %s=%.1f", nm[i], 0.01 * (double)h[i] / (prof_calls - 1));
            fprintf(stderr, "\n");
        }
        RR_CHECK_HIP(hipMemsetAsync(s.bar, 0, 64, c->stream));
        hipLaunchKernelGGL(rr_posterior_coop_kernel, dim3(PS_WG), dim3(PS_T), 0, c->stream, a);
        RR_CHECK_HIP(hipGetLastError());
        double *dm = s.dvec + Fp, *ddg = s.dvec + 2 * Fp, *dtr = s.dvec + 3 * Fp;
        RR_CHECK_HIP(hipMemsetAsync(dtr, 0, 8, c->stream));
        int rc = RR_OK;
        if (c->deterministic) {
            void *part = nullptr;
            rc = rr_det_scratch(c, (size_t)F * 8, &part);
            if (rc != RR_OK) return rc;
            hipLaunchKernelGGL(rr_posterior_rows_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, c->stream, dC, dG, db, ivar, F,
                               dm, ddg, (double *)part, (int64_t)1);
            rc = rr_det_reduce(c, (const double *)part, F, 1, 1, dtr);
            if (rc != RR_OK) return rc;
        } else {
            hipLaunchKernelGGL(rr_posterior_rows_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, c->stream, dC, dG, db, ivar, F,
                               dm, ddg, dtr);
        }
        RR_CHECK_HIP(hipGetLastError());
        std::vector<double> h((size_t)3 * Fp + 1);
        RR_CHECK_HIP(hipMemcpyAsync(h.data(), s.dvec, (size_t)(3 * Fp + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        double logdet = 0.0, mind = INFINITY;
        for (int64_t i = 0; i < F; ++i) {
            const double dval = h[(size_t)i];
            if (!(dval > 0.0) || !std::isfinite(dval)) {
                mind = -1.0;
                break;
            }
            logdet += 2.0 * std::log(dval);
            if (dval < mind) mind = dval;
        }
        scal[0] = logdet;
        scal[2] = mind;
        if (mind < 1e-5) {  // CHOLTHRESH, mathfun/linalg.py:31 (the kernel stopped behind the factorisation: C, m are not formed)
            rr_set_error("rr_posterior_dev: matrix is not safely positive definite (min diag of the factor %g)", mind);
            return RR_ERR_NOT_POSDEF;
        }
        memcpy(m, h.data() + Fp, (size_t)F * 8);
        memcpy(diagC, h.data() + 2 * Fp, (size_t)F * 8);
        scal[1] = h[(size_t)3 * Fp];
        return RR_OK;
    }
    const unsigned eb = (unsigned)((Fp * Fp + 255) / 256);
    hipLaunchKernelGGL(rr_assemble_ic_kernel, dim3(eb), dim3(256), 0, c->stream, dG, s.diL, ivar, F, Fp, s.W);
    hipLaunchKernelGGL(rr_set_identity_kernel, dim3(eb), dim3(256), 0, c->stream, s.Y, Fp);
    RR_CHECK_HIP(hipGetLastError());
    // ---- factor: W = U^T U (upper triangle of W), Uinv_j = U_jj^-1 on the side, and -- interleaved panel by panel --
    // ---- Y = U^-T by block forward substitution on the identity (Y lower triangular).
    // Step j of the substitution needs U_jj^-1 and the finished block row j of the factor only, not the trailing update
    // that follows it: it runs on the context's second stream while the factorisation goes on to panel j + 1 (round 3).
    // Both chains are sequences of small launches (a 128-column panel is 1-32 tiles), so running them side by side
    // shortens the call from 2 x 32 to ~32 dependent panel steps.  RR_POSDEF_OVERLAP=0: one stream, factor first (A/B).
    static const bool no_overlap = getenv("RR_POSDEF_OVERLAP") != nullptr && atoi(getenv("RR_POSDEF_OVERLAP")) == 0;
    const bool overlap = !no_overlap && nblk > 1;
    if (overlap && !c->stream2) {
        RR_CHECK_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        RR_CHECK_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    }
    // Look-ahead (round 3, last session; RR_POSDEF_LOOKAHEAD=0 for A/B runs).  The dependent chain of a panel step was
    // chol(j) -> solve of the whole block row -> update of the whole trailing matrix -> chol(j + 1), although chol(j + 1)
    // needs one 128 x 128 block of each.  Now the context's stream runs chol(j) -> solve of block (j, j + 1) -> update of
    // block (j + 1, j + 1) -> chol(j + 1), one-tile launches, and a third stream the rest of the row's solve and of the
    // trailing update (upper_only = 2: without the block done ahead) under the next diagonal block's factorisation.
    static const bool no_lookahead = getenv("RR_POSDEF_LOOKAHEAD") != nullptr && atoi(getenv("RR_POSDEF_LOOKAHEAD")) == 0;
    const bool lookahead = overlap && !no_lookahead && nblk > 2;
    static const bool no_pair = getenv("RR_POSDEF_PAIR") != nullptr && atoi(getenv("RR_POSDEF_PAIR")) == 0;
    const bool pair_on = lookahead && !no_pair;
    static const int64_t pair_min = getenv("RR_POSDEF_PAIR_MIN") ? atoll(getenv("RR_POSDEF_PAIR_MIN")) : 4096;
    if (lookahead && !c->stream3) RR_CHECK_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    const int64_t nev = lookahead ? 4 * nblk + 1 : nblk + 1;
    // events: [j] first block of row j solved (without look-ahead: the whole row) | [nblk] join | then per panel:
    // diagonal block factored, rest of the row solved, rest of the trailing matrix updated
    const int64_t E_JOIN = nblk, E_CHOL = nblk + 1, E_SOLVE = 2 * nblk + 1, E_TRAIL = 3 * nblk + 1;
    if (overlap && (int64_t)s.ev.size() < nev) {
        while ((int64_t)s.ev.size() < nev) {
            hipEvent_t e;
            RR_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            s.ev.push_back(e);
        }
    }
    const hipStream_t main_stream = c->stream;
    struct StreamGuard {  // rr_launch_gemm_tn_f64 takes the stream from the context: never leave it on the second one
        rr_ctx *c;
        hipStream_t s;
        ~StreamGuard() { c->stream = s; }
    } guard{c, main_stream};
    int rc = RR_OK;
    if (overlap) {  // the second stream starts behind everything queued so far (W and Y assembled)
        RR_CHECK_HIP(hipEventRecord(c->ev_fork, main_stream));
        RR_CHECK_HIP(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        if (lookahead) RR_CHECK_HIP(hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
    }
    for (int64_t j = 0; j < nblk && rc == RR_OK; ++j) {
        double *Ujj = s.W + j * PB * (ld + 1);
        double *Uij = s.Uinv + j * PB * PB;
        const int64_t rest = Fp - (j + 1) * PB;
        if (lookahead) {
            double *row = Ujj + PB;  // block row j right of the diagonal block
            // Panels in PAIRS (round 5; RR_POSDEF_PAIR=0 for A/B runs).  The bulk trailing update of a panel is a K = 128
            // product: every 128 x 128 block of the trailing matrix is read and written around a 128-deep k-loop (~40 % of
            // the f64 MFMA peak, tools/timeline.py).  Panel j (even) therefore updates only what panel j + 1 needs -- its
            // diagonal block (this stream) and the rest of its block row (a 128-row strip, third stream) -- and panel j + 1
            // updates everything below with BOTH block rows at once, K = 256: half the passes over the trailing matrix,
            // twice the k-loop per pass.
            // ... while the trailing matrix is wide (pair_min): below that a panel step lasts as long as its dependent chain
            // (diagonal block -> one-tile solve -> one-tile update), and pairing only adds launches.  (A first form had the
            // second panel update block (j + 1, j + 1) with both block rows at once -- a K = 256 one-tile product ON the chain:
            // F = 4096 measured 4.2 -> 4.65 ms with every panel paired; now the first panel's part comes from the third stream)
            const bool first = pair_on && (j % 2 == 0) && rest > PB && rest >= pair_min;
            const bool second = pair_on && (j % 2 == 1) && rest > 0 && rest + PB >= pair_min;
            double *row2 = s.W + (j - 1) * PB * ld + (j + 1) * PB;  // second: block rows j - 1 and j, columns right of block j
            launch_chol_diag(main_stream, Ujj, ld, Uij);
            RR_CHECK_HIP(hipEventRecord(s.ev[E_CHOL + j], main_stream));
            if (rest > 0) {
                // block (j, j + 1) carries the third stream's update of step j - 1
                if (j > 0) RR_CHECK_HIP(hipStreamWaitEvent(main_stream, s.ev[E_TRAIL + j - 1], 0));
                rc = rr_launch_gemm_tn_f64(c, Uij, PB, row, ld, row, ld, PB, PB, PB, 0, 0);
                if (rc != RR_OK) break;
                RR_CHECK_HIP(hipEventRecord(s.ev[j], main_stream));
                // block (j + 1, j + 1) gets THIS block row's part here, on the chain (one K = 128 tile as without pairs); for the
                // second panel of a pair the first one's part was applied by the third stream during the first panel (below)
                rc = rr_launch_gemm_tn_f64(c, row, ld, row, ld, Ujj + PB * (ld + 1), ld, PB, PB, PB, 1, 1);
                if (rc != RR_OK) break;
                if (rest > PB) {
                    c->stream = c->stream3;
                    RR_CHECK_HIP(hipStreamWaitEvent(c->stream3, s.ev[E_CHOL + j], 0));
                    rc = rr_launch_gemm_tn_f64(c, Uij, PB, row + PB, ld, row + PB, ld, PB, PB, rest - PB, 0, 0);
                    if (rc == RR_OK) {
                        RR_CHECK_HIP(hipEventRecord(s.ev[E_SOLVE + j], c->stream3));
                        RR_CHECK_HIP(hipStreamWaitEvent(c->stream3, s.ev[j], 0));
                        if (first) {      // the strip: block row j + 1 right of its diagonal block ...
                            rc = rr_launch_gemm_tn_f64(c, row, ld, row + PB, ld, Ujj + PB * (ld + 1) + PB, ld, PB, PB, rest - PB, 1, 0);
                            // ... and this block row's part of block (j + 2, j + 2), which the pair's K = 256 update leaves out
                            if (rc == RR_OK)
                                rc = rr_launch_gemm_tn_f64(c, row + PB, ld, row + PB, ld, Ujj + 2 * PB * (ld + 1), ld, PB, PB, PB, 1, 1);
                        }
                        else if (second)  // everything below the pair, K = 256
                            rc = rr_launch_gemm_tn_f64(c, row2, ld, row2, ld, Ujj + PB * (ld + 1), ld, 2 * PB, rest, rest, 1, 2);
                        else
                            rc = rr_launch_gemm_tn_f64(c, row, ld, row, ld, Ujj + PB * (ld + 1), ld, PB, rest, rest, 1, 2);
                    }
                    if (rc == RR_OK) RR_CHECK_HIP(hipEventRecord(s.ev[E_TRAIL + j], c->stream3));
                    c->stream = main_stream;
                    if (rc != RR_OK) break;
                }
            }
            // substitution, step j (second stream): U_jj^-1 and the whole of block row j
            RR_CHECK_HIP(hipStreamWaitEvent(c->stream2, s.ev[rest > 0 ? j : E_CHOL + j], 0));
            if (rest > PB) RR_CHECK_HIP(hipStreamWaitEvent(c->stream2, s.ev[E_SOLVE + j], 0));
            c->stream = c->stream2;
            double *Yj = s.Y + j * PB * ld;
            const int64_t width = (j + 1) * PB;
            rc = rr_launch_gemm_tn_f64(c, Uij, PB, Yj, ld, Yj, ld, PB, PB, width, 0, 0);
            // the same pairing for Y: the first panel of a pair updates block row j + 1 of Y alone, the second everything
            // below with both block rows (K = 256; block row j - 1 of Y is zero in the columns block row j adds)
            if (rc == RR_OK && first) rc = rr_launch_gemm_tn_f64(c, row, ld, Yj, ld, Yj + PB * ld, ld, PB, PB, width, 1, 0);
            else if (rc == RR_OK && second) rc = rr_launch_gemm_tn_f64(c, row2, ld, Yj - PB * ld, ld, Yj + PB * ld, ld, 2 * PB, rest, width, 1, 0);
            else if (rc == RR_OK && rest > 0) rc = rr_launch_gemm_tn_f64(c, row, ld, Yj, ld, Yj + PB * ld, ld, PB, rest, width, 1, 0);
            c->stream = main_stream;
            continue;
        }
        // factor, panel j (context's stream)
        launch_chol_diag(main_stream, Ujj, ld, Uij);
        if (rest > 0) rc = rr_launch_gemm_tn_f64(c, Uij, PB, Ujj + PB, ld, Ujj + PB, ld, PB, PB, rest, 0, 0);  // panel <- U_jj^-T panel
        if (rc != RR_OK) break;
        if (overlap) RR_CHECK_HIP(hipEventRecord(s.ev[j], main_stream));  // block row j of the factor is final
        if (rest > 0)
            rc = rr_launch_gemm_tn_f64(c, Ujj + PB, ld, Ujj + PB, ld, Ujj + PB * (ld + 1), ld, PB, rest, rest, 1, 1);  // trailing update
        if (rc != RR_OK) break;
        // substitution, step j (second stream, or behind the whole factorisation when not overlapped: see below)
        if (overlap) {
            RR_CHECK_HIP(hipStreamWaitEvent(c->stream2, s.ev[j], 0));
            c->stream = c->stream2;
            double *Yj = s.Y + j * PB * ld;
            const int64_t width = (j + 1) * PB;  // non-zero columns of block row j
            rc = rr_launch_gemm_tn_f64(c, Uij, PB, Yj, ld, Yj, ld, PB, PB, width, 0, 0);  // Y_j <- U_jj^-T Y_j
            if (rc == RR_OK && rest > 0) rc = rr_launch_gemm_tn_f64(c, Ujj + PB, ld, Yj, ld, Yj + PB * ld, ld, PB, rest, width, 1, 0);
            c->stream = main_stream;
        }
    }
    if (rc != RR_OK) {
        if (overlap) (void)hipStreamSynchronize(c->stream2);  // nothing of this call may outlive it on the other streams
        if (lookahead) (void)hipStreamSynchronize(c->stream3);
        return rc;
    }
    hipLaunchKernelGGL(rr_get_diag_kernel, dim3((unsigned)((Fp + 255) / 256)), dim3(256), 0, main_stream, s.W, Fp, ld, s.dvec);
    if (overlap) {  // the context's stream continues behind the substitution
        RR_CHECK_HIP(hipEventRecord(s.ev[E_JOIN], c->stream2));  // (the third stream's work is behind it: every E_SOLVE / E_TRAIL was waited for)
        RR_CHECK_HIP(hipStreamWaitEvent(main_stream, s.ev[E_JOIN], 0));
    } else {
        for (int64_t j = 0; j < nblk && rc == RR_OK; ++j) {
            const double *Ujj = s.W + j * PB * (ld + 1);
            double *Yj = s.Y + j * PB * ld;
            const int64_t width = (j + 1) * PB;
            rc = rr_launch_gemm_tn_f64(c, s.Uinv + j * PB * PB, PB, Yj, ld, Yj, ld, PB, PB, width, 0, 0);
            const int64_t rest = Fp - (j + 1) * PB;
            if (rc == RR_OK && rest > 0)
                rc = rr_launch_gemm_tn_f64(c, Ujj + PB, ld, Yj, ld, Yj + PB * ld, ld, PB, rest, width, 1, 0);
        }
        if (rc != RR_OK) return rc;
    }
    RR_CHECK_HIP(hipGetLastError());
    // The diagonal decides whether the rest is worth computing.  Up to Fp = 4096 the rest is cheaper than finding out: C = Y^T Y
    // is a few hundred microseconds there and the question a host round trip of ~35 us in the middle of a chain of dependent
    // launches (4 % of an evaluation at BASELINE config 1) -- so it is queued regardless, the pivots come back with the results,
    // and a matrix that is not safely positive definite costs the (rare) caller the wasted product before its SVD route.
    // (RR_POSDEF_EARLY_CHECK=1: always ask first.)
    static const bool early_env = getenv("RR_POSDEF_EARLY_CHECK") != nullptr;
    const bool ask_first = early_env || Fp > 4096;
    std::vector<double> h((size_t)3 * Fp + 1);
    auto pivots = [&]() -> int {  // h[0, F): the factor's diagonal
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
        if (mind < 1e-5) {  // CHOLTHRESH, mathfun/linalg.py:31
            rr_set_error("rr_posterior_dev: matrix is not safely positive definite (min diag of the factor %g)", mind);
            return RR_ERR_NOT_POSDEF;
        }
        return RR_OK;
    };
    if (ask_first) {
        RR_CHECK_HIP(hipMemcpyAsync(h.data(), s.dvec, (size_t)Fp * 8, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        rc = pivots();
        if (rc != RR_OK) return rc;
    }
    // ---- C = Y^T Y ----
    RR_CHECK_HIP(hipMemsetAsync(s.Cp, 0, (size_t)Fp * Fp * 8, c->stream));
    rc = rr_launch_syrk_f64(c, s.Y, Fp, ld, (int)Fp, s.Cp, 1);  // upper triangle of the (Fp, Fp) product; Y is lower triangular
    if (rc != RR_OK) return rc;
    rc = rr_symmetrize_dev(c, s.Cp, Fp);
    if (rc != RR_OK) return rc;
    hipLaunchKernelGGL(rr_extract_kernel, dim3((unsigned)((F * F + 255) / 256)), dim3(256), 0, c->stream, s.Cp, Fp, dC, F);
    double *dm = s.dvec + Fp, *ddg = s.dvec + 2 * Fp, *dtr = s.dvec + 3 * Fp;
    RR_CHECK_HIP(hipMemsetAsync(dtr, 0, 8, c->stream));
    if (c->deterministic) {
        void *part = nullptr;
        rc = rr_det_scratch(c, (size_t)F * 8, &part);
        if (rc != RR_OK) return rc;
        hipLaunchKernelGGL(rr_posterior_rows_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, c->stream, dC, dG, db, ivar, F,
                           dm, ddg, (double *)part, (int64_t)1);
        rc = rr_det_reduce(c, (const double *)part, F, 1, 1, dtr);
        if (rc != RR_OK) return rc;
    } else {
        hipLaunchKernelGGL(rr_posterior_rows_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, c->stream, dC, dG, db, ivar, F,
                           dm, ddg, dtr);
    }
    RR_CHECK_HIP(hipGetLastError());
    if (ask_first) {
        RR_CHECK_HIP(hipMemcpyAsync(h.data() + Fp, dm, (size_t)(2 * Fp + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    } else {  // [diagonal of the factor | m | diag C | sum(G o C)] in one copy
        RR_CHECK_HIP(hipMemcpyAsync(h.data(), s.dvec, (size_t)(3 * Fp + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        rc = pivots();
        if (rc != RR_OK) return rc;
    }
    memcpy(m, h.data() + Fp, (size_t)F * 8);
    memcpy(diagC, h.data() + 2 * Fp, (size_t)F * 8);
    scal[1] = h[(size_t)3 * Fp];
    return RR_OK;
}

}  // extern "C"
