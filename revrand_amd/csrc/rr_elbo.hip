// Second data pass of StandardLinearModel._elbo and predict_moments on the device.
//
// With the posterior (m, C) from the host Cholesky, both need  U = Phi C  (rows x F x F):
//   _elbo (slm.py:160-162,193-197):  Err = y - Phi m,  sqErr = sum Err^2,
//        dhyp_i = -( m.(Err.dPhi_i) - sum(dPhi_i^T Phi o C) ) / var
//      For a random Fourier basis dPhi_i[r, :] = dz_i[r, :] o [-Phi_s, Phi_c][r, :],  dz_i[r, f] = -x_ri W_if / l_i^2,
//      so with  A[r, f] = Err_r (Phi_c m_s - Phi_s m_c)[r, f] - (Phi_c U_s - Phi_s U_c)[r, f]   (c/s = cos/sin halves)
//      and  T = X^T A  (d x n):   dhyp_i = T[i, :] . W[i, :] / (var l_i^2).   The (N, 2n, d) tensor never exists.
//   predict_moments (slm.py:240-244):  Ey = Phi m,  Vf = rowsum(U o Phi).
//
// Kernels: rr_rff_features_t_kernel (feature-major Phi^T + Phi m), rr_gemm_tn_f32_kernel (U = (Phi^T)^T C on the
// SYRK kernel's LDS-DMA / MFMA structure, plain f32 stores), rr_rowdot_kernel, rr_grad_t_kernel.
#include "rr_internal.h"
#include "rr_mfma_tile.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>

// ---------------------------------------------------------------------------------------------
// Feature-major features:  Pt[f][r] = cos/sqrt(n), Pt[n+f][r] = sin/sqrt(n)  (GEMM operand, K-major),
// and per row  dot[r] = Phi_r . m.  One row per thread (x in registers), frequencies looped with the
// transposed weights Wt[f][DMAX] and m through the scalar cache, stores coalesced along r.  Pt == nullptr: the dot
// products only (the predictive mean without the variance, rr_rff_predict_mean_dev).
// ---------------------------------------------------------------------------------------------
template <int DMAX, typename TX>
__global__ void __launch_bounds__(256)
rr_rff_features_t_kernel(const TX *__restrict__ X, int64_t N, int64_t Npad, int64_t ldx,
                         const float *__restrict__ Wt, const float *__restrict__ mvec, int n,
                         float *__restrict__ Pt, int64_t ldt, float *__restrict__ dot, float scale) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = r < N;
    float x[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) x[i] = valid ? (float)X[r * ldx + i] : 0.f;
    float acc = 0.f;
    for (int f = 0; f < n; ++f) {
        const float *w = Wt + (size_t)f * DMAX;  // wave-uniform
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) z = fmaf(x[i], w[i], z);
        const float fr = z - __builtin_rintf(z);
        const float c = valid ? __builtin_amdgcn_cosf(fr) * scale : 0.f;
        const float s = valid ? __builtin_amdgcn_sinf(fr) * scale : 0.f;
        if (Pt && r < Npad) {
            Pt[(size_t)f * ldt + r] = c;
            Pt[(size_t)(n + f) * ldt + r] = s;
        }
        if (mvec) acc = fmaf(c, mvec[f], fmaf(s, mvec[n + f], acc));
    }
    if (dot && valid) dot[r] = acc;
}

// The same for FEW rows (fewer than two of the blocks above per CU: BASELINE config 1's 10 000 rows are 40 of them, each
// looping over all n frequencies on one CU -- 75 us of a 0.9 ms evaluation): 64 rows per workgroup, wave q takes the q-th
// quarter of the frequencies for all of them (stores still 64 consecutive rows per wave), the four partial dot products meet
// in LDS and are added in wave order.
template <int DMAX, typename TX>
__global__ void __launch_bounds__(256)
rr_rff_features_t4_kernel(const TX *__restrict__ X, int64_t N, int64_t Npad, int64_t ldx,
                          const float *__restrict__ Wt, const float *__restrict__ mvec, int n,
                          float *__restrict__ Pt, int64_t ldt, float *__restrict__ dot, float scale) {
    __shared__ float part[4][64];
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), rl = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 64 + rl;
    const bool valid = r < N;
    float x[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) x[i] = valid ? (float)X[r * ldx + i] : 0.f;
    const int chunk = (n + 3) / 4, f0 = q * chunk, f1 = f0 + chunk < n ? f0 + chunk : n;
    float acc = 0.f;
    for (int f = f0; f < f1; ++f) {
        const float *w = Wt + (size_t)f * DMAX;  // wave-uniform
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) z = fmaf(x[i], w[i], z);
        const float fr = z - __builtin_rintf(z);
        const float c = valid ? __builtin_amdgcn_cosf(fr) * scale : 0.f;
        const float s = valid ? __builtin_amdgcn_sinf(fr) * scale : 0.f;
        if (Pt && r < Npad) {
            Pt[(size_t)f * ldt + r] = c;
            Pt[(size_t)(n + f) * ldt + r] = s;
        }
        if (mvec) acc = fmaf(c, mvec[f], fmaf(s, mvec[n + f], acc));
    }
    part[q][rl] = acc;
    __syncthreads();
    if (q == 0 && dot && valid) dot[r] = ((part[0][rl] + part[1][rl]) + part[2][rl]) + part[3][rl];
}

// zero rows [F, Fp) of the feature-major matrix (pad features)
__global__ void __launch_bounds__(256) rr_zero_rows_kernel(float *P, int64_t row0, int64_t row1, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (row1 - row0) * ld) P[row0 * ld + i] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// D[M][N] = A^T B,  A: (K, lda) with M columns, B: (K, ldb) with N columns, all f32 row-major,
// K % 32 == 0, M % 256 == 0, N % 256 == 0.  One workgroup per 256x256 block of D, whole K loop;
// identical tile / operand / MFMA schedule to rr_syrk_f32_kernel, plain stores instead of atomics.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    const float *A, *B;
    float *D;
    int64_t lda, ldb, ldd;
    int K, ntb;  // ntb = N / 256 column tiles (fastest-varying in blockIdx)
    int kb_per_split = 0;  // > 0: blockIdx.y owns k-blocks [y * kb_per_split, ...) and ADDS into a zeroed D (f32 atomics)
    int upper_b = 0;  // B (K, N) with K == N is upper triangular: column tile tb needs only k < 256 (tb + 1)
    int pair_upper = 0;  // with upper_b: rr_gemm_pair_f32_kernel, one workgroup per PAIR of column tiles (ntb - 1 - q, q)
    int diag_skip = getenv("RR_PREDICT_NO_DIAG_SKIP") ? 0 : 1;  // that kernel: no MFMAs on the zero quarters of the diagonal blocks
    // TRIG epilogue (random Fourier features of Xdim > 128, rr_rff.hip): the product is the phase matrix Z in
    // revolutions; D is the feature matrix P: P[r][c] = cos(2 pi Z[r][c]) scale, P[r][n + c] = sin(..) scale for
    // c < n, zero rows for nvalid <= r < nout, nothing beyond; bvec += P^T y.
    int n = 0;
    float scale = 0.f;
    int64_t nvalid = 0, nout = 0;
    const void *y = nullptr;
    int y_f64 = 0;
    double *bvec = nullptr;
    // prediction with the factor form (Vf = rowsum((Phi B)^2), slm.py:240-244): rowsq[r] += sum over the block's columns of
    // D[r][c]^2 and D is NOT stored (rowsq zeroed by the caller; rows >= rowsq_rows skipped)
    double *rowsq = nullptr;
    int64_t rowsq_rows = 0;
    int spread = rr_dma_spread_env();  // rr_dma_slot
};

template <bool TRIG>
__device__ __forceinline__ void rr_gemm_tn_f32_body(const GemmArgs &p) {
    __shared__ float lds[2 * GR_KB * GR_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ta = blockIdx.x / p.ntb;
    // upper_b: tile cost grows with tb and block b runs on XCD b % 8 -- rotate the column tiles by the row tile so that
    // every XCD sees every cost
    const int tb = p.upper_b ? (int)((blockIdx.x % p.ntb + ta) % p.ntb) : (int)(blockIdx.x % p.ntb);
    const int64_t ca = ta * GR_TC;
    const int cb = tb * GR_TC;

    const int wr = wave >> 2, wc_ = wave & 3;
    const unsigned aoff = 4u * ((lane >> 5) * GR_LD + wr * 128 + (lane & 31));
    const unsigned boff = 4u * ((lane >> 5) * GR_LD + GR_TC + wc_ * 64 + (lane & 31));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const unsigned voff = 16u * lane;  // the requests' lane part; everything else is scalar (rr_dma_kblock)
    auto dma_tile = [&](float *buf, int kb0) {
        rr_dma_kblock(p.A + (int64_t)kb0 * p.lda + ca, p.lda, p.B + (int64_t)kb0 * p.ldb + cb, p.ldb, buf, wave, voff);
    };

    int nkb = p.K / GR_KB, kb_first = 0;
    if (p.upper_b) {
        const int kmax = (tb + 1) * (GR_TC / GR_KB);
        nkb = nkb < kmax ? nkb : kmax;
    }
    if (p.kb_per_split > 0) {  // split-K: few output tiles, long K (Edws = dfs Phi)
        kb_first = blockIdx.y * p.kb_per_split;
        nkb = nkb - kb_first < p.kb_per_split ? nkb - kb_first : p.kb_per_split;
    }
    dma_tile(lds, kb_first * GR_KB);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cbuf = kb & 1;
        gram_consume_staggered(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff, rr_dma_slot(wave, p.spread), [&]() {
            if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_LD), (kb_first + kb + 1) * GR_KB);
        });
        __syncthreads();
    }

    const int hi = lane >> 5;
    if constexpr (TRIG) {
        float bc[2] = {0.f, 0.f}, bs[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // the 16 targets of a row block are requested together (read one by one between the stores below, hipcc waits
            // for each before the next store: 64 serialised round trips per lane)
            float yrow[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                yrow[e] = 0.f;
                if (p.y && gr < p.nvalid) yrow[e] = p.y_f64 ? (float)((const double *)p.y)[gr] : ((const float *)p.y)[gr];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                const bool valid = gr < p.nvalid;
                const float yv = yrow[e];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int gc = cb + wc_ * 64 + j * 32 + (lane & 31);
                    if (gc < p.n) {
                        const float z = acc[i][j][e];
                        const float fr = z - __builtin_rintf(z);
                        const float cv = valid ? __builtin_amdgcn_cosf(fr) * p.scale : 0.f;
                        const float sv = valid ? __builtin_amdgcn_sinf(fr) * p.scale : 0.f;
                        if (gr < p.nout) {
                            p.D[gr * p.ldd + gc] = cv;
                            p.D[gr * p.ldd + p.n + gc] = sv;
                        }
                        bc[j] = fmaf(cv, yv, bc[j]);
                        bs[j] = fmaf(sv, yv, bs[j]);
                    }
                }
            }
        }
        if (p.y) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gc = cb + wc_ * 64 + j * 32 + (lane & 31);
                if (gc < p.n) {
                    unsafeAtomicAdd(&p.bvec[gc], (double)bc[j]);
                    unsafeAtomicAdd(&p.bvec[p.n + gc], (double)bs[j]);
                }
            }
        }
    } else if (p.rowsq) {
        // 32 lanes of a half wave hold the 32 columns of a row: squares summed over the lane's two column blocks, then over
        // the half wave (xor butterfly), one f64 atomic per row, half wave and column quarter
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = fmaf(acc[i][0][e], acc[i][0][e], acc[i][1][e] * acc[i][1][e]);
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if ((lane & 31) == 0 && gr < p.rowsq_rows) unsafeAtomicAdd(&p.rowsq[gr], (double)v);
            }
    } else {
        const bool atomic = p.kb_per_split > 0;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int gc = cb + wc_ * 64 + j * 32 + (lane & 31);
                    if (atomic)
                        unsafeAtomicAdd(&p.D[gr * p.ldd + gc], acc[i][j][e]);
                    else
                        p.D[gr * p.ldd + gc] = acc[i][j][e];
                }
            }
    }
}

__global__ void __launch_bounds__(GR_THREADS, 2) rr_gemm_tn_f32_kernel(const GemmArgs p) { rr_gemm_tn_f32_body<false>(p); }
__global__ void __launch_bounds__(GR_THREADS, 2) rr_gemm_trig_f32_kernel(const GemmArgs p) { rr_gemm_tn_f32_body<true>(p); }

// Prediction's triangular product (B upper triangular: column tile tb needs k < 256 (tb + 1) only) with EQUAL-COST workgroups:
// a workgroup owns the column tiles ntb - 1 - q and q of its row tile -- their k-ranges add up to the same for every q.  With
// one tile per workgroup the costs spread 1 : ntb, and the dispatcher, which hands workgroups to the 8 XCDs strictly in turn,
// left CUs idle behind XCDs still busy with long tiles: the matrix pipe was 94 % busy on the CUs that had work, yet the launch
// reached 0.84 of the peak on its issued MFMAs (profiles/r04_predict).  The second tile's first k-block is requested before
// the first tile's epilogue.  Epilogue: row sums of squares (p.rowsq) or a plain store.
__global__ void __launch_bounds__(GR_THREADS, 2) rr_gemm_pair_f32_kernel(const GemmArgs p) {
    __shared__ float lds[2 * GR_KB * GR_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = (p.ntb + 1) / 2;
    const int64_t ta = blockIdx.x / nq;
    const int q = (int)((blockIdx.x % nq + ta) % nq);  // (rotated by the row tile: neighbouring blocks share B's column tiles in L2)
    const int ntile = p.ntb - 1 - q > q ? 2 : 1;
    const int64_t ca = ta * GR_TC;
    // The two waves of a SIMD (w, w + 4) take the column quarters wc and 3 - wc: in a tile's DIAGONAL block (its last 8
    // k-blocks) the quarter wc of the upper-triangular B is zero from k-block 2 (wc + 1) on, a wave skips those k-blocks, and
    // with complementary quarters every SIMD is left with one wave in the block's second half (10 of its 16 wave x k-block
    // units of MFMAs; with equal quarters the SIMD of quarter 3 would keep all 16 and set the pace at the barriers).
    const int wr = wave >> 2, wc_ = __builtin_amdgcn_readfirstlane(p.diag_skip ? (wr ? 3 - (wave & 3) : (wave & 3)) : (wave & 3)), hi = lane >> 5;
    const unsigned aoff = 4u * ((lane >> 5) * GR_LD + wr * 128 + (lane & 31));
    const unsigned boff = 4u * ((lane >> 5) * GR_LD + GR_TC + wc_ * 64 + (lane & 31));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned voff = 16u * lane;
    const int slot = __builtin_amdgcn_readfirstlane(rr_dma_slot(wave, p.spread));
    const int nkb_all = p.K / GR_KB;
    const int kskip = p.diag_skip ? 2 * (wc_ + 1) : 1 << 30;  // k-blocks of a diagonal block this wave's quarter is non-zero in
    auto dma_tile = [&](float *buf, int kb0, int cbt) {
        rr_dma_kblock(p.A + (int64_t)kb0 * p.lda + ca, p.lda, p.B + (int64_t)kb0 * p.ldb + cbt, p.ldb, buf, wave, voff);
    };
    dma_tile(lds, 0, (ntile == 2 ? p.ntb - 1 - q : q) * GR_TC);
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const int tb = (t == 0 && ntile == 2) ? p.ntb - 1 - q : q;  // the long tile first
        const int cb = tb * GR_TC;
        int nkb = (tb + 1) * (GR_TC / GR_KB);
        nkb = nkb < nkb_all ? nkb : nkb_all;
        floatx16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        __syncthreads();  // this tile's first k-block has landed
        const int klim = tb * (GR_TC / GR_KB) + kskip;  // (wave-uniform) first k-block whose MFMAs would only add zeros
        for (int kb = 0; kb < nkb; ++kb) {
            const int cbuf = kb & 1;
            if (kb < klim) {
                gram_consume_staggered(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff, slot, [&]() {
                    if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_LD), (kb + 1) * GR_KB, cb);
                });
            } else if (kb + 1 < nkb) {
                dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_LD), (kb + 1) * GR_KB, cb);
            }
            __syncthreads();
        }
        if (t + 1 < ntile) dma_tile(lds, 0, q * GR_TC);  // (every wave is past the last barrier: both buffers are free)
        if (p.rowsq) {
            // one f64 atomic per row and half wave through a buffer descriptor over rowsq: lanes that do not hold a row's sum
            // and rows past the end get an offset outside its extent and the hardware drops them (no lane mask per element:
            // 64 of them would push the k-loop's scalars out of the SGPR file, see rr_gemm_gradt_f32_kernel's flush)
            const rr_rsrc_t qrs = rr_make_rsrc(p.rowsq, (unsigned)p.rowsq_rows * 8u);
            unsigned qo = (lane & 31) == 0 ? (unsigned)(ca + wr * 128 + 4 * hi) * 8u : 0x40000000u;
            asm volatile("" : "+v"(qo));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = fmaf(acc[i][0][e], acc[i][0][e], acc[i][1][e] * acc[i][1][e]);
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    const double dv = (double)v;
                    const unsigned off = qo + (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * 8u;
                    asm volatile("buffer_atomic_add_f64 %0, %1, %2, 0 offen" : : "v"(dv), "v"(off), "s"(qrs) : "memory");
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
#pragma unroll
                    for (int j = 0; j < 2; ++j) p.D[gr * p.ldd + cb + wc_ * 64 + j * 32 + (lane & 31)] = acc[i][j][e];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The GLM step's third product WITH its consumer (glm.py:275-283 for a random Fourier basis): the 256x256 block of
//   EdPhi = A^T B   (A = dfs^T (K = kl, rows), B = ws / (K L) (kl, F = 2n))
// never leaves the registers.  Column block cb of the cos half (cb < n) pairs with the sin half of P and vice versa:
//   R[r][c] = -EdPhi[r][c] P[r][c + n]  (c < n),   R[r][c] = +EdPhi[r][c] P[r][c - n]  (c >= n)
//   T[i][c mod n] += sum_r X[r][i] R[r][c]                                  (what rr_glm_grad_t_kernel forms from HBM)
// The accumulator layout of v_mfma_f32_32x32x2_f32 IS its B-operand layout for the row pair (r, r + 4): element e of
// lane l is row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31 -- so R goes straight back into the matrix pipe as B
// with A = X[r + 4 (l >> 5)][l & 31] (one coalesced 128-byte row per half wave): 128 MFMAs per wave and tile (one
// k-block's worth beside the tile's K / 32), no LDS, no barrier.  A workgroup keeps its column block and walks over
// row tiles g, g + G, ..: T stays in 32 registers per lane until one f64 atomic flush at the end.
// Saves EdPhi's write (rows x F floats), its re-read and P's second read by the contraction kernel.
// ---------------------------------------------------------------------------------------------
struct GradtArgs {
    const float *A, *B;
    int64_t lda, ldb;
    int K, ntb, nta;  // k rows; column tiles of B (= 2n / 256); row tiles of A
    const float *P;   // feature matrix (rows, ldp), columns [0, 2n)
    int64_t ldp;
    const float *X;   // the child's inputs (rows, ldx), d <= 128 valid columns
    int64_t ldx, rows;
    int n, d;
    double *T;        // (d, n), accumulated into
    // ERR (the linear model's second pass, slm.py:193-195): the product is U = Phi C and what is contracted is
    // A = Err m^T - U, i.e. R = -(U[r][c] - err[r] mvec[c]) P[r][partner] with the signs above: sign = -1
    const float *err = nullptr, *mvec = nullptr;
    float sign = 1.f;
    int spread = rr_dma_spread_env();  // rr_dma_slot
};

// NXB: 32-column blocks of X (d <= 32 NXB).  NXB == 1: T stays in registers over all of a workgroup's row tiles; NXB > 1:
// the epilogue runs once per block of X (P's block is read again -- from L2 / the Infinity Cache) and T is flushed per tile.
template <bool ERR, int NXB>
__global__ void __launch_bounds__(GR_THREADS, 2) rr_gemm_gradt_f32_kernel(const GradtArgs p) {
    __shared__ float lds[2 * GR_KB * GR_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % p.ntb));
    const int g0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / p.ntb)), G = __builtin_amdgcn_readfirstlane((int)(gridDim.x / p.ntb));
    const int cb = tb * GR_TC;
    const bool cosblk = cb < p.n;
    const int pcol = cosblk ? cb + p.n : cb - p.n;  // partner column block in P
    const int tcol = cosblk ? cb : cb - p.n;        // column block in T
    const float sgn = (cosblk ? -1.f : 1.f) * p.sign;

    const int wr = wave >> 2, wc_ = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const unsigned aoff = 4u * ((lane >> 5) * GR_LD + wr * 128 + (lane & 31));
    const unsigned boff = 4u * ((lane >> 5) * GR_LD + GR_TC + wc_ * 64 + (lane & 31));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const int nkb = p.K / GR_KB;
    const int tlane = tcol + wc_ * 64 + l31;
    // T[32 xb + i][tlane + 32 j] += t[j][e] as f64 atomics through a buffer descriptor over the (d, n) matrix: the rows of a
    // partial last block (32 xb + i >= d) fall outside its extent and the hardware drops them -- no branch and no lane mask
    // per element.  (With `if (i < dl) unsafeAtomicAdd(..)` hipcc formed the 16 lane masks of a flush up front: 32 SGPRs,
    // which inside the tile loop of the NXB > 1 variants pushed the k-loop's scalars out into VGPR lanes.)
    const rr_rsrc_t trs = rr_make_rsrc(p.T, (unsigned)p.d * (unsigned)p.n * 8u);
    const unsigned n8 = (unsigned)p.n * 8u;
    auto flush = [&](floatx16 (&t)[2], int xb) {
        unsigned to = (unsigned)(32 * xb + 4 * hi) * n8 + (unsigned)tlane * 8u;
        // (opaque: the 32 offsets are formed here -- hoisted to the kernel's start they would spill)
        asm volatile("" : "+v"(to));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned i = (unsigned)((e & 3) + 8 * (e >> 2));
                const double v = (double)(sgn * t[j][e]);
                const unsigned off = to + i * n8 + (unsigned)(j * 256);
                asm volatile("buffer_atomic_add_f64 %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(trs) : "memory");
            }
    };

    floatx16 tacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tacc[j][e] = 0.f;
    float mcol[2] = {0.f, 0.f};  // ERR: the lane's two entries of mvec, the same for every row tile
    if (ERR) {
        mcol[0] = p.mvec[cb + wc_ * 64 + l31];
        mcol[1] = p.mvec[cb + wc_ * 64 + 32 + l31];
    }

    // LDS-DMA requests through buffer descriptors, staggered (rr_dma_kblock's scheme) with as little scalar state as it takes:
    // the wave's four rows of a k-block start at (A + ca + 4 w lda) + kb (32 lda) -- ONE descriptor per side, rebuilt per
    // k-block by scalar instructions, the row step k lda in the scalar offset -- and the slot is made provably uniform, so
    // that its tests are s_cmp and not four precomputed lane masks.  (Round 3 kept the flat requests for NXB > 1: with
    // rr_dma_kblock's eight row offsets and the masks the kernel ran out of SGPRs and read them back with v_readlane between
    // the MFMAs -- 254 instead of 242 ms per launch at config 3's shape.)
#ifndef RR_GT_FLATDMA
    constexpr bool BUFDMA = true;
#else
    constexpr bool BUFDMA = NXB == 1;
#endif
    const unsigned voff = 16u * lane;  // the requests' lane part; everything else is scalar
    const unsigned lda4 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)p.lda * 4u));
    const unsigned ldb4 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)p.ldb * 4u));
    const int slot = __builtin_amdgcn_readfirstlane(rr_dma_slot(wave, p.spread));
    const uint64_t bw = (uint64_t)(p.B + cb) + (uint64_t)(4 * wave) * ldb4;  // this wave's first row of B's k-block 0
    auto dma_tile = [&](float *buf, int64_t ca, int kb0) {
        if constexpr (BUFDMA) {
            const uint64_t aw = (uint64_t)(p.A + ca) + (uint64_t)(4 * wave) * lda4 + (uint64_t)(unsigned)kb0 * lda4;
            const rr_rsrc_t ra = rr_make_rsrc((const void *)aw, 0x7fffffffu);
            const rr_rsrc_t rb = rr_make_rsrc((const void *)(bw + (uint64_t)(unsigned)kb0 * ldb4), 0x7fffffffu);
            float *dst = buf + 4 * wave * GR_LD;  // wave-uniform
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(dst + k * GR_LD), 16, voff, (unsigned)k * lda4, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + k * GR_LD + GR_TC), 16, voff, (unsigned)k * ldb4, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lr = 4 * wave + k;
                const float *sa = p.A + (int64_t)(kb0 + lr) * p.lda + ca + 4 * lane;
                const float *sb = p.B + (int64_t)(kb0 + lr) * p.ldb + cb + 4 * lane;
                float *dst = buf + lr * GR_LD;
                __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(dst + GR_TC), 16, 0, 0);
            }
        }
    };

    if (g0 < p.nta) dma_tile(lds, (int64_t)g0 * GR_TC, 0);
    for (int ta = g0; ta < p.nta; ta += G) {
        const int64_t ca = (int64_t)ta * GR_TC;
        floatx16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        __syncthreads();  // k-block 0 of this tile has landed (requested before the previous tile's epilogue)
        for (int kb = 0; kb + 1 < nkb; ++kb) {
            const int cbuf = kb & 1;
            if constexpr (BUFDMA) {
                gram_consume_staggered(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff, slot,
                                       [&]() { dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_LD), ca, (kb + 1) * GR_KB); });
            } else {
                dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_LD), ca, (kb + 1) * GR_KB);
                gram_consume(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff);
            }
            __syncthreads();
        }

        // The tile's block of P (partner columns), its rows of X (and of Err) through buffer descriptors: the base is
        // wave-uniform (SGPRs), the lane part a 32-bit offset, and rows past the end of a partial last tile fall outside
        // the descriptor's extent and read as zero (their product rows are zero as well).
        // The epilogue works through the lane's 64 elements of each column block in 8 groups of 8 (group hb: row block
        // i = hb >> 1, elements 8 (hb & 1) ..): NS register sets rotate, so that the loads of group hb + NS are in flight
        // under the MFMAs of groups hb + 1 .. hb + NS - 1, and the first set is requested BEFORE the tile's last k-block.
        const unsigned tr = (unsigned)(p.rows - ca < GR_TC ? p.rows - ca : GR_TC);  // rows of this tile that exist
        const rr_rsrc_t prs = rr_make_rsrc(p.P + ca * p.ldp + pcol, (tr * (unsigned)p.ldp - (unsigned)pcol) * 4u);
        const rr_rsrc_t xrs = rr_make_rsrc(p.X + ca * p.ldx, tr * (unsigned)p.ldx * 4u);
        const unsigned ldp4 = (unsigned)p.ldp * 4u, ldx4 = (unsigned)p.ldx * 4u;
        unsigned pofs = (unsigned)(wr * 128 + 4 * hi) * ldp4 + (unsigned)(wc_ * 64 + l31) * 4u;
        unsigned xofs = (unsigned)(wr * 128 + 4 * hi) * ldx4 + (unsigned)(l31 < p.d ? l31 : 0) * 4u;
        unsigned eofs = (unsigned)(wr * 128 + 4 * hi) * 4u;
        // (opaque per tile: the lane offsets below are the same for every tile, and hoisted out of the tile loop their
        // 192 registers would spill)
        asm volatile("" : "+v"(pofs), "+v"(xofs), "+v"(eofs));
        rr_rsrc_t ers;
        if (ERR) ers = rr_make_rsrc(p.err + ca, tr * 4u);
        constexpr int NS = ERR ? 2 : 3;
        float xv[NS][8], pv[NS][2][8], ev[NS][8];
        // X's 32-column block xb (a RUN-TIME loop for NXB > 1: unrolled, the blocks' epilogues together want more registers
        // than there are, and what hipcc spills then are the k-loop's lane offsets -- reloaded with `s_waitcnt vmcnt(0)`
        // behind every LDS-DMA request): xm = 1 for the lanes whose column of X exists (the others re-read a valid column
        // and are masked in the product), xadd = the block's byte offset in a row of X
        float xm = l31 < p.d ? 1.f : 0.f;
        unsigned xadd = 0;
        auto load_group = [&](int set, int hb) {  // (set, hb: compile-time after unrolling)
            const int i = hb >> 1;
            unsigned pb = pofs + (unsigned)(32 * i) * ldp4, xo = xofs + (unsigned)(32 * i) * ldx4 + xadd;
            asm volatile("" : "+v"(pb), "+v"(xo));  // (per group: see above)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = 8 * (hb & 1) + k;
                const unsigned ro = (unsigned)((e & 3) + 8 * (e >> 2));
                // (row steps in the VGPR offset, not in soffset: the hardware's range check does not see soffset)
                xv[set][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo + ro * ldx4, 0, 0));
                pv[set][0][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, pb + ro * ldp4, 0, 0));
                pv[set][1][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, pb + ro * ldp4 + 128u, 0, 0));
                if (ERR)
                    ev[set][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ers, eofs + (unsigned)(128 * i) + ro * 4u, 0, 0));
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        load_group(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        {   // the tile's last k-block (nothing left to request for this tile)
            const int cbuf = (nkb - 1) & 1;
            gram_consume(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff);
            __syncthreads();
        }
#pragma unroll 1
        for (int xb = 0; xb < NXB; ++xb) {
            __builtin_amdgcn_sched_barrier(0);
            if (xb > 0) {
                xm = l31 + 32 * xb < p.d ? 1.f : 0.f;
                xadd = xm != 0.f ? 128u * (unsigned)xb : 0u;
                load_group(0, 0);
            }
#pragma unroll
            for (int q = 1; q < NS; ++q) load_group(q, q);
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                const int set = hb % NS, i = hb >> 1;
                __builtin_amdgcn_sched_barrier(0);  // (the sets' loads stay where they are: the scheduler would pull them together)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = 8 * (hb & 1) + k;
                    const float xe = xv[set][k] * xm;  // (a multiply, not a select: the loads stay unconditional and batched)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float a = ERR ? fmaf(-ev[set][k], mcol[j], acc[i][j][e]) : acc[i][j][e];
                        tacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xe, a * pv[set][j][k], tacc[j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (hb + NS < 8) load_group(set, hb + NS);
            }
            if (NXB > 1) {  // this block of T leaves per tile (the registers serve the next block of X)
                flush(tacc, xb);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) tacc[j][e] = 0.f;
            }
        }
        // (requested only now: while an LDS-DMA is pending hipcc waits with vmcnt(0) for ANY load result, which would
        // serialise the rotating sets above)
        if (ta + G < p.nta) dma_tile(lds, (int64_t)(ta + G) * GR_TC, 0);
    }

    if (NXB == 1) flush(tacc, 0);
}

// one workgroup per CU: each keeps its column block and walks over row tiles (d <= 128)
static int launch_gemm_gradt(rr_ctx *c, const GradtArgs &g) {
    int G = c->num_cu / g.ntb;
    if (G < 1) G = 1;
    if (G > g.nta) G = g.nta;
    const dim3 grid((unsigned)(G * g.ntb));
#define RR_GT2(E, X) hipLaunchKernelGGL((rr_gemm_gradt_f32_kernel<E, X>), grid, dim3(GR_THREADS), 0, c->stream, g)
    const int nxb = (g.d + 31) / 32;
    if (g.err) {
        if (nxb <= 1) RR_GT2(true, 1);
        else if (nxb == 2) RR_GT2(true, 2);
        else RR_GT2(true, 4);
    } else {
        if (nxb <= 1) RR_GT2(false, 1);
        else if (nxb == 2) RR_GT2(false, 2);
        else RR_GT2(false, 4);
    }
#undef RR_GT2
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------------------------
// Row reductions: one wave per row.
//   MODE 0:  out[r] = sum_j U[r][j] P[r][j]                        (Vf of predict_moments)
//   MODE 1:  err[r] = y[r] - dot[r];  *sq += sum_r err[r]^2         (Err, sqErr of _elbo)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rr_rowdot_kernel(const float *__restrict__ U, const float *__restrict__ P, int64_t rows, int F, int64_t ld,
                 double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float *u = U + r * ld, *q = P + r * ld;
    float acc = 0.f;
    for (int j = lane; j < F; j += 64) acc = fmaf(u[j], q[j], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) out[r] = (double)acc;
}

template <typename TX>
__global__ void __launch_bounds__(256)
rr_err_kernel(const TX *__restrict__ y, const float *__restrict__ dot, int64_t N, float *__restrict__ err,
              double *__restrict__ sq, int64_t det = 0) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float e = 0.f;
    if (r < N) {
        e = (float)((double)y[r] - (double)dot[r]);
        err[r] = e;
    }
    double acc = (double)e * (double)e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rr_acc_out(sq, det, blockIdx.x, 0, part[0] + part[1] + part[2] + part[3]);
}

// ---------------------------------------------------------------------------------------------
// T[i][f] += sum_r x[r][i] * A[r][f],  A = err (P_c m_s - P_s m_c) - (P_c U_s - P_s U_c).
// One frequency per thread; x rows, err and m through the scalar cache; P and U coalesced along f.
// ---------------------------------------------------------------------------------------------
template <int DMAX, typename TX>
__global__ void __launch_bounds__(256)
rr_grad_t_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const float *__restrict__ P,
                 const float *__restrict__ U, int64_t ldp, const float *__restrict__ err,
                 const float *__restrict__ mvec, int n, int d, double *__restrict__ T, int rows_per_block, int64_t tdet = 0) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    const int fc = fvalid ? f : 0;
    const float mc = mvec[fc], ms = mvec[n + fc];
    float t[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) t[i] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const float pc = P[r * ldp + fc], ps = P[r * ldp + n + fc];
        const float uc = U[r * ldp + fc], us = U[r * ldp + n + fc];
        const float a = err[r] * (pc * ms - ps * mc) - (pc * us - ps * uc);
        const TX *xr = X + r * ldx;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) t[i] = fmaf((float)xr[i], a, t[i]);
    }
    if (fvalid) {
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < d) rr_acc_out(T, tdet, blockIdx.y, (int64_t)i * n + f, (double)t[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// Generic basis-gradient contraction  sum_{r,j} E[r][j] dPhi_i[r][j]  for a random Fourier basis without
// materialising dPhi:  T[i][f] += sum_r x[r][i] * scale * (E[r][n+f] cos - E[r][f] sin)(2 pi z_rf),
// then (host)  d_i = -(1 / l_i^2) sum_f W[i][f] T[i][f].  This is the consumer of basis.grad in
// slm.py:193-197 (E = Err m^T - Phi C) and glm.py:274-275 (E = EdPhi).  One frequency per thread.
// ---------------------------------------------------------------------------------------------
template <int DMAX, typename TX, typename TE>
__global__ void __launch_bounds__(256)
rr_grad_contract_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const float *__restrict__ Ws,
                        const TE *__restrict__ E, int64_t lde, int n, int npad, int d, double *__restrict__ T,
                        float scale, int rows_per_block, int64_t tdet = 0) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    const int fc = fvalid ? f : 0;
    float w[DMAX], t[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
        w[i] = Ws[(size_t)i * npad + fc];
        t[i] = 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const TX *xr = X + r * ldx;
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) z = fmaf((float)xr[i], w[i], z);
        const float fr = z - __builtin_rintf(z);
        const float a = scale * ((float)E[r * lde + n + fc] * __builtin_amdgcn_cosf(fr) -
                                 (float)E[r * lde + fc] * __builtin_amdgcn_sinf(fr));
#pragma unroll
        for (int i = 0; i < DMAX; ++i) t[i] = fmaf((float)xr[i], a, t[i]);
    }
    if (fvalid) {
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < d) rr_acc_out(T, tdet, blockIdx.y, (int64_t)i * n + f, (double)t[i]);
    }
}

// C32 (Fp, Fp) f32, zero padded  <-  C (F, F) f64 on the device (the posterior of rr_posterior_dev)
// tri: the upper-triangular form of the (symmetric) matrix with doubled off-diagonal entries -- x^T C x == x^T Ctri x,
// which is all predict_moments needs from C, and Phi Ctri costs half the k-blocks of Phi C (GemmArgs::upper_b)
__global__ void __launch_bounds__(256)
rr_c64_to_c32_kernel(const double *__restrict__ C, int64_t F, float *__restrict__ C32, int64_t Fp, int tri = 0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Fp * Fp) return;
    const int64_t r = i / Fp, c = i % Fp;
    float v = (r < F && c < F) ? (float)C[r * F + c] : 0.f;
    if (tri) v = r < c ? 2.f * v : (r == c ? v : 0.f);
    C32[i] = v;
}

void rr_launch_c64_to_c32_tri(rr_ctx *c, const double *dC, int64_t F, float *dB, int64_t Fb) {
    hipLaunchKernelGGL(rr_c64_to_c32_kernel, dim3((unsigned)((Fb * Fb + 255) / 256)), dim3(256), 0, c->stream, dC, F, dB, Fb, 1);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Scratch of the second pass, kept by the basis between calls (a fit makes ~100 of them): grow-only.
struct Pass2Scratch {
    float *P = nullptr, *Pt = nullptr, *U = nullptr, *C32 = nullptr, *m32 = nullptr, *dot = nullptr, *err = nullptr;
    void *Ab = nullptr, *Cb = nullptr;  // split-bf16 engine: K-blocked copies of Pt and C32 (rr_launch_gemm_tn_bf16)
    double *acc = nullptr;  // [sqErr | T (d*n)] or Vf
    float *mean = nullptr;  // rr_rff_predict_mean_dev: [m (F) | Phi m (rows)], grow-only -- no hipMalloc / hipFree (a
    size_t mean_count = 0;  // device-wide synchronisation) per `predict` call
    int64_t chunk = 0, Fp = 0;
    size_t nacc = 0;
    std::vector<float> hC, hm;  // host staging for the f64 -> f32 posterior
    void release() {
        void *q[] = {P, Pt, U, C32, m32, dot, err, acc, Ab, Cb, mean};
        for (void *x : q)
            if (x) (void)hipFree(x);
        P = Pt = U = C32 = m32 = dot = err = nullptr;
        Ab = Cb = nullptr;
        acc = nullptr;
        mean = nullptr;
        mean_count = 0;
        chunk = Fp = 0;
        nacc = 0;
    }
};

void rr_pass2_scratch_free(void *p) {
    if (!p) return;
    Pass2Scratch *s = (Pass2Scratch *)p;
    s->release();
    delete s;
}

int rr_features_rowmajor_f32(rr_basis *b, const void *dX, int x_dtype, int64_t m, int64_t mpad, int64_t ldx,
                             float *P, int64_t ldp, bool zero_pad_cols, float *Pt = nullptr, int64_t ldt = 0,
                             bool *pt_written = nullptr, double scale_mult = 1.0);  // rr_rff.hip
int rr_launch_gemm_tn_bf16(rr_ctx *c, int nprod, const float *A, int64_t lda, const float *B, int64_t ldb, float *D,
                           int64_t ldd, int64_t K, int64_t M, int64_t N, void *sa, void *sb, bool sb_ready,
                           bool upper_b = false);  // rr_syrk16.hip

template <typename TX>
static int launch_features_t(rr_basis *b, const TX *X, int64_t N, int64_t Npad, int64_t ldx, const float *m32,
                             float *Pt, int64_t ldt, float *dot) {
    rr_ctx *c = b->ctx;
    const float scale = (float)(1.0 / sqrt((double)b->n));
    const dim3 grid((unsigned)(Npad / 256));
    if (b->large) {  // Xdim > 128: no feature-major kernel; transpose the row-major features the caller just made
        rr_set_error("pass2: internal: feature-major kernel called for Xdim > 128");
        return RR_ERR_INVALID;
    }
    static const bool no_t4 = getenv("RR_FEATURES_T_NO_SPLIT") != nullptr;  // (A/B runs)
    const bool few = !no_t4 && Npad / 256 < 2 * (int64_t)c->num_cu && b->n >= 16;
    const dim3 grid4((unsigned)(Npad / 64));
#define RR_FT(DM)                                                                                                  \
    if (few)                                                                                                       \
        hipLaunchKernelGGL((rr_rff_features_t4_kernel<DM, TX>), grid4, dim3(256), 0, c->stream, X, N, Npad, ldx, b->dWt32, \
                           m32, b->n, Pt, ldt, dot, scale);                                                        \
    else                                                                                                           \
        hipLaunchKernelGGL((rr_rff_features_t_kernel<DM, TX>), grid, dim3(256), 0, c->stream, X, N, Npad, ldx, b->dWt32, \
                           m32, b->n, Pt, ldt, dot, scale)
    switch (b->dpad) {
        case 8: RR_FT(8); break;
        case 16: RR_FT(16); break;
        case 32: RR_FT(32); break;
        case 64: RR_FT(64); break;
        case 128: RR_FT(128); break;
        default: rr_set_error("pass2: d=%d is not supported", b->d); return RR_ERR_UNSUPPORTED;
    }
#undef RR_FT
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

template <typename TX>
static int launch_grad_t(rr_basis *b, const TX *X, int64_t N, int64_t ldx, const float *P, const float *U, int64_t ldp,
                         const float *err, const float *m32, double *T) {
    rr_ctx *c = b->ctx;
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = (N * fblocks + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
    if (rpb < 64) rpb = 64;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((N + rpb - 1) / rpb));
    // deterministic mode: every row block stores its (d, n) partial into its own slot, added to T in order afterwards
    const int64_t tcount = (int64_t)b->d * b->n, tdet = c->deterministic ? tcount : 0;
    double *Tacc = T;
    if (tdet) {
        RR_REQUIRE(!b->large, "second pass: deterministic mode does not cover Xdim > 128");
        void *part = nullptr;
        int rc = rr_det_scratch(c, (size_t)grid.y * (size_t)tcount * 8, &part);
        if (rc != RR_OK) return rc;
        T = (double *)part;
    }
#define RR_GT(DM)                                                                                                 \
    hipLaunchKernelGGL((rr_grad_t_kernel<DM, TX>), grid, dim3(256), 0, c->stream, X, N, ldx, P, U, ldp, err, m32,   \
                       b->n, b->d, T, (int)rpb, tdet)
    switch (b->dpad) {
        case 8: RR_GT(8); break;
        case 16: RR_GT(16); break;
        case 32: RR_GT(32); break;
        case 64: RR_GT(64); break;
        case 128: RR_GT(128); break;
        default:  // Xdim > 128: 128 input dimensions per launch (A is recomputed from P and U, 4 loads per (row, frequency))
            for (int i0 = 0; i0 < b->d; i0 += 128)
                hipLaunchKernelGGL((rr_grad_t_kernel<128, TX>), grid, dim3(256), 0, c->stream, X + i0, N, ldx, P, U, ldp, err,
                                   m32, b->n, b->d - i0 < 128 ? b->d - i0 : 128, T + (size_t)i0 * b->n, (int)rpb);
    }
#undef RR_GT
    RR_CHECK_HIP(hipGetLastError());
    return tdet ? rr_det_reduce(c, T, grid.y, tcount, tcount, Tacc) : RR_OK;
}

__global__ void rr_transpose_f32_kernel(const float *__restrict__ P, int64_t rows, int64_t ldp, float *__restrict__ Pt,
                                        int64_t ldt);
__global__ void rr_rowvec_kernel(const float *__restrict__ P, const float *__restrict__ mvec, int64_t rows, int F, int64_t ld,
                                 float *__restrict__ dot);

// Common driver.  MODE_ELBO: out = [sqErr | T(d*n)] accumulated over row chunks (host doubles).
//                 MODE_PRED: Ey, Vf per row (host doubles, length N).
// Bprep (pred only): a prepared DEVICE (Fp, Fp) f32 upper-triangular prediction factor (rr_variance_factor_dev) instead of
// C; form 1: Vf = rowsum((Phi B)^2), form 0: B is the triangular form of C and Vf = rowsum((Phi B) o Phi).
template <typename TX>
static int pass2_run(rr_basis *b, bool pred, const TX *dX, const TX *dy, int64_t N, int64_t ldx, const double *mh,
                     const double *Ch, double *out0, double *out1, bool c_on_device = false, const float *Bprep = nullptr,
                     int form = 0) {
    rr_ctx *c = b->ctx;
    const int F = 2 * b->n, n = b->n;
    const int64_t Fp = ((int64_t)F + 255) / 256 * 256;
    // rows per chunk: P, Pt, U  (3 x 4 Fp bytes per row) within ~24 GiB, multiple of 256
    int64_t chunk = (int64_t)(((size_t)24 << 30) / ((size_t)12 * Fp));
    const char *cenv = getenv("RR_PASS2_CHUNK_ROWS");
    if (cenv && atoll(cenv) >= 256) chunk = atoll(cenv);
    else if (N > chunk) {
        // several chunks: equal ones, like the Gram pass' (N = 1M at F = 4096: 2 x 500 000 instead of 524 288 + 475 712).
        // Besides the balance, a 2^19-row chunk makes the leading dimension of Pt exactly 2 MiB: the K rows of one operand
        // tile then sit a power of two apart and alias in the L2
        const int64_t nchunks = (N + chunk - 1) / chunk;
        chunk = (N + nchunks - 1) / nchunks;
    }
    if (chunk > N) chunk = N;
    chunk = (chunk + 255) / 256 * 256;
    if (!b->pass2) b->pass2 = new Pass2Scratch();
    Pass2Scratch &s = *(Pass2Scratch *)b->pass2;
    const size_t nacc = pred ? (size_t)chunk : (size_t)1 + (size_t)b->d * n;
    if (s.chunk < chunk || s.Fp != Fp || s.nacc < nacc) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        s.release();
        hipError_t ea = hipMalloc((void **)&s.P, (size_t)chunk * Fp * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Pt, (size_t)Fp * chunk * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.U, (size_t)chunk * Fp * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.C32, (size_t)Fp * Fp * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.m32, (size_t)F * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.dot, (size_t)chunk * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.err, (size_t)chunk * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.acc, nacc * 8);
        if (ea != hipSuccess) {
            (void)hipGetLastError();
            s.release();
            rr_set_error("pass2: device allocation failed (%lld rows per chunk)", (long long)chunk);
            return RR_ERR_OOM;
        }
        s.chunk = chunk;
        s.Fp = Fp;
        s.nacc = nacc;
    }
    if (c->gram_engine != 0 && !s.Ab) {  // the U = Phi C GEMM on the split-bf16 engine
        hipError_t eb = hipMalloc(&s.Ab, (size_t)Fp * s.chunk * 4);
        if (eb == hipSuccess) eb = hipMalloc(&s.Cb, (size_t)Fp * Fp * 4);
        if (eb != hipSuccess) {
            (void)hipGetLastError();
            s.release();
            rr_set_error("pass2: device allocation failed (split-bf16 operands)");
            return RR_ERR_OOM;
        }
    }
    const int64_t step = chunk;  // rows per launch; a scratch kept from a larger call does not change the partition
    chunk = s.chunk;             // the allocated leading dimension of Pt
    hipError_t e = hipSuccess;
    int rc = RR_OK;
    const char *nfz = getenv("RR_PASS2_NO_FUSE");
    const bool fuse_t = !pred && sizeof(TX) == 4 && c->gram_engine == 0 && !c->deterministic && !b->large && !b->phase64 &&
                        n % 256 == 0 && b->d <= 128 && Fp < (1 << 21) && ldx < (1 << 21) && !(nfz && atoi(nfz) != 0);
    {   // posterior to the device in f32: m (F), C padded to (Fp, Fp)
        s.hm.resize(F);
        for (int i = 0; i < F; ++i) s.hm[i] = (float)mh[i];
        e = hipMemcpy(s.m32, s.hm.data(), (size_t)F * 4, hipMemcpyHostToDevice);
        // prediction needs only the quadratic form phi^T C phi: C goes up in its upper-triangular form (doubled
        // off-diagonals), and the GEMM skips the k-blocks below the diagonal -- half the product
        if (Bprep) {
            // nothing to convert
        } else if (c_on_device) {
            hipLaunchKernelGGL(rr_c64_to_c32_kernel, dim3((unsigned)((Fp * Fp + 255) / 256)), dim3(256), 0, c->stream, Ch,
                               (int64_t)F, s.C32, Fp, pred ? 1 : 0);
        } else {
            s.hC.assign((size_t)Fp * Fp, 0.f);
            for (int i = 0; i < F; ++i) {
                const double *src = Ch + (size_t)i * F;
                float *dst = s.hC.data() + (size_t)i * Fp;
                if (pred) {
                    dst[i] = (float)src[i];
                    for (int j = i + 1; j < F; ++j) dst[j] = 2.f * (float)src[j];
                } else {
                    for (int j = 0; j < F; ++j) dst[j] = (float)src[j];
                }
            }
            if (e == hipSuccess) e = hipMemcpy(s.C32, s.hC.data(), s.hC.size() * 4, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && !pred) e = hipMemsetAsync(s.acc, 0, nacc * 8, c->stream);
        if (e == hipSuccess && Fp > F) {  // pad feature rows of Pt are never written by the kernel
            const int64_t cnt = (Fp - F) * chunk;
            hipLaunchKernelGGL(rr_zero_rows_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, s.Pt,
                               (int64_t)F, Fp, chunk);
        }
        if (e != hipSuccess) {
            rr_set_error("pass2: upload failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += step) {
        const int64_t mrows = (N - r0 < step) ? N - r0 : step;
        const int64_t mpad = (mrows + 255) / 256 * 256;
        const TX *Xc = dX + r0 * ldx;
        bool fused_vf = false;
        // prediction in the factor form sums the squares of Phi B in the product's epilogue: nothing reads the row-major P
        static const bool no_predict_fuse = getenv("RR_PREDICT_NO_FUSE") != nullptr;
        const bool will_fuse_vf = pred && Bprep && form == 1 && !c->deterministic && c->gram_engine == 0 && !no_predict_fuse;
        const bool need_p = !(will_fuse_vf && !b->large && !b->phase64);
        // row-major P (epilogues) and feature-major Pt + Phi m (GEMM operand)
        if (need_p) rc = rr_features_rowmajor_f32(b, Xc, sizeof(TX) == 4 ? RR_F32 : RR_F64, mrows, mpad, ldx, s.P, Fp, true);
        if (rc != RR_OK) break;
        if (b->large || b->phase64) {  // no feature-major kernel for these: transpose the row-major features
            hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(Fp / 64), (unsigned)(mpad / 64)), dim3(256), 0, c->stream,
                               s.P, mrows, Fp, s.Pt, chunk);
            hipLaunchKernelGGL(rr_rowvec_kernel, dim3((unsigned)((mrows + 3) / 4)), dim3(256), 0, c->stream, s.P, s.m32, mrows,
                               F, Fp, s.dot);
        } else {
            rc = launch_features_t<TX>(b, Xc, mrows, mpad, ldx, s.m32, s.Pt, chunk, s.dot);
        }
        if (rc != RR_OK) break;
        // The gradient pass with whole [cos | sin] tiles: Err first, then U = P C contracts itself with P, Err m^T and X
        // block by block (rr_gemm_gradt_f32_kernel<true, ..>) -- U is neither written nor read back, P is read once.
        if (fuse_t) {
            const TX *yc = dy + r0;
            hipLaunchKernelGGL(rr_err_kernel<TX>, dim3((unsigned)(mpad / 256)), dim3(256), 0, c->stream, yc, s.dot, mrows, s.err,
                               s.acc);
            GradtArgs g;
            g.A = s.Pt; g.lda = chunk; g.B = s.C32; g.ldb = Fp; g.K = (int)(((int64_t)F + GR_KB - 1) / GR_KB * GR_KB);
            g.ntb = (int)(Fp / 256); g.nta = (int)(mpad / 256);
            g.P = s.P; g.ldp = Fp; g.X = (const float *)Xc; g.ldx = ldx; g.rows = mrows;
            g.n = n; g.d = b->d; g.T = s.acc + 1; g.err = s.err; g.mvec = s.m32; g.sign = -1.f;
            if (launch_gemm_gradt(c, g) != RR_OK || (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
                rr_set_error("pass2: fused gemm failed");
                rc = RR_ERR_HIP;
                break;
            }
            continue;
        }
        // U = P C  as  (Pt)^T C : A = Pt (K = Fp, M = mpad columns), B = C32 (K = Fp, N = Fp)
        if (c->gram_engine != 0) {
            rc = rr_launch_gemm_tn_bf16(c, c->gram_engine, s.Pt, chunk, Bprep ? Bprep : s.C32, Fp, s.U, Fp, Fp, mpad, Fp, s.Ab, s.Cb,
                                        r0 > 0, pred);
            if (rc != RR_OK) break;
        } else {
            GemmArgs g;
            g.A = s.Pt; g.B = Bprep ? Bprep : s.C32; g.D = s.U; g.lda = chunk; g.ldb = Fp; g.ldd = Fp; g.K = (int)(((int64_t)F + GR_KB - 1) / GR_KB * GR_KB); g.ntb = (int)(Fp / 256);  // K: the valid feature rows only (the rest of Fp is zero padding)
            g.upper_b = pred ? 1 : 0;
            if (will_fuse_vf) {  // Vf = rowsum((Phi B)^2) summed in the product's epilogue
                e = hipMemsetAsync(s.acc, 0, (size_t)mrows * 8, c->stream);
                g.rowsq = s.acc;
                g.rowsq_rows = mrows;
                fused_vf = true;
            }
            static const bool no_pair = getenv("RR_PREDICT_NO_PAIR") != nullptr;
            g.pair_upper = (g.upper_b && !no_pair) ? 1 : 0;
            const int64_t wg_per_row_tile = g.pair_upper ? (g.ntb + 1) / 2 : g.ntb;
            if (g.pair_upper) hipLaunchKernelGGL(rr_gemm_pair_f32_kernel, dim3((unsigned)((mpad / 256) * wg_per_row_tile)), dim3(GR_THREADS), 0, c->stream, g);
            else hipLaunchKernelGGL(rr_gemm_tn_f32_kernel, dim3((unsigned)((mpad / 256) * wg_per_row_tile)), dim3(GR_THREADS), 0, c->stream, g);
        }
        if (hipGetLastError() != hipSuccess) {
            rr_set_error("pass2: gemm launch failed");
            rc = RR_ERR_HIP;
            break;
        }
        if (pred) {
            if (!fused_vf)
                hipLaunchKernelGGL(rr_rowdot_kernel, dim3((unsigned)((mrows + 3) / 4)), dim3(256), 0, c->stream, s.U,
                                   (Bprep && form == 1) ? (const float *)s.U : (const float *)s.P, mrows, F, Fp, s.acc);
            std::vector<float> dot(mrows);
            e = hipMemcpyAsync(out1 + r0, s.acc, (size_t)mrows * 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dot.data(), s.dot, (size_t)mrows * 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) {
                rr_set_error("pass2: download failed: %s", hipGetErrorString(e));
                rc = RR_ERR_HIP;
                break;
            }
            for (int64_t i = 0; i < mrows; ++i) out0[r0 + i] = (double)dot[i];
        } else {
            const TX *yc = dy + r0;
            if (c->deterministic) {
                void *part = nullptr;
                rc = rr_det_scratch(c, (size_t)(mpad / 256) * 8, &part);
                if (rc != RR_OK) break;
                hipLaunchKernelGGL(rr_err_kernel<TX>, dim3((unsigned)(mpad / 256)), dim3(256), 0, c->stream, yc, s.dot, mrows,
                                   s.err, (double *)part, (int64_t)1);
                rc = rr_det_reduce(c, (const double *)part, mpad / 256, 1, 1, s.acc);
                if (rc != RR_OK) break;
            } else {
                hipLaunchKernelGGL(rr_err_kernel<TX>, dim3((unsigned)(mpad / 256)), dim3(256), 0, c->stream, yc, s.dot, mrows,
                                   s.err, s.acc);
            }
            rc = launch_grad_t<TX>(b, Xc, mrows, ldx, s.P, s.U, Fp, s.err, s.m32, s.acc + 1);
            if (rc == RR_OK && (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
                rr_set_error("pass2: kernel failed: %s", hipGetErrorString(e));
                rc = RR_ERR_HIP;
            }
        }
    }
    if (rc == RR_OK && !pred) {
        std::vector<double> acc(nacc);
        e = hipMemcpy(acc.data(), s.acc, nacc * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            rr_set_error("pass2: download failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        } else {
            *out0 = acc[0];
            memcpy(out1, acc.data() + 1, (nacc - 1) * 8);
        }
    }
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

static int pass2_checks(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                        int n_ls, const double *m, const double *C, const char *who) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "%s: not an RFF basis", who);
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "%s: bad dtype", who);
    RR_REQUIRE(N >= 1 && dX != nullptr && m != nullptr && C != nullptr, "%s: null/empty argument", who);
    RR_REQUIRE(ldx >= b->dpad, "%s: device X needs ldx >= rr_rff_padded_dim() = %d with zero pad columns", who, b->dpad);
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    return RR_OK;
}

// ---------------------------------------------------------------------------------------------
// Second pass over a device feature matrix (concatenated bases): the same Err / U = P C / gradient
// contraction with P assembled by the children (rr_featmat_put_*).
// ---------------------------------------------------------------------------------------------
// Pt[c][r] = P[r][c] (r < rows; zero for rows <= r < rows256): the K-major GEMM operand.  64x64 tiles via LDS.
__global__ void __launch_bounds__(256)
rr_transpose_f32_kernel(const float *__restrict__ P, int64_t rows, int64_t ldp, float *__restrict__ Pt, int64_t ldt) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t r = r0 + ty + 4 * k;
        tile[ty + 4 * k][tx] = r < rows ? P[r * ldp + c0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) Pt[(c0 + ty + 4 * k) * ldt + r0 + tx] = tile[tx][ty + 4 * k];
}

// dot[r] = P[r][0:F] . m   (one wave per row)
__global__ void __launch_bounds__(256)
rr_rowvec_kernel(const float *__restrict__ P, const float *__restrict__ mvec, int64_t rows, int F, int64_t ld,
                 float *__restrict__ dot) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float *q = P + r * ld;
    float acc = 0.f;
    for (int j = lane; j < F; j += 64) acc = fmaf(q[j], mvec[j], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) dot[r] = acc;
}

// ---------------------------------------------------------------------------------------------
// GLM SVI step (glm.py:296-322) on a device feature matrix.  FSt[r][kl] = Phi_r . ws_kl for all K*L weight
// samples; this kernel turns it IN PLACE into dfs = d loglike / d f and reduces, per mixture component
// k = kl / L,  sum(loglike) (without its f-independent constant) and the likelihood's auxiliary sum
// (Gaussian: sum (y - f)^2).  One weight sample per thread (coalesced along kl), rows looped; rows >= M and
// samples >= K*L are written as zero so the following GEMMs can run over the padded shapes.
// likelihoods.py: Bernoulli :46-104, Binomial :171-233, Gaussian :298-396, Poisson :456-521.
// ---------------------------------------------------------------------------------------------
// log(1 + t) for 0 <= t <= 1 without the library's log1pf (whose code keeps hipcc from unrolling the tile epilogue of
// rr_gemm_lik_f32_kernel over a lane's 128 elements): log(u) t / (u - 1) with u = fl(1 + t) undoes the rounding of 1 + t
// (u - 1 is exact), and t itself where u == 1
__device__ __forceinline__ float rr_log1p01(float t) {
    const float u = 1.f + t, dd = u - 1.f;
    return dd == 0.f ? t : __logf(u) * __fdividef(t, dd);
}
__device__ __forceinline__ float rr_softplus(float f) { return fmaxf(f, 0.f) + rr_log1p01(__expf(-fabsf(f))); }
__device__ __forceinline__ float rr_expit(float f) {
    const float t = __expf(-fabsf(f));
    return f >= 0.f ? 1.f / (1.f + t) : t / (1.f + t);
}

template <int LIK, typename TY>
__global__ void __launch_bounds__(256)
rr_glm_lik_kernel(float *__restrict__ FSt, int64_t M, int64_t rows256, int64_t ld, const TY *__restrict__ y,
                  const TY *__restrict__ rowarg, float par, float fscale, int KL, int L, double *__restrict__ llsum,
                  double *__restrict__ aux, int rows_per_block, const double *__restrict__ par_dev = nullptr) {
    if (LIK == RR_LIK_GAUSSIAN && par_dev) par = (float)par_dev[0];  // the variance is an optimiser coordinate in HBM (resident SVI loop)
    const int kl = blockIdx.x * 256 + threadIdx.x;
    const bool kvalid = kl < KL;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows256) r1 = rows256;
    float ll = 0.f, ax = 0.f;
    const float ipar = LIK == RR_LIK_GAUSSIAN ? 1.f / par : 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        float df = 0.f;
        if (kvalid && r < M) {
            const float f = FSt[r * ld + kl] * fscale;  // fs was formed with ws / (K L)
            const float yr = (float)y[r];
            if (LIK == RR_LIK_BERNOULLI) {
                df = yr - rr_expit(f);
                ll += yr * f - rr_softplus(f);
            } else if (LIK == RR_LIK_BINOMIAL) {
                const float n = (float)rowarg[r];
                df = yr - n * rr_expit(f);
                ll += yr * f - n * rr_softplus(f);
            } else if (LIK == RR_LIK_GAUSSIAN) {
                const float e = yr - f;
                df = e * ipar;
                ax = fmaf(e, e, ax);
            } else if (LIK == RR_LIK_POISSON_EXP) {
                const float g = __expf(f);
                df = yr - g;
                ll += yr * f - g;
            } else {  // Poisson, softplus link
                const float g = fmaxf(rr_softplus(f), 1e-37f);
                df = rr_expit(f) * (yr / g - 1.f);
                ll += yr * __logf(g) - g;
            }
        }
        FSt[r * ld + kl] = df;
    }
    // per-component sums: LDS first (a block spans at most 256 / L + 2 components), then one global atomic each
    __shared__ float sacc[258];
    const int k0 = (blockIdx.x * 256) / L;
    for (int i = threadIdx.x; i < 258; i += 256) sacc[i] = 0.f;
    __syncthreads();
    if (kvalid) atomicAdd(&sacc[kl / L - k0], LIK == RR_LIK_GAUSSIAN ? ax : ll);
    __syncthreads();
    const int klast = blockIdx.x * 256 + 255 < KL ? blockIdx.x * 256 + 255 : KL - 1;
    const int nk = klast >= blockIdx.x * 256 ? klast / L - k0 + 1 : 0;
    for (int i = threadIdx.x; i < nk; i += 256) {
        const double v = (double)sacc[i];
        if (LIK == RR_LIK_GAUSSIAN) {
            unsafeAtomicAdd(&aux[k0 + i], v);
            unsafeAtomicAdd(&llsum[k0 + i], -0.5 * v * (double)ipar);
        } else {
            unsafeAtomicAdd(&llsum[k0 + i], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The step's first product WITH the likelihood kernel above as its epilogue: the 256x256 block of fs = P WS^T turns into
// dfs in the accumulators and is stored twice -- row-major (the A operand of Ed = dfs Phi) and transposed (kl, rows: the
// A operand of EdPhi = dfs^T ws), four consecutive rows of a column as one 16-byte store -- and the per-component sums
// go through LDS to one f64 atomic per component and workgroup.  Saves the likelihood kernel's read + write of fs and
// the transposing pass over dfs (three passes over rows x kl floats).  Whole K per workgroup (no split-K).
// ---------------------------------------------------------------------------------------------
// d loglike / d f and the log-likelihood term of one element (the formulas of rr_glm_lik_kernel)
template <int LIK>
__device__ __forceinline__ void rr_lik_elem(float f, float yr, float nn, float ipar, float &df, float &ll) {
    if (LIK == RR_LIK_BERNOULLI) {
        df = yr - rr_expit(f);
        ll = yr * f - rr_softplus(f);
    } else if (LIK == RR_LIK_BINOMIAL) {
        df = yr - nn * rr_expit(f);
        ll = yr * f - nn * rr_softplus(f);
    } else if (LIK == RR_LIK_GAUSSIAN) {
        const float er = yr - f;
        df = er * ipar;
        ll = er * er;
    } else if (LIK == RR_LIK_POISSON_EXP) {
        const float g = __expf(f);
        df = yr - g;
        ll = yr * f - g;
    } else {  // Poisson, softplus link
        const float g = fmaxf(rr_softplus(f), 1e-37f);
        df = rr_expit(f) * (yr / g - 1.f);
        ll = yr * __logf(g) - g;
    }
}

struct GemmLikArgs {
    const float *A, *B;   // A = P^T (K = Fp, rows), B = WS^T (Fp, kl)
    float *D, *Dt;        // dfs (rows256, ldd) and dfs^T (kl, ldt); BOTH null: objective only, nothing stored
    int64_t lda, ldb, ldd, ldt;
    int K, ntb;
    const void *y, *rowarg;
    int y_f64;
    int64_t M;            // valid rows
    float par, fscale;
    int KL, L;
    double *llsum, *aux;
    int spread = rr_dma_spread_env();  // rr_dma_slot
    const double *par_dev = nullptr;   // Gaussian: the variance in device memory instead of `par` (resident SVI loop)
};

template <int LIK, bool ST>  // ST: store dfs and dfs^T (false: objective only)
__global__ void __launch_bounds__(GR_THREADS, 2) rr_gemm_lik_f32_kernel(const GemmLikArgs p) {
    __shared__ float lds[2 * GR_KB * GR_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ta = blockIdx.x / p.ntb;
    const int tb = (int)(blockIdx.x % p.ntb);
    const int64_t ca = ta * GR_TC;
    const int cb = tb * GR_TC;
    const int wr = wave >> 2, wc_ = wave & 3;
    const unsigned aoff = 4u * ((lane >> 5) * GR_LD + wr * 128 + (lane & 31));
    const unsigned boff = 4u * ((lane >> 5) * GR_LD + GR_TC + wc_ * 64 + (lane & 31));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const unsigned voff = 16u * lane;  // the requests' lane part; everything else is scalar (rr_dma_kblock)
    auto dma_tile = [&](float *buf, int kb0) {
        rr_dma_kblock(p.A + (int64_t)kb0 * p.lda + ca, p.lda, p.B + (int64_t)kb0 * p.ldb + cb, p.ldb, buf, wave, voff);
    };
    const int nkb = p.K / GR_KB;
    dma_tile(lds, 0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cbuf = kb & 1;
        gram_consume_staggered(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff, rr_dma_slot(wave, p.spread), [&]() {
            if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_LD), (kb + 1) * GR_KB);
        });
        __syncthreads();
    }

    // the tile buffers are free now: [0, 258) per-component sums, [512, 768) the tile's targets, [768, 1024) its row
    // argument, [1024, 1280) 1 / 0 for rows that exist / do not
    float *sacc = lds, *sy = lds + 512, *sn = lds + 768, *sm = lds + 1024;
    for (int t = tid; t < 258; t += GR_THREADS) sacc[t] = 0.f;
    if (tid < GR_TC) {
        const int64_t r = ca + tid;
        float yv = 0.f, nv = 0.f;
        sm[tid] = r < p.M ? 1.f : 0.f;
        if (r < p.M) {
            yv = p.y_f64 ? (float)((const double *)p.y)[r] : ((const float *)p.y)[r];
            if (LIK == RR_LIK_BINOMIAL) nv = p.y_f64 ? (float)((const double *)p.rowarg)[r] : ((const float *)p.rowarg)[r];
        }
        sy[tid] = yv;
        sn[tid] = nv;
    }
    __syncthreads();

    const int hi = lane >> 5, l31 = lane & 31;
    constexpr int lik = LIK;
    const float ipar = lik == RR_LIK_GAUSSIAN ? 1.f / (p.par_dev ? (float)p.par_dev[0] : p.par) : 0.f;
    float red[2] = {0.f, 0.f};
    const int rl0 = wr * 128 + 4 * hi;
    const float cm[2] = {cb + wc_ * 64 + l31 < p.KL ? 1.f : 0.f, cb + wc_ * 64 + 32 + l31 < p.KL ? 1.f : 0.f};
    // Four rows of a column block at a time: likelihood terms, one 16-byte store into dfs^T, four 4-byte stores into dfs --
    // through buffer descriptors (tile bases and row steps in SGPRs, one 32-bit lane offset each).  The accumulators are
    // only read.  Row / column validity are weights (1 / 0), not predicates: 128 lane masks would live in SGPR pairs.
    typedef float float4v __attribute__((ext_vector_type(4)));
    constexpr bool st = ST;
    const unsigned ldd4 = (unsigned)p.ldd * 4u, ldt4 = (unsigned)p.ldt * 4u;
    rr_rsrc_t drs, trs;
    unsigned dofs = 0, tofs = 0;
    if (st) {
        drs = rr_make_rsrc(p.D + ca * p.ldd + cb, 256u * ldd4 - (unsigned)cb * 4u);
        trs = rr_make_rsrc(p.Dt + (int64_t)cb * p.ldt + ca, 256u * ldt4 - (unsigned)ca * 4u);
        dofs = (unsigned)(wr * 128 + 4 * hi) * ldd4 + (unsigned)(wc_ * 64 + l31) * 4u;
        tofs = (unsigned)(wc_ * 64 + l31) * ldt4 + (unsigned)(wr * 128 + 4 * hi) * 4u;
    }
    // (the targets / weights come through asm reads pinned in program order: instruction selection otherwise emits all 48
    // LDS reads of the epilogue first -- 192 registers -- and spills)
    const unsigned srow = lds0 + 4u * (unsigned)rl0;  // byte address of lds[rl0]; sy / sn / sm sit 2048 / 3072 / 4096 bytes on
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4v y4, m4, n4 = {0.f, 0.f, 0.f, 0.f};  // rows rl0 + 32 i + 8 q + (0..3)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y4) : "v"(srow), "i"(2048 + 4 * (32 * i + 8 * q)));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m4) : "v"(srow), "i"(4096 + 4 * (32 * i + 8 * q)));
            if (LIK == RR_LIK_BINOMIAL)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(n4) : "v"(srow), "i"(3072 + 4 * (32 * i + 8 * q)));
            // (the wait names the registers it is for: the compiler does not count asm loads, and nothing else keeps their
            // first use behind it)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y4), "+v"(m4), "+v"(n4)::"memory");
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4v v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float df, ll;
                    rr_lik_elem<LIK>(acc[i][j][4 * q + r] * p.fscale, y4[r], n4[r], ipar, df, ll);  // fs was formed with ws / (K L)
                    const float w = m4[r] * cm[j];
                    v[r] = df * w;
                    red[j] = fmaf(ll, w, red[j]);
                }
                // (pinned before the next group's reads: the scheduler otherwise sinks all 128 log-likelihood terms to the
                // end of the epilogue and spills their inputs)
                asm volatile("" : "+v"(red[j]));
                if (st) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rr_u4_t, v), trs, tofs + (unsigned)(8 * q) * 4u,
                                                           (unsigned)(j * 32) * ldt4, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), drs, dofs + (unsigned)(j * 128),
                                                              (unsigned)(r + 8 * q) * ldd4, 0);
                }
            }
        }
        dofs += 32u * ldd4;
        tofs += 128u;
    }
    // per-component sums: LDS first (the tile's 256 columns span at most 256 / L + 2 components), then one atomic each
    const int k0 = cb / p.L;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gc = cb + wc_ * 64 + j * 32 + l31;
        if (gc < p.KL) atomicAdd(&sacc[gc / p.L - k0], red[j]);
    }
    __syncthreads();
    const int klast = cb + 255 < p.KL ? cb + 255 : p.KL - 1;
    const int nk = klast >= cb ? klast / p.L - k0 + 1 : 0;
    for (int t = tid; t < nk; t += GR_THREADS) {
        const double v = (double)sacc[t];
        if (lik == RR_LIK_GAUSSIAN) {
            unsafeAtomicAdd(&p.aux[k0 + t], v);
            unsafeAtomicAdd(&p.llsum[k0 + t], -0.5 * v * (double)ipar);
        } else {
            unsafeAtomicAdd(&p.llsum[k0 + t], v);
        }
    }
}

// T[i][f] += sum_r x[r][i] * (E[r][n+f] P[r][f] - E[r][f] P[r][n+f]):  sum(E o dPhi_i) = -(1/l_i^2) W[i,:].T[i,:]
template <int DMAX, typename TX>
__global__ void __launch_bounds__(256)
rr_glm_grad_t_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const float *__restrict__ P,
                     const float *__restrict__ E, int64_t ldp, int n, int d, double *__restrict__ T,
                     int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    const int fc = fvalid ? f : 0;
    float t[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) t[i] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const float a = E[r * ldp + n + fc] * P[r * ldp + fc] - E[r * ldp + fc] * P[r * ldp + n + fc];
        const TX *xr = X + r * ldx;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) t[i] = fmaf((float)xr[i], a, t[i]);
    }
    if (fvalid) {
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < d) unsafeAtomicAdd(&T[(size_t)i * n + f], (double)t[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// Opt-in device-side reparameterisation draws for the GLM step ("device-side RNG", the host-generated
// draws of rr_featmat_glm_step remain the parity route).  Counter-based: the normal for (step, sample kl, feature
// f) is a pure function of (seed, step, kl F + f), two SplitMix64 rounds + Box-Muller.
//   E[kl][f] = e;   WSs[kl][f] = (m[f][k] + sqrt(C[f][k]) e) / (K L),  k = kl / L
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rr_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256)
rr_glm_draw_kernel(const double *__restrict__ mdev, const double *__restrict__ Cdev, int F, int K, int L, int64_t Fp,
                   int64_t klp, uint64_t seed, uint64_t step, const float *__restrict__ Egiven, float *__restrict__ E,
                   float *__restrict__ WSs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= klp * Fp) return;
    const int64_t kl = i / Fp;
    const int f = (int)(i % Fp);
    float e = 0.f, w = 0.f;
    if (kl < (int64_t)K * L && f < F) {
        if (Egiven) {  // the caller's draws (K L, F): the parity route
            e = Egiven[(size_t)kl * F + f];
        } else {
            const uint64_t ctr = (uint64_t)kl * (uint64_t)F + (uint64_t)f;
            const uint64_t h = rr_splitmix64(rr_splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull)) ^ ctr);
            const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
            const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
            e = sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);  // cos of u2 revolutions
        }
        const int k = (int)(kl / L);
        w = (float)((mdev[(size_t)f * K + k] + sqrt(Cdev[(size_t)f * K + k]) * (double)e) / ((double)K * (double)L));
    }
    E[i] = e;
    WSs[i] = w;
}

// Edm[k][f] = sum_l Ed[kL + l][f] / L;   EdC[k][f] = sum_l Ed[kL + l][f] e[kL + l][f] / (L sqrt(C[f][k]))   (glm.py:309-310)
__global__ void __launch_bounds__(256)
rr_glm_reduce_kernel(const float *__restrict__ Ed, const float *__restrict__ E, const double *__restrict__ Cdev, int F, int K,
                     int L, int64_t Fp, double *__restrict__ Edm, double *__restrict__ EdC) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)K * F) return;
    const int k = (int)(i / F), f = (int)(i % F);
    double a = 0.0, b = 0.0;
    for (int l = 0; l < L; ++l) {
        const size_t o = (size_t)(k * L + l) * Fp + f;
        const double ed = (double)Ed[o];
        a += ed;
        b += ed * (double)E[o];
    }
    Edm[i] = a / L;
    EdC[i] = b / (L * sqrt(Cdev[(size_t)f * K + k]));
}

struct FmPass2 {
    float *Pt = nullptr, *U = nullptr, *C32 = nullptr, *m32 = nullptr, *dot = nullptr, *err = nullptr;
    double *sq = nullptr, *vf = nullptr;
    std::vector<float> hC, hm;
    bool have_rows = false;
    void *Ab = nullptr, *Cb = nullptr;  // split-bf16 engine: K-blocked copies of Pt and C32
    bool cb_ready = false;              // Cb matches C32
    bool tri_c = false;                 // prediction: C32 holds the upper-triangular form, the GEMM stops at the diagonal
    bool sq_form = false;               // prediction: C32 holds the factor M (C = M M^T): Vf = rowsum((P M)^2)
    // GLM step / projection: FSt (max_rows, klp), its transpose DFS (klp, max_rows), the sample matrices
    float *FSt = nullptr, *DFS = nullptr, *WSt = nullptr, *WSs = nullptr, *Ed = nullptr, *Ee = nullptr;
    double *mc = nullptr;  // [m (F K) | C (F K) | Edm (K F) | EdC (K F)] of the device-sampled step
    double *kacc = nullptr;  // [llsum (K) | aux (K)]
    int64_t klp = 0;
    int kcap = 0;
    bool have_edphi = false;
    // rr_featmat_pass2_plan_rff: children whose contraction the next rr_featmat_pass2_rows_planned may fuse into U = P C
    struct Pass2Plan {
        rr_basis *b;
        const void *dX;
        int x_dtype;
        int64_t ldx, col0;
        double *dT;
    };
    std::vector<Pass2Plan> plans, fused;  // armed for the next rows call / contracted by the last one
    // rr_featmat_glm_plan_rff: the next step's EdPhi product may contract itself with this child (rr_gemm_gradt_f32_kernel)
    struct {
        rr_basis *b = nullptr;
        const void *dX = nullptr;
        int64_t ldx = 0, col0 = 0;
        double *dT = nullptr;
        bool armed = false;  // planned, not yet consumed by a step
        bool done = false;   // the last step accumulated dT itself: rr_featmat_glm_rff for this child has nothing left to do
        bool take = false, take_lik = false;  // glm_pipeline's decisions of its first phase, for the later ones
    } fuse;
};

float *rr_fm_pass2_pt(void *p) { return p ? ((FmPass2 *)p)->Pt : nullptr; }

void rr_fm_pass2_free(void *p) {
    if (!p) return;
    FmPass2 *s = (FmPass2 *)p;
    void *q[] = {s->Pt, s->U, s->C32, s->m32, s->dot, s->err, s->sq, s->vf, s->FSt, s->DFS, s->WSt, s->WSs, s->Ed, s->Ee, s->mc,
                 s->Ab, s->Cb};
    for (void *x : q)
        if (x) (void)hipFree(x);
    delete s;
}

// Err-independent part for the rows currently in the matrix: dot = P m, Pt = P^T, U = P C.
static int fm_pass2_products(rr_featmat *fm, FmPass2 &s, double *rowsq = nullptr) {
    rr_ctx *c = fm->ctx;
    const int64_t rows256 = (fm->rows + 255) / 256 * 256;
    hipLaunchKernelGGL(rr_rowvec_kernel, dim3((unsigned)((fm->rows + 3) / 4)), dim3(256), 0, c->stream, fm->P, s.m32,
                       fm->rows, fm->F, fm->ld, s.dot);
    if (!(fm->pt_rows == fm->rows && fm->pt_covered >= fm->F)) {  // (else: the children wrote P^T next to P)
        hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(fm->ld / 64), (unsigned)(rows256 / 64)), dim3(256), 0,
                           c->stream, fm->P, fm->rows, fm->ld, s.Pt, fm->max_rows);
        fm->pt_rows = fm->rows;
    }
    if (c->gram_engine != 0) {  // split-bf16 engine (rr_rff.hip)
        if (!s.Ab) {
            RR_CHECK_HIP(hipMalloc(&s.Ab, (size_t)fm->ld * fm->max_rows * 4));
            RR_CHECK_HIP(hipMalloc(&s.Cb, (size_t)fm->ld * fm->ld * 4));
        }
        int rc = rr_launch_gemm_tn_bf16(c, c->gram_engine, s.Pt, fm->max_rows, s.C32, fm->ld, s.U, fm->ld, fm->ld, rows256,
                                        fm->ld, s.Ab, s.Cb, s.cb_ready, s.tri_c);
        s.cb_ready = true;
        return rc;
    }
    GemmArgs g;
    g.A = s.Pt; g.B = s.C32; g.D = s.U; g.lda = fm->max_rows; g.ldb = fm->ld; g.ldd = fm->ld;
    g.K = (int)(((int64_t)fm->F + GR_KB - 1) / GR_KB * GR_KB);  // the valid feature rows only: F = 8257 -> 8288 of the 8448 padded ones
    g.ntb = (int)(fm->ld / 256);
    g.upper_b = s.tri_c ? 1 : 0;
    g.rowsq = rowsq;
    g.rowsq_rows = fm->rows;
    static const bool no_pair = getenv("RR_PREDICT_NO_PAIR") != nullptr;
    g.pair_upper = (g.upper_b && !no_pair) ? 1 : 0;  // equal-cost workgroups: column tiles (q, ntb - 1 - q) together
    const int64_t wg_per_row_tile = g.pair_upper ? (g.ntb + 1) / 2 : g.ntb;
    if (g.pair_upper) hipLaunchKernelGGL(rr_gemm_pair_f32_kernel, dim3((unsigned)((rows256 / 256) * wg_per_row_tile)), dim3(GR_THREADS), 0, c->stream, g);
    else hipLaunchKernelGGL(rr_gemm_tn_f32_kernel, dim3((unsigned)((rows256 / 256) * wg_per_row_tile)), dim3(GR_THREADS), 0, c->stream, g);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

static int fm_pass2_scratch(rr_featmat *fm) {
    const int64_t Fp = fm->ld;
    // (rr_dma_kblock: 31 rows x 4 ld bytes of a k-block must stay inside a 32-bit scalar offset)
    RR_REQUIRE(fm->max_rows < (1 << 25) && Fp < (1 << 25),
               "feature matrix: second pass / GLM step support up to 2^25 rows per matrix (%lld): process the data in row chunks",
               (long long)fm->max_rows);
    if (!fm->pass2) {
        FmPass2 *s = new FmPass2();
        fm->pass2 = s;
        fm->pt_rows = -1;  // a fresh P^T: no padding laid out yet
        hipError_t ea = hipMalloc((void **)&s->Pt, (size_t)Fp * fm->max_rows * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->U, (size_t)fm->max_rows * Fp * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->C32, (size_t)Fp * Fp * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->m32, (size_t)Fp * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->dot, (size_t)fm->max_rows * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->err, (size_t)fm->max_rows * 4);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->sq, 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s->vf, (size_t)fm->max_rows * 8);
        if (ea != hipSuccess) {
            (void)hipGetLastError();
            rr_fm_pass2_free(s);
            fm->pass2 = nullptr;
            rr_set_error("featmat second pass: device allocation failed");
            return RR_ERR_OOM;
        }
    }
    return RR_OK;
}

// scratch of the GLM step / projection for klp (multiple of 256) sample columns and K components
static int fm_glm_scratch(rr_featmat *fm, int64_t klp, int K) {
    int rc = fm_pass2_scratch(fm);
    if (rc != RR_OK) return rc;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    if (s.klp >= klp && s.kcap >= K) return RR_OK;
    RR_CHECK_HIP(hipStreamSynchronize(fm->ctx->stream));
    if (klp < s.klp) klp = s.klp;  // grow-only in both dimensions
    if (K < s.kcap) K = s.kcap;
    void *q[] = {s.FSt, s.DFS, s.WSt, s.WSs, s.Ed, s.Ee, s.mc};  // (kacc lives behind mc)
    for (void *x : q)
        if (x) (void)hipFree(x);
    s.FSt = s.DFS = s.WSt = s.WSs = s.Ed = s.Ee = nullptr;
    s.kacc = nullptr;
    s.mc = nullptr;
    s.klp = 0;
    s.kcap = 0;
    hipError_t ea = hipMalloc((void **)&s.FSt, (size_t)fm->max_rows * klp * 4);
    if (ea == hipSuccess) ea = hipMalloc((void **)&s.DFS, (size_t)klp * fm->max_rows * 4);
    if (ea == hipSuccess) ea = hipMalloc((void **)&s.WSt, (size_t)fm->ld * klp * 4);
    if (ea == hipSuccess) ea = hipMalloc((void **)&s.WSs, (size_t)klp * fm->ld * 4);
    if (ea == hipSuccess) ea = hipMalloc((void **)&s.Ed, (size_t)klp * fm->ld * 4);
    if (ea == hipSuccess) ea = hipMalloc((void **)&s.Ee, (size_t)klp * fm->ld * 4);
    // [m | C | Edm | EdC] and, right behind them, [llsum | aux]: with K == kcap (one K per fit) everything a step sums over rows
    // besides dT is ONE contiguous run -- one all-reduce per step of a device group (rr_glm_sgd_group_step)
    const size_t kc = (size_t)(K > 1 ? K : 1);
    // (+ 2: [llconst | rows] of the minibatch when they are sums over RANKS -- rr_glm_sgd_dist_step)
    if (ea == hipSuccess) ea = hipMalloc((void **)&s.mc, (4 * kc * (size_t)fm->F + 2 * kc + 2) * 8);
    if (ea == hipSuccess) s.kacc = s.mc + 4 * kc * (size_t)fm->F;
    if (ea != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("featmat GLM step: device allocation failed");
        return RR_ERR_OOM;
    }
    s.klp = klp;
    s.kcap = K;
    return RR_OK;
}

// Small products (the reference's default minibatch of 10 rows: 256 x 1024 x 256 once padded).  A 256 x 256 tile of the kernel
// above is one workgroup walking its k-blocks at the rate of ONE CU -- 7 us per 32-row k-block, 60-70 us for K = 256, most of it
// spent on padding -- and the GLM step has three such products on its dependent chain (200 of its 340 us).  Here a workgroup
// owns a 32 x 32 block of D (2 x 2 per thread, k in steps of 32 through LDS, plain FMAs): hundreds of workgroups instead of a
// handful, a few microseconds.  Same operands, zero padding included; summation order differs from the tile kernel's.
__global__ void __launch_bounds__(256)
rr_gemm_tn_small_f32_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, float *__restrict__ D,
                            int64_t ldd, int K) {
    __shared__ float As[32][33], Bs[32][33];
    const int ti = blockIdx.y * 32, tj = blockIdx.x * 32;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * 256 + threadIdx.x, kk = e >> 5, cc = e & 31;
            As[kk][cc] = A[(int64_t)(k0 + kk) * lda + ti + cc];
            Bs[kk][cc] = B[(int64_t)(k0 + kk) * ldb + tj + cc];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float a0 = As[kk][2 * ty], a1 = As[kk][2 * ty + 1], b0 = Bs[kk][2 * tx], b1 = Bs[kk][2 * tx + 1];
            acc[0][0] = fmaf(a0, b0, acc[0][0]);
            acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]);
            acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) D[(int64_t)(ti + 2 * ty + i) * ldd + tj + 2 * tx + j] = acc[i][j];
}

// Products BETWEEN the two (late round 6): a GLM step on a minibatch of a few thousand rows -- D = 4096 x 512, K = 512 -- is 32
// tiles of 256 x 256: an eighth of the CUs walk 8-16 k-blocks each (73 us per product, 126 us for the EdPhi product with its
// contraction, for 2 GFLOP = 14 us of the matrix cores), and there is too much of it for the FMA kernel above.  Here a workgroup
// of four waves owns a 128 x 128 block (64 x 64 per wave: 2 x 2 v_mfma_f32_32x32x2f32 accumulators), k-blocks of 32 rows
// through registers into a double-buffered LDS tile [32][A 128 | B 128] -- rr_syrk_f32_small_kernel's loop with two operands
// -- two workgroups per CU (64 KiB of LDS each), K split over blockIdx.y into a zeroed D (f32 atomics) when the tiles alone do
// not fill the chip.  M, N % 128 == 0, K % 32 == 0, 16-byte aligned rows.
struct GemmMidArgs {
    const float *A, *B;
    float *D;
    int64_t lda, ldb, ldd;
    int K, ntb;          // ntb = N / 128 column tiles (fastest-varying in blockIdx.x)
    int kb_per_split;    // > 0: blockIdx.y owns that many k-blocks and ADDS
};

__global__ void __launch_bounds__(256, 2) rr_gemm_tn_mid_f32_kernel(const GemmMidArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2][32 * 256];  // 2 x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int64_t ca = (int64_t)(blockIdx.x / p.ntb) * 128, cb = (int64_t)(blockIdx.x % p.ntb) * 128;
    int kb0 = 0, nkb = p.K / 32;
    if (p.kb_per_split > 0) {
        kb0 = blockIdx.y * p.kb_per_split;
        nkb = nkb - kb0 < p.kb_per_split ? nkb - kb0 : p.kb_per_split;
    }
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // this thread's part of a k-block: rows lr + 8 u (u < 4), float4 number lc of the A side and of the B side
    const int lr = tid >> 5, lc = tid & 31;
    float4 ra[4], rb[4];
    auto gload = [&](int kb) {
        const int64_t r = (int64_t)(kb0 + kb) * 32 + lr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = *(const float4 *)(p.A + (r + 8 * u) * p.lda + ca + 4 * lc);
            rb[u] = *(const float4 *)(p.B + (r + 8 * u) * p.ldb + cb + 4 * lc);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *(float4 *)&lds[buf][(lr + 8 * u) * 256 + 4 * lc] = ra[u];
            *(float4 *)&lds[buf][(lr + 8 * u) * 256 + 128 + 4 * lc] = rb[u];
        }
    };
    if (nkb > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    const int i32 = lane & 31, kk = lane >> 5;
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) gload(kb + 1);   // in flight under this block's products
        const float *L = lds[cur];
#pragma unroll
        for (int k2 = 0; k2 < 32; k2 += 2) {
            const float *row = L + (k2 + kk) * 256;
            const float a0 = row[wr * 64 + i32], a1 = row[wr * 64 + 32 + i32];
            const float b0 = row[128 + wc * 64 + i32], b1 = row[128 + wc * 64 + 32 + i32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kb + 1 < nkb) lstore(cur ^ 1);
        __syncthreads();
    }
    // C/D of the 32x32 forms: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const bool atomic = p.kb_per_split > 0;  // (uniform)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t gc = cb + wc * 64 + 32 * j + i32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (atomic) unsafeAtomicAdd(&p.D[gr * p.ldd + gc], acc[i][j][e]);
                else p.D[gr * p.ldd + gc] = acc[i][j][e];
            }
        }
}

// (tiles of 256 x 256 would leave three quarters of the CUs without one, and K is short enough that splitting it cannot make up
// for that: with K = 16 384 the tile kernel's 16-32 K-splits of 32 k-blocks fill the chip at its better rate per flop -- a GLM
// step of 16 384 x 2048 measured 6 % slower with this kernel on its Ed product.  RR_GEMM_MID=0: the tile kernel -- A/B runs)
static inline bool fm_gemm_is_mid(rr_ctx *c, int64_t Kd, int64_t Md, int64_t Nd) {
    static const bool off = getenv("RR_GEMM_MID") != nullptr && atoi(getenv("RR_GEMM_MID")) == 0;
    // (not in deterministic mode: its K-split adds float32 atomics where the tile kernel had none)
    return !off && !c->deterministic && (Md / 256) * (Nd / 256) * 4 <= (int64_t)c->num_cu && Kd <= 4096 && Kd % 32 == 0 &&
           Md % 128 == 0 && Nd % 128 == 0;
}

static inline bool fm_gemm_is_small(int64_t Kd, int64_t Md, int64_t Nd) {
    static const bool off = getenv("RR_GEMM_SMALL") != nullptr && atoi(getenv("RR_GEMM_SMALL")) == 0;  // (A/B runs)
    return !off && Kd * Md * Nd <= ((int64_t)1 << 27) && Md / 32 < 65536;
}

static int fm_gemm(rr_ctx *c, const float *A, int64_t lda, const float *B, int64_t ldb, float *D, int64_t ldd, int64_t Kd,
                   int64_t Md, int64_t Nd, bool allow_mid = true) {
    RR_REQUIRE(lda < (1 << 25) && ldb < (1 << 25), "GEMM: leading dimensions up to 2^25 floats (rr_dma_kblock's 32-bit row offsets)");
    if (fm_gemm_is_small(Kd, Md, Nd)) {
        hipLaunchKernelGGL(rr_gemm_tn_small_f32_kernel, dim3((unsigned)(Nd / 32), (unsigned)(Md / 32)), dim3(256), 0, c->stream, A, lda, B,
                           ldb, D, ldd, (int)Kd);
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    if (allow_mid && fm_gemm_is_mid(c, Kd, Md, Nd)) {
        GemmMidArgs m;
        m.A = A; m.B = B; m.D = D; m.lda = lda; m.ldb = ldb; m.ldd = ldd; m.K = (int)Kd; m.ntb = (int)(Nd / 128); m.kb_per_split = 0;
        const int64_t tiles = (Md / 128) * m.ntb, nkb = Kd / 32;
        // K-splits for RR_GEMM_MID_ROUNDS (default 2: two workgroups fit a CU) workgroups per CU, at least 4 k-blocks each
        static const int rounds = getenv("RR_GEMM_MID_ROUNDS") ? atoi(getenv("RR_GEMM_MID_ROUNDS")) : 2;
        int64_t want = (rounds * (int64_t)c->num_cu + tiles - 1) / tiles;
        if (want > nkb / 4) want = nkb / 4;
        unsigned splits = 1;
        if (want > 1) {
            m.kb_per_split = (int)((nkb + want - 1) / want);
            splits = (unsigned)((nkb + m.kb_per_split - 1) / m.kb_per_split);
            RR_CHECK_HIP(hipMemsetAsync(D, 0, (size_t)Md * ldd * sizeof(float), c->stream));
        }
        hipLaunchKernelGGL(rr_gemm_tn_mid_f32_kernel, dim3((unsigned)tiles, splits), dim3(256), 0, c->stream, m);
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    GemmArgs g;
    g.A = A; g.B = B; g.D = D; g.lda = lda; g.ldb = ldb; g.ldd = ldd; g.K = (int)Kd; g.ntb = (int)(Nd / 256);
    const int64_t tiles = (Md / 256) * g.ntb, nkb = Kd / GR_KB;
    unsigned splits = 1;
    if (tiles < 2 * (int64_t)c->num_cu && nkb >= 16) {  // too few tiles to fill the chip: split K, >= 8 k-blocks each
        // workgroups for RR_GEMM_SPLIT_ROUNDS rounds over the CUs (default 1: one K-split per CU and half the atomic
        // flushes of two rounds -- config 5's Ed = dfs Phi: 1.17 -> 1.12 ms)
        const char *sr = getenv("RR_GEMM_SPLIT_ROUNDS");
        const int64_t rounds = sr && atoi(sr) >= 1 ? atoi(sr) : 1;
        int64_t want = (rounds * (int64_t)c->num_cu + tiles - 1) / tiles;
        if (want > nkb / 8) want = nkb / 8;
        g.kb_per_split = (int)((nkb + want - 1) / want);
        splits = (unsigned)((nkb + g.kb_per_split - 1) / g.kb_per_split);
        RR_CHECK_HIP(hipMemsetAsync(D, 0, (size_t)Md * ldd * sizeof(float), c->stream));
    }
    hipLaunchKernelGGL(rr_gemm_tn_f32_kernel, dim3((unsigned)tiles, splits), dim3(GR_THREADS), 0, c->stream, g);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_launch_gemm_tn_f32(rr_ctx *c, const float *A, int64_t lda, const float *B, int64_t ldb, float *D, int64_t ldd,
                          int64_t K, int64_t M, int64_t N) {  // D (M, N) = A^T B, A (K, M), B (K, N); M, N % 256 == 0, K % 32 == 0
    // (the phase matrix of a basis with Xdim > 128, in row sub-chunks: a row's phases must not depend on how many rows came
    // with it -- tests/test_gpu_large_xdim.py holds a sub-chunked transform to the one-pass one BIT FOR BIT -- and the 128 x 128
    // kernel rounds differently from the other two (3.7e-7), so it stays out of this product)
    return fm_gemm(c, A, lda, B, ldb, D, ldd, K, M, N, false);
}

// The same product with the random Fourier feature epilogue (GemmArgs: TRIG): P (nout rows, ldp) <- cos / sin of the
// phases A^T B, rows >= nvalid zero, bvec += P^T y (y: nvalid values, float or double; may be null).
int rr_launch_gemm_trig_f32(rr_ctx *c, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t K, int64_t M, int64_t N,
                            float *P, int64_t ldp, int n, float scale, int64_t nvalid, int64_t nout, const void *y, int y_f64,
                            double *bvec) {
    GemmArgs g;
    g.A = A; g.B = B; g.D = P; g.lda = lda; g.ldb = ldb; g.ldd = ldp; g.K = (int)K; g.ntb = (int)(N / 256);
    g.n = n; g.scale = scale; g.nvalid = nvalid; g.nout = nout; g.y = y; g.y_f64 = y_f64; g.bvec = bvec;
    hipLaunchKernelGGL(rr_gemm_trig_f32_kernel, dim3((unsigned)((M / 256) * g.ntb)), dim3(GR_THREADS), 0, c->stream, g);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------------------------
// The second pass in f64 arithmetic (dtype = "f64" bases): same data flow, f64 features (sincospi), f64 MFMA GEMM
// (rr_gemm_tn_f64_kernel, rr_rff.hip), f64 epilogues.  Small kernels are written for clarity, not tuned: the pass
// is bound by the GEMM.
// ---------------------------------------------------------------------------------------------
int rr_features_rowmajor_f64(rr_basis *b, const void *dX, int x_dtype, int64_t m, int64_t mpad, int64_t ldx,
                             double *P, int64_t ldp, bool zero_pad_cols = true);  // rr_rff.hip
int rr_launch_gemm_gradt_f64(rr_ctx *c, const double *Pt, int64_t lda, const double *C, int64_t ldb, int64_t K, int64_t mpad,
                             int64_t Fp, const double *P, int64_t ldp, const double *X, int64_t ldx, int64_t rows, int n, int d,
                             const double *err, const double *mvec, double *T);  // rr_rff.hip
int rr_launch_gemm_tn_f64(rr_ctx *c, const double *A, int64_t lda, const double *B, int64_t ldb, double *D, int64_t ldd,
                          int64_t K, int64_t M, int64_t N, int subtract, int upper_only);

__global__ void __launch_bounds__(256)
rr_transpose_f64_kernel(const double *__restrict__ P, int64_t rows, int64_t ldp, double *__restrict__ Pt, int64_t ldt) {
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t r = r0 + ty + 4 * k;
        tile[ty + 4 * k][tx] = r < rows ? P[r * ldp + c0 + tx] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) Pt[(c0 + ty + 4 * k) * ldt + r0 + tx] = tile[tx][ty + 4 * k];
}

// one wave per row: dot[r] = P[r] . m;  MODE 1 additionally out[r] = U[r] . P[r]
template <int MODE>
__global__ void __launch_bounds__(256)
rr_rows64_kernel(const double *__restrict__ P, const double *__restrict__ U, const double *__restrict__ mvec, int64_t rows,
                 int F, int64_t ld, double *__restrict__ dot, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const double *q = P + r * ld;
    double a0 = 0.0, a1 = 0.0;
    for (int j = lane; j < F; j += 64) {
        a0 = fma(q[j], mvec[j], a0);
        if (MODE == 1) a1 = fma(q[j], U[r * ld + j], a1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64);
        if (MODE == 1) a1 += __shfl_down(a1, o, 64);
    }
    if (lane == 0) {
        dot[r] = a0;
        if (MODE == 1) out[r] = a1;
    }
}

template <typename TX>
__global__ void __launch_bounds__(256)
rr_err64_kernel(const TX *__restrict__ y, const double *__restrict__ dot, int64_t N, double *__restrict__ err,
                double *__restrict__ sq, int64_t det = 0) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double e = 0.0;
    if (r < N) {
        e = (double)y[r] - dot[r];
        err[r] = e;
    }
    double acc = e * e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rr_acc_out(sq, det, blockIdx.x, 0, part[0] + part[1] + part[2] + part[3]);
}

template <int DMAX, typename TX>
__global__ void __launch_bounds__(256)
rr_grad_t64_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const double *__restrict__ P,
                   const double *__restrict__ U, int64_t ldp, const double *__restrict__ err,
                   const double *__restrict__ mvec, int n, int d, double *__restrict__ T, int rows_per_block, int64_t tdet = 0) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    const int fc = fvalid ? f : 0;
    const double mc = mvec[fc], ms = mvec[n + fc];
    double t[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) t[i] = 0.0;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const double pc = P[r * ldp + fc], ps = P[r * ldp + n + fc];
        const double uc = U[r * ldp + fc], us = U[r * ldp + n + fc];
        const double a = err[r] * (pc * ms - ps * mc) - (pc * us - ps * uc);
        const TX *xr = X + r * ldx;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) t[i] = fma((double)xr[i], a, t[i]);
    }
    if (fvalid) {
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < d) rr_acc_out(T, tdet, blockIdx.y, (int64_t)i * n + f, t[i]);
    }
}

// Cp (Fp, Fp) zero padded <- C (F, F), both f64 on the device
__global__ void __launch_bounds__(256)
rr_pad_c64_kernel(const double *__restrict__ C, int64_t F, double *__restrict__ Cp, int64_t Fp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Fp * Fp) return;
    const int64_t r = i / Fp, c = i % Fp;
    Cp[i] = (r < F && c < F) ? C[r * F + c] : 0.0;
}

struct Pass2Scratch64 {
    double *P = nullptr, *Pt = nullptr, *U = nullptr, *Cp = nullptr, *Craw = nullptr, *m = nullptr, *dot = nullptr,
           *err = nullptr, *acc = nullptr;
    int64_t chunk = 0, Fp = 0;
    size_t nacc = 0;
    void release() {
        void *q[] = {P, Pt, U, Cp, Craw, m, dot, err, acc};
        for (void *x : q)
            if (x) (void)hipFree(x);
        P = Pt = U = Cp = Craw = m = dot = err = acc = nullptr;
        chunk = Fp = 0;
        nacc = 0;
    }
};

void rr_pass2d_scratch_free(void *p) {
    if (!p) return;
    Pass2Scratch64 *s = (Pass2Scratch64 *)p;
    s->release();
    delete s;
}

template <typename TX>
static int pass2_run64(rr_basis *b, bool pred, const TX *dX, const TX *dy, int64_t N, int64_t ldx, const double *mh,
                       const double *Ch, double *out0, double *out1, bool c_on_device) {
    rr_ctx *c = b->ctx;
    const int F = 2 * b->n, n = b->n;
    const int64_t Fp = ((int64_t)F + 127) / 128 * 128;
    int64_t chunk = (int64_t)(((size_t)24 << 30) / ((size_t)24 * Fp));  // P, Pt, U in f64
    const char *cenv = getenv("RR_PASS2_CHUNK_ROWS");
    if (cenv && atoll(cenv) >= 128) chunk = atoll(cenv);
    else if (N > chunk) {  // equal chunks (see the f32 driver)
        const int64_t nchunks = (N + chunk - 1) / chunk;
        chunk = (N + nchunks - 1) / nchunks;
    }
    if (chunk > N) chunk = N;
    chunk = (chunk + 127) / 128 * 128;
    if (!b->pass2d) b->pass2d = new Pass2Scratch64();
    Pass2Scratch64 &s = *(Pass2Scratch64 *)b->pass2d;
    const size_t nacc = pred ? (size_t)chunk : (size_t)1 + (size_t)b->d * n;
    if (s.chunk < chunk || s.Fp != Fp || s.nacc < nacc) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        s.release();
        hipError_t ea = hipMalloc((void **)&s.P, (size_t)chunk * Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Pt, (size_t)Fp * chunk * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.U, (size_t)chunk * Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Cp, (size_t)Fp * Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.Craw, (size_t)F * F * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.m, (size_t)Fp * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.dot, (size_t)chunk * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.err, (size_t)chunk * 8);
        if (ea == hipSuccess) ea = hipMalloc((void **)&s.acc, nacc * 8);
        if (ea != hipSuccess) {
            (void)hipGetLastError();
            s.release();
            rr_set_error("pass2 (f64): device allocation failed (%lld rows per chunk)", (long long)chunk);
            return RR_ERR_OOM;
        }
        s.chunk = chunk;
        s.Fp = Fp;
        s.nacc = nacc;
    }
    const int64_t step = chunk;  // rows per launch (see the f32 driver)
    chunk = s.chunk;
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemset(s.m, 0, (size_t)Fp * 8));
    RR_CHECK_HIP(hipMemcpy(s.m, mh, (size_t)F * 8, hipMemcpyHostToDevice));
    const double *Csrc = Ch;
    if (!c_on_device) {
        RR_CHECK_HIP(hipMemcpy(s.Craw, Ch, (size_t)F * F * 8, hipMemcpyHostToDevice));
        Csrc = s.Craw;
    }
    hipLaunchKernelGGL(rr_pad_c64_kernel, dim3((unsigned)((Fp * Fp + 255) / 256)), dim3(256), 0, c->stream, Csrc, (int64_t)F,
                       s.Cp, Fp);
    if (!pred) RR_CHECK_HIP(hipMemsetAsync(s.acc, 0, nacc * 8, c->stream));
    RR_CHECK_HIP(hipGetLastError());
    int rc = RR_OK;
    // (opt-in, RR_PASS2_FUSE_F64=1: measured equal to the stored route -- 103.1 vs 102.6 ms per 200 000 x 4096 second pass; the
    // float64 tile loop runs two workgroups per CU, which already hide each other's store epilogues and the 2.3 ms
    // contraction pass is paid back by the fused epilogue's exposed loads)
    const char *fz64 = getenv("RR_PASS2_FUSE_F64");
    const bool fuse_t64 = !pred && sizeof(TX) == 8 && !c->deterministic && !b->large && n % 128 == 0 && b->d <= 32 &&
                          fz64 && atoi(fz64) != 0;
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += step) {
        const int64_t mrows = (N - r0 < step) ? N - r0 : step;
        const int64_t mpad = (mrows + 127) / 128 * 128;
        const TX *Xc = dX + r0 * ldx;
        rc = rr_features_rowmajor_f64(b, Xc, sizeof(TX) == 4 ? RR_F32 : RR_F64, mrows, mpad, ldx, s.P, Fp);
        if (rc != RR_OK) break;
        hipLaunchKernelGGL(rr_transpose_f64_kernel, dim3((unsigned)(Fp / 64), (unsigned)(mpad / 64)), dim3(256), 0, c->stream,
                           s.P, mrows, Fp, s.Pt, chunk);
        // The gradient pass with whole [cos | sin] tiles, float64 X: Phi m and Err first, then U = Phi C contracts itself with
        // Phi, Err m^T and X in registers (rr_gemm_gradt_f64_kernel, rr_rff.hip) -- U is neither written nor read back.
        if (fuse_t64) {
            hipLaunchKernelGGL(rr_rows64_kernel<0>, dim3((unsigned)((mrows + 3) / 4)), dim3(256), 0, c->stream, s.P, s.U, s.m,
                               mrows, F, Fp, s.dot, s.acc);
            hipLaunchKernelGGL(rr_err64_kernel<TX>, dim3((unsigned)((mrows + 255) / 256)), dim3(256), 0, c->stream, dy + r0, s.dot,
                               mrows, s.err, s.acc);
            rc = rr_launch_gemm_gradt_f64(c, s.Pt, chunk, s.Cp, Fp, ((int64_t)F + 15) / 16 * 16, mpad, Fp, s.P, Fp, (const double *)Xc,
                                          ldx, mrows, n, b->d, s.err, s.m, s.acc + 1);
            if (rc != RR_OK) break;
            RR_CHECK_HIP(hipStreamSynchronize(c->stream));
            continue;
        }
        rc = rr_launch_gemm_tn_f64(c, s.Pt, chunk, s.Cp, Fp, s.U, Fp, ((int64_t)F + 15) / 16 * 16, mpad, Fp, 0, 0);
        if (rc != RR_OK) break;
        if (pred) {
            hipLaunchKernelGGL(rr_rows64_kernel<1>, dim3((unsigned)((mrows + 3) / 4)), dim3(256), 0, c->stream, s.P, s.U, s.m,
                               mrows, F, Fp, s.dot, s.acc);
            RR_CHECK_HIP(hipMemcpyAsync(out0 + r0, s.dot, (size_t)mrows * 8, hipMemcpyDeviceToHost, c->stream));
            RR_CHECK_HIP(hipMemcpyAsync(out1 + r0, s.acc, (size_t)mrows * 8, hipMemcpyDeviceToHost, c->stream));
            RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        } else {
            hipLaunchKernelGGL(rr_rows64_kernel<0>, dim3((unsigned)((mrows + 3) / 4)), dim3(256), 0, c->stream, s.P, s.U, s.m,
                               mrows, F, Fp, s.dot, s.acc);
            const int64_t eb = (mrows + 255) / 256;
            if (c->deterministic) {
                void *part = nullptr;
                rc = rr_det_scratch(c, (size_t)eb * 8, &part);
                if (rc != RR_OK) break;
                hipLaunchKernelGGL(rr_err64_kernel<TX>, dim3((unsigned)eb), dim3(256), 0, c->stream, dy + r0, s.dot, mrows, s.err,
                                   (double *)part, (int64_t)1);
                rc = rr_det_reduce(c, (const double *)part, eb, 1, 1, s.acc);
                if (rc != RR_OK) break;
            } else {
                hipLaunchKernelGGL(rr_err64_kernel<TX>, dim3((unsigned)eb), dim3(256), 0, c->stream, dy + r0, s.dot, mrows, s.err,
                                   s.acc);
            }
            const int fblocks = (n + 255) / 256;
            int64_t rpb = (mrows * fblocks + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
            if (rpb < 64) rpb = 64;
            if ((mrows + rpb - 1) / rpb > 65535) rpb = (mrows + 65534) / 65535;
            const dim3 grid(fblocks, (unsigned)((mrows + rpb - 1) / rpb));
            const int64_t tcount = (int64_t)b->d * n, tdet = c->deterministic ? tcount : 0;
            double *Tdst = s.acc + 1;
            if (tdet) {
                if (b->large) {
                    rr_set_error("second pass: deterministic mode does not cover Xdim > 128");
                    rc = RR_ERR_UNSUPPORTED;
                    break;
                }
                void *part = nullptr;
                rc = rr_det_scratch(c, (size_t)grid.y * (size_t)tcount * 8, &part);
                if (rc != RR_OK) break;
                Tdst = (double *)part;
            }
#define RR_GT64(DM)                                                                                               \
    hipLaunchKernelGGL((rr_grad_t64_kernel<DM, TX>), grid, dim3(256), 0, c->stream, Xc, mrows, ldx, s.P, s.U, Fp, s.err, \
                       s.m, n, b->d, Tdst, (int)rpb, tdet)
            switch (b->dpad) {
                case 8: RR_GT64(8); break;
                case 16: RR_GT64(16); break;
                case 32: RR_GT64(32); break;
                case 64: RR_GT64(64); break;
                case 128: RR_GT64(128); break;
                default:
                    for (int i0 = 0; i0 < b->d; i0 += 128)
                        hipLaunchKernelGGL((rr_grad_t64_kernel<128, TX>), grid, dim3(256), 0, c->stream, Xc + i0, mrows, ldx, s.P,
                                           s.U, Fp, s.err, s.m, n, b->d - i0 < 128 ? b->d - i0 : 128,
                                           s.acc + 1 + (size_t)i0 * n, (int)rpb);
            }
#undef RR_GT64
            RR_CHECK_HIP(hipGetLastError());
            if (tdet) {
                rc = rr_det_reduce(c, Tdst, grid.y, tcount, tcount, s.acc + 1);
                if (rc != RR_OK) break;
            }
            RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        }
    }
    if (rc == RR_OK && !pred) {
        std::vector<double> acc(nacc);
        RR_CHECK_HIP(hipMemcpy(acc.data(), s.acc, nacc * 8, hipMemcpyDeviceToHost));
        *out0 = acc[0];
        memcpy(out1, acc.data() + 1, (nacc - 1) * 8);
    }
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

// rr_grad_contract_kernel from finished features (Xdim > 128): a = E_s P_c - E_c P_s, T[i][f] += sum_r x[r][i] a
template <int DMAX, typename TX, typename TE>
__global__ void __launch_bounds__(256)
rr_grad_contract_p_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const float *__restrict__ P, int64_t ldp,
                          const TE *__restrict__ E, int64_t lde, int n, int d, double *__restrict__ T, int rows_per_block,
                          int64_t tdet = 0) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    const int fc = fvalid ? f : 0;
    float t[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) t[i] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const float a = (float)E[r * lde + n + fc] * P[r * ldp + fc] - (float)E[r * lde + fc] * P[r * ldp + n + fc];
        const TX *xr = X + r * ldx;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) t[i] = fmaf((float)xr[i], a, t[i]);
    }
    if (fvalid) {
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < d) rr_acc_out(T, tdet, blockIdx.y, (int64_t)i * n + f, (double)t[i]);
    }
}

static int large_scratch(rr_basis *b, size_t bytes) {  // the Gram feature scratch, reused (grow-only)
    if (b->zbuf_bytes >= bytes) return RR_OK;
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    if (b->zbuf) (void)hipFree(b->zbuf);
    b->zbuf = nullptr;
    b->zbuf_bytes = 0;
    RR_CHECK_HIP(hipMalloc(&b->zbuf, bytes));
    b->zbuf_bytes = bytes;
    return RR_OK;
}

template <typename TX, typename TE>
static int launch_grad_contract(rr_basis *b, const TX *dX, int64_t N, int64_t ldx, const TE *dE, int64_t lde, double *dT) {
    rr_ctx *c = b->ctx;
    const float scale = (float)(1.0 / sqrt((double)b->n));
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = (N * fblocks + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
    if (rpb < 32) rpb = 32;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((N + rpb - 1) / rpb));
    const int64_t tcount = (int64_t)b->d * b->n, tdet = c->deterministic ? tcount : 0;
    double *Tacc = dT;
    if (tdet) {
        RR_REQUIRE(!b->large, "rr_rff_grad_contract: deterministic mode does not cover Xdim > 128");
        void *part = nullptr;
        int rc = rr_det_scratch(c, (size_t)grid.y * (size_t)tcount * 8, &part);
        if (rc != RR_OK) return rc;
        dT = (double *)part;
    }
    if (b->phase64) {  // RR_F32P64: features by the float64-phase kernel first (row-major f32 scratch), then the contraction
        const int64_t ldp = 2 * (int64_t)b->n;
        const int64_t Npad = (N + 31) / 32 * 32;
        int rc = large_scratch(b, (size_t)Npad * ldp * 4);
        if (rc == RR_OK)
            rc = rr_features_rowmajor_f32(b, dX, sizeof(TX) == 4 ? RR_F32 : RR_F64, N, N, ldx, (float *)b->zbuf, ldp, false);
        if (rc != RR_OK) return rc;
#define RR_GCP(DM)                                                                                                  \
    hipLaunchKernelGGL((rr_grad_contract_p_kernel<DM, TX, TE>), grid, dim3(256), 0, c->stream, dX, N, ldx,            \
                       (const float *)b->zbuf, ldp, dE, lde, b->n, b->d, dT, (int)rpb, tdet)
        switch (b->dpad) {
            case 8: RR_GCP(8); break;
            case 16: RR_GCP(16); break;
            case 32: RR_GCP(32); break;
            case 64: RR_GCP(64); break;
            default: RR_GCP(128); break;
        }
#undef RR_GCP
        RR_CHECK_HIP(hipGetLastError());
        return tdet ? rr_det_reduce(c, dT, grid.y, tcount, tcount, Tacc) : RR_OK;
    }
#define RR_GC(DM)                                                                                              \
    hipLaunchKernelGGL((rr_grad_contract_kernel<DM, TX, TE>), grid, dim3(256), 0, c->stream, dX, N, ldx, b->dWs32, \
                       dE, lde, b->n, b->npad, b->d, dT, scale, (int)rpb, tdet)
    switch (b->dpad) {
        case 8: RR_GC(8); break;
        case 16: RR_GC(16); break;
        case 32: RR_GC(32); break;
        case 64: RR_GC(64); break;
        case 128: RR_GC(128); break;
        default: {  // Xdim > 128: features once (row-major f32 scratch), then 128 input dimensions per launch
            const int64_t ldp = 2 * (int64_t)b->n;
            int rc = large_scratch(b, (size_t)N * ldp * 4);
            if (rc == RR_OK)
                rc = rr_features_rowmajor_f32(b, dX, sizeof(TX) == 4 ? RR_F32 : RR_F64, N, N, ldx, (float *)b->zbuf, ldp, false);
            if (rc != RR_OK) return rc;
            for (int i0 = 0; i0 < b->d; i0 += 128)
                hipLaunchKernelGGL((rr_grad_contract_p_kernel<128, TX, TE>), grid, dim3(256), 0, c->stream, dX + i0, N, ldx,
                                   (const float *)b->zbuf, ldp, dE, lde, b->n, b->d - i0 < 128 ? b->d - i0 : 128,
                                   dT + (size_t)i0 * b->n, (int)rpb);
        }
    }
#undef RR_GC
    RR_CHECK_HIP(hipGetLastError());
    return tdet ? rr_det_reduce(c, dT, grid.y, tcount, tcount, Tacc) : RR_OK;
}

template <typename TY>
static void glm_launch_lik(rr_ctx *c, int lik, float *FSt, int64_t M, int64_t rows256, int64_t klp, const void *dy,
                           const void *drow, float par, int KL, int L, double *llsum, double *aux, const double *par_dev = nullptr) {
    int64_t rpb = (rows256 * (klp / 256) + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
    if (rpb < 16) rpb = 16;
    if ((rows256 + rpb - 1) / rpb > 65535) rpb = (rows256 + 65534) / 65535;
    const dim3 grid((unsigned)(klp / 256), (unsigned)((rows256 + rpb - 1) / rpb));
#define RR_LK(ID)                                                                                                   \
    hipLaunchKernelGGL((rr_glm_lik_kernel<ID, TY>), grid, dim3(256), 0, c->stream, FSt, M, rows256, klp, (const TY *)dy, \
                       (const TY *)drow, par, (float)KL, KL, L, llsum, aux, (int)rpb, par_dev)
    switch (lik) {
        case RR_LIK_BERNOULLI: RR_LK(RR_LIK_BERNOULLI); break;
        case RR_LIK_BINOMIAL: RR_LK(RR_LIK_BINOMIAL); break;
        case RR_LIK_GAUSSIAN: RR_LK(RR_LIK_GAUSSIAN); break;
        case RR_LIK_POISSON_EXP: RR_LK(RR_LIK_POISSON_EXP); break;
        default: RR_LK(RR_LIK_POISSON_SOFTPLUS); break;
    }
#undef RR_LK
}

// rows of a host feature matrix (f32 | f64) into the zero-padded f64 layout of the f64 GEMM
template <typename TS>
__global__ void __launch_bounds__(256)
rr_pad_rows64_kernel(const TS *__restrict__ src, int64_t rows, int F, double *__restrict__ dst, int64_t ldp, int64_t rows_pad) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ldp) return;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    for (int64_t r = r0; r < r0 + 64 && r < rows_pad; ++r)
        dst[r * ldp + c] = (r < rows && c < F) ? (double)src[r * (int64_t)F + c] : 0.0;
}

extern "C" {

int rr_rff_grad_contract(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                         int n_ls, const void *E, int e_dtype, int64_t lde, double *T) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "rr_rff_grad_contract: not an RFF basis");
    RR_REQUIRE((x_dtype == RR_F32 || x_dtype == RR_F64) && (e_dtype == RR_F32 || e_dtype == RR_F64),
               "rr_rff_grad_contract: bad dtype");
    RR_REQUIRE(N >= 0 && ldx >= b->d && lde >= 2 * (int64_t)b->n && T != nullptr, "rr_rff_grad_contract: bad argument");
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    const size_t tcount = (size_t)b->d * b->n;
    memset(T, 0, tcount * sizeof(double));
    if (N == 0) return RR_OK;
    RR_REQUIRE(X != nullptr && E != nullptr, "rr_rff_grad_contract: null buffer");
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t xs = x_dtype == RR_F32 ? 4 : 8, es = e_dtype == RR_F32 ? 4 : 8;
    const int64_t F = 2 * (int64_t)b->n;
    int64_t chunk = (int64_t)(((size_t)1 << 30) / ((size_t)b->dpad * xs + (size_t)F * es));
    if (chunk < 1) chunk = 1;
    if (chunk > N) chunk = N;
    void *dX = nullptr, *dE = nullptr;
    double *dT = nullptr;
    hipError_t e = hipMalloc(&dX, (size_t)chunk * b->dpad * xs);
    if (e == hipSuccess) e = hipMalloc(&dE, (size_t)chunk * F * es);
    if (e == hipSuccess) e = hipMalloc((void **)&dT, tcount * 8);
    if (e == hipSuccess) e = hipMemsetAsync(dX, 0, (size_t)chunk * b->dpad * xs, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dT, 0, tcount * 8, c->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_rff_grad_contract: device allocation failed");
        rc = RR_ERR_OOM;
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += chunk) {
        const int64_t m = (N - r0 < chunk) ? N - r0 : chunk;
        e = hipMemcpy2DAsync(dX, (size_t)b->dpad * xs, (const char *)X + (size_t)r0 * ldx * xs, (size_t)ldx * xs,
                             (size_t)b->d * xs, (size_t)m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(dE, (size_t)F * es, (const char *)E + (size_t)r0 * lde * es, (size_t)lde * es,
                                 (size_t)F * es, (size_t)m, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_grad_contract: upload failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        if (x_dtype == RR_F32)
            rc = e_dtype == RR_F32 ? launch_grad_contract<float, float>(b, (const float *)dX, m, b->dpad, (const float *)dE, F, dT)
                                   : launch_grad_contract<float, double>(b, (const float *)dX, m, b->dpad, (const double *)dE, F, dT);
        else
            rc = e_dtype == RR_F32 ? launch_grad_contract<double, float>(b, (const double *)dX, m, b->dpad, (const float *)dE, F, dT)
                                   : launch_grad_contract<double, double>(b, (const double *)dX, m, b->dpad, (const double *)dE, F, dT);
        if (rc == RR_OK && (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            rr_set_error("rr_rff_grad_contract: kernel failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    if (rc == RR_OK && hipMemcpy(T, dT, tcount * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        rr_set_error("rr_rff_grad_contract: download failed");
        rc = RR_ERR_HIP;
    }
    if (dX) (void)hipFree(dX);
    if (dE) (void)hipFree(dE);
    if (dT) (void)hipFree(dT);
    return rc;
}

int rr_rff_elbo_pass2_dev(rr_basis *b, const void *dX, const void *dy, int x_dtype, int64_t N, int64_t ldx,
                          const double *lenscale, int n_ls, const double *m, const double *C, double *sqerr,
                          double *T) {
    int rc = pass2_checks(b, dX, x_dtype, N, ldx, lenscale, n_ls, m, C, "rr_rff_elbo_pass2_dev");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(dy != nullptr && sqerr != nullptr && T != nullptr, "rr_rff_elbo_pass2_dev: null argument");
    if (b->compute == RR_F64)
        return x_dtype == RR_F32
                   ? pass2_run64<float>(b, false, (const float *)dX, (const float *)dy, N, ldx, m, C, sqerr, T, false)
                   : pass2_run64<double>(b, false, (const double *)dX, (const double *)dy, N, ldx, m, C, sqerr, T, false);
    return x_dtype == RR_F32 ? pass2_run<float>(b, false, (const float *)dX, (const float *)dy, N, ldx, m, C, sqerr, T)
                             : pass2_run<double>(b, false, (const double *)dX, (const double *)dy, N, ldx, m, C, sqerr, T);
}

int rr_rff_elbo_pass2_devc(rr_basis *b, const void *dX, const void *dy, int x_dtype, int64_t N, int64_t ldx,
                           const double *lenscale, int n_ls, const double *m, const double *dC, double *sqerr,
                           double *T) {
    int rc = pass2_checks(b, dX, x_dtype, N, ldx, lenscale, n_ls, m, dC, "rr_rff_elbo_pass2_devc");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(dy != nullptr && sqerr != nullptr && T != nullptr, "rr_rff_elbo_pass2_devc: null argument");
    if (b->compute == RR_F64)
        return x_dtype == RR_F32
                   ? pass2_run64<float>(b, false, (const float *)dX, (const float *)dy, N, ldx, m, dC, sqerr, T, true)
                   : pass2_run64<double>(b, false, (const double *)dX, (const double *)dy, N, ldx, m, dC, sqerr, T, true);
    return x_dtype == RR_F32
               ? pass2_run<float>(b, false, (const float *)dX, (const float *)dy, N, ldx, m, dC, sqerr, T, true)
               : pass2_run<double>(b, false, (const double *)dX, (const double *)dy, N, ldx, m, dC, sqerr, T, true);
}

int rr_rff_predict_dev(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                       int n_ls, const double *m, const double *C, double *Ey, double *Vf) {
    int rc = pass2_checks(b, dX, x_dtype, N, ldx, lenscale, n_ls, m, C, "rr_rff_predict_dev");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(Ey != nullptr && Vf != nullptr, "rr_rff_predict_dev: null argument");
    if (b->compute == RR_F64)
        return x_dtype == RR_F32 ? pass2_run64<float>(b, true, (const float *)dX, nullptr, N, ldx, m, C, Ey, Vf, false)
                                 : pass2_run64<double>(b, true, (const double *)dX, nullptr, N, ldx, m, C, Ey, Vf, false);
    return x_dtype == RR_F32 ? pass2_run<float>(b, true, (const float *)dX, nullptr, N, ldx, m, C, Ey, Vf)
                             : pass2_run<double>(b, true, (const double *)dX, nullptr, N, ldx, m, C, Ey, Vf);
}

int rr_rff_predict_devc(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                        int n_ls, const double *m, const double *dC, double *Ey, double *Vf) {
    int rc = pass2_checks(b, dX, x_dtype, N, ldx, lenscale, n_ls, m, dC, "rr_rff_predict_devc");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(Ey != nullptr && Vf != nullptr, "rr_rff_predict_devc: null argument");
    if (b->compute == RR_F64)
        return x_dtype == RR_F32 ? pass2_run64<float>(b, true, (const float *)dX, nullptr, N, ldx, m, dC, Ey, Vf, true)
                                 : pass2_run64<double>(b, true, (const double *)dX, nullptr, N, ldx, m, dC, Ey, Vf, true);
    return x_dtype == RR_F32 ? pass2_run<float>(b, true, (const float *)dX, nullptr, N, ldx, m, dC, Ey, Vf, true)
                             : pass2_run<double>(b, true, (const double *)dX, nullptr, N, ldx, m, dC, Ey, Vf, true);
}

int rr_rff_predict_devb(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale, int n_ls,
                        const double *m, const float *dB, int form, double *Ey, double *Vf) {
    int rc = pass2_checks(b, dX, x_dtype, N, ldx, lenscale, n_ls, m, (const double *)dB, "rr_rff_predict_devb");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(Ey != nullptr && Vf != nullptr && (form == 0 || form == 1), "rr_rff_predict_devb: bad argument");
    RR_REQUIRE(b->compute == RR_F32, "rr_rff_predict_devb: float64 bases keep the float64 quadratic form (rr_rff_predict_devc)");
    return x_dtype == RR_F32
               ? pass2_run<float>(b, true, (const float *)dX, nullptr, N, ldx, m, nullptr, Ey, Vf, true, dB, form)
               : pass2_run<double>(b, true, (const double *)dX, nullptr, N, ldx, m, nullptr, Ey, Vf, true, dB, form);
}

int rr_rff_predict_mean_dev(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale, int n_ls,
                            const double *m, double *Ey) {
    int rc = pass2_checks(b, dX, x_dtype, N, ldx, lenscale, n_ls, m, m, "rr_rff_predict_mean_dev");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(Ey != nullptr, "rr_rff_predict_mean_dev: null argument");
    if (b->compute != RR_F32 || b->large || b->phase64) {
        rr_set_error("rr_rff_predict_mean_dev: f32 bases of Xdim <= 128 only (the others: rr_rff_predict_dev)");
        return RR_ERR_UNSUPPORTED;
    }
    rr_ctx *c = b->ctx;
    const int F = 2 * b->n;
    const int64_t Npad = (N + 255) / 256 * 256;
    // [m (F) | Phi m (Npad)] in the basis' grow-only scratch: a hipMalloc / hipFree pair per call would synchronise the
    // whole device -- other contexts' streams too -- on the one route meant for small, frequent queries
    if (!b->pass2) b->pass2 = new Pass2Scratch();
    Pass2Scratch &ps = *(Pass2Scratch *)b->pass2;
    const size_t want = (size_t)F + (size_t)Npad;
    if (ps.mean_count < want) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (ps.mean) (void)hipFree(ps.mean);
        ps.mean = nullptr;
        ps.mean_count = 0;
        const size_t grow = want + want / 4;
        if (hipMalloc((void **)&ps.mean, grow * 4) != hipSuccess) {
            (void)hipGetLastError();
            rr_set_error("rr_rff_predict_mean_dev: device allocation failed");
            return RR_ERR_OOM;
        }
        ps.mean_count = grow;
    }
    float *buf = ps.mean;
    std::vector<float> hmv((size_t)F), h((size_t)N);  // separate staging: the weights need not have landed before the result
    for (int i = 0; i < F; ++i) hmv[i] = (float)m[i];
    hipError_t e = hipMemcpyAsync(buf, hmv.data(), (size_t)F * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = x_dtype == RR_F32 ? launch_features_t<float>(b, (const float *)dX, N, Npad, ldx, buf, nullptr, 0, buf + F)
                               : launch_features_t<double>(b, (const double *)dX, N, Npad, ldx, buf, nullptr, 0, buf + F);
        if (rc == RR_OK) e = hipMemcpyAsync(h.data(), buf + F, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    if (rc != RR_OK) return rc;
    if (e != hipSuccess) {
        rr_set_error("rr_rff_predict_mean_dev: %s", hipGetErrorString(e));
        return RR_ERR_HIP;
    }
    for (int64_t i = 0; i < N; ++i) Ey[i] = (double)h[i];
    return RR_OK;
}

int rr_dense_predict(rr_ctx *c, const void *Phi, int dtype, int64_t N, int64_t F, int64_t ldphi, const double *m,
                     const double *C, double *Ey, double *Vf) {
    RR_REQUIRE(c != nullptr && m != nullptr && C != nullptr && Ey != nullptr && Vf != nullptr, "rr_dense_predict: null argument");
    RR_REQUIRE(dtype == RR_F32 || dtype == RR_F64, "rr_dense_predict: bad dtype");
    RR_REQUIRE(N >= 0 && F >= 1 && ldphi >= F && F < 46340, "rr_dense_predict: bad shape");
    if (N == 0) return RR_OK;
    RR_REQUIRE(Phi != nullptr, "rr_dense_predict: null Phi");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = dtype == RR_F32 ? 4 : 8;
    const int64_t Fp = (F + 127) / 128 * 128;
    int64_t chunk = (int64_t)(((size_t)2 << 30) / ((size_t)32 * Fp));  // raw rows + P, Pt, U in f64: ~2 GiB
    if (chunk > N) chunk = N;
    chunk = (chunk + 127) / 128 * 128;
    void *raw = nullptr;
    double *P = nullptr, *Pt = nullptr, *U = nullptr, *Cp = nullptr, *Craw = nullptr, *mv = nullptr, *dot = nullptr, *vf = nullptr;
    hipError_t e = hipMalloc(&raw, (size_t)chunk * F * es);
    if (e == hipSuccess) e = hipMalloc((void **)&P, (size_t)chunk * Fp * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&Pt, (size_t)Fp * chunk * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&U, (size_t)chunk * Fp * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&Cp, (size_t)Fp * Fp * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&Craw, (size_t)F * F * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&mv, (size_t)Fp * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&dot, (size_t)chunk * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&vf, (size_t)chunk * 8);
    int rc = RR_OK;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_dense_predict: device allocation failed: %s", hipGetErrorString(e));
        rc = RR_ERR_OOM;
    }
    auto hip_ok = [&](hipError_t he, const char *what) {
        if (he != hipSuccess && rc == RR_OK) {
            rr_set_error("rr_dense_predict: %s failed: %s", what, hipGetErrorString(he));
            rc = RR_ERR_HIP;
        }
        return he == hipSuccess;
    };
    if (rc == RR_OK) {
        hip_ok(hipMemsetAsync(mv, 0, (size_t)Fp * 8, c->stream), "memset");
        hip_ok(hipMemcpyAsync(mv, m, (size_t)F * 8, hipMemcpyHostToDevice, c->stream), "upload of m");
        hip_ok(hipMemcpyAsync(Craw, C, (size_t)F * F * 8, hipMemcpyHostToDevice, c->stream), "upload of C");
        hipLaunchKernelGGL(rr_pad_c64_kernel, dim3((unsigned)((Fp * Fp + 255) / 256)), dim3(256), 0, c->stream, Craw, F, Cp, Fp);
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += chunk) {
        const int64_t rows = (N - r0 < chunk) ? N - r0 : chunk;
        const int64_t rpad = (rows + 127) / 128 * 128;
        if (!hip_ok(hipMemcpy2DAsync(raw, (size_t)F * es, (const char *)Phi + (size_t)r0 * ldphi * es, (size_t)ldphi * es,
                                     (size_t)F * es, (size_t)rows, hipMemcpyHostToDevice, c->stream), "upload of Phi")) break;
        const dim3 pg((unsigned)((Fp + 255) / 256), (unsigned)((rpad + 63) / 64));
        if (dtype == RR_F32)
            hipLaunchKernelGGL(rr_pad_rows64_kernel<float>, pg, dim3(256), 0, c->stream, (const float *)raw, rows, (int)F, P, Fp, rpad);
        else
            hipLaunchKernelGGL(rr_pad_rows64_kernel<double>, pg, dim3(256), 0, c->stream, (const double *)raw, rows, (int)F, P, Fp, rpad);
        hipLaunchKernelGGL(rr_transpose_f64_kernel, dim3((unsigned)(Fp / 64), (unsigned)(rpad / 64)), dim3(256), 0, c->stream, P,
                           rows, Fp, Pt, chunk);
        rc = rr_launch_gemm_tn_f64(c, Pt, chunk, Cp, Fp, U, Fp, Fp, rpad, Fp, 0, 0);
        if (rc != RR_OK) break;
        hipLaunchKernelGGL(rr_rows64_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, P, U, mv, rows, (int)F,
                           Fp, dot, vf);
        if (!hip_ok(hipGetLastError(), "kernel launch")) break;
        hip_ok(hipMemcpyAsync(Ey + r0, dot, (size_t)rows * 8, hipMemcpyDeviceToHost, c->stream), "download");
        hip_ok(hipMemcpyAsync(Vf + r0, vf, (size_t)rows * 8, hipMemcpyDeviceToHost, c->stream), "download");
        hip_ok(hipStreamSynchronize(c->stream), "kernel");
    }
    (void)hipStreamSynchronize(c->stream);
    void *q[] = {raw, P, Pt, U, Cp, Craw, mv, dot, vf};
    for (void *x : q)
        if (x) (void)hipFree(x);
    return rc;
}

static int fm_pass2_begin(rr_featmat *fm, const double *m, const double *C, bool c_on_device, bool tri = false) {
    RR_REQUIRE(fm != nullptr && m != nullptr && C != nullptr, "rr_featmat_pass2_begin: null argument");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int F = fm->F;
    const int64_t Fp = fm->ld;
    {
        int rc0 = fm_pass2_scratch(fm);
        if (rc0 != RR_OK) return rc0;
    }
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    s.plans.clear();  // (plans hold for one rows call of ONE pass)
    s.fused.clear();
    s.hm.assign((size_t)Fp, 0.f);
    for (int i = 0; i < F; ++i) s.hm[i] = (float)m[i];
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(s.m32, s.hm.data(), (size_t)Fp * 4, hipMemcpyHostToDevice));
    if (c_on_device) {
        hipLaunchKernelGGL(rr_c64_to_c32_kernel, dim3((unsigned)((Fp * Fp + 255) / 256)), dim3(256), 0, c->stream, C,
                           (int64_t)F, s.C32, Fp, tri ? 1 : 0);
        RR_CHECK_HIP(hipGetLastError());
    } else {
        s.hC.assign((size_t)Fp * Fp, 0.f);
        for (int i = 0; i < F; ++i) {
            const double *src = C + (size_t)i * F;
            float *dst = s.hC.data() + (size_t)i * Fp;
            if (tri) {  // upper-triangular form with doubled off-diagonals: same quadratic form, half the product
                dst[i] = (float)src[i];
                for (int j = i + 1; j < F; ++j) dst[j] = 2.f * (float)src[j];
            } else {
                for (int j = 0; j < F; ++j) dst[j] = (float)src[j];
            }
        }
        RR_CHECK_HIP(hipMemcpy(s.C32, s.hC.data(), s.hC.size() * 4, hipMemcpyHostToDevice));
    }
    RR_CHECK_HIP(hipMemsetAsync(s.sq, 0, 8, c->stream));
    s.have_rows = false;
    s.cb_ready = false;
    s.tri_c = tri;
    s.sq_form = false;
    return RR_OK;
}

int rr_featmat_pass2_begin(rr_featmat *fm, const double *m, const double *C) { return fm_pass2_begin(fm, m, C, false); }
int rr_featmat_pass2_begin_devc(rr_featmat *fm, const double *m, const double *dC) { return fm_pass2_begin(fm, m, dC, true); }
int rr_featmat_predict_begin(rr_featmat *fm, const double *m, const double *C, int c_on_device) {
    return fm_pass2_begin(fm, m, C, c_on_device != 0, true);
}

int rr_featmat_predict_begin_b(rr_featmat *fm, const double *m, const float *dB, int form) {
    RR_REQUIRE(fm != nullptr && m != nullptr && dB != nullptr && (form == 0 || form == 1), "rr_featmat_predict_begin_b: bad argument");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int F = fm->F;
    const int64_t Fp = fm->ld;
    {
        int rc0 = fm_pass2_scratch(fm);
        if (rc0 != RR_OK) return rc0;
    }
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    s.plans.clear();  // (plans hold for one rows call of ONE pass)
    s.fused.clear();
    s.hm.assign((size_t)Fp, 0.f);
    for (int i = 0; i < F; ++i) s.hm[i] = (float)m[i];
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(s.m32, s.hm.data(), (size_t)Fp * 4, hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpyAsync(s.C32, dB, (size_t)Fp * Fp * 4, hipMemcpyDeviceToDevice, c->stream));
    RR_CHECK_HIP(hipMemsetAsync(s.sq, 0, 8, c->stream));
    s.have_rows = false;
    s.cb_ready = false;
    s.tri_c = true;
    s.sq_form = form == 1;
    return RR_OK;
}

int rr_featmat_pass2_rows(rr_featmat *fm, const void *dy, int y_dtype) {
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr, "rr_featmat_pass2_rows: call rr_featmat_pass2_begin first");
    RR_FM_REQUIRE_FILLED(fm, "rr_featmat_pass2_rows");
    RR_REQUIRE(dy == nullptr || y_dtype == RR_F32 || y_dtype == RR_F64, "rr_featmat_pass2_rows: bad dtype");
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    s.have_rows = true;
    s.plans.clear();  // (announced, but the caller took the stored route: nothing is left armed)
    s.fused.clear();
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = fm_pass2_products(fm, s);
    if (rc != RR_OK || dy == nullptr) return rc;
    const dim3 grid((unsigned)((fm->rows + 255) / 256));
    double *sq = s.sq;
    const int64_t det = c->deterministic ? 1 : 0;
    if (det) {
        void *part = nullptr;
        rc = rr_det_scratch(c, (size_t)grid.x * 8, &part);
        if (rc != RR_OK) return rc;
        sq = (double *)part;
    }
    if (y_dtype == RR_F32)
        hipLaunchKernelGGL(rr_err_kernel<float>, grid, dim3(256), 0, c->stream, (const float *)dy, s.dot, fm->rows, s.err, sq, det);
    else
        hipLaunchKernelGGL(rr_err_kernel<double>, grid, dim3(256), 0, c->stream, (const double *)dy, s.dot, fm->rows, s.err, sq, det);
    RR_CHECK_HIP(hipGetLastError());
    return det ? rr_det_reduce(c, sq, grid.x, 1, 1, s.sq) : RR_OK;
}

int rr_featmat_pass2_plan_rff(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, int64_t col0, double *dT) {
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr, "rr_featmat_pass2_plan_rff: call rr_featmat_pass2_begin first");
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF && dT != nullptr, "rr_featmat_pass2_plan_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_pass2_plan_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat_pass2_plan_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad, "rr_featmat_pass2_plan_rff: device X needs ldx >= rr_rff_padded_dim() = %d", b->dpad);
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    s.plans.push_back({b, dX, x_dtype, ldx, col0, dT});
    return RR_OK;
}

// rr_featmat_pass2_rows when the planned children are the ONLY consumers of U = P C (the caller's promise): if every plan
// sits in whole 256-column tiles (col0 % 256 == 0, n % 256 == 0, d <= 128, float32 X, f32 engine, not deterministic mode),
// Err is formed first and each child's columns of U are contracted with P, Err m^T and X block by block in registers
// (rr_gemm_gradt_f32_kernel<true, ..>) -- U is never written, columns without a consumer (a linear child's) never computed.
// Otherwise: rr_featmat_pass2_rows.
int rr_featmat_pass2_rows_planned(rr_featmat *fm, const void *dy, int y_dtype) {
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr, "rr_featmat_pass2_rows_planned: call rr_featmat_pass2_begin first");
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    std::vector<FmPass2::Pass2Plan> plans;
    plans.swap(s.plans);
    s.fused.clear();
    rr_ctx *c = fm->ctx;
    const char *nfz = getenv("RR_PASS2_NO_FUSE");
    bool ok = dy != nullptr && !plans.empty() && c->gram_engine == 0 && !c->deterministic && !s.tri_c && fm->ld < (1 << 21) &&
              !(nfz && atoi(nfz) != 0);
    for (const auto &pl : plans)
        ok = ok && pl.x_dtype == RR_F32 && !pl.b->large && pl.col0 % 256 == 0 && pl.b->n % 256 == 0 && pl.b->d <= 128 &&
             pl.ldx < (1 << 21) && pl.dX != nullptr;
    if (!ok) return rr_featmat_pass2_rows(fm, dy, y_dtype);
    RR_FM_REQUIRE_FILLED(fm, "rr_featmat_pass2_rows_planned");
    RR_REQUIRE(y_dtype == RR_F32 || y_dtype == RR_F64, "rr_featmat_pass2_rows_planned: bad dtype");
    s.have_rows = true;
    if (fm->rows == 0) return RR_OK;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int64_t rows256 = (fm->rows + 255) / 256 * 256;
    hipLaunchKernelGGL(rr_rowvec_kernel, dim3((unsigned)((fm->rows + 3) / 4)), dim3(256), 0, c->stream, fm->P, s.m32,
                       fm->rows, fm->F, fm->ld, s.dot);
    if (!(fm->pt_rows == fm->rows && fm->pt_covered >= fm->F)) {  // (else: the children wrote P^T next to P)
        hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(fm->ld / 64), (unsigned)(rows256 / 64)), dim3(256), 0,
                           c->stream, fm->P, fm->rows, fm->ld, s.Pt, fm->max_rows);
        fm->pt_rows = fm->rows;
    }
    const dim3 grid((unsigned)((fm->rows + 255) / 256));
    if (y_dtype == RR_F32)
        hipLaunchKernelGGL(rr_err_kernel<float>, grid, dim3(256), 0, c->stream, (const float *)dy, s.dot, fm->rows, s.err, s.sq,
                           (int64_t)0);
    else
        hipLaunchKernelGGL(rr_err_kernel<double>, grid, dim3(256), 0, c->stream, (const double *)dy, s.dot, fm->rows, s.err, s.sq,
                           (int64_t)0);
    for (const auto &pl : plans) {
        GradtArgs g;
        g.A = s.Pt; g.lda = fm->max_rows; g.B = s.C32 + pl.col0; g.ldb = fm->ld;
        g.K = (int)(((int64_t)fm->F + GR_KB - 1) / GR_KB * GR_KB);
        g.ntb = 2 * pl.b->n / 256; g.nta = (int)(rows256 / 256);
        g.P = fm->P + pl.col0; g.ldp = fm->ld; g.X = (const float *)pl.dX; g.ldx = pl.ldx; g.rows = fm->rows;
        g.n = pl.b->n; g.d = pl.b->d; g.T = pl.dT; g.err = s.err; g.mvec = s.m32 + pl.col0; g.sign = -1.f;
        int rcg = launch_gemm_gradt(c, g);
        if (rcg != RR_OK) return rcg;
    }
    s.fused = plans;
    return RR_OK;
}

int rr_featmat_pass2_rff(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, int64_t col0, double *dT) {
    if (fm != nullptr && fm->pass2 != nullptr) {
        FmPass2 &sf = *(FmPass2 *)fm->pass2;
        for (size_t i = 0; i < sf.fused.size(); ++i)
            if (sf.fused[i].b == b && sf.fused[i].dX == dX && sf.fused[i].col0 == col0 && sf.fused[i].dT == dT) {
                sf.fused.erase(sf.fused.begin() + (long)i);  // contracted by rr_featmat_pass2_rows_planned already
                return RR_OK;
            }
    }
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr && ((FmPass2 *)fm->pass2)->have_rows,
               "rr_featmat_pass2_rff: call rr_featmat_pass2_rows first");
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF && dT != nullptr, "rr_featmat_pass2_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_pass2_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat_pass2_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad, "rr_featmat_pass2_rff: device X needs ldx >= rr_rff_padded_dim() = %d", b->dpad);
    if (fm->rows == 0) return RR_OK;
    RR_REQUIRE(dX != nullptr, "rr_featmat_pass2_rff: null X");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    if (x_dtype == RR_F32)
        return launch_grad_t<float>(b, (const float *)dX, fm->rows, ldx, fm->P + col0, s.U + col0, fm->ld, s.err,
                                    s.m32 + col0, dT);
    return launch_grad_t<double>(b, (const double *)dX, fm->rows, ldx, fm->P + col0, s.U + col0, fm->ld, s.err,
                                 s.m32 + col0, dT);
}

int rr_featmat_pass2_end(rr_featmat *fm, double *sqErr) {
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr && sqErr != nullptr, "rr_featmat_pass2_end: bad argument");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    RR_CHECK_HIP(hipStreamSynchronize(fm->ctx->stream));
    RR_CHECK_HIP(hipMemcpy(sqErr, ((FmPass2 *)fm->pass2)->sq, 8, hipMemcpyDeviceToHost));
    return RR_OK;
}

int rr_featmat_predict_rows(rr_featmat *fm, double *Ey, double *Vf) {
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr, "rr_featmat_predict_rows: call rr_featmat_pass2_begin first");
    RR_FM_REQUIRE_FILLED(fm, "rr_featmat_predict_rows");
    RR_REQUIRE(Ey != nullptr && Vf != nullptr, "rr_featmat_predict_rows: null output");
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    const bool fused_vf = s.sq_form && c->gram_engine == 0 && !c->deterministic && !getenv("RR_PREDICT_NO_FUSE");  // (see GemmArgs::rowsq)
    if (fused_vf) RR_CHECK_HIP(hipMemsetAsync(s.vf, 0, (size_t)fm->rows * 8, c->stream));
    int rc = fm_pass2_products(fm, s, fused_vf ? s.vf : nullptr);
    if (rc != RR_OK) return rc;
    if (!fused_vf)
        hipLaunchKernelGGL(rr_rowdot_kernel, dim3((unsigned)((fm->rows + 3) / 4)), dim3(256), 0, c->stream, s.U,
                           s.sq_form ? (const float *)s.U : (const float *)fm->P, fm->rows, fm->F, fm->ld, s.vf);
    RR_CHECK_HIP(hipGetLastError());
    std::vector<float> dot((size_t)fm->rows);
    RR_CHECK_HIP(hipMemcpyAsync(Vf, s.vf, (size_t)fm->rows * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(dot.data(), s.dot, (size_t)fm->rows * 4, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < fm->rows; ++i) Ey[i] = (double)dot[i];
    return RR_OK;
}

static int glm_step_checks(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param, int K,
                           int L, const char *who) {
    RR_REQUIRE(fm != nullptr && dy != nullptr, "%s: null argument", who);
    RR_FM_REQUIRE_FILLED(fm, "rr_featmat_glm_step");
    RR_REQUIRE(dtype == RR_F32 || dtype == RR_F64, "%s: bad dtype", who);
    RR_REQUIRE(lik >= RR_LIK_BERNOULLI && lik <= RR_LIK_POISSON_SOFTPLUS, "%s: unknown likelihood %d", who, lik);
    RR_REQUIRE(lik != RR_LIK_BINOMIAL || drowarg != nullptr, "%s: the binomial needs its per-row n", who);
    RR_REQUIRE(lik != RR_LIK_GAUSSIAN || lik_param > 0.0, "%s: the Gaussian variance must be > 0", who);
    RR_REQUIRE(K >= 1 && L >= 1 && (int64_t)K * L < (1 << 24), "%s: bad K, L", who);
    RR_REQUIRE(fm->rows >= 1, "%s: the feature matrix is empty", who);
    return RR_OK;
}

// The GLM step's GEMMs: f32 MFMA, or -- with a split Gram engine selected -- bf16 hi + lo operands on the 16-bit matrix
// pipe (rr_launch_gemm_tn_bf16; both operands are converted per call into context scratch).  The Monte-Carlo noise of
// the step (L samples) is orders of magnitude above either arithmetic's error.
static int glm_gemm(rr_ctx *c, const float *A, int64_t lda, const float *B, int64_t ldb, float *D, int64_t ldd, int64_t K,
                    int64_t M, int64_t N) {
    if (c->gram_engine == 0 || (K % 64) != 0) return fm_gemm(c, A, lda, B, ldb, D, ldd, K, M, N);
    const size_t na = (size_t)K * lda * 4, nb = (size_t)K * ldb * 4;
    if (c->gsa_bytes < na || c->gsb_bytes < nb) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (c->gsa_bytes < na) {
            if (c->gsa) (void)hipFree(c->gsa);
            c->gsa = nullptr; c->gsa_bytes = 0;
            RR_CHECK_HIP(hipMalloc(&c->gsa, na));
            c->gsa_bytes = na;
        }
        if (c->gsb_bytes < nb) {
            if (c->gsb) (void)hipFree(c->gsb);
            c->gsb = nullptr; c->gsb_bytes = 0;
            RR_CHECK_HIP(hipMalloc(&c->gsb, nb));
            c->gsb_bytes = nb;
        }
    }
    return rr_launch_gemm_tn_bf16(c, c->gram_engine, A, lda, B, ldb, D, ldd, K, M, N, c->gsa, c->gsb, false);
}

// With WSs (kl_ld, Fp) = ws / (K L) on the device: fs, likelihood derivatives and sums, Ed = dfs Phi, EdPhi.
// phases: 1 = fs and the likelihood kernel (dfs), 2 = Ed = dfs Phi, 4 = EdPhi (stored, or contracted with the planned random
// Fourier child) -- 7: all three in this order; the resident SVI loop runs 1, 4, 2 as separate calls (the length scales'
// gradient first, so that the next step's features can be made while Ed is formed).
static int glm_pipeline(rr_featmat *fm, FmPass2 &s, const void *dy, const void *drowarg, int dtype, int lik,
                        double lik_param, int K, int L, bool objective_only = false, const double *par_dev = nullptr,
                        int phases = 7) {
    rr_ctx *c = fm->ctx;
    const int KL = K * L;
    const int64_t Fp = fm->ld, kl_ld = s.klp;
    const int64_t rows256 = (fm->rows + 255) / 256 * 256;
    int rc = RR_OK;
    if (phases & 1) {
        // a plan is good for one step; it is taken when the whole matrix is that child's [cos | sin] block in whole tiles
        // (RR_GLM_NO_FUSE=1: never -- the EdPhi GEMM and rr_glm_grad_t_kernel as separate passes, for A/B runs)
        const char *nf = getenv("RR_GLM_NO_FUSE");
        const bool no_fuse = nf && atoi(nf) != 0;
        // (not where 256 x 256 tiles leave most CUs idle: the EdPhi product then goes through fm_gemm's 128 x 128 kernel and
        // rr_glm_grad_t_kernel contracts it)
        s.fuse.take = s.fuse.armed && !objective_only && !no_fuse && c->gram_engine == 0 && !c->deterministic &&
                      !fm_gemm_is_mid(c, kl_ld, rows256, Fp) &&
                      s.fuse.col0 == 0 && 2 * (int64_t)s.fuse.b->n == fm->F && fm->F == Fp && s.fuse.b->n % 256 == 0 &&
                      s.fuse.b->d <= 128 && Fp < (1 << 21) && s.fuse.ldx < (1 << 21);
        s.fuse.armed = false;
        s.fuse.done = false;
        RR_CHECK_HIP(hipMemsetAsync(s.kacc, 0, (size_t)2 * s.kcap * 8, c->stream));
        // WSt (Fp, kl_ld) = WSs^T (the 1 / (K L) scale is undone in the likelihood kernel's read of fs)
        hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(Fp / 64), (unsigned)(kl_ld / 64)), dim3(256), 0, c->stream,
                           s.WSs, kl_ld, Fp, s.WSt, kl_ld);
        // Pt = P^T (unless every child wrote its block of it while writing P: rr_featmat_put_rff);  FSt (rows256, kl) = P WS^T
        if (!(fm->pt_rows == fm->rows && fm->pt_covered >= fm->F)) {
            hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(Fp / 64), (unsigned)(rows256 / 64)), dim3(256), 0, c->stream,
                               fm->P, fm->rows, Fp, s.Pt, fm->max_rows);
            fm->pt_rows = fm->rows;  // P^T's padding is now laid out for this row count
        }
        // With enough output tiles to fill the chip without a K-split, the product's epilogue IS the likelihood kernel and
        // stores dfs in both layouts (rr_gemm_lik_f32_kernel); RR_GLM_FUSE_LIK=force takes that route for any shape (tests).
        const char *fl = getenv("RR_GLM_FUSE_LIK");
        const int64_t tiles1 = (rows256 / 256) * (kl_ld / 256);
        const bool lik_force = fl && !strcmp(fl, "force"), lik_off = fl && !strcmp(fl, "0");
        const bool lik_auto = tiles1 >= 2 * (int64_t)c->num_cu || Fp / GR_KB < 16;  // (fm_gemm would not split K)
        // (a small product goes to the small kernel and the likelihood kernel behind it: one tile's k-loop would take longer)
        const bool small = !lik_force && fm_gemm_is_small(Fp, rows256, kl_ld);
        s.fuse.take_lik = !no_fuse && c->gram_engine == 0 && !c->deterministic && !lik_off && (lik_force || lik_auto) && !small &&
                          fm->max_rows < (1 << 20) && kl_ld < (1 << 20);
        if (s.fuse.take_lik) {
            GemmLikArgs g;
            g.A = s.Pt; g.lda = fm->max_rows; g.B = s.WSt; g.ldb = kl_ld; g.K = (int)Fp; g.ntb = (int)(kl_ld / 256);
            g.D = objective_only ? nullptr : s.FSt; g.ldd = kl_ld;
            g.Dt = objective_only ? nullptr : s.DFS; g.ldt = fm->max_rows;
            g.y = dy; g.rowarg = drowarg; g.y_f64 = dtype == RR_F64; g.M = fm->rows;
            g.par = (float)lik_param; g.fscale = (float)KL; g.KL = KL; g.L = L; g.llsum = s.kacc; g.aux = s.kacc + s.kcap;
            g.par_dev = par_dev;
#define RR_GL(ID, ST) hipLaunchKernelGGL((rr_gemm_lik_f32_kernel<ID, ST>), dim3((unsigned)tiles1), dim3(GR_THREADS), 0, c->stream, g)
#define RR_GLS(ID)                         \
    if (objective_only) RR_GL(ID, false);  \
    else RR_GL(ID, true)
            switch (lik) {
                case RR_LIK_BERNOULLI: RR_GLS(RR_LIK_BERNOULLI); break;
                case RR_LIK_BINOMIAL: RR_GLS(RR_LIK_BINOMIAL); break;
                case RR_LIK_GAUSSIAN: RR_GLS(RR_LIK_GAUSSIAN); break;
                case RR_LIK_POISSON_EXP: RR_GLS(RR_LIK_POISSON_EXP); break;
                default: RR_GLS(RR_LIK_POISSON_SOFTPLUS); break;
            }
#undef RR_GLS
#undef RR_GL
            RR_CHECK_HIP(hipGetLastError());
        } else {
            rc = glm_gemm(c, s.Pt, fm->max_rows, s.WSt, kl_ld, s.FSt, kl_ld, Fp, rows256, kl_ld);
            if (rc != RR_OK) return rc;
            // dfs in place + per-component reductions
            if (dtype == RR_F32)
                glm_launch_lik<float>(c, lik, s.FSt, fm->rows, rows256, kl_ld, dy, drowarg, (float)lik_param, KL, L, s.kacc, s.kacc + s.kcap, par_dev);
            else
                glm_launch_lik<double>(c, lik, s.FSt, fm->rows, rows256, kl_ld, dy, drowarg, (float)lik_param, KL, L, s.kacc, s.kacc + s.kcap, par_dev);
            RR_CHECK_HIP(hipGetLastError());
        }
        if (objective_only) {  // the log-likelihood sums are all the objective needs: no gradient GEMMs
            s.have_edphi = false;
            return RR_OK;
        }
    }
    if (phases & 2) {  // Ed (kl, Fp) = dfs Phi
        rc = glm_gemm(c, s.FSt, kl_ld, fm->P, Fp, s.Ed, Fp, rows256, kl_ld, Fp);
        if (rc != RR_OK) return rc;
    }
    if (phases & 4) {
        // EdPhi (rows256, Fp) = dfs^T ws / (K L), kept in U for the gradient contraction
        if (!s.fuse.take_lik)
            hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(kl_ld / 64), (unsigned)(rows256 / 64)), dim3(256), 0, c->stream,
                               s.FSt, rows256, kl_ld, s.DFS, fm->max_rows);
        if (s.fuse.take) {  // ... or contracted with the one random Fourier child block by block, without ever reaching HBM
            GradtArgs g;
            g.A = s.DFS; g.lda = fm->max_rows; g.B = s.WSs; g.ldb = Fp; g.K = (int)kl_ld;
            g.ntb = (int)(Fp / 256); g.nta = (int)(rows256 / 256);
            g.P = fm->P; g.ldp = Fp; g.X = (const float *)s.fuse.dX; g.ldx = s.fuse.ldx; g.rows = fm->rows;
            g.n = s.fuse.b->n; g.d = s.fuse.b->d; g.T = s.fuse.dT;
            rc = launch_gemm_gradt(c, g);
            if (rc != RR_OK) return rc;
            s.have_edphi = false;
            s.fuse.done = true;
        } else {
            rc = glm_gemm(c, s.DFS, fm->max_rows, s.WSs, Fp, s.U, Fp, kl_ld, rows256, Fp);
            if (rc != RR_OK) return rc;
            s.have_edphi = true;
        }
    }
    return RR_OK;
}

int rr_featmat_glm_step(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param,
                        const double *WS, int K, int L, double *Edws, double *llsum, double *aux) {
    int rc = glm_step_checks(fm, dy, drowarg, dtype, lik, lik_param, K, L, "rr_featmat_glm_step");
    if (rc != RR_OK) return rc;
    RR_REQUIRE(WS != nullptr && Edws != nullptr && llsum != nullptr && aux != nullptr, "rr_featmat_glm_step: null argument");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int KL = K * L, F = fm->F;
    const int64_t klp = ((int64_t)KL + 255) / 256 * 256, Fp = fm->ld;
    rc = fm_glm_scratch(fm, klp, K);
    if (rc != RR_OK) return rc;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    const int64_t kl_ld = s.klp;
    // weight samples: WSs (kl_ld, Fp) = ws / (K L)
    std::vector<float> wsn((size_t)kl_ld * Fp, 0.f);
    const float inv = (float)(1.0 / ((double)K * (double)L));
    for (int i = 0; i < KL; ++i) {
        const double *src = WS + (size_t)i * F;
        float *dn = wsn.data() + (size_t)i * Fp;
        for (int j = 0; j < F; ++j) dn[j] = (float)src[j] * inv;
    }
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(s.WSs, wsn.data(), wsn.size() * 4, hipMemcpyHostToDevice));
    rc = glm_pipeline(fm, s, dy, drowarg, dtype, lik, lik_param, K, L);
    if (rc != RR_OK) return rc;
    std::vector<float> ed((size_t)KL * Fp);
    std::vector<double> acc((size_t)2 * s.kcap);
    RR_CHECK_HIP(hipMemcpyAsync(ed.data(), s.Ed, ed.size() * 4, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(acc.data(), s.kacc, acc.size() * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < KL; ++i)
        for (int j = 0; j < F; ++j) Edws[(size_t)i * F + j] = (double)ed[(size_t)i * Fp + j];
    for (int k = 0; k < K; ++k) {
        llsum[k] = acc[k];
        aux[k] = acc[s.kcap + k];
    }
    return RR_OK;
}

static int glm_step_reduced(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param,
                            const double *m, const double *C, int K, int L, uint64_t seed, uint64_t step,
                            const float *Ehost, double *Edm, double *EdC, double *llsum, double *aux,
                            const float *Edev = nullptr) {
    int rc = glm_step_checks(fm, dy, drowarg, dtype, lik, lik_param, K, L, "rr_featmat_glm_step_sampled");
    if (rc != RR_OK) return rc;
    const bool objective_only = (Edm == nullptr && EdC == nullptr);  // llsum / aux only (random starts)
    RR_REQUIRE(m != nullptr && C != nullptr && (objective_only || (Edm != nullptr && EdC != nullptr)) && llsum != nullptr &&
               aux != nullptr, "rr_featmat_glm_step_sampled: null argument");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int KL = K * L, F = fm->F;
    const int64_t klp = ((int64_t)KL + 255) / 256 * 256, Fp = fm->ld;
    rc = fm_glm_scratch(fm, klp, K);
    if (rc != RR_OK) return rc;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    const int64_t kl_ld = s.klp;
    const size_t fk = (size_t)F * K;
    RR_CHECK_HIP(hipMemcpyAsync(s.mc, m, fk * 8, hipMemcpyHostToDevice, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(s.mc + fk, C, fk * 8, hipMemcpyHostToDevice, c->stream));
    const float *Egiven = nullptr;
    if (Ehost) {  // the caller's draws go up as float32 (K L, F), staged in the (not yet used) Ed buffer
        RR_CHECK_HIP(hipMemcpyAsync(s.Ed, Ehost, (size_t)KL * F * 4, hipMemcpyHostToDevice, c->stream));
        Egiven = s.Ed;
    } else if (Edev) {  // the caller's draws are on the device already
        Egiven = Edev;
    }
    hipLaunchKernelGGL(rr_glm_draw_kernel, dim3((unsigned)((kl_ld * Fp + 255) / 256)), dim3(256), 0, c->stream, s.mc, s.mc + fk,
                       F, K, L, Fp, kl_ld, seed, step, Egiven, s.Ee, s.WSs);
    RR_CHECK_HIP(hipGetLastError());
    rc = glm_pipeline(fm, s, dy, drowarg, dtype, lik, lik_param, K, L, objective_only);
    if (rc != RR_OK) return rc;
    std::vector<double> acc((size_t)2 * s.kcap);
    if (!objective_only) {
        hipLaunchKernelGGL(rr_glm_reduce_kernel, dim3((unsigned)((fk + 255) / 256)), dim3(256), 0, c->stream, s.Ed, s.Ee,
                           s.mc + fk, F, K, L, Fp, s.mc + 2 * fk, s.mc + 3 * fk);
        RR_CHECK_HIP(hipGetLastError());
        RR_CHECK_HIP(hipMemcpyAsync(Edm, s.mc + 2 * fk, fk * 8, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipMemcpyAsync(EdC, s.mc + 3 * fk, fk * 8, hipMemcpyDeviceToHost, c->stream));
    }
    RR_CHECK_HIP(hipMemcpyAsync(acc.data(), s.kacc, acc.size() * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < K; ++k) {
        llsum[k] = acc[k];
        aux[k] = acc[s.kcap + k];
    }
    return RR_OK;
}

int rr_featmat_glm_step_sampled(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param,
                                const double *m, const double *C, int K, int L, uint64_t seed, uint64_t step, double *Edm,
                                double *EdC, double *llsum, double *aux) {
    return glm_step_reduced(fm, dy, drowarg, dtype, lik, lik_param, m, C, K, L, seed, step, nullptr, Edm, EdC, llsum, aux);
}

int rr_featmat_glm_step_draws(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param,
                              const double *m, const double *C, int K, int L, const float *E, double *Edm, double *EdC,
                              double *llsum, double *aux) {
    RR_REQUIRE(E != nullptr, "rr_featmat_glm_step_draws: null draws");
    return glm_step_reduced(fm, dy, drowarg, dtype, lik, lik_param, m, C, K, L, 0, 0, E, Edm, EdC, llsum, aux);
}

int rr_featmat_glm_step_draws_dev(rr_featmat *fm, const void *dy, const void *drowarg, int dtype, int lik, double lik_param,
                                  const double *m, const double *C, int K, int L, const float *dE, double *Edm, double *EdC,
                                  double *llsum, double *aux) {
    RR_REQUIRE(dE != nullptr, "rr_featmat_glm_step_draws_dev: null draws");
    return glm_step_reduced(fm, dy, drowarg, dtype, lik, lik_param, m, C, K, L, 0, 0, nullptr, Edm, EdC, llsum, aux, dE);
}

int rr_featmat_glm_plan_rff(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, int64_t col0, double *dT) {
    RR_REQUIRE(fm != nullptr, "rr_featmat_glm_plan_rff: null feature matrix");
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF && dT != nullptr && dX != nullptr, "rr_featmat_glm_plan_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_glm_plan_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat_glm_plan_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad, "rr_featmat_glm_plan_rff: device X needs ldx >= rr_rff_padded_dim() = %d", b->dpad);
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    int rc = fm_pass2_scratch(fm);
    if (rc != RR_OK) return rc;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    s.fuse.armed = false;
    if (x_dtype != RR_F32 || b->large) return RR_OK;  // (declined: the step forms EdPhi and rr_featmat_glm_rff contracts it)
    s.fuse.b = b; s.fuse.dX = dX; s.fuse.ldx = ldx; s.fuse.col0 = col0; s.fuse.dT = dT;
    s.fuse.armed = true;
    return RR_OK;
}

int rr_featmat_glm_rff(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, int64_t col0, double *dT) {
    if (fm != nullptr && fm->pass2 != nullptr) {
        FmPass2 &sf = *(FmPass2 *)fm->pass2;
        if (sf.fuse.done && sf.fuse.b == b && sf.fuse.dX == dX && sf.fuse.col0 == col0 && sf.fuse.dT == dT) {
            sf.fuse.done = false;  // the step's fused product has added this child's contraction to dT already
            return RR_OK;
        }
    }
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr && ((FmPass2 *)fm->pass2)->have_edphi,
               "rr_featmat_glm_rff: call rr_featmat_glm_step first");
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF && dT != nullptr && dX != nullptr, "rr_featmat_glm_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_glm_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat_glm_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad, "rr_featmat_glm_rff: device X needs ldx >= rr_rff_padded_dim() = %d", b->dpad);
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    const int64_t N = fm->rows;
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = (N * fblocks + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
    if (rpb < 64) rpb = 64;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((N + rpb - 1) / rpb));
#define RR_GG(DM, TX)                                                                                                   \
    hipLaunchKernelGGL((rr_glm_grad_t_kernel<DM, TX>), grid, dim3(256), 0, c->stream, (const TX *)dX, N, ldx, fm->P + col0, \
                       s.U + col0, fm->ld, b->n, b->d, dT, (int)rpb)
#define RR_GGD(DM)                          \
    if (x_dtype == RR_F32) RR_GG(DM, float); \
    else RR_GG(DM, double)
    switch (b->dpad) {
        case 8: RR_GGD(8); break;
        case 16: RR_GGD(16); break;
        case 32: RR_GGD(32); break;
        case 64: RR_GGD(64); break;
        case 128: RR_GGD(128); break;
        default:
            for (int i0 = 0; i0 < b->d; i0 += 128) {
                const int dd = b->d - i0 < 128 ? b->d - i0 : 128;
                if (x_dtype == RR_F32)
                    hipLaunchKernelGGL((rr_glm_grad_t_kernel<128, float>), grid, dim3(256), 0, c->stream, (const float *)dX + i0,
                                       N, ldx, fm->P + col0, s.U + col0, fm->ld, b->n, dd, dT + (size_t)i0 * b->n, (int)rpb);
                else
                    hipLaunchKernelGGL((rr_glm_grad_t_kernel<128, double>), grid, dim3(256), 0, c->stream, (const double *)dX + i0,
                                       N, ldx, fm->P + col0, s.U + col0, fm->ld, b->n, dd, dT + (size_t)i0 * b->n, (int)rpb);
            }
    }
#undef RR_GGD
#undef RR_GG
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_featmat_glm_edphi(rr_featmat *fm, int64_t col0, int64_t ncols, double *E) {
    RR_REQUIRE(fm != nullptr && fm->pass2 != nullptr && ((FmPass2 *)fm->pass2)->have_edphi,
               "rr_featmat_glm_edphi: call rr_featmat_glm_step first");
    RR_REQUIRE(E != nullptr && col0 >= 0 && ncols >= 1 && col0 + ncols <= fm->F, "rr_featmat_glm_edphi: columns out of range");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    std::vector<float> h((size_t)fm->rows * ncols);
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy2D(h.data(), (size_t)ncols * 4, s.U + col0, (size_t)fm->ld * 4, (size_t)ncols * 4, (size_t)fm->rows,
                             hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) E[i] = (double)h[i];
    return RR_OK;
}

int rr_featmat_project(rr_featmat *fm, const double *W, int S, double *out) {
    RR_REQUIRE(fm != nullptr && W != nullptr && out != nullptr && S >= 1 && S < (1 << 24), "rr_featmat_project: bad argument");
    RR_FM_REQUIRE_FILLED(fm, "rr_featmat_project");
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int F = fm->F;
    if (S == 1 && fm->max_rows >= 2) {
        // one vector (the estimator's `predict`: Phi m): a dot product per row, no transposing pass and no GEMM.  m and the
        // result sit in the product scratch U, which every product overwrites anyway
        int rc1 = fm_pass2_scratch(fm);
        if (rc1 != RR_OK) return rc1;
        FmPass2 &s1 = *(FmPass2 *)fm->pass2;
        float *mv = s1.U, *dv = s1.U + fm->ld;
        std::vector<float> h((size_t)(F > fm->rows ? F : fm->rows));
        for (int j = 0; j < F; ++j) h[j] = (float)W[j];
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        s1.have_rows = false;
        s1.have_edphi = false;
        RR_CHECK_HIP(hipMemcpy(mv, h.data(), (size_t)F * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rr_rowvec_kernel, dim3((unsigned)((fm->rows + 3) / 4)), dim3(256), 0, c->stream, fm->P, mv, fm->rows, F,
                           fm->ld, dv);
        RR_CHECK_HIP(hipGetLastError());
        RR_CHECK_HIP(hipMemcpyAsync(h.data(), dv, (size_t)fm->rows * 4, hipMemcpyDeviceToHost, c->stream));
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (int64_t i = 0; i < fm->rows; ++i) out[i] = (double)h[i];
        return RR_OK;
    }
    const int64_t sp = ((int64_t)S + 255) / 256 * 256, Fp = fm->ld;
    const int64_t rows256 = (fm->rows + 255) / 256 * 256;
    int rc = fm_glm_scratch(fm, sp, 1);
    if (rc != RR_OK) return rc;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    const int64_t ldw = s.klp;
    std::vector<float> w((size_t)Fp * ldw, 0.f);
    for (int j = 0; j < F; ++j)
        for (int i = 0; i < S; ++i) w[(size_t)j * ldw + i] = (float)W[(size_t)j * S + i];
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(s.WSt, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rr_transpose_f32_kernel, dim3((unsigned)(Fp / 64), (unsigned)(rows256 / 64)), dim3(256), 0, c->stream,
                       fm->P, fm->rows, Fp, s.Pt, fm->max_rows);
    fm->pt_rows = fm->rows;
    rc = fm_gemm(c, s.Pt, fm->max_rows, s.WSt, ldw, s.FSt, ldw, Fp, rows256, ldw);
    if (rc != RR_OK) return rc;
    s.have_edphi = false;
    std::vector<float> h((size_t)fm->rows * S);
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy2D(h.data(), (size_t)S * 4, s.FSt, (size_t)ldw * 4, (size_t)S * 4, (size_t)fm->rows, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) out[i] = (double)h[i];
    return RR_OK;
}

}  // extern "C"

// =============================================================================================
// The SVI loop of GeneralizedLinearModel.fit with its PARAMETERS RESIDENT (round 5; glm.py:141-203 and :205-294,
// optimize/sgd.py:337-425, optimize/decorators.py:329-408).  The step above returns (Edm, EdC, sums, dT) to the host, which
// forms the mixture-entropy terms, assembles the gradient, applies the log trick, the bounds and the updater, and sends the new
// (m, C, length scales) back: ~1.2 ms of serial host work and two synchronisations around 3.6 ms of kernels at config 5.
// Here the optimiser's vector
//     z = [m (F, K) | C (F, K) | regularisers (one per child) | likelihood parameter (Gaussian: variance) | length scales, child by child]
// (the flat vector of structured_sgd over a basis or a concatenation of random Fourier and linear children, Positive coordinates
// in log space) lives in HBM together with the updater's state:
//   rr_glm_sgd_from_log_kernel   x = from_log(z)                                           decorators.py:377-381
//   rr_scale_w_dev_kernel, feature kernels, the step's kernels (glm_pipeline) with m, C, lengths read from x
//   rr_glm_sgd_sums_kernel       the K(K+1)/2 mixture cross terms of _qmatrix, sum(m^2 + C) per regulariser slice, W[i,:].T[i,:]
//                                per length scale                                                  glm.py:697-712,265,274-275
//   rr_glm_sgd_update_kernel     log N_kl, log z_k, alpha; dm, dC, dreg, dlik, dl (glm.py:238-283); the chain rule of the log
//                                trick; |grad|^2 partials; bound truncation, updater, clip (sgd.py:404-420)
//   rr_glm_sgd_finish_kernel     the gradient norm and -ELBO of the step (glm.py:285-292) into per-step arrays
// Nothing is read back and the host never waits for a step: it queues step t while t - 1 runs (at most two in flight: the
// caller's minibatch buffers are reused in turn).  Float64 throughout, multiply-add contraction off: the arithmetic of the
// NumPy expressions it stands for, reduction order aside.
// Order within a step, and the second stream: the length scales' gradient comes from the EdPhi product alone, (m, C)'s from
// Ed = dfs Phi.  So the step runs fs -> EdPhi (contracted, or stored and contracted per child) -> the LENGTH SCALES' update -> Ed
// -> the rest of the update, and as soon as the length scales of step t + 1 exist its features (an HBM-write-bound 0.2 ms at
// config 5) are made on a second stream into the OTHER of two feature matrices while the matrix cores form step t's Ed.
// =============================================================================================
#define RR_SGD_MAXK 64
#define RR_SGD_MAXCHILD 16

struct SgdHRow {  // one basis-parameter gradient: W[i, :] . T[i, :] of a random Fourier child; a spectral-mixture component has
    const double *T, *W;  // two contractions (T+, T-: phases VX +- mX) and two parameters per input dimension --
    int n;                //   dmean_i = -sum_f (T+ - T-)[i, f]           (W == null: the constants c1, c2 as weights)
    const double *T2 = nullptr, *W2 = nullptr;  //   dl_i = sum_f W[i, f] (T+ + T-)[i, f] / l_i^2
    double c1 = 0.0, c2 = 0.0;
};

struct rr_glm_sgd {
    rr_featmat *fm = nullptr;
    std::vector<rr_glm_sgd_child> kids;
    std::vector<int> col0, width, ls0;  // per child: first column, columns, first length-scale coordinate (relative to the ls block)
    std::vector<double *> dTk;          // per child (random Fourier): its (d, n) contraction inside dT
    int nkids = 0, K = 0, F = 0, n_ls = 0, n_lik = 0, updater = 0, n_h = 0;
    int64_t fk = 0, np = 0, maxiter = 0, t = 0, dT_count = 0;
    double up[4] = {0, 0, 0, 0};
    double *z = nullptr, *x = nullptr, *s1 = nullptr, *s2 = nullptr, *lower = nullptr, *upper = nullptr;
    unsigned char *islog = nullptr;
    double *red = nullptr;    // [Q (K, K) | R (children) | H (length-scale gradients)]
    double *npart = nullptr;  // |grad|^2 per block of the update kernel: [main blocks | length-scale blocks]
    double *objs = nullptr, *norms = nullptr;
    double *dT = nullptr;     // the children's (d, n) blocks X^T (E_s o P_c - E_c o P_s) of the step
    int *slice_of_f = nullptr, *slice_lo = nullptr, *slice_hi = nullptr, *h_of_ls = nullptr;  // device tables
    unsigned char *ls_plain = nullptr;                                                         // (SgdUpdArgs::ls_plain)
    SgdHRow *hrows = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    rr_featmat *fm2 = nullptr;                  // steps alternate between fm and fm2 (owned)
    hipStream_t sfeat = nullptr;                // features of the next step
    hipEvent_t e_ls[2] = {nullptr, nullptr};    // step t's length scales are updated (recorded on the context's stream)
    hipEvent_t e_feat[2] = {nullptr, nullptr};  // step t's features are in its matrix (recorded on sfeat)
    hipEvent_t e_in = nullptr;                  // what the caller queued on the context's stream before this step (its row gathers)
    bool overlap = true;                        // RR_GLM_SGD_OVERLAP=0: one stream, one matrix (A/B runs)
    // RR_GLM_GROUP_TIMING=1 (measurement): host time inside rr_glm_sgd_group_step, and of it the wait for step t - 2
    int64_t call_ns = 0, wait_ns = 0, group_steps = 0;
};

static inline int64_t sgd_now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The members of a device group are queued CONCURRENTLY: member 0 on the calling thread, member i > 0 on worker i of this
// pool -- a part of a step is ~10-15 launches per member (~150 us of host time per step and member), which one thread
// queueing 8 members in turn would stretch to more than the members' kernels take.  A context is still served by one thread at
// a time (the caller waits for all workers before it goes on); a worker's error text is handed back to the caller's thread.
// Workers spin for a millisecond after a job (the next part of the step is tens of microseconds away), then sleep.
class SgdGroupWorkers {
  public:
    static SgdGroupWorkers &get() {
        static SgdGroupWorkers *pool = nullptr;
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (!pool || pool->pid_ != getpid()) pool = new SgdGroupWorkers();  // (a fork's inheritance is abandoned, not joined)
        return *pool;
    }
    // fn(i) for i in [0, n); the first failure's status (its message restored on this thread)
    int run(int n, const std::function<int(int)> &fn) {
        static const bool serial = [] { const char *e = getenv("RR_GLM_GROUP_THREADS"); return e && atoi(e) == 0; }();
        if (n <= 1 || serial) {
            for (int i = 0; i < n; ++i) {
                const int rc = fn(i);
                if (rc != RR_OK) return rc;
            }
            return RR_OK;
        }
        std::lock_guard<std::mutex> one(call_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            while ((int)slots_.size() < n - 1) {
                slots_.emplace_back(new Slot());
                const int id = (int)slots_.size() - 1;
                std::thread([this, id] { loop(id); }).detach();
            }
            job_ = &fn;
            want_ = n - 1;
            remaining_.store(n - 1, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        int rc = fn(0);
        for (int spin = 0; remaining_.load(std::memory_order_acquire) > 0; ++spin) {
            if (spin < 1 << 16) std::this_thread::yield();
            else {
                std::unique_lock<std::mutex> lk(mu_);
                done_.wait_for(lk, std::chrono::microseconds(200), [this] { return remaining_.load(std::memory_order_acquire) == 0; });
            }
        }
        if (rc != RR_OK) return rc;
        for (int i = 0; i < n - 1; ++i)
            if (slots_[(size_t)i]->rc != RR_OK) {
                rr_set_error("%s", slots_[(size_t)i]->err.c_str());
                return slots_[(size_t)i]->rc;
            }
        return RR_OK;
    }

  private:
    struct Slot {
        int rc = RR_OK;
        std::string err;
    };
    SgdGroupWorkers() : pid_(getpid()) {}
    void loop(int id) {
        int64_t seen = 0;
        for (;;) {
            int64_t g = gen_.load(std::memory_order_acquire);
            const auto t0 = std::chrono::steady_clock::now();
            while (g == seen) {
                std::this_thread::yield();
                g = gen_.load(std::memory_order_acquire);
                if (g == seen && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1000)) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                    g = gen_.load(std::memory_order_acquire);
                }
            }
            seen = g;
            const std::function<int(int)> *fn;
            Slot *slot;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (id >= want_) continue;
                fn = job_;
                slot = slots_[(size_t)id].get();
            }
            slot->rc = (*fn)(id + 1);
            if (slot->rc != RR_OK) slot->err = rr_last_error();
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_.notify_all();
            }
        }
    }
    pid_t pid_;
    std::mutex mu_, call_;
    std::condition_variable cv_, done_;
    std::vector<std::unique_ptr<Slot>> slots_;
    const std::function<int(int)> *job_ = nullptr;
    int want_ = 0;
    std::atomic<int> remaining_{0};
    std::atomic<int64_t> gen_{0};
};

__global__ void __launch_bounds__(256)
rr_glm_sgd_from_log_kernel(const double *__restrict__ z, const unsigned char *__restrict__ islog, int64_t np, double *__restrict__ x) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < np) x[p] = islog[p] ? exp(z[p]) : z[p];
}

__device__ __forceinline__ double rr_block_sum256(double v, double *sh) {  // fixed tree: the same bits every launch
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w) sh[tid] += sh[tid + w];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// blocks [0, npairs): the mixture's cross terms; [npairs, npairs + nkids): sum (m^2 + C) over a child's rows of (m, C);
// then one block per length-scale gradient
__global__ void __launch_bounds__(256)
rr_glm_sgd_sums_kernel(const double *__restrict__ x, int F, int K, int nkids, const int *__restrict__ slice_lo,
                       const int *__restrict__ slice_hi, const SgdHRow *__restrict__ hrows, int n_h, double *__restrict__ red,
                       int first_block) {
#pragma clang fp contract(off)
    __shared__ double sh[256];
    const int npairs = K * (K + 1) / 2, tid = threadIdx.x;
    const int64_t fk = (int64_t)F * K;
    int bid = blockIdx.x + first_block;
    double acc = 0.0;
    if (bid < npairs) {  // sum_f log(C_fk + C_fl) + (m_fk - m_fl)^2 / (C_fk + C_fl)
        int k = 0;
        while (bid >= K - k) {
            bid -= K - k;
            ++k;
        }
        const int l = k + bid;
        for (int f = tid; f < F; f += 256) {
            const double dc = x[fk + (int64_t)f * K + k] + x[fk + (int64_t)f * K + l];
            const double dm = x[(int64_t)f * K + k] - x[(int64_t)f * K + l];
            acc += log(dc) + dm * dm / dc;
        }
        const double q = rr_block_sum256(acc, sh);
        if (tid == 0) red[k * K + l] = red[l * K + k] = q;
    } else if (bid < npairs + nkids) {  // sum (m^2 + C) over the child's slice
        const int s = bid - npairs;
        for (int64_t p = (int64_t)slice_lo[s] * K + tid; p < (int64_t)slice_hi[s] * K; p += 256) acc += x[p] * x[p] + x[fk + p];
        const double r = rr_block_sum256(acc, sh);
        if (tid == 0) red[K * K + s] = r;
    } else {  // W[i, :] . T[i, :]
        const int i = bid - npairs - nkids;
        if (i < n_h) {
            const SgdHRow h = hrows[i];
            for (int f = tid; f < h.n; f += 256) {
                acc += h.T[f] * (h.W ? h.W[f] : h.c1);
                if (h.T2) acc += h.T2[f] * (h.W2 ? h.W2[f] : h.c2);
            }
            const double v = rr_block_sum256(acc, sh);
            if (tid == 0) red[K * K + nkids + i] = v;
        }
    }
}

struct SgdUpdArgs {
    const double *x, *red, *Edm, *EdC, *aux, *lower, *upper;
    const unsigned char *islog;
    const int *slice_of_f, *slice_lo, *slice_hi, *h_of_ls;
    const unsigned char *ls_plain = nullptr;  // per basis parameter: 1 = its gradient is the sum itself (a mixture's mean), 0 = / l^2
    double *z, *s1, *s2, *npart;
    int F, K, nkids, n_lik, n_ls, updater, L;
    int64_t np, p0, p1;  // all coordinates; this launch's are [p0, p1)
    double bmag, nrows, up[4], b1t, b2t;
    const double *tot = nullptr;  // device [llconst | rows] summed over ranks (rr_glm_sgd_dist_step); null: nrows above
};

__global__ void __launch_bounds__(256) rr_glm_sgd_update_kernel(const SgdUpdArgs a) {
#pragma clang fp contract(off)
    __shared__ double logN[RR_SGD_MAXK * RR_SGD_MAXK], alpha[RR_SGD_MAXK * RR_SGD_MAXK], logz[RR_SGD_MAXK], sh[256];
    const int tid = threadIdx.x, K = a.K, F = a.F;
    const int64_t fk = (int64_t)F * K;
    if (a.p0 < 2 * fk) {  // (a launch over the length scales alone needs none of the mixture's terms)
        for (int i = tid; i < K * K; i += 256) logN[i] = -0.5 * ((double)F * 1.8378770664093453 + a.red[i]);  // log(2 pi)
        __syncthreads();
        if (tid < K) {  // logsumexp over the first index (glm.py:222)
            double mx = -INFINITY;
            for (int j = 0; j < K; ++j) mx = fmax(mx, logN[j * K + tid]);
            double sm = 0.0;
            for (int j = 0; j < K; ++j) sm += exp(logN[j * K + tid] - mx);
            logz[tid] = log(sm) + mx;
        }
        __syncthreads();
        for (int i = tid; i < K * K; i += 256) {
            const int k = i / K, l = i % K;
            alpha[i] = exp(logN[l * K + k] - logz[k]) + exp(logN[l * K + k] - logz[l]);
        }
        __syncthreads();
    }
    const int64_t p = a.p0 + (int64_t)blockIdx.x * 256 + tid;
    double g = 0.0;
    const bool live = p < a.p1;
    if (live) {
        if (p < 2 * fk) {
            const bool cov = p >= fk;
            const int64_t q = cov ? p - fk : p;
            const int f = (int)(q / K), k = (int)(q % K);
            const double reg = a.x[2 * fk + a.slice_of_f[f]];  // the regulariser of this feature's child (glm.py:218-219)
            const double mk = a.x[q], Ck = a.x[fk + q];
            double mix = 0.0;
            for (int l = 0; l < K; ++l) {
                const double ic = 1.0 / (Ck + a.x[fk + (int64_t)f * K + l]);
                const double dm = mk - a.x[(int64_t)f * K + l];
                const double e = dm * ic;
                mix += (cov ? ic - e * e : ic * dm) * alpha[k * K + l];
            }
            if (!cov) g = -((a.bmag * a.Edm[(int64_t)k * F + f] - mk / reg + mix) / K);
            else g = -((a.bmag * a.EdC[(int64_t)k * F + f] - 1.0 / reg + mix) / (2 * K));
        } else if (p < 2 * fk + a.nkids) {  // dreg of the child's slice (glm.py:265-268)
            const int s = (int)(p - 2 * fk);
            const double iL = 1.0 / a.x[p];
            g = -(0.5 * (a.red[K * K + s] * (iL * iL) / K - (double)(a.slice_hi[s] - a.slice_lo[s]) * iL));
        } else if (p < 2 * fk + a.nkids + a.n_lik) {  // Gaussian variance: dp = ((y - f)^2 / var^2 - 1 / var) / 2  (likelihoods.py:360-381)
            const double ivar = 1.0 / a.x[p];
            double sm = 0.0;
            const double nrows = a.tot ? a.tot[1] : a.nrows;
            for (int k = 0; k < K; ++k) sm += 0.5 * (a.aux[k] * ivar * ivar - ivar * nrows * a.L) / a.L;
            g = 0.0 - sm / K;
        } else {  // -(EdPhi o dPhi_i).sum() = W[i,:].T[i,:] / l_i^2; isotropic: input dimension 0 only, as the reference
            const int j = (int)(p - (2 * fk + a.nkids + a.n_lik));
            const double l = a.x[p];
            g = a.red[K * K + a.nkids + a.h_of_ls[j]];
            if (!(a.ls_plain && a.ls_plain[j])) g = g / (1.0 * (l * l));
        }
        if (a.islog[p]) g *= a.x[p];  // d/dz through x = exp(z)
    }
    const double n2 = rr_block_sum256(live ? g * g : 0.0, sh);
    if (tid == 0) a.npart[blockIdx.x] = n2;
    if (!live) return;
    const double zz = a.z[p], lo = a.lower[p], hi = a.upper[p];
    if (zz <= lo) g = (g <= 0.0 || g != g) ? g : 0.0;  // np.minimum(grad, 0) on a coordinate sitting on its lower bound
    if (zz >= hi) g = (g >= 0.0 || g != g) ? g : 0.0;
    double zn;
    switch (a.updater) {
        case RR_UPD_SGD: zn = zz - a.up[0] * g; break;
        case RR_UPD_ADADELTA: {  // up = (rho, epsilon); s1 = E[g^2], s2 = E[dx^2]
            const double eg2 = a.up[0] * a.s1[p] + (1 - a.up[0]) * (g * g);
            const double dx = -g * sqrt(a.s2[p] + a.up[1]) / sqrt(eg2 + a.up[1]);
            a.s1[p] = eg2;
            a.s2[p] = a.up[0] * a.s2[p] + (1 - a.up[0]) * (dx * dx);
            zn = zz + dx;
        } break;
        case RR_UPD_ADAGRAD: {  // up = (eta, epsilon); s1 = sum g^2
            const double h = a.s1[p] + g * g;
            a.s1[p] = h;
            zn = zz - a.up[0] * g / (a.up[1] + sqrt(h));
        } break;
        case RR_UPD_MOMENTUM: {  // up = (rho, eta); s1 = dx
            const double dx = a.up[0] * a.s1[p] - a.up[1] * g;
            a.s1[p] = dx;
            zn = zz + dx;
        } break;
        default: {  // Adam: up = (alpha, beta1, beta2, epsilon); b1t = 1 - beta1^t, b2t = 1 - beta2^t
            const double m = a.up[1] * a.s1[p] + (1 - a.up[1]) * g;
            const double v = a.up[2] * a.s2[p] + (1 - a.up[2]) * (g * g);
            a.s1[p] = m;
            a.s2[p] = v;
            zn = zz - a.up[0] * (m / a.b1t) / (sqrt(v / a.b2t) + a.up[3]);
        }
    }
    a.z[p] = zn < lo ? lo : (zn > hi ? hi : zn);
}

__global__ void __launch_bounds__(256)
rr_glm_sgd_finish_kernel(const double *__restrict__ x, const double *__restrict__ red, const double *__restrict__ llsum,
                         const double *__restrict__ npart, int nblocks, int F, int K, int L, int nkids, const int *__restrict__ slice_lo,
                         const int *__restrict__ slice_hi, int n_lik, double llconst, double nrows, double bmag,
                         double *__restrict__ obj, double *__restrict__ norm, const double *__restrict__ tot) {
#pragma clang fp contract(off)
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < nblocks; i += 256) acc += npart[i];
    const double n2 = rr_block_sum256(acc, sh);
    double lz = 0.0;
    if (tid < K) {
        double mx = -INFINITY;
        for (int j = 0; j < K; ++j) mx = fmax(mx, -0.5 * ((double)F * 1.8378770664093453 + red[j * K + tid]));
        double sm = 0.0;
        for (int j = 0; j < K; ++j) sm += exp(-0.5 * ((double)F * 1.8378770664093453 + red[j * K + tid]) - mx);
        lz = log(sm) + mx;
    }
    const double logzsum = rr_block_sum256(lz, sh);
    if (tid == 0) {
        const int64_t fk = (int64_t)F * K;
        if (tot) {  // the whole job's minibatch: sums over the ranks
            llconst = tot[0];
            nrows = tot[1];
        }
        if (n_lik) llconst = -0.5 * log(2.0 * 3.141592653589793 * x[2 * fk + nkids]) * nrows;  // Gaussian (likelihoods.py:295-317)
        double ell = 0.0;
        for (int k = 0; k < K; ++k) ell += llsum[k] / L + llconst;
        double logL = 0.0, quad = 0.0;  // log(L).sum() and ((m^2 + C) iL).sum(), child by child (glm.py:288-289)
        for (int s = 0; s < nkids; ++s) {
            const double reg = x[2 * fk + s];
            logL += (double)(slice_hi[s] - slice_lo[s]) * log(reg);
            quad += red[K * K + s] / reg;
        }
        const double elbo = (ell * bmag - 0.5 * F * K * 1.8378770664093453 - 0.5 * K * logL - 0.5 * quad - logzsum + log((double)K)) / K;
        *obj = -elbo;
        *norm = sqrt(n2);
    }
}

static void sgd_free(rr_glm_sgd *o) {
    void *q[] = {o->z, o->x, o->s1, o->s2, o->lower, o->upper, o->islog, o->red, o->npart, o->objs, o->norms, o->dT,
                 o->slice_of_f, o->slice_lo, o->slice_hi, o->h_of_ls, o->hrows, o->ls_plain};
    for (void *v : q)
        if (v) (void)hipFree(v);
    for (hipEvent_t e : {o->ev[0], o->ev[1], o->e_ls[0], o->e_ls[1], o->e_feat[0], o->e_feat[1], o->e_in})
        if (e) (void)hipEventDestroy(e);
    if (o->sfeat) (void)hipStreamDestroy(o->sfeat);
    if (o->fm2) rr_featmat_destroy(o->fm2);
    delete o;
}

int rr_basis_raw_w(rr_basis *b);  // rr_api.hip: W as given, resident (dWraw)

extern "C" {

int rr_glm_sgd_create(rr_featmat *fm, int n_children, const rr_glm_sgd_child *children, int K, int n_lik, const double *z0,
                      const double *lower, const double *upper, const unsigned char *is_log, int updater, const double *upd_par,
                      int64_t maxiter, rr_glm_sgd **out) {
    RR_REQUIRE(fm != nullptr && children != nullptr && out != nullptr && z0 != nullptr && lower != nullptr && upper != nullptr &&
               is_log != nullptr && upd_par != nullptr, "rr_glm_sgd_create: null argument");
    *out = nullptr;
    RR_REQUIRE(n_children >= 1 && n_children <= RR_SGD_MAXCHILD, "rr_glm_sgd_create: 1 <= children <= %d", RR_SGD_MAXCHILD);
    RR_REQUIRE(K >= 1 && K <= RR_SGD_MAXK, "rr_glm_sgd_create: 1 <= K <= %d", RR_SGD_MAXK);
    RR_REQUIRE(n_lik == 0 || n_lik == 1, "rr_glm_sgd_create: at most one likelihood parameter");
    RR_REQUIRE(updater >= RR_UPD_SGD && updater <= RR_UPD_ADAM, "rr_glm_sgd_create: unknown updater %d", updater);
    RR_REQUIRE(maxiter >= 1 && maxiter < ((int64_t)1 << 31), "rr_glm_sgd_create: bad maxiter");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    rr_glm_sgd *o = new rr_glm_sgd();
    o->fm = fm; o->K = K; o->F = fm->F; o->n_lik = n_lik; o->updater = updater; o->nkids = n_children;
    std::vector<int> h_slice(fm->F, 0), h_lo, h_hi, h_of_ls;
    std::vector<unsigned char> h_plain;
    std::vector<SgdHRow> hrows;
    int col = 0, nls = 0;
    int64_t dT_count = 0;
    for (int s = 0; s < n_children; ++s) {
        const rr_glm_sgd_child &k = children[s];
        int w = 0;
        if (k.kind == RR_SGD_CHILD_RFF) {
            rr_basis *b = k.basis;
            if (!(b != nullptr && b->kind == RR_KIND_RFF && !b->large && b->d <= 128 && b->ctx == fm->ctx &&
                  (k.n_ls == 1 || k.n_ls == b->d))) {
                delete o;
                rr_set_error("rr_glm_sgd_create: child %d must be a random Fourier basis of Xdim <= 128 on the matrix' context with 1 or "
                             "Xdim length scales", s);
                return RR_ERR_INVALID;
            }
            w = 2 * b->n;
            dT_count += (int64_t)b->d * b->n;
        } else if (k.kind == RR_SGD_CHILD_GM) {
            rr_basis *b = k.basis;
            if (!(b != nullptr && b->kind == RR_KIND_RFF && !b->large && b->d <= 128 && b->ctx == fm->ctx && k.n_ls == 2 * b->d)) {
                delete o;
                rr_set_error("rr_glm_sgd_create: child %d must be the dense equivalent of a spectral-mixture component of Xdim <= 128 on "
                             "the matrix' context with 2 Xdim parameters (mean, length scales)", s);
                return RR_ERR_INVALID;
            }
            w = 4 * b->n;
            dT_count += 2 * (int64_t)b->d * b->n;
        } else if (k.kind == RR_SGD_CHILD_LINEAR) {
            if (!(k.d >= 1 && k.n_ls == 0)) {
                delete o;
                rr_set_error("rr_glm_sgd_create: child %d: a linear child has d >= 1 columns and no length scale", s);
                return RR_ERR_INVALID;
            }
            w = k.d + (k.onescol ? 1 : 0);
        } else {
            delete o;
            rr_set_error("rr_glm_sgd_create: child %d: unknown kind %d", s, k.kind);
            return RR_ERR_INVALID;
        }
        if (col + w > fm->F) {
            delete o;
            rr_set_error("rr_glm_sgd_create: the children are wider than the feature matrix (%d columns)", fm->F);
            return RR_ERR_INVALID;
        }
        o->kids.push_back(k);
        o->col0.push_back(col);
        o->width.push_back(w);
        o->ls0.push_back(nls);
        h_lo.push_back(col);
        h_hi.push_back(col + w);
        for (int f = col; f < col + w; ++f) h_slice[f] = s;
        col += w;
        nls += k.n_ls;
    }
    if (col != fm->F) {
        delete o;
        rr_set_error("rr_glm_sgd_create: the children cover %d of the feature matrix' %d columns", col, fm->F);
        return RR_ERR_INVALID;
    }
    o->n_ls = nls;
    o->fk = (int64_t)fm->F * K;
    o->np = 2 * o->fk + n_children + n_lik + nls;
    o->maxiter = maxiter;
    o->dT_count = dT_count > 0 ? dT_count : 1;
    for (int i = 0; i < 4; ++i) o->up[i] = upd_par[i];
    const size_t nb = (size_t)o->np * 8, nblocks = (size_t)((o->np + 255) / 256) + 1;  // (main and length-scale launches)
    const char *ov = getenv("RR_GLM_SGD_OVERLAP");
    // (the second stream and matrix pay when a step's features are worth hiding: not for the reference's default batch of 10
    // rows, where a step is ~35 launches of a few microseconds and two more events only add to the host's part.
    // RR_GLM_SGD_OVERLAP=1 forces it, for tests)
    o->overlap = ov ? atoi(ov) != 0 : fm->max_rows * (int64_t)fm->F >= ((int64_t)1 << 22);
    hipError_t e = hipMalloc((void **)&o->z, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->x, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->s1, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->s2, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->lower, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->upper, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->islog, (size_t)o->np);
    if (e == hipSuccess) e = hipMalloc((void **)&o->red, (size_t)(K * K + n_children + (nls > 0 ? nls : 1)) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&o->npart, nblocks * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&o->objs, (size_t)maxiter * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&o->norms, (size_t)maxiter * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&o->dT, (size_t)o->dT_count * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&o->slice_of_f, (size_t)fm->F * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&o->slice_lo, (size_t)n_children * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&o->slice_hi, (size_t)n_children * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&o->h_of_ls, (size_t)(nls > 0 ? nls : 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&o->hrows, (size_t)(nls > 0 ? nls : 1) * sizeof(SgdHRow));
    if (e == hipSuccess) e = hipMalloc((void **)&o->ls_plain, (size_t)(nls > 0 ? nls : 1));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&o->ev[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&o->ev[1], hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&o->e_ls[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&o->e_feat[i], hipEventDisableTiming);
    }
    if (e == hipSuccess && o->overlap) e = hipStreamCreateWithFlags(&o->sfeat, hipStreamNonBlocking);
    if (e == hipSuccess && o->overlap) e = hipEventCreateWithFlags(&o->e_in, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sgd_free(o);
        rr_set_error("rr_glm_sgd_create: device allocation failed");
        return RR_ERR_OOM;
    }
    if (o->overlap) {
        const int rc2 = rr_featmat_create(c, fm->max_rows, fm->F, &o->fm2);
        if (rc2 != RR_OK) {
            sgd_free(o);
            return rc2;
        }
    }
    // the length-scale gradients: one W[i, :] . T[i, :] per ARD coordinate, dimension 0 alone for an isotropic child (the
    // reference's quirk, basis_functions.py:866-901)
    double *dTp = o->dT;
    for (int s = 0; s < n_children; ++s) {
        const rr_glm_sgd_child &k = o->kids[(size_t)s];
        o->dTk.push_back(nullptr);
        if (k.kind != RR_SGD_CHILD_RFF && k.kind != RR_SGD_CHILD_GM) continue;
        const int rcw = rr_basis_raw_w(k.basis);
        if (rcw != RR_OK) {
            sgd_free(o);
            return rcw;
        }
        o->dTk.back() = dTp;
        const int64_t n = k.basis->n, dn = (int64_t)k.basis->d * n;
        if (k.kind == RR_SGD_CHILD_GM) {  // [mean (d) | length scales (d)]: T+ at dTp, T- behind it
            for (int i = 0; i < k.basis->d; ++i) {
                h_of_ls.push_back((int)hrows.size());
                h_plain.push_back(1);
                SgdHRow r{dTp + i * n, nullptr, (int)n};
                r.T2 = dTp + dn + i * n; r.c1 = -1.0; r.c2 = 1.0;
                hrows.push_back(r);
            }
            for (int i = 0; i < k.basis->d; ++i) {
                h_of_ls.push_back((int)hrows.size());
                h_plain.push_back(0);
                SgdHRow r{dTp + i * n, k.basis->dWraw + i * n, (int)n};
                r.T2 = dTp + dn + i * n; r.W2 = r.W;
                hrows.push_back(r);
            }
            dTp += 2 * dn;
            continue;
        }
        for (int i = 0; i < k.n_ls; ++i) {
            h_of_ls.push_back((int)hrows.size());
            h_plain.push_back(0);
            hrows.push_back(SgdHRow{dTp + (int64_t)i * n, k.basis->dWraw + (int64_t)i * n, (int)n});
        }
        dTp += dn;
    }
    o->n_h = (int)hrows.size();
    hipError_t h = hipStreamSynchronize(c->stream);
    if (h == hipSuccess) h = hipMemcpy(o->z, z0, nb, hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemcpy(o->lower, lower, nb, hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemcpy(o->upper, upper, nb, hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemcpy(o->islog, is_log, (size_t)o->np, hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemcpy(o->slice_of_f, h_slice.data(), (size_t)fm->F * sizeof(int), hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemcpy(o->slice_lo, h_lo.data(), (size_t)n_children * sizeof(int), hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemcpy(o->slice_hi, h_hi.data(), (size_t)n_children * sizeof(int), hipMemcpyHostToDevice);
    if (h == hipSuccess && nls > 0) h = hipMemcpy(o->h_of_ls, h_of_ls.data(), (size_t)nls * sizeof(int), hipMemcpyHostToDevice);
    if (h == hipSuccess && nls > 0) h = hipMemcpy(o->hrows, hrows.data(), hrows.size() * sizeof(SgdHRow), hipMemcpyHostToDevice);
    if (h == hipSuccess && nls > 0) h = hipMemcpy(o->ls_plain, h_plain.data(), h_plain.size(), hipMemcpyHostToDevice);
    if (h == hipSuccess) h = hipMemset(o->s1, 0, nb);
    if (h == hipSuccess) h = hipMemset(o->s2, 0, nb);
    if (h == hipSuccess) h = hipMemset(o->objs, 0, (size_t)maxiter * 8);
    if (h == hipSuccess) h = hipMemset(o->norms, 0, (size_t)maxiter * 8);
    if (h == hipSuccess) h = hipDeviceSynchronize();
    if (h != hipSuccess) {
        (void)hipGetLastError();
        sgd_free(o);
        rr_set_error("rr_glm_sgd_create: %s", hipGetErrorString(h));
        return RR_ERR_HIP;
    }
    *out = o;
    return RR_OK;
}

// One step in three parts, so that the members of a device group (rr_glm_sgd_group_step) can put their all-reduces between
// them; rr_glm_sgd_step runs the three back to back.  Everything a step sums over ROWS ends up in three buffers: dT (the
// length-scale contractions), [Edm | EdC] and [llsum | aux]; everything else is a function of z and of those.
struct SgdStepIn {
    const void *const *dX;
    const int *x_dtype;
    const int64_t *ldx;
    int64_t rows;        // of THIS loop's share of the minibatch (a group member's may be 0)
    int64_t rows_total;  // of the whole minibatch
    const void *dy, *drowarg;
    int dtype, lik;
    double llconst, bmag;
    int L;
    const float *dE;
    uint64_t seed, key;
    bool totals_on_device = false;  // rr_glm_sgd_dist_step: llconst and the row count are summed over the ranks in HBM
};

__global__ void rr_glm_sgd_set_totals_kernel(double *tot, double llconst, double rows) {
    tot[0] = llconst;
    tot[1] = rows;
}

static rr_featmat *sgd_step_fm(rr_glm_sgd *o) { return (o->overlap && (o->t & 1)) ? o->fm2 : o->fm; }

static int sgd_step_check(rr_glm_sgd *o, const SgdStepIn &in, int64_t min_rows, const char *who) {
    RR_REQUIRE(o->t < o->maxiter, "%s: all %lld steps of this loop are done", who, (long long)o->maxiter);
    RR_REQUIRE((in.lik == RR_LIK_GAUSSIAN) == (o->n_lik == 1), "%s: likelihood %d with %d likelihood parameter(s)", who, in.lik, o->n_lik);
    RR_REQUIRE(in.rows >= min_rows && in.rows <= o->fm->max_rows, "%s: rows out of range", who);
    if (in.rows > 0) {
        RR_REQUIRE(in.dX != nullptr && in.x_dtype != nullptr && in.ldx != nullptr && in.dy != nullptr, "%s: null argument", who);
        for (int s = 0; s < o->nkids; ++s)
            RR_REQUIRE(in.dX[s] != nullptr && (in.x_dtype[s] == RR_F32 || in.x_dtype[s] == RR_F64), "%s: child %d: bad rows", who, s);
    }
    return RR_OK;
}

// features, draws, fs and the likelihood terms, EdPhi and its contraction with every child: dT is complete
static int sgd_step_front(rr_glm_sgd *o, const SgdStepIn &in) {
    rr_featmat *fm = o->fm;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int K = o->K, F = o->F, nk = o->nkids, L = in.L;
    const int64_t fk = o->fk;
    // at most two steps in flight: the event of step t - 2 (which also was the last user of this step's feature matrix)
    if (o->t >= 2) {
        const int64_t t0 = sgd_now_ns();
        RR_CHECK_HIP(hipEventSynchronize(o->ev[o->t & 1]));
        o->wait_ns += sgd_now_ns() - t0;
    }
    const int par = (int)(o->t & 1);
    fm = sgd_step_fm(o);
    const int64_t n_main = 2 * fk + nk + o->n_lik;
    const unsigned nb_main = (unsigned)((n_main + 255) / 256), nb_ls = (unsigned)((o->n_ls + 255) / 256);
    const double *xls = o->x + n_main, *xpar = o->n_lik ? o->x + 2 * fk + nk : nullptr;
    int rc = RR_OK;
    hipStream_t s0 = c->stream;
    const int KL = K * L;
    const int64_t klp = ((int64_t)KL + 255) / 256 * 256, Fp = fm->ld;
    if (in.rows == 0) {  // (a group member without rows of this minibatch: its sums are zero, its parameters follow the others')
        if (nb_ls)
            hipLaunchKernelGGL(rr_glm_sgd_from_log_kernel, dim3(nb_ls), dim3(256), 0, s0, o->z + n_main, o->islog + n_main, (int64_t)o->n_ls,
                               o->x + n_main);
        hipLaunchKernelGGL(rr_glm_sgd_from_log_kernel, dim3(nb_main), dim3(256), 0, s0, o->z, o->islog, n_main, o->x);
        RR_CHECK_HIP(hipGetLastError());
        rc = fm_glm_scratch(fm, klp, K);
        if (rc != RR_OK) return rc;
        FmPass2 &s = *(FmPass2 *)fm->pass2;
        RR_CHECK_HIP(hipMemsetAsync(o->dT, 0, (size_t)o->dT_count * 8, s0));
        RR_CHECK_HIP(hipMemsetAsync(s.mc + 2 * fk, 0, (size_t)(2 * fk) * 8, s0));
        RR_CHECK_HIP(hipMemsetAsync(s.kacc, 0, (size_t)2 * s.kcap * 8, s0));
        return RR_OK;
    }
    // ---- this step's features: after the previous step's length-scale update, on the second stream, into this step's
    //      matrix -- while the previous step's Ed product runs on the first
    hipStream_t sf = o->overlap ? o->sfeat : s0;
    if (o->overlap) {
        // stream order of dX / dy / drowarg / dE: whatever the caller queued on the context's stream before this call (row
        // gathers of a minibatch that was not prefetched, uploads) is ordered BEFORE the feature kernels on the second stream
        RR_CHECK_HIP(hipEventRecord(o->e_in, s0));
        RR_CHECK_HIP(hipStreamWaitEvent(sf, o->e_in, 0));
        if (o->t >= 1) RR_CHECK_HIP(hipStreamWaitEvent(sf, o->e_ls[1 - par], 0));
    }
    {
        // every launch helper below reads the context's stream when it is called; restored on every way out of this block.
        // (A context serves ONE host thread at a time -- include/revrand_hip.h -- so nobody else reads the field meanwhile.)
        struct StreamScope {
            rr_ctx *c;
            hipStream_t prev;
            StreamScope(rr_ctx *c_, hipStream_t s_) : c(c_), prev(c_->stream) { c->stream = s_; }
            ~StreamScope() { c->stream = prev; }
        } scope(c, sf);
        if (nb_ls)
            hipLaunchKernelGGL(rr_glm_sgd_from_log_kernel, dim3(nb_ls), dim3(256), 0, sf, o->z + n_main, o->islog + n_main, (int64_t)o->n_ls,
                               o->x + n_main);
        rc = rr_featmat_begin(fm, in.rows);
        for (int s = 0; s < nk && rc == RR_OK; ++s) {
            const rr_glm_sgd_child &k = o->kids[(size_t)s];
            if (k.kind == RR_SGD_CHILD_RFF) rc = rr_fm_put_rff_dev(fm, k.basis, in.dX[s], in.x_dtype[s], in.ldx[s], xls + o->ls0[(size_t)s], k.n_ls, o->col0[(size_t)s]);
            else if (k.kind == RR_SGD_CHILD_GM) {
                // [cos | sin](VX + mX) at col0, [cos | sin](VX - mX) at col0 + 2n: the random Fourier kernels with every frequency
                // moved by +- mean (stream order: the second rescaling of W follows the first block's feature kernel)
                const double *mean = xls + o->ls0[(size_t)s], *ls = mean + k.basis->d;
                rc = rr_fm_put_rff_dev(fm, k.basis, in.dX[s], in.x_dtype[s], in.ldx[s], ls, k.basis->d, o->col0[(size_t)s], mean, 1.0);
                if (rc == RR_OK)
                    rc = rr_fm_put_rff_dev(fm, k.basis, in.dX[s], in.x_dtype[s], in.ldx[s], ls, k.basis->d, o->col0[(size_t)s] + 2 * k.basis->n, mean, -1.0);
            } else rc = rr_featmat_put_linear(fm, in.dX[s], in.x_dtype[s], in.ldx[s], k.d, k.onescol, o->col0[(size_t)s]);
        }
    }
    if (rc != RR_OK) return rc;
    RR_CHECK_HIP(hipGetLastError());
    if (o->overlap) {
        RR_CHECK_HIP(hipEventRecord(o->e_feat[par], sf));
        RR_CHECK_HIP(hipStreamWaitEvent(s0, o->e_feat[par], 0));
    }
    // ---- the step proper
    hipLaunchKernelGGL(rr_glm_sgd_from_log_kernel, dim3(nb_main), dim3(256), 0, s0, o->z, o->islog, n_main, o->x);
    RR_CHECK_HIP(hipGetLastError());
    rc = glm_step_checks(fm, in.dy, in.drowarg, in.dtype, in.lik, 1.0, K, L, "rr_glm_sgd_step");
    if (rc != RR_OK) return rc;
    rc = fm_glm_scratch(fm, klp, K);
    if (rc != RR_OK) return rc;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    const int64_t kl_ld = s.klp;
    RR_CHECK_HIP(hipMemsetAsync(o->dT, 0, (size_t)o->dT_count * 8, s0));
    const bool lone_rff = nk == 1 && o->kids[0].kind == RR_SGD_CHILD_RFF;
    if (lone_rff) {  // the EdPhi product may contract itself with the one child (rr_gemm_gradt_f32_kernel)
        rc = rr_featmat_glm_plan_rff(fm, o->kids[0].basis, in.dX[0], in.x_dtype[0], in.ldx[0], 0, o->dTk[0]);
        if (rc != RR_OK) return rc;
    }
    hipLaunchKernelGGL(rr_glm_draw_kernel, dim3((unsigned)((kl_ld * Fp + 255) / 256)), dim3(256), 0, s0, o->x, o->x + fk, F, K,
                       L, Fp, kl_ld, in.seed, in.key, in.dE, s.Ee, s.WSs);
    RR_CHECK_HIP(hipGetLastError());
    rc = glm_pipeline(fm, s, in.dy, in.drowarg, in.dtype, in.lik, 1.0, K, L, false, xpar, 1);      // fs, likelihood terms
    if (rc == RR_OK && o->n_h) rc = glm_pipeline(fm, s, in.dy, in.drowarg, in.dtype, in.lik, 1.0, K, L, false, xpar, 4);  // EdPhi, contracted or stored
    for (int ch = 0; ch < nk && rc == RR_OK; ++ch)  // (returns at once when the step contracted EdPhi itself)
        if (o->kids[(size_t)ch].kind == RR_SGD_CHILD_RFF)
            rc = rr_featmat_glm_rff(fm, o->kids[(size_t)ch].basis, in.dX[ch], in.x_dtype[ch], in.ldx[ch], o->col0[(size_t)ch], o->dTk[(size_t)ch]);
        else if (o->kids[(size_t)ch].kind == RR_SGD_CHILD_GM) {  // T+ and T-: the two blocks' contractions
            rr_basis *b = o->kids[(size_t)ch].basis;
            rc = rr_featmat_glm_rff(fm, b, in.dX[ch], in.x_dtype[ch], in.ldx[ch], o->col0[(size_t)ch], o->dTk[(size_t)ch]);
            if (rc == RR_OK)
                rc = rr_featmat_glm_rff(fm, b, in.dX[ch], in.x_dtype[ch], in.ldx[ch], o->col0[(size_t)ch] + 2 * b->n,
                                        o->dTk[(size_t)ch] + (int64_t)b->d * b->n);
        }
    return rc;
}

static void sgd_update_args(rr_glm_sgd *o, const SgdStepIn &in, FmPass2 &s, SgdUpdArgs &a) {
    const int64_t fk = o->fk;
    a.x = o->x; a.red = o->red; a.Edm = s.mc + 2 * fk; a.EdC = s.mc + 3 * fk; a.aux = s.kacc + s.kcap; a.lower = o->lower; a.upper = o->upper;
    a.islog = o->islog; a.z = o->z; a.s1 = o->s1; a.s2 = o->s2;
    a.slice_of_f = o->slice_of_f; a.slice_lo = o->slice_lo; a.slice_hi = o->slice_hi; a.h_of_ls = o->h_of_ls; a.ls_plain = o->ls_plain;
    a.F = o->F; a.K = o->K; a.nkids = o->nkids; a.n_lik = o->n_lik; a.n_ls = o->n_ls; a.updater = o->updater; a.L = in.L; a.np = o->np;
    a.bmag = in.bmag; a.nrows = (double)in.rows_total;
    a.tot = in.totals_on_device ? s.kacc + 2 * s.kcap : nullptr;
    for (int i = 0; i < 4; ++i) a.up[i] = o->up[i];
    const double tt = (double)(o->t + 1);
    a.b1t = 1.0 - pow(o->up[1], tt);
    a.b2t = 1.0 - pow(o->up[2], tt);
}

// (dT summed over all rows:) the length scales' gradient and update -- the next step's features may start --, then
// Ed = dfs Phi and its reductions over the samples: [Edm | EdC] and [llsum | aux] of this loop's rows are complete
static int sgd_step_middle(rr_glm_sgd *o, const SgdStepIn &in) {
    rr_featmat *fm = sgd_step_fm(o);
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int K = o->K, F = o->F, nk = o->nkids, L = in.L, par = (int)(o->t & 1);
    const int64_t fk = o->fk, n_main = 2 * fk + nk + o->n_lik;
    const unsigned nb_main = (unsigned)((n_main + 255) / 256), nb_ls = (unsigned)((o->n_ls + 255) / 256);
    const double *xpar = o->n_lik ? o->x + 2 * fk + nk : nullptr;
    hipStream_t s0 = c->stream;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    SgdUpdArgs a;
    sgd_update_args(o, in, s, a);
    const int npairs = K * (K + 1) / 2;
    if (o->n_ls) {
        hipLaunchKernelGGL(rr_glm_sgd_sums_kernel, dim3((unsigned)o->n_h), dim3(256), 0, s0, o->x, F, K, nk, o->slice_lo, o->slice_hi,
                           o->hrows, o->n_h, o->red, npairs + nk);
        a.p0 = n_main; a.p1 = o->np; a.npart = o->npart + nb_main;
        hipLaunchKernelGGL(rr_glm_sgd_update_kernel, dim3(nb_ls), dim3(256), 0, s0, a);
        RR_CHECK_HIP(hipGetLastError());
    }
    RR_CHECK_HIP(hipEventRecord(o->e_ls[par], s0));
    if (in.rows == 0) return RR_OK;  // (zeroed by the front part)
    int rc = glm_pipeline(fm, s, in.dy, in.drowarg, in.dtype, in.lik, 1.0, K, L, false, xpar, 2);
    if (rc != RR_OK) return rc;
    hipLaunchKernelGGL(rr_glm_reduce_kernel, dim3((unsigned)((fk + 255) / 256)), dim3(256), 0, s0, s.Ed, s.Ee, o->x + fk, F, K,
                       L, fm->ld, s.mc + 2 * fk, s.mc + 3 * fk);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// ([Edm | EdC], [llsum | aux] summed over all rows:) the mixture's sums, the update of (m, C, regularisers, variance), the
// step's record
static int sgd_step_back(rr_glm_sgd *o, const SgdStepIn &in) {
    rr_featmat *fm = sgd_step_fm(o);
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int K = o->K, F = o->F, nk = o->nkids, L = in.L;
    const int64_t fk = o->fk, n_main = 2 * fk + nk + o->n_lik;
    const unsigned nb_main = (unsigned)((n_main + 255) / 256), nb_ls = (unsigned)((o->n_ls + 255) / 256);
    hipStream_t s0 = c->stream;
    FmPass2 &s = *(FmPass2 *)fm->pass2;
    SgdUpdArgs a;
    sgd_update_args(o, in, s, a);
    const int npairs = K * (K + 1) / 2;
    hipLaunchKernelGGL(rr_glm_sgd_sums_kernel, dim3((unsigned)(npairs + nk)), dim3(256), 0, s0, o->x, F, K, nk, o->slice_lo, o->slice_hi,
                       o->hrows, o->n_h, o->red, 0);
    a.p0 = 0; a.p1 = n_main; a.npart = o->npart;
    hipLaunchKernelGGL(rr_glm_sgd_update_kernel, dim3(nb_main), dim3(256), 0, s0, a);
    RR_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rr_glm_sgd_finish_kernel, dim3(1), dim3(256), 0, s0, o->x, o->red, s.kacc, o->npart, (int)(nb_main + nb_ls), F, K, L,
                       nk, o->slice_lo, o->slice_hi, o->n_lik, in.llconst, (double)in.rows_total, in.bmag, o->objs + o->t, o->norms + o->t,
                       a.tot);
    RR_CHECK_HIP(hipGetLastError());
    RR_CHECK_HIP(hipEventRecord(o->ev[o->t & 1], c->stream));
    o->t += 1;
    return RR_OK;
}

int rr_glm_sgd_step(rr_glm_sgd *o, const void *const *dX, const int *x_dtype, const int64_t *ldx, int64_t rows, const void *dy,
                    const void *drowarg, int dtype, int lik, double llconst, double bmag, int L, const float *dE, uint64_t seed,
                    uint64_t key) {
    RR_REQUIRE(o != nullptr && dX != nullptr && x_dtype != nullptr && ldx != nullptr && dy != nullptr, "rr_glm_sgd_step: null argument");
    const SgdStepIn in = {dX, x_dtype, ldx, rows, rows, dy, drowarg, dtype, lik, llconst, bmag, L, dE, seed, key};
    int rc = sgd_step_check(o, in, 1, "rr_glm_sgd_step");
    if (rc == RR_OK) rc = sgd_step_front(o, in);
    if (rc == RR_OK) rc = sgd_step_middle(o, in);
    if (rc == RR_OK) rc = sgd_step_back(o, in);
    return rc;
}

// The same step with the minibatch's rows spread over the n members of a device group (rr_comm_init_all): member i runs the
// products on ITS rows, the three row sums are added over the members in HBM (dT before the length scales' update, [Edm | EdC]
// and [llsum | aux] before the main update), and every member then makes the SAME update of its own copy of z -- the copies
// stay bit-identical (the all-reduce leaves the same bits with every member).  One host thread queues everything.
int rr_glm_sgd_group_step(int n, rr_glm_sgd *const *loops, rr_comm *const *comms, const rr_glm_sgd_batch *batches, int dtype, int lik,
                          double llconst, double bmag, int L, uint64_t seed, uint64_t key) {
    RR_REQUIRE(n >= 1 && loops != nullptr && comms != nullptr && batches != nullptr, "rr_glm_sgd_group_step: null argument");
    std::vector<SgdStepIn> in((size_t)n);
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        RR_REQUIRE(loops[i] != nullptr && comms[i] != nullptr, "rr_glm_sgd_group_step: member %d: null loop or communicator", i);
        RR_REQUIRE(batches[i].rows >= 0, "rr_glm_sgd_group_step: member %d: rows out of range", i);
        total += batches[i].rows;
    }
    RR_REQUIRE(total >= 1, "rr_glm_sgd_group_step: a minibatch without rows");
    rr_glm_sgd *o0 = loops[0];
    for (int i = 0; i < n; ++i) {
        rr_glm_sgd *o = loops[i];
        RR_REQUIRE(o->np == o0->np && o->t == o0->t && o->K == o0->K && o->F == o0->F && o->dT_count == o0->dT_count,
                   "rr_glm_sgd_group_step: member %d's loop is not a copy of member 0's (shape or step count)", i);
        RR_REQUIRE(rr_comm_ctx(comms[i]) == o->fm->ctx, "rr_glm_sgd_group_step: member %d: loop and communicator live on different contexts", i);
        const rr_glm_sgd_batch &b = batches[i];
        in[(size_t)i] = SgdStepIn{b.dX, b.x_dtype, b.ldx, b.rows, total, b.dy, b.drowarg, dtype, lik, llconst, bmag, L, b.dE, seed, key};
        const int rc = sgd_step_check(o, in[(size_t)i], 0, "rr_glm_sgd_group_step");
        if (rc != RR_OK) return rc;
    }
    std::vector<double *> bufs((size_t)n);
    static const bool timing = getenv("RR_GLM_GROUP_TIMING") != nullptr;
    const int64_t t_call = timing ? sgd_now_ns() : 0;
    SgdGroupWorkers &pool = SgdGroupWorkers::get();
    int rc = pool.run(n, [&](int i) { return sgd_step_front(loops[i], in[(size_t)i]); });
    if (rc != RR_OK) return rc;
    if (o0->n_ls) {
        for (int i = 0; i < n; ++i) bufs[(size_t)i] = loops[i]->dT;
        rc = rr_comm_group_allreduce_dev(comms, n, bufs.data(), o0->dT_count, RR_COMM_SUM);
        if (rc != RR_OK) return rc;
    }
    rc = pool.run(n, [&](int i) { return sgd_step_middle(loops[i], in[(size_t)i]); });
    if (rc != RR_OK) return rc;
    // [Edm | EdC] and [llsum | aux]: one run of 2 F K + 2 K doubles when the scratch was made for this K (fm_glm_scratch), else two
    const int kcap = ((FmPass2 *)sgd_step_fm(o0)->pass2)->kcap;
    bool one_run = true;
    for (int i = 0; i < n; ++i) {
        FmPass2 *s = (FmPass2 *)sgd_step_fm(loops[i])->pass2;
        RR_REQUIRE(s->kcap == kcap, "rr_glm_sgd_group_step: member %d's scratch differs", i);
        bufs[(size_t)i] = s->mc + 2 * o0->fk;
        one_run = one_run && s->kacc == s->mc + 4 * o0->fk;
    }
    rc = rr_comm_group_allreduce_dev(comms, n, bufs.data(), 2 * o0->fk + (one_run ? 2 * (int64_t)kcap : 0), RR_COMM_SUM);
    if (rc != RR_OK) return rc;
    if (!one_run) {
        for (int i = 0; i < n; ++i) bufs[(size_t)i] = ((FmPass2 *)sgd_step_fm(loops[i])->pass2)->kacc;
        rc = rr_comm_group_allreduce_dev(comms, n, bufs.data(), 2 * (int64_t)kcap, RR_COMM_SUM);
    }
    if (rc == RR_OK) rc = pool.run(n, [&](int i) { return sgd_step_back(loops[i], in[(size_t)i]); });
    if (timing) {
        o0->call_ns += sgd_now_ns() - t_call;
        o0->group_steps += 1;
    }
    return rc;
}

// The same step on ONE rank of a one-process-per-GPU job (rr_comm_init_rank): this rank's minibatch -- rows of ITS shard of the
// data, cut by its own generator -- and two all-reduces over the ranks, as between the members of a group: dT, then
// [Edm | EdC | llsum | aux | llconst | rows] (the f-independent constant and the row count are sums over ranks too: they
// travel in the same message and the Gaussian's terms / the step's record read them from HBM).  Every rank makes the same
// update of its copy of z.  Asynchronous like rr_glm_sgd_step: nothing is waited for but step t - 2's event.
int rr_glm_sgd_dist_step(rr_glm_sgd *o, rr_comm *comm, const void *const *dX, const int *x_dtype, const int64_t *ldx, int64_t rows,
                         const void *dy, const void *drowarg, int dtype, int lik, double llconst, double bmag, int L, const float *dE,
                         uint64_t seed, uint64_t key) {
    RR_REQUIRE(o != nullptr && comm != nullptr, "rr_glm_sgd_dist_step: null argument");
    RR_REQUIRE(rr_comm_ctx(comm) == o->fm->ctx, "rr_glm_sgd_dist_step: loop and communicator live on different contexts");
    SgdStepIn in = {dX, x_dtype, ldx, rows, rows, dy, drowarg, dtype, lik, llconst, bmag, L, dE, seed, key};
    in.totals_on_device = true;
    int rc = sgd_step_check(o, in, 0, "rr_glm_sgd_dist_step");
    if (rc == RR_OK) rc = sgd_step_front(o, in);
    if (rc != RR_OK) return rc;
    if (o->n_ls) {
        rc = rr_comm_allreduce_dev(comm, o->dT, o->dT_count, RR_COMM_SUM);
        if (rc != RR_OK) return rc;
    }
    rc = sgd_step_middle(o, in);
    if (rc != RR_OK) return rc;
    rr_ctx *c = o->fm->ctx;
    FmPass2 *s = (FmPass2 *)sgd_step_fm(o)->pass2;
    double *tot = s->kacc + 2 * s->kcap;
    hipLaunchKernelGGL(rr_glm_sgd_set_totals_kernel, dim3(1), dim3(1), 0, c->stream, tot, llconst, (double)rows);
    RR_CHECK_HIP(hipGetLastError());
    if (s->kacc == s->mc + 4 * o->fk) {  // one run (the scratch was made for this K)
        rc = rr_comm_allreduce_dev(comm, s->mc + 2 * o->fk, 2 * o->fk + 2 * (int64_t)s->kcap + 2, RR_COMM_SUM);
    } else {
        rc = rr_comm_allreduce_dev(comm, s->mc + 2 * o->fk, 2 * o->fk, RR_COMM_SUM);
        if (rc == RR_OK) rc = rr_comm_allreduce_dev(comm, s->kacc, 2 * (int64_t)s->kcap + 2, RR_COMM_SUM);
    }
    if (rc == RR_OK) rc = sgd_step_back(o, in);
    return rc;
}

int rr_glm_sgd_read(rr_glm_sgd *o, double *z, double *objs, double *norms, int64_t *steps) {
    RR_REQUIRE(o != nullptr, "rr_glm_sgd_read: null argument");
    rr_ctx *c = o->fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (z) RR_CHECK_HIP(hipMemcpy(z, o->z, (size_t)o->np * 8, hipMemcpyDeviceToHost));
    if (objs && o->t) RR_CHECK_HIP(hipMemcpy(objs, o->objs, (size_t)o->t * 8, hipMemcpyDeviceToHost));
    if (norms && o->t) RR_CHECK_HIP(hipMemcpy(norms, o->norms, (size_t)o->t * 8, hipMemcpyDeviceToHost));
    if (steps) *steps = o->t;
    return RR_OK;
}

int rr_glm_sgd_objective(rr_glm_sgd *o, int64_t step, double *obj) {
    RR_REQUIRE(o != nullptr && obj != nullptr && step >= 0 && step < o->t, "rr_glm_sgd_objective: no such step");
    rr_ctx *c = o->fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemcpy(obj, o->objs + step, 8, hipMemcpyDeviceToHost));
    return RR_OK;
}

void rr_glm_sgd_destroy(rr_glm_sgd *o) {
    if (!o) return;
    (void)hipSetDevice(o->fm->ctx->device);
    (void)hipStreamSynchronize(o->fm->ctx->stream);
    if (o->sfeat) (void)hipStreamSynchronize(o->sfeat);
    if (o->group_steps > 0)
        fprintf(stderr, "rr_glm_sgd_group_step: %lld steps, %.1f us of host time per step, %.1f us of it waiting for step t - 2 (member 0)\n",
                (long long)o->group_steps, 1e-3 * (double)o->call_ns / (double)o->group_steps, 1e-3 * (double)o->wait_ns / (double)o->group_steps);
    sgd_free(o);
}

}  // extern "C"

// =============================================================================================
// The float64 feature matrix of a concatenated basis (round 3): BasisCat.transform's hstack (basis_functions.py:1599-1627)
// in float64 in HBM, reduced to Phi^T Phi / Phi^T y (slm.py:146,157) by the f64 MFMA SYRK and consumed by the second pass
// of _elbo (slm.py:160-197) and predict_moments (slm.py:240-244) in float64 -- the reference's arithmetic end to end for
// concatenations with a dtype="f64" child (north star: 1e-5 relative in fp64).  The same data flow as rr_featmat /
// pass2_run64: every child writes its column block, P^T by a transposing pass, U = P C on rr_gemm_tn_f64_kernel, the
// float64 epilogue kernels above.  Written for the resident fit (CatFitState): no GLM step, no split engines.
// =============================================================================================
int rr_launch_syrk_f64(rr_ctx *c, const double *P, int64_t rows, int64_t ldp, int F, double *dG, int lower_tri = 0);  // rr_rff.hip

struct rr_featmat64 {
    rr_ctx *ctx = nullptr;
    double *P = nullptr;
    int64_t max_rows = 0, ld = 0, rows = 0, rows_pad = 0;  // ld % 128 == 0; rows_pad = rows rounded up to 128
    int F = 0;
    int64_t covered = 0;
    std::vector<std::pair<int64_t, int64_t>> spans;  // column intervals put since begin (overlaps refused)
    // second pass / prediction scratch (allocated by the first pass2_begin)
    double *Pt = nullptr, *U = nullptr, *Cp = nullptr, *Craw = nullptr, *m = nullptr, *dot = nullptr, *err = nullptr,
           *sq = nullptr, *vf = nullptr;
    bool have_rows = false;
};

template <typename TX>
__global__ void __launch_bounds__(256)
rr_fm64_linear_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, int d, int onescol, double *__restrict__ P, int64_t ldp) {
    const int w = d + onescol;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * w) return;
    const int64_t r = i / w;
    const int c = (int)(i % w);
    P[r * ldp + c] = (onescol && c == 0) ? 1.0 : (double)X[r * ldx + (c - onescol)];
}

template <typename TS>
__global__ void __launch_bounds__(256)
rr_fm64_copy_cols_kernel(const TS *__restrict__ src, int64_t N, int64_t lds_, int ncols, double *__restrict__ P, int64_t ldp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * ncols) return;
    P[(i / ncols) * ldp + (i % ncols)] = (double)src[(i / ncols) * lds_ + (i % ncols)];
}

__global__ void __launch_bounds__(256) rr_fm64_zero_padcols_kernel(double *P, int64_t rows, int64_t ld, int F) {
    const int64_t w = ld - F;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows * w) P[(i / w) * ld + F + (i % w)] = 0.0;
}

template <typename TY>
__global__ void __launch_bounds__(256)
rr_fm64_gemv_t_kernel(const double *__restrict__ P, const TY *__restrict__ y, int64_t rows, int F, int64_t ldp,
                      double *__restrict__ bvec, int rows_per_block, int64_t bdet) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (c >= F) return;
    double acc = 0.0;
    for (int64_t r = r0; r < r1; ++r) acc = fma(P[r * ldp + c], (double)y[r], acc);
    rr_acc_out(bvec, bdet, blockIdx.y, c, acc);
}

template <typename TY>
__global__ void __launch_bounds__(256) rr_fm64_yty_kernel(const TY *__restrict__ y, int64_t N, double *out, int64_t det) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)y[i];
        acc = fma(v, v, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rr_acc_out(out, det, blockIdx.x, 0, part[0] + part[1] + part[2] + part[3]);
}

static int fm64_claim(rr_featmat64 *fm, int64_t col0, int64_t width, const char *who) {
    const int64_t c1 = col0 + width;
    size_t pos = 0;
    while (pos < fm->spans.size() && fm->spans[pos].first < col0) ++pos;
    const bool clash = (pos > 0 && fm->spans[pos - 1].second > col0) || (pos < fm->spans.size() && fm->spans[pos].first < c1);
    RR_REQUIRE(!clash, "%s: columns [%lld, %lld) overlap a block already written since rr_featmat64_begin", who, (long long)col0,
               (long long)c1);
    fm->spans.insert(fm->spans.begin() + (std::ptrdiff_t)pos, std::make_pair(col0, c1));
    fm->covered += width;
    return RR_OK;
}
#define RR_FM64_REQUIRE_FILLED(fm, who)                                                                             \
    RR_REQUIRE((fm)->rows == 0 || (fm)->covered == (fm)->F,                                                         \
               who ": only %lld of the %d columns were written since rr_featmat64_begin", (long long)(fm)->covered, (fm)->F)

static void fm64_free_scratch(rr_featmat64 *fm) {
    void *q[] = {fm->Pt, fm->U, fm->Cp, fm->Craw, fm->m, fm->dot, fm->err, fm->sq, fm->vf};
    for (void *x : q)
        if (x) (void)hipFree(x);
    fm->Pt = fm->U = fm->Cp = fm->Craw = fm->m = fm->dot = fm->err = fm->sq = fm->vf = nullptr;
}

static int fm64_scratch(rr_featmat64 *fm) {
    if (fm->Pt) return RR_OK;
    const int64_t Fp = fm->ld;
    hipError_t ea = hipMalloc((void **)&fm->Pt, (size_t)Fp * fm->max_rows * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->U, (size_t)fm->max_rows * Fp * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->Cp, (size_t)Fp * Fp * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->Craw, (size_t)fm->F * fm->F * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->m, (size_t)Fp * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->dot, (size_t)fm->max_rows * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->err, (size_t)fm->max_rows * 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->sq, 8);
    if (ea == hipSuccess) ea = hipMalloc((void **)&fm->vf, (size_t)fm->max_rows * 8);
    if (ea != hipSuccess) {
        (void)hipGetLastError();
        fm64_free_scratch(fm);
        rr_set_error("float64 feature matrix: device allocation of the second pass' scratch failed");
        return RR_ERR_OOM;
    }
    return RR_OK;
}

// dot = P m, P^T, U = P C for the rows currently in the matrix
static int fm64_products(rr_featmat64 *fm) {
    rr_ctx *c = fm->ctx;
    const int64_t Fp = fm->ld, rp = fm->rows_pad;
    hipLaunchKernelGGL(rr_transpose_f64_kernel, dim3((unsigned)(Fp / 64), (unsigned)(rp / 64)), dim3(256), 0, c->stream, fm->P,
                       fm->rows, Fp, fm->Pt, fm->max_rows);
    RR_CHECK_HIP(hipGetLastError());
    return rr_launch_gemm_tn_f64(c, fm->Pt, fm->max_rows, fm->Cp, Fp, fm->U, Fp, ((int64_t)fm->F + 15) / 16 * 16, rp, Fp, 0, 0);
}

extern "C" {

int rr_featmat64_create(rr_ctx *ctx, int64_t max_rows, int64_t F, rr_featmat64 **out) {
    RR_REQUIRE(ctx != nullptr && out != nullptr, "rr_featmat64_create: null argument");
    *out = nullptr;
    RR_REQUIRE(max_rows >= 1 && F >= 1 && F < 46340, "rr_featmat64_create: bad shape");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    rr_featmat64 *fm = new rr_featmat64();
    fm->ctx = ctx;
    fm->F = (int)F;
    fm->ld = (F + 127) / 128 * 128;
    fm->max_rows = (max_rows + 127) / 128 * 128;
    if (hipMalloc((void **)&fm->P, (size_t)fm->max_rows * fm->ld * 8) != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_featmat64_create: hipMalloc(%zu bytes) failed", (size_t)fm->max_rows * fm->ld * 8);
        delete fm;
        return RR_ERR_OOM;
    }
    *out = fm;
    return RR_OK;
}

void rr_featmat64_destroy(rr_featmat64 *fm) {
    if (!fm) return;
    (void)hipSetDevice(fm->ctx->device);
    (void)hipStreamSynchronize(fm->ctx->stream);
    if (fm->P) (void)hipFree(fm->P);
    fm64_free_scratch(fm);
    delete fm;
}

int rr_featmat64_begin(rr_featmat64 *fm, int64_t rows) {
    RR_REQUIRE(fm != nullptr && rows >= 0 && rows <= fm->max_rows, "rr_featmat64_begin: rows out of range");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    fm->rows = rows;
    fm->rows_pad = (rows + 127) / 128 * 128;
    fm->covered = 0;
    fm->spans.clear();
    fm->have_rows = false;
    // the children overwrite every column of [0, F) for the rows [0, rows) (checked by the consumers); the padding is ours:
    // pad rows up to the next multiple of 128 (SYRK k-blocks, GEMM tiles) and the pad columns [F, ld) of the data rows
    if (fm->rows_pad > rows)
        RR_CHECK_HIP(hipMemsetAsync(fm->P + rows * fm->ld, 0, (size_t)(fm->rows_pad - rows) * fm->ld * 8, c->stream));
    const int64_t w = fm->ld - fm->F;
    if (w > 0 && rows > 0) {
        hipLaunchKernelGGL(rr_fm64_zero_padcols_kernel, dim3((unsigned)((rows * w + 255) / 256)), dim3(256), 0, c->stream, fm->P, rows,
                           fm->ld, fm->F);
        RR_CHECK_HIP(hipGetLastError());
    }
    return RR_OK;
}

int rr_featmat64_put_rff(rr_featmat64 *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *lenscale,
                         int n_ls, int64_t col0) {
    RR_REQUIRE(fm != nullptr && b != nullptr && b->kind == RR_KIND_RFF, "rr_featmat64_put_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat64_put_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat64_put_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad, "rr_featmat64_put_rff: device X needs ldx >= rr_rff_padded_dim() = %d", b->dpad);
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK || fm->rows == 0) return rc;
    RR_REQUIRE(dX != nullptr, "rr_featmat64_put_rff: null X");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    rc = fm64_claim(fm, col0, 2 * (int64_t)b->n, "rr_featmat64_put_rff");
    if (rc != RR_OK) return rc;
    // float64 features whatever the basis' own arithmetic (its W is resident in float64 as well); whole 16-row tiles: the
    // rows up to the next multiple of 16 are written as zeros, like begin() left them
    return rr_features_rowmajor_f64(b, dX, x_dtype, fm->rows, (fm->rows + 15) / 16 * 16, ldx, fm->P + col0, fm->ld, false);
}

int rr_featmat64_put_linear(rr_featmat64 *fm, const void *dX, int x_dtype, int64_t ldx, int d, int onescol, int64_t col0) {
    RR_REQUIRE(fm != nullptr && d >= 1 && ldx >= d, "rr_featmat64_put_linear: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat64_put_linear: bad dtype");
    const int w = d + (onescol ? 1 : 0);
    RR_REQUIRE(col0 >= 0 && col0 + w <= fm->F, "rr_featmat64_put_linear: columns out of range");
    if (fm->rows == 0) return RR_OK;
    RR_REQUIRE(dX != nullptr, "rr_featmat64_put_linear: null X");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    int rc = fm64_claim(fm, col0, w, "rr_featmat64_put_linear");
    if (rc != RR_OK) return rc;
    const dim3 grid((unsigned)((fm->rows * w + 255) / 256));
    if (x_dtype == RR_F32)
        hipLaunchKernelGGL(rr_fm64_linear_kernel<float>, grid, dim3(256), 0, fm->ctx->stream, (const float *)dX, fm->rows, ldx, d,
                           onescol ? 1 : 0, fm->P + col0, fm->ld);
    else
        hipLaunchKernelGGL(rr_fm64_linear_kernel<double>, grid, dim3(256), 0, fm->ctx->stream, (const double *)dX, fm->rows, ldx, d,
                           onescol ? 1 : 0, fm->P + col0, fm->ld);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_featmat64_put_host(rr_featmat64 *fm, const void *Phi, int dtype, int64_t ncols, int64_t ldphi, int64_t col0) {
    RR_REQUIRE(fm != nullptr && ncols >= 1 && ldphi >= ncols, "rr_featmat64_put_host: bad argument");
    RR_REQUIRE(dtype == RR_F32 || dtype == RR_F64, "rr_featmat64_put_host: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + ncols <= fm->F, "rr_featmat64_put_host: columns out of range");
    if (fm->rows == 0) return RR_OK;
    RR_REQUIRE(Phi != nullptr, "rr_featmat64_put_host: null Phi");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = dtype == RR_F32 ? 4 : 8;
    void *raw = nullptr;
    RR_CHECK_HIP(hipMalloc(&raw, (size_t)fm->rows * ncols * es));
    hipError_t e = hipMemcpy2DAsync(raw, (size_t)ncols * es, Phi, (size_t)ldphi * es, (size_t)ncols * es, (size_t)fm->rows,
                                    hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((fm->rows * ncols + 255) / 256));
        if (dtype == RR_F32)
            hipLaunchKernelGGL(rr_fm64_copy_cols_kernel<float>, grid, dim3(256), 0, c->stream, (const float *)raw, fm->rows, ncols,
                               (int)ncols, fm->P + col0, fm->ld);
        else
            hipLaunchKernelGGL(rr_fm64_copy_cols_kernel<double>, grid, dim3(256), 0, c->stream, (const double *)raw, fm->rows, ncols,
                               (int)ncols, fm->P + col0, fm->ld);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(raw);
    if (e != hipSuccess) {
        rr_set_error("rr_featmat64_put_host: copy failed: %s", hipGetErrorString(e));
        return RR_ERR_HIP;
    }
    return fm64_claim(fm, col0, ncols, "rr_featmat64_put_host");
}

int rr_featmat64_gram(rr_featmat64 *fm, const void *dy, int y_dtype, double *dG, double *db, double *dyty) {
    RR_REQUIRE(fm != nullptr && dG != nullptr, "rr_featmat64_gram: null argument");
    RR_REQUIRE((dy == nullptr) == (db == nullptr) && (dy == nullptr) == (dyty == nullptr),
               "rr_featmat64_gram: y, b and yty must be given together");
    RR_REQUIRE(dy == nullptr || y_dtype == RR_F32 || y_dtype == RR_F64, "rr_featmat64_gram: bad dtype");
    RR_FM64_REQUIRE_FILLED(fm, "rr_featmat64_gram");
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = RR_OK;
    if (dy) {
        const int rpb = 512;
        const dim3 gg((unsigned)((fm->F + 255) / 256), (unsigned)((fm->rows + rpb - 1) / rpb));
        int yb = (int)((fm->rows + 255) / 256);
        if (yb > c->num_cu * 8) yb = c->num_cu * 8;
        const bool det = c->deterministic != 0;
        double *bdst = db, *ydst = dyty;
        void *part = nullptr;
        if (det) {
            rc = rr_det_scratch(c, (size_t)gg.y * (size_t)fm->F * 8, &part);
            if (rc != RR_OK) return rc;
            bdst = (double *)part;
        }
        if (y_dtype == RR_F32)
            hipLaunchKernelGGL(rr_fm64_gemv_t_kernel<float>, gg, dim3(256), 0, c->stream, fm->P, (const float *)dy, fm->rows, fm->F,
                               fm->ld, bdst, rpb, (int64_t)(det ? fm->F : 0));
        else
            hipLaunchKernelGGL(rr_fm64_gemv_t_kernel<double>, gg, dim3(256), 0, c->stream, fm->P, (const double *)dy, fm->rows, fm->F,
                               fm->ld, bdst, rpb, (int64_t)(det ? fm->F : 0));
        RR_CHECK_HIP(hipGetLastError());
        if (det) {
            rc = rr_det_reduce(c, bdst, gg.y, fm->F, fm->F, db);
            if (rc == RR_OK) rc = rr_det_scratch(c, (size_t)yb * 8, &part);
            if (rc != RR_OK) return rc;
            ydst = (double *)part;
        }
        if (y_dtype == RR_F32)
            hipLaunchKernelGGL(rr_fm64_yty_kernel<float>, dim3(yb), dim3(256), 0, c->stream, (const float *)dy, fm->rows, ydst,
                               (int64_t)(det ? 1 : 0));
        else
            hipLaunchKernelGGL(rr_fm64_yty_kernel<double>, dim3(yb), dim3(256), 0, c->stream, (const double *)dy, fm->rows, ydst,
                               (int64_t)(det ? 1 : 0));
        RR_CHECK_HIP(hipGetLastError());
        if (det) {
            rc = rr_det_reduce(c, ydst, yb, 1, 1, dyty);
            if (rc != RR_OK) return rc;
        }
    }
    return rr_launch_syrk_f64(c, fm->P, fm->rows_pad, fm->ld, fm->F, dG);
}

// m: host (F); C: host (F, F) or -- c_on_device -- device (F, F), float64
int rr_featmat64_pass2_begin(rr_featmat64 *fm, const double *m, const double *C, int c_on_device) {
    RR_REQUIRE(fm != nullptr && m != nullptr && C != nullptr, "rr_featmat64_pass2_begin: null argument");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = fm64_scratch(fm);
    if (rc != RR_OK) return rc;
    const int64_t F = fm->F, Fp = fm->ld;
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    RR_CHECK_HIP(hipMemset(fm->m, 0, (size_t)Fp * 8));
    RR_CHECK_HIP(hipMemcpy(fm->m, m, (size_t)F * 8, hipMemcpyHostToDevice));
    const double *Csrc = C;
    if (!c_on_device) {
        RR_CHECK_HIP(hipMemcpy(fm->Craw, C, (size_t)F * F * 8, hipMemcpyHostToDevice));
        Csrc = fm->Craw;
    }
    hipLaunchKernelGGL(rr_pad_c64_kernel, dim3((unsigned)((Fp * Fp + 255) / 256)), dim3(256), 0, c->stream, Csrc, F, fm->Cp, Fp);
    RR_CHECK_HIP(hipGetLastError());
    RR_CHECK_HIP(hipMemsetAsync(fm->sq, 0, 8, c->stream));
    fm->have_rows = false;
    return RR_OK;
}

int rr_featmat64_pass2_rows(rr_featmat64 *fm, const void *dy, int y_dtype) {
    RR_REQUIRE(fm != nullptr && fm->Pt != nullptr, "rr_featmat64_pass2_rows: call rr_featmat64_pass2_begin first");
    RR_FM64_REQUIRE_FILLED(fm, "rr_featmat64_pass2_rows");
    RR_REQUIRE(dy != nullptr && (y_dtype == RR_F32 || y_dtype == RR_F64), "rr_featmat64_pass2_rows: bad y");
    fm->have_rows = true;
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = fm64_products(fm);
    if (rc != RR_OK) return rc;
    hipLaunchKernelGGL(rr_rows64_kernel<0>, dim3((unsigned)((fm->rows + 3) / 4)), dim3(256), 0, c->stream, fm->P, fm->U, fm->m,
                       fm->rows, fm->F, fm->ld, fm->dot, fm->vf);
    const int64_t eb = (fm->rows + 255) / 256;
    double *sq = fm->sq;
    const int64_t det = c->deterministic ? 1 : 0;
    if (det) {
        void *part = nullptr;
        rc = rr_det_scratch(c, (size_t)eb * 8, &part);
        if (rc != RR_OK) return rc;
        sq = (double *)part;
    }
    if (y_dtype == RR_F32)
        hipLaunchKernelGGL(rr_err64_kernel<float>, dim3((unsigned)eb), dim3(256), 0, c->stream, (const float *)dy, fm->dot, fm->rows,
                           fm->err, sq, det);
    else
        hipLaunchKernelGGL(rr_err64_kernel<double>, dim3((unsigned)eb), dim3(256), 0, c->stream, (const double *)dy, fm->dot, fm->rows,
                           fm->err, sq, det);
    RR_CHECK_HIP(hipGetLastError());
    return det ? rr_det_reduce(c, sq, eb, 1, 1, fm->sq) : RR_OK;
}

// dT (d, n) += X^T A for the random Fourier child at columns [col0, col0 + 2n): its hyper-gradient contraction
int rr_featmat64_pass2_rff(rr_featmat64 *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, int64_t col0, double *dT) {
    RR_REQUIRE(fm != nullptr && fm->Pt != nullptr && fm->have_rows, "rr_featmat64_pass2_rff: call rr_featmat64_pass2_rows first");
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF && dT != nullptr, "rr_featmat64_pass2_rff: bad argument");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat64_pass2_rff: bad dtype");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat64_pass2_rff: columns out of range");
    RR_REQUIRE(ldx >= b->dpad && !b->large, "rr_featmat64_pass2_rff: device X needs ldx >= rr_rff_padded_dim() = %d, Xdim <= 128", b->dpad);
    if (fm->rows == 0) return RR_OK;
    RR_REQUIRE(dX != nullptr, "rr_featmat64_pass2_rff: null X");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int n = b->n;
    const int64_t mrows = fm->rows;
    const int fblocks = (n + 255) / 256;
    int64_t rpb = (mrows * fblocks + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
    if (rpb < 64) rpb = 64;
    if ((mrows + rpb - 1) / rpb > 65535) rpb = (mrows + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((mrows + rpb - 1) / rpb));
    const int64_t tcount = (int64_t)b->d * n, tdet = c->deterministic ? tcount : 0;
    double *Tdst = dT;
    if (tdet) {
        void *part = nullptr;
        int rc = rr_det_scratch(c, (size_t)grid.y * (size_t)tcount * 8, &part);
        if (rc != RR_OK) return rc;
        Tdst = (double *)part;
    }
#define RR_FGT(DM, TX)                                                                                                    \
    hipLaunchKernelGGL((rr_grad_t64_kernel<DM, TX>), grid, dim3(256), 0, c->stream, (const TX *)dX, mrows, ldx, fm->P + col0, \
                       fm->U + col0, fm->ld, fm->err, fm->m + col0, n, b->d, Tdst, (int)rpb, tdet)
#define RR_FGTD(DM)                          \
    if (x_dtype == RR_F32) RR_FGT(DM, float); \
    else RR_FGT(DM, double)
    switch (b->dpad) {
        case 8: RR_FGTD(8); break;
        case 16: RR_FGTD(16); break;
        case 32: RR_FGTD(32); break;
        case 64: RR_FGTD(64); break;
        default: RR_FGTD(128); break;
    }
#undef RR_FGTD
#undef RR_FGT
    RR_CHECK_HIP(hipGetLastError());
    return tdet ? rr_det_reduce(c, Tdst, grid.y, tcount, tcount, dT) : RR_OK;
}

int rr_featmat64_pass2_end(rr_featmat64 *fm, double *sqErr) {
    RR_REQUIRE(fm != nullptr && fm->Pt != nullptr && sqErr != nullptr, "rr_featmat64_pass2_end: bad argument");
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    RR_CHECK_HIP(hipMemcpyAsync(sqErr, fm->sq, 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    return RR_OK;
}

// (Ey, Vf) = (P m, rowsum((P C) o P)) for the rows currently in the matrix (after rr_featmat64_pass2_begin), host float64
int rr_featmat64_predict_rows(rr_featmat64 *fm, double *Ey, double *Vf) {
    RR_REQUIRE(fm != nullptr && fm->Pt != nullptr && Ey != nullptr && Vf != nullptr, "rr_featmat64_predict_rows: bad argument");
    RR_FM64_REQUIRE_FILLED(fm, "rr_featmat64_predict_rows");
    if (fm->rows == 0) return RR_OK;
    rr_ctx *c = fm->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = fm64_products(fm);
    if (rc != RR_OK) return rc;
    hipLaunchKernelGGL(rr_rows64_kernel<1>, dim3((unsigned)((fm->rows + 3) / 4)), dim3(256), 0, c->stream, fm->P, fm->U, fm->m,
                       fm->rows, fm->F, fm->ld, fm->dot, fm->vf);
    RR_CHECK_HIP(hipGetLastError());
    RR_CHECK_HIP(hipMemcpyAsync(Ey, fm->dot, (size_t)fm->rows * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipMemcpyAsync(Vf, fm->vf, (size_t)fm->rows * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    return RR_OK;
}

}  // extern "C"
