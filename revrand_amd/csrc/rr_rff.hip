// Random Fourier feature kernels for gfx950 (MI355X, CDNA4): Phi, dPhi/dl and the fused
// Phi -> Phi^T Phi / Phi^T y accumulation.  HIP source, wave64, f32-input MFMA.
//
// Phase convention: the host uploads Ws[i][f] = W[i][f] / (l_i * 2 pi), so the projection
// t = sum_i x_i Ws[i][f] is the phase in REVOLUTIONS; after the exact reduction
// t - rint(t) in [-0.5, 0.5] the hardware v_sin_f32 / v_cos_f32 (which take revolutions)
// give sin/cos directly.
#include "rr_syrk_args.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>



// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ void sincos_rev(float t, float &s, float &c) {
    const float f = t - __builtin_rintf(t);
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}

__device__ __forceinline__ void sincos_rev(double t, double &s, double &c) { rr_sincos_rev_f64(t, s, c); }
typedef double doublex4 __attribute__((ext_vector_type(4)));

// z = sum_i x[i] * w[i] with x wave-uniform (scalar loads) and w in registers.
// GUARD == false: the row has at least DMAX readable, finite elements (the caller padded X with
// zero columns up to DMAX, see rr_rff_padded_dim) and w[i] == 0 for i >= d, so the loop is
// branch-free and the scalar loads merge.  GUARD == true: read exactly d elements.
template <int DMAX, bool GUARD, typename TX, typename TC>
__device__ __forceinline__ TC project_row(const TX *__restrict__ xr, int d, const TC (&w)[DMAX]) {
    TC z = 0;
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
        if (!GUARD || i < d) z = fma((TC)xr[i], w[i], z);
    }
    return z;
}

// Ws has DMAX rows (rows >= d are zero) and npad columns (columns >= n are zero).
template <int DMAX, typename TC>
__device__ __forceinline__ void load_w(TC (&w)[DMAX], const TC *__restrict__ Ws, int npad, int f) {
#pragma unroll
    for (int i = 0; i < DMAX; ++i) w[i] = Ws[(size_t)i * npad + f];
}

// float64 kernels: rows of X staged through LDS, RS rows at a time (coalesced vector loads by the whole block, then
// broadcast ds_reads).  Through the scalar cache -- as the f32 VALU kernels read x -- every new row is a miss the wave
// waits out with nothing else to do (s_waitcnt lgkmcnt(0) admits no prefetch): the f64 feature kernel ran at 29 % of its
// arithmetic, 2.6 TB/s of Phi.
constexpr int RR_XS_ROWS = 16;
template <int DMAX, typename TX, typename TC>
__device__ __forceinline__ void stage_x_rows(TC (*xs)[DMAX], const TX *__restrict__ X, int64_t rb, int64_t rvalid, int64_t ldx) {
    for (int e = threadIdx.x; e < RR_XS_ROWS * DMAX; e += 256) {
        const int rr = e / DMAX, i = e % DMAX;
        xs[rr][i] = (rb + rr < rvalid) ? (TC)X[(rb + rr) * ldx + i] : (TC)0;
    }
}
template <int DMAX, typename TC>
__device__ __forceinline__ TC project_lds(const TC *__restrict__ xr, const TC (&w)[DMAX]) {
    TC z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
    for (int i = 0; i < DMAX; i += 4) {
        z0 = fma(xr[i], w[i], z0);
        z1 = fma(xr[i + 1], w[i + 1], z1);
        z2 = fma(xr[i + 2], w[i + 2], z2);
        z3 = fma(xr[i + 3], w[i + 3], z3);
    }
    return (z0 + z1) + (z2 + z3);
}

// ---------------------------------------------------------------------------------------
// Phi = [cos, sin] / sqrt(n)        (_RandomKernelBasis.transform, basis_functions.py:838-864)
// grid.x = frequency blocks of 256, grid.y = row blocks; one frequency per thread, W column in
// registers, X rows through the scalar cache, stores coalesced along the frequency axis.
// ---------------------------------------------------------------------------------------
template <int DMAX, typename TX, typename TC, typename TO>
__global__ void __launch_bounds__(256)
rr_rff_transform_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const TC *__restrict__ Ws,
                        int n, int npad, TO *__restrict__ Phi, int64_t ldphi, TC scale,
                        int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    TC w[DMAX];
    load_w<DMAX, TC>(w, Ws, npad, fvalid ? f : 0);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    if constexpr (sizeof(TC) == 8 && DMAX >= 8) {
        __shared__ TC xs[RR_XS_ROWS][DMAX];
        for (int64_t rb = r0; rb < r1; rb += RR_XS_ROWS) {
            __syncthreads();
            stage_x_rows<DMAX, TX, TC>(xs, X, rb, r1, ldx);
            __syncthreads();
            const int nr = (int)(r1 - rb < RR_XS_ROWS ? r1 - rb : RR_XS_ROWS);
            for (int rr = 0; rr < nr; ++rr) {
                TC s, c;
                sincos_rev(project_lds<DMAX, TC>(xs[rr], w), s, c);
                if (fvalid) {
                    TO *o = Phi + (rb + rr) * ldphi;
                    o[f] = (TO)(c * scale);
                    o[n + f] = (TO)(s * scale);
                }
            }
        }
        return;
    }
    for (int64_t r = r0; r < r1; ++r) {
        const TC t = project_row<DMAX, false, TX, TC>(X + r * ldx, DMAX, w);
        TC s, c;
        sincos_rev(t, s, c);
        if (fvalid) {
            TO *o = Phi + r * ldphi;
            o[f] = (TO)(c * scale);
            o[n + f] = (TO)(s * scale);
        }
    }
}

// ---------------------------------------------------------------------------------------
// dPhi/dl_i = [ -sin(z) dz_i , cos(z) dz_i ] / sqrt(n),  dz_i = -x_i W[i][f] / l_i^2
//           (_RandomKernelBasis.grad, basis_functions.py:866-901)
// nout == 1: (N, 2n), dimension 0 only (the reference's isotropic quirk);
// nout == d: (N, 2n, d) C-order.
// ---------------------------------------------------------------------------------------
template <int DMAX, typename TX, typename TC, typename TO>
__global__ void __launch_bounds__(256)
rr_rff_grad_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const TC *__restrict__ Ws,
                   const TC *__restrict__ gfac, int n, int npad, int nout, TO *__restrict__ out,
                   TC scale, int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    TC w[DMAX];
    load_w<DMAX, TC>(w, Ws, npad, fvalid ? f : 0);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const TX *xr = X + r * ldx;
        const TC t = project_row<DMAX, false, TX, TC>(xr, DMAX, w);
        TC s, c;
        sincos_rev(t, s, c);
        if (fvalid) {
            TO *oc = out + ((size_t)r * 2 * n + f) * nout;
            TO *os = out + ((size_t)r * 2 * n + n + f) * nout;
#pragma unroll
            for (int i = 0; i < DMAX; ++i) {
                if (i < nout) {
                    const TC dz = -(TC)xr[i] * w[i] * gfac[i] * scale;
                    oc[i] = (TO)(-s * dz);
                    os[i] = (TO)(c * dz);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Spectral-mixture component (FastFoodGM, basis_functions.py:1386-1562) on the dense equivalent of its
// FastFood chain (W = _makeVX(I_d), see rr_fastfood.hip):
//   Phi = [cos(z + mX), sin(z + mX), cos(z - mX), sin(z - mX)] / sqrt(2 n),  z = (x / l) . W[:, f],  mX = x . mean
// mu[i] = mean_i / (2 pi) so that both phases are in revolutions.
// ---------------------------------------------------------------------------------------
template <int DMAX, typename TX, typename TC, typename TO>
__global__ void __launch_bounds__(256)
rr_gm_transform_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const TC *__restrict__ Ws,
                       const TC *__restrict__ mu, int n, int npad, TO *__restrict__ Phi, int64_t ldphi, TC scale,
                       int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    TC w[DMAX];
    load_w<DMAX, TC>(w, Ws, npad, fvalid ? f : 0);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const TX *xr = X + r * ldx;
        const TC z = project_row<DMAX, false, TX, TC>(xr, DMAX, w);
        TC mx = 0;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) mx = fma((TC)xr[i], mu[i], mx);
        TC sp, cp, sm, cm;
        sincos_rev(z + mx, sp, cp);
        sincos_rev(z - mx, sm, cm);
        if (fvalid) {
            TO *o = Phi + r * ldphi;
            o[f] = (TO)(cp * scale);
            o[n + f] = (TO)(sp * scale);
            o[2 * n + f] = (TO)(cm * scale);
            o[3 * n + f] = (TO)(sm * scale);
        }
    }
}

// dPhi/dmean_i and dPhi/dl_i (basis_functions.py:1477-1537), each (N, 4n, d) C-order:
//   dmean_i = x_i [-sin(z+m), cos(z+m),  sin(z-m), -cos(z-m)] / sqrt(2n)
//   dlen_i  = dz_i [-sin(z+m), cos(z+m), -sin(z-m),  cos(z-m)] / sqrt(2n),  dz_i = -x_i W[i][f] / l_i^2
template <int DMAX, typename TX, typename TC, typename TO>
__global__ void __launch_bounds__(256)
rr_gm_grad_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const TC *__restrict__ Ws,
                  const TC *__restrict__ mu, const TC *__restrict__ gfac, int n, int npad, int d,
                  TO *__restrict__ dmean, TO *__restrict__ dlen, TC scale, int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    TC w[DMAX];
    load_w<DMAX, TC>(w, Ws, npad, fvalid ? f : 0);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const TX *xr = X + r * ldx;
        const TC z = project_row<DMAX, false, TX, TC>(xr, DMAX, w);
        TC mx = 0;
#pragma unroll
        for (int i = 0; i < DMAX; ++i) mx = fma((TC)xr[i], mu[i], mx);
        TC sp, cp, sm, cm;
        sincos_rev(z + mx, sp, cp);
        sincos_rev(z - mx, sm, cm);
        if (fvalid) {
            const size_t row = (size_t)r * 4 * n;
#pragma unroll
            for (int i = 0; i < DMAX; ++i) {
                if (i < d) {
                    const TC xi = (TC)xr[i] * scale;
                    const TC dz = -(TC)xr[i] * w[i] * gfac[i] * scale;
                    TO *om = dmean + (row + f) * d + i;
                    TO *ol = dlen + (row + f) * d + i;
                    const size_t blk = (size_t)n * d;
                    om[0] = (TO)(-sp * xi);
                    om[blk] = (TO)(cp * xi);
                    om[2 * blk] = (TO)(sm * xi);
                    om[3 * blk] = (TO)(-cm * xi);
                    ol[0] = (TO)(-sp * dz);
                    ol[blk] = (TO)(cp * dz);
                    ol[2 * blk] = (TO)(-sm * dz);
                    ol[3 * blk] = (TO)(cm * dz);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Phi^T Phi / Phi^T y in two kernels per row chunk.
//
//  (A) rr_rff_features_kernel:  P[r][f] = cos(2 pi z)/sqrt(n), P[r][n+f] = sin(2 pi z)/sqrt(n),
//      z = x_r . Ws[:, f]; f32, (rows rounded up to 32, Fp = 2n rounded up to 256) row-major HBM
//      scratch with zero pad rows/columns.  The same pass accumulates b = Phi^T y.
//  (B) rr_syrk_f32_kernel: G(upper) += P^T P for any such zero-padded f32 feature matrix.  One
//      workgroup (8 waves) owns a 256x256 block of G (column blocks ta <= tb) for one K-split of
//      rows.  Per k-block of 32 rows the [32][256 | 256] tile arrives by LDS-DMA
//      (global_load_lds_dwordx4, one 1 KiB row-segment per wave-instruction, no VGPRs, no
//      VALU), double-buffered with one barrier per k-block; waves accumulate 128x64 sub-blocks
//      with v_mfma_f32_32x32x2_f32, operands straight from LDS by conflict-free ds_read_b32
//      (lane -> column, lane>>5 -> row of the 2-row k-step == the 32x32x2 A/B operand layout),
//      prefetched one k-step ahead.  f32 accumulation inside a K-split, f64 atomics across
//      K-splits into the upper triangle of G.
//
// Why two kernels: on gfx950 the f32-input MFMA runs at the f32 VALU rate and (measured:
// SQ_VALU_MFMA_COEXEC_CYCLES = 0, produce/consume times purely additive) does not overlap with
// VALU work of either wave on the SIMD, so every VALU instruction in the Gram kernel is lost
// MFMA time.  The projection and the transcendentals therefore run once per row in (A) instead
// of once per (row, tile) inside (B), and (B)'s instruction stream is DMA + ds_read + MFMA only.
// ---------------------------------------------------------------------------------------

template <int DMAX, bool HAS_Y, typename TX, typename TC>
__global__ void __launch_bounds__(256)
rr_rff_features_kernel(const TX *__restrict__ X, const TX *__restrict__ y, int64_t N, int64_t Npad,
                       int64_t ldx, const TC *__restrict__ Ws, int n, int npad, TC *__restrict__ P,
                       int64_t ldp, double *__restrict__ bvec, TC scale, int rows_per_block, int64_t bdet = 0) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    TC w[DMAX];
    load_w<DMAX, TC>(w, Ws, npad, f < npad ? f : 0);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > Npad) r1 = Npad;
    TC bc = 0, bs = 0;
    if constexpr (sizeof(TC) == 8 && DMAX >= 8) {
        __shared__ TC xs[RR_XS_ROWS][DMAX];
        for (int64_t rb = r0; rb < r1; rb += RR_XS_ROWS) {
            __syncthreads();
            stage_x_rows<DMAX, TX, TC>(xs, X, rb, N, ldx);
            __syncthreads();
            const int nr = (int)(r1 - rb < RR_XS_ROWS ? r1 - rb : RR_XS_ROWS);
            for (int rr = 0; rr < nr; ++rr) {
                const int64_t r = rb + rr;
                TC c = 0, s = 0;
                if (r < N) {  // uniform
                    sincos_rev(project_lds<DMAX, TC>(xs[rr], w), s, c);
                    c *= scale;
                    s *= scale;
                    if (HAS_Y) {
                        const TC yv = (TC)y[r];
                        bc = fma(c, yv, bc);
                        bs = fma(s, yv, bs);
                    }
                }
                if (fvalid) {
                    P[r * ldp + f] = c;
                    P[r * ldp + n + f] = s;
                }
            }
        }
        if (HAS_Y && fvalid) {
            rr_acc_out(bvec, bdet, blockIdx.y, f, (double)bc);
            rr_acc_out(bvec, bdet, blockIdx.y, n + f, (double)bs);
        }
        return;
    }
    for (int64_t r = r0; r < r1; ++r) {
        TC c = 0, s = 0;
        if (r < N) {  // uniform
            const TC t = project_row<DMAX, false, TX, TC>(X + r * ldx, DMAX, w);
            sincos_rev(t, s, c);
            c *= scale;
            s *= scale;
            if (HAS_Y) {
                const TC yv = (TC)y[r];
                bc = fma(c, yv, bc);
                bs = fma(s, yv, bs);
            }
        }
        if (fvalid) {
            P[r * ldp + f] = c;
            P[r * ldp + n + f] = s;
        }
    }
    if (HAS_Y && fvalid) {
        rr_acc_out(bvec, bdet, blockIdx.y, f, (double)bc);
        rr_acc_out(bvec, bdet, blockIdx.y, n + f, (double)bs);
    }
}

// (A') The same feature pass with the projection X Ws on the matrix cores (f32 X, f32 features).
// A wave keeps the Ws operands of CB column blocks of 32 frequencies in registers (B of
// v_mfma_f32_32x32x2_f32: lane (j, h) holds Ws[h KS + t][c0 + j] for k-step t, KS = DMAX / 2) and streams
// 32-row tiles of X (A: lane (i, h) holds X[r0 + i][h KS + t], i.e. half a row per lane, read with
// dwordx4 loads -- any pairing of the k index works as long as A and B agree).  Per tile and column
// block: KS MFMAs give z for 32 x 32 (row, frequency) pairs, lane (j, h) owning frequency c0 + j and
// rows r0 + (e & 3) + 8 (e >> 2) + 4 h; then rint / v_sin / v_cos / scale per value and two stores
// whose half-waves each cover 128 contiguous bytes of one row of P.  Per (row, frequency) this costs
// 1/64 of a 64-cycle MFMA instead of DMAX VALU FMAs.
// WT (plain f32 output): the same tile is ALSO written feature-major into Pt (rows = features, ldt floats per row):
// lane (j, h) holds 4 consecutive data rows per e >> 2, i.e. one 16-byte store per group -- the consumer that needs both
// layouts (the GLM step: P as the K-major operand of dfs Phi, P^T as the one of P WS^T) then skips its transposing pass.
template <int DMAX, int CB, bool HAS_Y, bool WT, typename TX, typename TO>
__global__ void __launch_bounds__(256, 2)
rr_rff_features_mfma_kernel(const TX *__restrict__ X, const TX *__restrict__ y, int64_t N, int64_t Npad,
                            int64_t ldx, const float *__restrict__ Ws, int n, int npad, TO *__restrict__ P,
                            int64_t ldp, double *__restrict__ bvec, float scale, int tiles_per_block,
                            float *__restrict__ Pt, int64_t ldt, int64_t bdet = 0) {
    constexpr int KS = DMAX / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // row tiles, hence store bases, are wave-uniform
    const int j = lane & 31, h = lane >> 5;
    const int c0 = blockIdx.x * (32 * CB);
    // Ws rows are npad wide, X rows at least DMAX; P holds whole 32-row tiles of ldp columns and every store offset
    // (a 32-bit byte offset from the tile base) must fit
    RR_DEV_ASSERT(c0 + 32 * CB <= npad && DMAX <= ldx && Npad >= N);
    RR_DEV_ASSERT((std::is_same<TO, rr_pb_t>::value || std::is_same<TO, rr_pf_t>::value) ||
                  ((int64_t)2 * n <= ldp && (uint64_t)sizeof(TO) * 36u * (uint64_t)ldp < (1ull << 32)));
    float bw[CB][KS];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int t = 0; t < KS; ++t) bw[cb][t] = Ws[(size_t)(h * KS + t) * npad + c0 + 32 * cb + j];
    float bc[CB], bs[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bc[cb] = bs[cb] = 0.f;
    const int64_t ntiles = (Npad + 31) / 32;
    const int64_t tile0 = (int64_t)blockIdx.y * tiles_per_block;
    int64_t tile1 = tile0 + tiles_per_block;
    if (tile1 > ntiles) tile1 = ntiles;
    for (int64_t tl = tile0 + wave; tl < tile1; tl += 4) {
        const int64_t r0 = tl * 32;
        float a[KS];
        {
            const int64_t ra = r0 + j;
            const TX *src = X + (ra < N ? ra : 0) * ldx + h * KS;  // KS contiguous elements: dwordx4 loads
#pragma unroll
            for (int t = 0; t < KS; ++t) a[t] = (float)src[t];
            if (ra >= N) {
#pragma unroll
                for (int t = 0; t < KS; ++t) a[t] = 0.f;
            }
        }
        // this lane's rows are r0 + 4 h + rr, rr = (e & 3) + 8 (e >> 2); data rows while rr < lim.  The scratch
        // has whole 32-row tiles (caller contract), so pad rows are stored (as zeros) without a guard.
        const int64_t d0 = N - r0 - 4 * h;
        const int lim = (int)(d0 > 32 ? 32 : (d0 < 0 ? 0 : d0));
        float yv[16];
        if (HAS_Y) {  // one coalesced load of the tile's 32 targets, then the lane's 16 rows by cross-lane reads
            const int64_t ry = r0 + j;
            float yt = (float)y[ry < N ? ry : N - 1];
            yt = ry < N ? yt : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2);
                yv[e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(16 * h + 4 * rr, __builtin_bit_cast(int, yt)));
            }
        }
        // stores: wave-uniform tile bases (cos and sin halves) in SGPRs + one 32-bit byte offset per row of the
        // lane (written as asm: left alone, the compiler keeps a 64-bit pointer induction variable per store)
        constexpr bool F16 = std::is_same<TO, rr_pf_t>::value;
        constexpr bool SPLIT = std::is_same<TO, rr_pb_t>::value || F16;
        const float s16 = F16 ? f16_store_scale(scale) : 1.f;
        // SPLIT: the tile is k-steps 2 tl and 2 tl + 1 of Pb; byte addresses, column stride 64
        const char *tile_c = SPLIT ? (const char *)P + ((2 * tl) * ldp + c0) * 64 : (const char *)(P + r0 * ldp + c0);
        const char *tile_s = tile_c + (int64_t)n * (SPLIT ? 64 : (int64_t)sizeof(TO));
        const char *tile_c1 = tile_c + ldp * 64, *tile_s1 = tile_s + ldp * 64;  // SPLIT: second k-step
        constexpr unsigned ES = SPLIT ? 4 : sizeof(TO);
        const unsigned lane_off = SPLIT ? (unsigned)(j * 64 + h * 16) : ES * ((unsigned)(4 * h) * (unsigned)ldp + (unsigned)j);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int t = 0; t < KS; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bw[cb][t], acc, 0, 0, 0);
            if (SPLIT) {
                if (c0 + 32 * cb + j < n) {
                    float cvv[16], svv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rr = (e & 3) + 8 * (e >> 2);
                        float sv, cv;
                        sincos_rev(acc[e], sv, cv);
                        cvv[e] = rr < lim ? cv * scale : 0.f;
                        svv[e] = rr < lim ? sv * scale : 0.f;
                        if (HAS_Y) {
                            bc[cb] = fmaf(cvv[e], yv[e], bc[cb]);
                            bs[cb] = fmaf(svv[e], yv[e], bs[cb]);
                        }
                    }
                    // lane (j, h) holds rows {0-3, 8-11} + 4 h of each 16-row k-step: exactly granule h of each part.
                    // s_nop after each store: a >8-byte store still reads its data registers for two more cycles and
                    // the compiler's hazard recogniser does not look inside asm.
                    const unsigned off = lane_off + 2048u * cb;
                    uintx4 g_hi, g_lo;
                    auto split8 = [&](const float *v) {
                        if (F16) split_f16x8(v, s16, g_hi, g_lo);
                        else split_bf16x8(v, g_hi, g_lo);
                    };
                    split8(cvv);
                    asm volatile("global_store_dwordx4 %0, %1, %2" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_hi), "s"(tile_c) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:32" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_lo), "s"(tile_c) : "memory");
                    split8(cvv + 8);
                    asm volatile("global_store_dwordx4 %0, %1, %2" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_hi), "s"(tile_c1) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:32" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_lo), "s"(tile_c1) : "memory");
                    split8(svv);
                    asm volatile("global_store_dwordx4 %0, %1, %2" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_hi), "s"(tile_s) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:32" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_lo), "s"(tile_s) : "memory");
                    split8(svv + 8);
                    asm volatile("global_store_dwordx4 %0, %1, %2" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_hi), "s"(tile_s1) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:32" RR_NT_ASM "\n\ts_nop 1" ::"v"(off), "v"(g_lo), "s"(tile_s1) : "memory");
                }
            } else if (c0 + 32 * cb + j < n) {  // one divergent region per column block (ragged n only)
                float ct[WT ? 16 : 1], st[WT ? 16 : 1];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2);
                    float sv, cv;
                    sincos_rev(acc[e], sv, cv);
                    cv = rr < lim ? cv * scale : 0.f;
                    sv = rr < lim ? sv * scale : 0.f;
                    if (WT) {
                        ct[e] = cv;
                        st[e] = sv;
                    }
                    const unsigned off = lane_off + ES * (unsigned)rr * (unsigned)ldp;
                    if constexpr (ES == 4) {
                        asm volatile("global_store_dword %0, %1, %2 offset:%3" RR_NT_ASM ::"v"(off), "v"(cv), "s"(tile_c), "i"(128 * cb) : "memory");
                        asm volatile("global_store_dword %0, %1, %2 offset:%3" RR_NT_ASM ::"v"(off), "v"(sv), "s"(tile_s), "i"(128 * cb) : "memory");
                    } else {
                        const double cd = (double)cv, sd = (double)sv;
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" RR_NT_ASM ::"v"(off), "v"(cd), "s"(tile_c), "i"(256 * cb) : "memory");
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" RR_NT_ASM ::"v"(off), "v"(sd), "s"(tile_s), "i"(256 * cb) : "memory");
                    }
                    if (HAS_Y) {
                        bc[cb] = fmaf(cv, yv[e], bc[cb]);
                        bs[cb] = fmaf(sv, yv[e], bs[cb]);
                    }
                }
                if constexpr (WT) {
                    typedef float float4v __attribute__((ext_vector_type(4)));
                    float *tc = Pt + (int64_t)(c0 + 32 * cb + j) * ldt + r0 + 4 * h;  // feature row, this lane's first data row
                    float *ts = tc + (int64_t)n * ldt;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        *reinterpret_cast<float4v *>(tc + 8 * g) = float4v{ct[4 * g], ct[4 * g + 1], ct[4 * g + 2], ct[4 * g + 3]};
                        *reinterpret_cast<float4v *>(ts + 8 * g) = float4v{st[4 * g], st[4 * g + 1], st[4 * g + 2], st[4 * g + 3]};
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one column block at a time: 16 accumulators live, not 64
        }
    }
    if (HAS_Y) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int col = c0 + 32 * cb + j;
            if (col < n) {  // deterministic mode: one slot per (row block, wave, half-wave)
                rr_acc_out(bvec, bdet, ((int64_t)blockIdx.y * 4 + wave) * 2 + h, col, (double)bc[cb]);
                rr_acc_out(bvec, bdet, ((int64_t)blockIdx.y * 4 + wave) * 2 + h, n + col, (double)bs[cb]);
            }
        }
    }
}

// launch (A') when its preconditions hold (16-byte aligned X rows); false = use the VALU kernel.  The output
// must have whole 32-row tiles: (mpad + 31) / 32 * 32 rows.
template <typename TX>
static bool rr_features_mfma_ok(rr_basis *b, const TX *X, int64_t m, int64_t ldx) {
    static const bool disabled = getenv("RR_FEATURES_NO_MFMA") != nullptr;
    return !(disabled || b->large || ((ldx * sizeof(TX)) & 15) != 0 || ((uintptr_t)X & 15) != 0 || b->dpad < 8 || m < 1);
}

template <typename TX, typename TO>
static bool rr_features_mfma_launch(rr_basis *b, const TX *X, const TX *y, int64_t m, int64_t mpad, int64_t ldx,
                                    TO *P, int64_t ldp, double *db, float scale, float *Pt = nullptr, int64_t ldt = 0) {
    if (!rr_features_mfma_ok<TX>(b, X, m, ldx)) return false;
    // feature-major second output: plain f32 tiles without targets, 16-byte aligned rows of Pt
    constexpr bool CAN_WT = std::is_same<TO, float>::value;
    if (Pt != nullptr && (!CAN_WT || y != nullptr || (ldt & 3) != 0 || ((uintptr_t)Pt & 15) != 0)) return false;
    rr_ctx *c = b->ctx;
    const int64_t ntiles = (mpad + 31) / 32;
#define RR_FM(DM, CBK)                                                                                              \
    do {                                                                                                            \
        const int cgroups = (b->n + 32 * CBK - 1) / (32 * CBK);                                                     \
        int64_t tpb = 64;                                                                                           \
        /* a wave loads its CB x KS weight operands once and then walks its tiles: down to 16 tiles per workgroup (4 per */ \
        /* wave) for four workgroups per CU, below that only to give every CU one -- at 4 (one tile per wave) the operand */ \
        /* loads cost 10 % of a statistics pass of 44 484 or 200 000 rows at F = 1024 (RR_FEAT_TPB: A/B runs)            */ \
        while (tpb > 16 && cgroups * ((ntiles + tpb - 1) / tpb) < 4 * (int64_t)c->num_cu) tpb >>= 1;                \
        while (tpb > 4 && cgroups * ((ntiles + tpb - 1) / tpb) < (int64_t)c->num_cu) tpb >>= 1;                     \
        if (getenv("RR_FEAT_TPB")) tpb = atoi(getenv("RR_FEAT_TPB"));                                               \
        if ((ntiles + tpb - 1) / tpb > 65535) tpb = (ntiles + 65534) / 65535;                                       \
        const dim3 grid(cgroups, (unsigned)((ntiles + tpb - 1) / tpb));                                             \
        if (y && c->deterministic) {                                                                                \
            void *part = nullptr;                                                                                   \
            const int64_t nslots = (int64_t)grid.y * 8, F2 = 2 * (int64_t)b->n;                                     \
            if (rr_det_scratch(c, (size_t)nslots * F2 * 8, &part) != RR_OK) return false;                           \
            hipLaunchKernelGGL((rr_rff_features_mfma_kernel<DM, CBK, true, false, TX, TO>), grid, dim3(256), 0, c->stream, \
                               X, y, m, mpad, ldx, b->dWs32, b->n, b->npad, P, ldp, (double *)part, scale, (int)tpb, nullptr, 0, F2); \
            if (rr_det_reduce(c, (const double *)part, nslots, F2, F2, db) != RR_OK) return false;                  \
        } else if (y) hipLaunchKernelGGL((rr_rff_features_mfma_kernel<DM, CBK, true, false, TX, TO>), grid, dim3(256), 0, c->stream, \
                                  X, y, m, mpad, ldx, b->dWs32, b->n, b->npad, P, ldp, db, scale, (int)tpb, nullptr, 0); \
        else if (CAN_WT && Pt)                                                                                      \
            hipLaunchKernelGGL((rr_rff_features_mfma_kernel<DM, CBK, false, CAN_WT, TX, TO>), grid, dim3(256), 0, c->stream, \
                               X, y, m, mpad, ldx, b->dWs32, b->n, b->npad, P, ldp, db, scale, (int)tpb, Pt, ldt);  \
        else hipLaunchKernelGGL((rr_rff_features_mfma_kernel<DM, CBK, false, false, TX, TO>), grid, dim3(256), 0, c->stream, \
                                X, y, m, mpad, ldx, b->dWs32, b->n, b->npad, P, ldp, db, scale, (int)tpb, nullptr, 0); \
    } while (0)
    switch (b->dpad) {
        case 8: RR_FM(8, 4); break;
        case 16: RR_FM(16, 4); break;
        case 32: RR_FM(32, 4); break;
        case 64: RR_FM(64, 2); break;
        case 128: RR_FM(128, 1); break;
        default: return false;
    }
#undef RR_FM
    return true;
}

// (A'') The float64 feature pass with the projection on the f64 matrix cores (round 2).  The VALU kernel above reads x
// through LDS broadcast reads -- 16 ds_read_b128 per row and wave, which is what bounds it (the LDS pipe is shared by the
// four SIMDs: 3.2 TB/s of float64 Phi, 36 % of the f64 VALU rate).  Here a wave keeps the Ws operands of CB column blocks
// of 16 frequencies in registers (B of v_mfma_f64_16x16x4_f64: lane (j, g) holds Ws[g KS + t][c0 + j] for k-step t,
// KS = DMAX / 4) and streams 16-row tiles of x (A: lane (i, g) holds x[r0 + i][g KS + t]: a quarter row per lane, vector
// loads; any pairing of the k index works as long as A and B agree).  Per tile and column block KS MFMAs give the
// phases of 16 x 16 (row, frequency) pairs, lane (j, g) owning frequency c0 + j and rows r0 + g + 4 e; the VALU is left
// with the float64 sin / cos kernels only, and every store instruction writes 4 rows x 128 contiguous bytes.
template <int DMAX, int CB, bool HAS_Y, typename TX, typename TO>
__global__ void __launch_bounds__(256)
rr_rff_features_mfma64_kernel(const TX *__restrict__ X, const TX *__restrict__ y, int64_t N, int64_t Npad, int64_t ldx,
                              const double *__restrict__ Ws, int n, int npad, TO *__restrict__ P, int64_t ldp,
                              double *__restrict__ bvec, double scale, int tiles_per_block, int64_t bdet = 0) {
    constexpr int KS = DMAX / 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * (16 * CB);
    RR_DEV_ASSERT(DMAX <= ldx && Npad >= N && (int64_t)2 * n <= ldp && (uint64_t)sizeof(TO) * 20u * (uint64_t)ldp < (1ull << 32));
    double bw[CB][KS];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int col = c0 + 16 * cb + j;
#pragma unroll
        for (int t = 0; t < KS; ++t) bw[cb][t] = col < npad ? Ws[(size_t)(g * KS + t) * npad + col] : 0.0;
    }
    double bc[CB], bs[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bc[cb] = bs[cb] = 0.0;
    const int64_t ntiles = (Npad + 15) / 16;
    const int64_t tile0 = (int64_t)blockIdx.y * tiles_per_block;
    int64_t tile1 = tile0 + tiles_per_block;
    if (tile1 > ntiles) tile1 = ntiles;
    for (int64_t tl = tile0 + wave; tl < tile1; tl += 4) {
        const int64_t r0 = tl * 16;
        double a[KS];
        {
            const int64_t ra = r0 + j;
            const TX *src = X + (ra < N ? ra : 0) * ldx + g * KS;  // KS contiguous elements
#pragma unroll
            for (int t = 0; t < KS; ++t) a[t] = (double)src[t];
            if (ra >= N) {
#pragma unroll
                for (int t = 0; t < KS; ++t) a[t] = 0.0;
            }
        }
        // this lane's rows are r0 + g + 4 e; data rows while g + 4 e < lim.  The scratch has whole 16-row tiles (caller
        // contract), so pad rows are stored (as zeros) without a guard.
        const int64_t d0 = N - r0;
        const int lim = (int)(d0 > 16 ? 16 : (d0 < 0 ? 0 : d0));
        double yv[4];
        if (HAS_Y) {  // one coalesced load of the tile's 16 targets, then the lane's 4 rows by cross-lane reads
            const int64_t ry = r0 + j;
            double yt = (double)y[ry < N ? ry : N - 1];
            yt = ry < N ? yt : 0.0;
            const int lo = __double2loint(yt), hi = __double2hiint(yt);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int srcl = 4 * (g + 4 * e);  // byte address of lane g + 4 e (a lane of the first DPP row: holds y[r0 + g + 4 e])
                yv[e] = __hiloint2double(__builtin_amdgcn_ds_bpermute(srcl, hi), __builtin_amdgcn_ds_bpermute(srcl, lo));
            }
        }
        const char *tile_c = (const char *)(P + r0 * ldp + c0);
        const char *tile_s = tile_c + (int64_t)n * (int64_t)sizeof(TO);
        constexpr unsigned ES = sizeof(TO);
        const unsigned lane_off = ES * ((unsigned)g * (unsigned)ldp + (unsigned)j);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            doublex4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < KS; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bw[cb][t], acc, 0, 0, 0);
            if (c0 + 16 * cb + j < n) {  // one divergent region per column block (ragged n only)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rr = g + 4 * e;
                    const unsigned off = lane_off + ES * 4u * (unsigned)e * (unsigned)ldp;
                    if constexpr (ES == 8) {  // non-temporal: +7 % for these 8-byte stores (3.52 -> 3.28 ms per 500k x 4096)
                        double sv, cv;
                        rr_sincos_rev_f64(acc[e], sv, cv);
                        cv = rr < lim ? cv * scale : 0.0;
                        sv = rr < lim ? sv * scale : 0.0;
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 nt" ::"v"(off), "v"(cv), "s"(tile_c), "i"(128 * cb) : "memory");
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 nt" ::"v"(off), "v"(sv), "s"(tile_s), "i"(128 * cb) : "memory");
                        if (HAS_Y) {
                            bc[cb] = fma(cv, yv[e], bc[cb]);
                            bs[cb] = fma(sv, yv[e], bs[cb]);
                        }
                    } else {
                        // RR_F32P64: the float64 phase loses its whole revolutions in float64 (exact), what is left in
                        // [-0.5, 0.5] goes through the float32 hardware sin / cos like every f32 feature (2^-24 revolutions)
                        const float fr = (float)(acc[e] - rint(acc[e]));
                        const float fs = (float)scale;
                        float cf = __builtin_amdgcn_cosf(fr), sf = __builtin_amdgcn_sinf(fr);
                        cf = rr < lim ? cf * fs : 0.f;
                        sf = rr < lim ? sf * fs : 0.f;
                        asm volatile("global_store_dword %0, %1, %2 offset:%3" RR_NT_ASM ::"v"(off), "v"(cf), "s"(tile_c), "i"(64 * cb) : "memory");
                        asm volatile("global_store_dword %0, %1, %2 offset:%3" RR_NT_ASM ::"v"(off), "v"(sf), "s"(tile_s), "i"(64 * cb) : "memory");
                        if (HAS_Y) {
                            bc[cb] = fma((double)cf, yv[e], bc[cb]);
                            bs[cb] = fma((double)sf, yv[e], bs[cb]);
                        }
                    }
                }
            }
        }
    }
    if (HAS_Y) {
        // the four lanes (j, g = 0..3) hold partial sums of the same frequency
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) {
                bc[cb] += __hiloint2double(__shfl_xor(__double2hiint(bc[cb]), m, 64), __shfl_xor(__double2loint(bc[cb]), m, 64));
                bs[cb] += __hiloint2double(__shfl_xor(__double2hiint(bs[cb]), m, 64), __shfl_xor(__double2loint(bs[cb]), m, 64));
            }
            const int col = c0 + 16 * cb + j;
            if (g == 0 && col < n) {  // deterministic mode: one slot per (row block, wave)
                rr_acc_out(bvec, bdet, (int64_t)blockIdx.y * 4 + wave, col, bc[cb]);
                rr_acc_out(bvec, bdet, (int64_t)blockIdx.y * 4 + wave, n + col, bs[cb]);
            }
        }
    }
}

// launch (A'') when its preconditions hold (x rows aligned for the lanes' vector loads); false = use the VALU kernel.
// The output must have whole 16-row tiles.
template <typename TX, typename TO>
static bool rr_features_mfma64_launch(rr_basis *b, const TX *X, const TX *y, int64_t m, int64_t mpad, int64_t ldx, TO *P,
                                      int64_t ldp, double *db, double scale) {
    static const bool disabled = getenv("RR_FEATURES_NO_MFMA64") != nullptr;
    if (disabled || b->large || ((ldx * sizeof(TX)) & 15) != 0 || ((uintptr_t)X & 15) != 0 || b->dpad < 8 || m < 1) return false;
    if ((uint64_t)sizeof(TO) * 20u * (uint64_t)ldp >= (1ull << 32)) return false;
    rr_ctx *c = b->ctx;
    const int64_t ntiles = (mpad + 15) / 16;
#define RR_FM64(DM, CBK)                                                                                            \
    do {                                                                                                            \
        const int cgroups = (b->n + 16 * CBK - 1) / (16 * CBK);                                                     \
        int64_t tpb = 128;                                                                                          \
        while (tpb > 4 && cgroups * ((ntiles + tpb - 1) / tpb) < 8 * (int64_t)c->num_cu) tpb >>= 1;                 \
        if ((ntiles + tpb - 1) / tpb > 65535) tpb = (ntiles + 65534) / 65535;                                       \
        const dim3 grid(cgroups, (unsigned)((ntiles + tpb - 1) / tpb));                                             \
        if (y && c->deterministic) {                                                                                \
            void *part = nullptr;                                                                                   \
            const int64_t nslots = (int64_t)grid.y * 4, F2 = 2 * (int64_t)b->n;                                     \
            if (rr_det_scratch(c, (size_t)nslots * F2 * 8, &part) != RR_OK) return false;                           \
            hipLaunchKernelGGL((rr_rff_features_mfma64_kernel<DM, CBK, true, TX, TO>), grid, dim3(256), 0, c->stream, \
                               X, y, m, mpad, ldx, b->dWs64, b->n, b->npad, P, ldp, (double *)part, scale, (int)tpb, F2); \
            if (rr_det_reduce(c, (const double *)part, nslots, F2, F2, db) != RR_OK) return false;                  \
        } else if (y) hipLaunchKernelGGL((rr_rff_features_mfma64_kernel<DM, CBK, true, TX, TO>), grid, dim3(256), 0, c->stream, \
                                  X, y, m, mpad, ldx, b->dWs64, b->n, b->npad, P, ldp, db, scale, (int)tpb);        \
        else hipLaunchKernelGGL((rr_rff_features_mfma64_kernel<DM, CBK, false, TX, TO>), grid, dim3(256), 0, c->stream, \
                                X, y, m, mpad, ldx, b->dWs64, b->n, b->npad, P, ldp, db, scale, (int)tpb);          \
    } while (0)
    switch (b->dpad) {
        case 8: RR_FM64(8, 4); break;
        case 16: RR_FM64(16, 4); break;
        case 32: RR_FM64(32, 4); break;
        case 64: RR_FM64(64, 2); break;
        case 128: RR_FM64(128, 1); break;
        default: return false;
    }
#undef RR_FM64
    return true;
}

// zero the pad columns [F, Fp) of a feature matrix (the feature kernels only write [0, F))
template <typename TC>
__global__ void __launch_bounds__(256) rr_zero_padcols_kernel(TC *P, int64_t rows, int64_t ldp, int F) {
    const int w = (int)ldp - F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > 0 && i < rows * w) P[(i / w) * ldp + F + (i % w)] = TC(0);
}

// the same for the K-blocked split-bf16 layout: columns [F, ldp) of every k-step (64 B each)
__global__ void __launch_bounds__(256) rr_zero_padcols_pb_kernel(uintx4 *Pb, int64_t ksteps, int64_t ldp, int F) {
    const int64_t w = (ldp - F) * 4;  // granules per k-step
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > 0 && i < ksteps * w) Pb[(i / w) * ldp * 4 + (int64_t)F * 4 + (i % w)] = uintx4{0u, 0u, 0u, 0u};
}



// One k-block tile: 32 rows x (256 + 256) floats = 64 row-segments of 1 KiB; wave w moves rows
// 4w..4w+3 (both sides) with 8 LDS-DMA instructions, lane l carrying bytes [16 l, 16 l + 16).
__device__ __forceinline__ void syrk_dma_tile(const SyrkArgs &p, float *buf, int64_t kb0, int wave, int lane,
                                              int ca, int cb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lr = 4 * wave + k;
        RR_DEV_ASSERT(kb0 + lr < p.rows && ca + GR_TC <= p.ldp && cb + GR_TC <= p.ldp);  // whole padded tiles only
        const float *src = p.P + (kb0 + lr) * p.ldp + 4 * lane;
        float *dst = buf + lr * GR_LD;  // wave-uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(src + ca), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + cb), (lptr_t)(dst + GR_TC), 16, 0, 0);
    }
}

// MODE 0: flat LDS-DMA right after the barrier (rounds 1-2).  1: the same, staggered over the first k-step pairs.
// 2: buffer-descriptor LDS-DMA (rr_dma_kblock) right after the barrier.  3: buffer-descriptor LDS-DMA, staggered (the default).
template <int MODE>
__device__ __forceinline__ void rr_syrk_f32_body(const SyrkArgs &p, float *lds, const unsigned bid) {  // lds: two [32][512] tiles (128 KiB)
    constexpr bool STAG = (MODE & 1) != 0, BUF = (MODE & 2) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tile (ta <= tb) and K-split of this workgroup.  Workgroups are dispatched round-robin over
    // the 8 XCDs (block b -> XCD b % 8, observed); tile_map orders the tiles so that the ones an
    // XCD receives share column blocks of P and its 4 MiB L2 fetches each of them once.
    int tdx = bid % p.ntiles;
    const int ks = bid / p.ntiles;
    if (p.tile_map) tdx = p.tile_map[tdx];
    int ta = 0;
    const int od = p.offdiag_only;  // row ta then holds nb - ta - od tiles
    while (tdx >= p.nb - ta - od) {
        tdx -= p.nb - ta - od;
        ++ta;
    }
    const int tb = ta + tdx + od;
    const bool diag = (ta == tb);
    const int ca = ta * GR_TC, cb = tb * GR_TC;

    const int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;

    // consumer role: wave (wr, wc) -> rows [wr*128, +128) of side A, cols [wc*64, +64) of side B
    const int wr = wave >> 2, wc_ = wave & 3;
    const unsigned aoff = 4u * ((lane >> 5) * GR_LD + wr * 128 + (lane & 31));          // bytes
    const unsigned boff = 4u * ((lane >> 5) * GR_LD + GR_TC + wc_ * 64 + (lane & 31));  // bytes
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int64_t nkb = (row_end - row_begin) / GR_KB;  // rows and splits are multiples of 32
    RR_DEV_ASSERT(p.rows % GR_KB == 0 && p.rows_per_split % GR_KB == 0 && tb < p.nb && p.nb * GR_TC <= p.ldp);
    if (BUF && nkb > 0) {
        const unsigned voff = 16u * lane;
        const float *Pa = p.P + row_begin * p.ldp + ca, *Pb = p.P + row_begin * p.ldp + cb;  // (wave-uniform)
        rr_dma_kblock(Pa, p.ldp, Pb, p.ldp, lds, wave, voff);
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            float *nxt = lds + (cbuf ^ 1) * (GR_KB * GR_LD);
            const int64_t ro = (kb + 1) * GR_KB * p.ldp;
            if (STAG) {
                gram_consume_staggered(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff, rr_dma_slot(wave, p.spread), [&]() {
                    if (kb + 1 < nkb) rr_dma_kblock(Pa + ro, p.ldp, Pb + ro, p.ldp, nxt, wave, voff);
                });
            } else {
                if (kb + 1 < nkb) rr_dma_kblock(Pa + ro, p.ldp, Pb + ro, p.ldp, nxt, wave, voff);
                gram_consume(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff);
            }
            __syncthreads();
        }
    } else if (nkb > 0) {
        syrk_dma_tile(p, lds, row_begin, wave, lane, ca, cb);
        __syncthreads();  // drains the DMA (vmcnt(0)) and publishes tile 0
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            float *nxt = lds + (cbuf ^ 1) * (GR_KB * GR_LD);
            // tile kb+1 flies while tile kb is consumed (its buffer was last read before the
            // barrier that ended iteration kb-1)
            if (STAG) {  // the waves' DMA bursts staggered over the first k-step pairs
                const int slot = (wave < 4) ? wave : ((wave + 2) & 3);
                gram_consume_staggered(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff, slot, [&]() {
                    if (kb + 1 < nkb) syrk_dma_tile(p, nxt, row_begin + (kb + 1) * GR_KB, wave, lane, ca, cb);
                });
                __syncthreads();
                continue;
            }
            if (kb + 1 < nkb && !(p.ablate & 1)) syrk_dma_tile(p, nxt, row_begin + (kb + 1) * GR_KB, wave, lane, ca, cb);
            gram_consume(lds0 + cbuf * (4u * GR_KB * GR_LD), acc, aoff, boff);
            if (!(p.ablate & 2)) __syncthreads();
        }
    }

    // ---- flush: f32 partial -> f64 G (upper triangle only) ----
    if (p.ablate & 4) return;  // RR_GRAM_ABLATE bit 2: measure the k-loop alone
    const int64_t F = p.F;
    const int hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t gc = cb + wc_ * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (gr <= gc) rr_syrk_out(p, ks, gr, gc, acc[i][j][e]);
            }
        }
    }
    (void)diag;
    (void)F;
}

#define RR_SYRK_KERNEL(NAME, MODE)                                                       \
    __global__ void __launch_bounds__(GR_THREADS, 2) NAME(const SyrkArgs p) {             \
        __shared__ float lds[2 * GR_KB * GR_LD];                                          \
        rr_syrk_f32_body<MODE>(p, lds, blockIdx.x);                                       \
    }
RR_SYRK_KERNEL(rr_syrk_f32_kernel, 3)  // the default
RR_SYRK_KERNEL(rr_syrk_f32_flat_kernel, 0)
RR_SYRK_KERNEL(rr_syrk_f32_flatstag_kernel, 1)
RR_SYRK_KERNEL(rr_syrk_f32_buf_kernel, 2)
#undef RR_SYRK_KERNEL

// ---------------------------------------------------------------------------------------
// Ragged last column block (round 2).  When F is not a multiple of 256 the last column block holds only
// w = F - 256 (nb - 1) valid columns, yet every tile (ta, last) costs a full 256x256 tile in the kernel above
// (config 3: F = 8257, w = 65: 32 of 561 tiles, 4 % of the pass, three quarters of their MFMAs on zeros).  This kernel
// computes those tiles TRANSPOSED: the ragged block is loaded as the A side, whose 256 columns are split over the waves
// in 32-column blocks i (wave (wr, wc) owns blocks 4 wr + i), so a wave simply drops the blocks beyond w: NI = the
// number of its blocks that hold valid columns (0..4; waves with NI = 0 only help with the DMA and the barriers).  A
// SIMD then issues 2 NI_total MFMAs per k-step instead of 16 (config 3: 6), and the flush writes G[c_t][c_r] with the
// roles of row and column exchanged.  Launched when w <= 192; the main kernel then enumerates the tiles among the
// first nb - 1 blocks only.
// ---------------------------------------------------------------------------------------
template <int NI>
struct KOpsR {
    float2v a[NI], b[2];
    template <int P>
    __device__ __forceinline__ void load(const unsigned (&abase)[NI], const unsigned (&bbase)[2]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = lds_read2st64<32 * P, 32 * P + 16>(abase[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = lds_read2st64<32 * P, 32 * P + 16>(bbase[j]);
    }
};

// MFMAs [FIRST, LAST) of the 4 NI of a k-step pair, in (s, i, j) order
template <int NI, int FIRST, int LAST>
__device__ __forceinline__ void gram_mfma_r(const KOpsR<NI> &o, floatx16 (&acc)[NI][2]) {
#pragma unroll
    for (int q = FIRST; q < LAST; ++q) {
        constexpr int dummy = 0;
        (void)dummy;
        const int s_ = q / (2 * NI), r_ = q % (2 * NI);
        acc[r_ >> 1][r_ & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[r_ >> 1][s_], o.b[r_ & 1][s_], acc[r_ >> 1][r_ & 1], 0, 0, 0);
    }
}

#define RR_PAIRR(P, CUR, NXT)                                  \
    lds_wait();                                                \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma_r<NI, 0, 1>(CUR, acc);                           \
    __builtin_amdgcn_sched_barrier(0);                         \
    if ((P) + 1 < 8) NXT.template load<((P) + 1) & 7>(abase, bbase); \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma_r<NI, 1, 4 * NI>(CUR, acc);                      \
    __builtin_amdgcn_sched_barrier(0);

template <int NI>
__device__ __forceinline__ void syrk_ragged_loop(const SyrkArgs &p, float *lds, int wave, int lane, int ca, int cb,
                                                 int64_t row_begin, int64_t nkb, int ks) {
    const int wr = wave >> 2, wc_ = wave & 3;
    const unsigned aoff = 4u * ((lane >> 5) * GR_LD + wr * 128 + (lane & 31));
    const unsigned boff = 4u * ((lane >> 5) * GR_LD + GR_TC + wc_ * 64 + (lane & 31));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    constexpr int NA = NI > 0 ? NI : 1;
    floatx16 acc[NA][2];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nkb > 0) {
        syrk_dma_tile(p, lds, row_begin, wave, lane, ca, cb);
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            if (kb + 1 < nkb) syrk_dma_tile(p, lds + (cbuf ^ 1) * (GR_KB * GR_LD), row_begin + (kb + 1) * GR_KB, wave, lane, ca, cb);
            if constexpr (NI > 0) {
                const unsigned cur = lds0 + cbuf * (4u * GR_KB * GR_LD);
                unsigned abase[NI], bbase[2];
#pragma unroll
                for (int i = 0; i < NI; ++i) abase[i] = cur + aoff + i * 128;
#pragma unroll
                for (int j = 0; j < 2; ++j) bbase[j] = cur + boff + j * 128;
                KOpsR<NI> o0, o1;
                o0.template load<0>(abase, bbase);
                RR_PAIRR(0, o0, o1) RR_PAIRR(1, o1, o0) RR_PAIRR(2, o0, o1) RR_PAIRR(3, o1, o0)
                RR_PAIRR(4, o0, o1) RR_PAIRR(5, o1, o0) RR_PAIRR(6, o0, o1) RR_PAIRR(7, o1, o0)
            }
            __syncthreads();
        }
    }
    // flush, transposed: the A side is the ragged block (columns of G), the B side block ta (rows of G)
    if constexpr (NI > 0) {
        const int hi = lane >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t grow = cb + wc_ * 64 + j * 32 + (lane & 31);  // row of G: a column of block ta (all valid)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t gcol = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    rr_syrk_out(p, ks, grow, gcol, acc[i][j][e]);
                }
            }
        }
    }
}
#undef RR_PAIRR

__global__ void __launch_bounds__(GR_THREADS, 2)
rr_syrk_f32_ragged_kernel(const SyrkArgs p) {
    __shared__ float lds[2 * GR_KB * GR_LD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int last = p.nb - 1;
    const int ta = blockIdx.x % last;
    const int ks = blockIdx.x / last;
    const int ca = last * GR_TC;  // A side: the ragged block
    const int cb = ta * GR_TC;    // B side: block ta
    const int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;
    const int64_t nkb = (row_end - row_begin) / GR_KB;
    RR_DEV_ASSERT(p.rows % GR_KB == 0 && p.rows_per_split % GR_KB == 0 && p.nb * GR_TC == p.ldp && p.F > ca);
    const int w = p.F + (p.bcol ? 1 : 0) - ca;     // valid columns of the ragged block (+ the rider column)
    int ni = (w - (wave >> 2) * 128 + 31) / 32;    // this wave's 32-column blocks with valid columns
    ni = ni < 0 ? 0 : (ni > 4 ? 4 : ni);
    switch (ni) {  // wave-uniform; every path runs the same barriers
        case 0: syrk_ragged_loop<0>(p, lds, wave, lane, ca, cb, row_begin, nkb, ks); break;
        case 1: syrk_ragged_loop<1>(p, lds, wave, lane, ca, cb, row_begin, nkb, ks); break;
        case 2: syrk_ragged_loop<2>(p, lds, wave, lane, ca, cb, row_begin, nkb, ks); break;
        case 3: syrk_ragged_loop<3>(p, lds, wave, lane, ca, cb, row_begin, nkb, ks); break;
        default: syrk_ragged_loop<4>(p, lds, wave, lane, ca, cb, row_begin, nkb, ks); break;
    }
}

// ---------------------------------------------------------------------------------------
// Diagonal tiles (ta == tb) of the f32 SYRK.  Of the 64 32x32 blocks of a diagonal 256x256 tile
// only the 36 with block-row <= block-column are needed.  They are dealt to the 8 waves 5 + 4 per
// SIMD pair (waves w and w+4 share a SIMD: 9 MFMAs per k-step between them), so a diagonal
// workgroup issues 4.5 instead of 8 MFMAs per wave and k-step and needs only the A side of the tile
// (32 KiB of DMA per k-block).  Launched as its own kernel after the off-diagonal one.
// ---------------------------------------------------------------------------------------
__constant__ unsigned char RR_DIAG_I[8][5] = {{0, 0, 0, 0, 0}, {1, 1, 1, 1, 1}, {2, 2, 2, 2, 2}, {3, 3, 3, 3, 3},
                                              {0, 0, 0, 7, 0}, {1, 1, 6, 6, 1}, {2, 5, 5, 5, 2}, {4, 4, 4, 4, 4}};
__constant__ unsigned char RR_DIAG_J[8][5] = {{0, 1, 2, 3, 4}, {1, 2, 3, 4, 5}, {2, 3, 4, 5, 6}, {3, 4, 5, 6, 7},
                                              {5, 6, 7, 7, 5}, {6, 7, 6, 7, 6}, {7, 5, 6, 7, 7}, {4, 5, 6, 7, 4}};

template <int NB>
struct KOpsD {
    float2v a[NB], b[NB];
    template <int P>
    __device__ __forceinline__ void load(const unsigned (&abase)[NB], const unsigned (&bbase)[NB]) {
#pragma unroll
        for (int e = 0; e < NB; ++e) a[e] = lds_read2st64<16 * P, 16 * P + 8>(abase[e]);
#pragma unroll
        for (int e = 0; e < NB; ++e) b[e] = lds_read2st64<16 * P, 16 * P + 8>(bbase[e]);
    }
};

template <int NB, int FIRST, int LAST>
__device__ __forceinline__ void gram_mfma_d(const KOpsD<NB> &o, floatx16 (&acc)[NB]) {
#pragma unroll
    for (int q = FIRST; q < LAST; ++q)
        acc[q % NB] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[q % NB][q / NB], o.b[q % NB][q / NB], acc[q % NB], 0, 0, 0);
}

#define RR_PAIRD(P, CUR, NXT)                                  \
    lds_wait();                                                \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma_d<NB, 0, 1>(CUR, acc);                           \
    __builtin_amdgcn_sched_barrier(0);                         \
    if ((P) + 1 < 8) NXT.template load<((P) + 1) & 7>(abase, bbase); \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma_d<NB, 1, 2 * NB>(CUR, acc);                      \
    __builtin_amdgcn_sched_barrier(0);

// Everything a wave does in the diagonal kernel, for its NB blocks (5 for waves 0-3, 4 for waves 4-7: the two
// waves of a SIMD issue 9 MFMAs per k-step between them, none wasted).
template <int NB>
__device__ __forceinline__ void syrk_diag_body(const SyrkArgs &p, float *lds, int wave, int lane) {
    const int ta = blockIdx.x % p.nb;
    const int ks = blockIdx.x / p.nb;
    const int ca = ta * GR_TC;

    const int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned lane_off = 4u * ((lane >> 5) * GR_TC + (lane & 31));  // row stride 1024 B = 4 units of 256 B
    int bi[NB], bj[NB];
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        bi[e] = RR_DIAG_I[wave][e];
        bj[e] = RR_DIAG_J[wave][e];
    }
    floatx16 acc[NB];
#pragma unroll
    for (int e = 0; e < NB; ++e)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[e][k] = 0.f;

    // DMA: 32 row segments of 1 KiB per k-block; wave w moves rows 4w..4w+3
    auto dma_tile = [&](float *buf, int64_t kb0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = 4 * wave + k;
            RR_DEV_ASSERT(kb0 + lr < p.rows && ca + GR_TC <= p.ldp && p.rows % GR_KB == 0 && p.rows_per_split % GR_KB == 0);
            const float *src = p.P + (kb0 + lr) * p.ldp + ca + 4 * lane;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + lr * GR_TC), 16, 0, 0);
        }
    };

    const int64_t nkb = (row_end - row_begin) / GR_KB;
    if (nkb > 0) {
        dma_tile(lds, row_begin);
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (GR_KB * GR_TC), row_begin + (kb + 1) * GR_KB);
            unsigned abase[NB], bbase[NB];
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                abase[e] = lds0 + cbuf * (4u * GR_KB * GR_TC) + lane_off + 128u * bi[e];
                bbase[e] = lds0 + cbuf * (4u * GR_KB * GR_TC) + lane_off + 128u * bj[e];
            }
            KOpsD<NB> o0, o1;
            o0.template load<0>(abase, bbase);
            RR_PAIRD(0, o0, o1) RR_PAIRD(1, o1, o0) RR_PAIRD(2, o0, o1) RR_PAIRD(3, o1, o0)
            RR_PAIRD(4, o0, o1) RR_PAIRD(5, o1, o0) RR_PAIRD(6, o0, o1) RR_PAIRD(7, o1, o0)
            __syncthreads();
        }
    }

    const int64_t F = p.F;
    const int hi = lane >> 5;
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int64_t gc = ca + 32 * bj[e] + (lane & 31);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t gr = ca + 32 * bi[e] + (k & 3) + 8 * (k >> 2) + 4 * hi;
            if (gr <= gc) rr_syrk_out(p, ks, gr, gc, acc[e][k]);
        }
    }
    (void)F;
}
#undef RR_PAIRD

__global__ void __launch_bounds__(GR_THREADS, 2)
rr_syrk_f32_diag_kernel(const SyrkArgs p) {
    __shared__ float lds[2 * GR_KB * GR_TC];  // 64 KiB: two [32][256] tiles (A side only)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4)  // wave-uniform: both paths hit the same barriers
        syrk_diag_body<5>(p, lds, wave, lane);
    else
        syrk_diag_body<4>(p, lds, wave, lane);
}

// ---------------------------------------------------------------------------------------
// The diagonal-tile kernel with the eight DIAGONAL 32x32 blocks of a tile done as 16x16 sub-blocks (round 3).  Every wave
// of the kernel above owns exactly one block on the tile's diagonal (block-row == block-column), of which only the upper
// triangle is needed; as one v_mfma_f32_32x32x2_f32 per k-step it is computed whole (the "89 % bound" of DESIGN 9.4).
// v_mfma_f32_16x16x4_f32 has the same rate per flop (32 cycles for 16 x 16 x 4), so the block is split into its three
// upper 16x16 sub-blocks (0,0), (0,1), (1,1): 3 x 32 = 96 cycles per four rows instead of 2 x 64 = 128.  Its operands are
// one float per lane (lane l: row 4P + (l >> 4) of the k-step pair, column c + (l & 15)) and serve as A and B alike:
//   X0 = columns [c, c + 16), X1 = columns [c + 16, c + 32):  D00 += X0^T X0, D01 += X0^T X1, D11 += X1^T X1.
// A SIMD (waves w, w + 4) then issues 1088 instead of 1152 MFMA cycles per k-step pair (-5.6 %).  ED = the position of
// the diagonal block in the wave's list (RR_DIAG_I / RR_DIAG_J): compile time, one instantiation per distinct (NB, ED).
// ---------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ float lds_read_b32_off(unsigned addr) {
    float r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}

template <int NB, int ED>
struct KOpsD16 {
    float2v a[NB], b[NB];  // entry ED is not used
    float x0, x1;
    template <int P>
    __device__ __forceinline__ void load(const unsigned (&abase)[NB], const unsigned (&bbase)[NB], unsigned xbase) {
#pragma unroll
        for (int e = 0; e < NB; ++e)
            if (e != ED) a[e] = lds_read2st64<16 * P, 16 * P + 8>(abase[e]);
#pragma unroll
        for (int e = 0; e < NB; ++e)
            if (e != ED) b[e] = lds_read2st64<16 * P, 16 * P + 8>(bbase[e]);
        x0 = lds_read_b32_off<4096 * P>(xbase);       // rows 4P + (lane >> 4): 4 rows of 1 KiB per k-step pair
        x1 = lds_read_b32_off<4096 * P + 64>(xbase);  // 16 columns to the right
    }
};

// MFMAs [FIRST, LAST) of a k-step pair: the NB - 1 off-diagonal blocks' first k-step, the diagonal block's three 16x16x4
// products (both k-steps at once), the off-diagonal blocks' second k-step
template <int NB, int ED, int FIRST, int LAST>
__device__ __forceinline__ void gram_mfma_d16(const KOpsD16<NB, ED> &o, floatx16 (&acc)[NB], floatx4 (&dd)[3]) {
#pragma unroll
    for (int q = FIRST; q < LAST; ++q) {
        if (q < NB - 1 || q >= NB + 2) {
            const int s_ = q < NB - 1 ? 0 : 1;
            const int r_ = q < NB - 1 ? q : q - (NB + 2);
            const int e = r_ < ED ? r_ : r_ + 1;
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[e][s_], o.b[e][s_], acc[e], 0, 0, 0);
        } else {
            const int t = q - (NB - 1);  // 0: (0,0)  1: (0,1)  2: (1,1)
            dd[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t == 2 ? o.x1 : o.x0, t == 0 ? o.x0 : o.x1, dd[t], 0, 0, 0);
        }
    }
}

#define RR_PAIRD16(P, CUR, NXT)                                \
    lds_wait();                                                \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma_d16<NB, ED, 0, 1>(CUR, acc, dd);                 \
    __builtin_amdgcn_sched_barrier(0);                         \
    if ((P) + 1 < KB / 4) NXT.template load<((P) + 1) & (KB / 4 - 1)>(abase, bbase, xbase); \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma_d16<NB, ED, 1, 2 * NB + 1>(CUR, acc, dd);        \
    __builtin_amdgcn_sched_barrier(0);

// KB = rows per k-block: 32 (default) or 64 (RR_SYRK_DIAG_KB=64, an experiment kept for A/B runs).  The diagonal kernel's
// k-block carries half the MFMA cycles of the off-diagonal kernel's (one side of the tile, 36 of 64 blocks), so the
// per-k-block barrier + DMA burst should weigh twice as much; 64-row k-blocks (128 KiB of LDS -- the kernel runs one
// workgroup per CU either way: 152 VGPRs) halve their number -- and change nothing (87.9 vs 86.6 ms per 10M rows).
template <int NB, int ED, int KB>
__device__ __forceinline__ void syrk_diag16_body(const SyrkArgs &p, float *lds, int wave, int lane, const unsigned bid) {
    const int ta = bid % p.nb;
    const int ks = bid / p.nb;
    const int ca = ta * GR_TC;
    const int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned lane_off = 4u * ((lane >> 5) * GR_TC + (lane & 31));  // 32x32x2 operands: row stride 1024 B
    int bi[NB], bj[NB];
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        bi[e] = RR_DIAG_I[wave][e];
        bj[e] = RR_DIAG_J[wave][e];
    }
    RR_DEV_ASSERT(bi[ED] == bj[ED]);
    const unsigned lane_off16 = 4u * ((lane >> 4) * GR_TC + 32 * bi[ED] + (lane & 15));  // 16x16x4 operands
    floatx16 acc[NB];
    floatx4 dd[3];
#pragma unroll
    for (int e = 0; e < NB; ++e)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[e][k] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) dd[t] = floatx4{0.f, 0.f, 0.f, 0.f};

    // KB row segments of 1 KiB per k-block; wave w moves rows (KB/8) w ...; requests through a buffer descriptor (one constant
    // VGPR, scalar row offsets: no vector instruction per request, see rr_dma_kblock)
    const unsigned voff = 16u * lane, ldp4 = (unsigned)p.ldp * 4u;
    auto dma_tile = [&](float *buf, int64_t kb0) {
        RR_DEV_ASSERT(kb0 + KB <= p.rows && ca + GR_TC <= p.ldp && p.rows % KB == 0 && p.rows_per_split % KB == 0);
        const rr_rsrc_t ra = rr_make_rsrc(p.P + kb0 * p.ldp + ca, 0x7fffffffu);
#pragma unroll
        for (int k = 0; k < KB / 8; ++k) {
            const int lr = (KB / 8) * wave + k;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(buf + lr * GR_TC), 16, voff, (unsigned)lr * ldp4, 0, 0);
        }
    };

    const int64_t nkb = (row_end - row_begin) / KB;
    if (nkb > 0) {
        dma_tile(lds, row_begin);
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            const int slot = p.spread ? rr_dma_slot(wave, p.spread) : 0;  // (staggered like the off-diagonal kernel's)
            auto dma_next = [&]() {
                if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (KB * GR_TC), row_begin + (kb + 1) * KB);
            };
            if (slot == 0) dma_next();
            unsigned abase[NB], bbase[NB];
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                abase[e] = lds0 + cbuf * (4u * KB * GR_TC) + lane_off + 128u * bi[e];
                bbase[e] = lds0 + cbuf * (4u * KB * GR_TC) + lane_off + 128u * bj[e];
            }
            const unsigned xbase = lds0 + cbuf * (4u * KB * GR_TC) + lane_off16;
            KOpsD16<NB, ED> o0, o1;
            o0.template load<0>(abase, bbase, xbase);
            RR_PAIRD16(0, o0, o1)
            if (slot == 1) dma_next();
            RR_PAIRD16(1, o1, o0)
            if (slot == 2) dma_next();
            RR_PAIRD16(2, o0, o1)
            if (slot == 3) dma_next();
            RR_PAIRD16(3, o1, o0)
            RR_PAIRD16(4, o0, o1) RR_PAIRD16(5, o1, o0) RR_PAIRD16(6, o0, o1) RR_PAIRD16(7, o1, o0)
            if constexpr (KB == 64) {
                RR_PAIRD16(8, o0, o1) RR_PAIRD16(9, o1, o0) RR_PAIRD16(10, o0, o1) RR_PAIRD16(11, o1, o0)
                RR_PAIRD16(12, o0, o1) RR_PAIRD16(13, o1, o0) RR_PAIRD16(14, o0, o1) RR_PAIRD16(15, o1, o0)
            }
            __syncthreads();
        }
    }

    const int hi = lane >> 5;
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        if (e == ED) continue;
        const int64_t gc = ca + 32 * bj[e] + (lane & 31);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t gr = ca + 32 * bi[e] + (k & 3) + 8 * (k >> 2) + 4 * hi;
            if (gr <= gc) rr_syrk_out(p, ks, gr, gc, acc[e][k]);
        }
    }
    // the diagonal block's three 16x16 sub-blocks (C/D of the 16x16 forms: column = lane & 15, row = 4 (lane >> 4) + register)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int64_t gc = ca + 32 * bi[ED] + (t >= 1 ? 16 : 0) + (lane & 15);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t gr = ca + 32 * bi[ED] + (t == 2 ? 16 : 0) + 4 * (lane >> 4) + k;
            if (gr <= gc) rr_syrk_out(p, ks, gr, gc, dd[t][k]);
        }
    }
}
#undef RR_PAIRD16

template <int KB>
__device__ __forceinline__ void syrk_diag16_dispatch(const SyrkArgs &p, float *lds, const unsigned bid) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {  // wave-uniform: every path runs the same barriers; (NB, ED) as laid out in RR_DIAG_I / RR_DIAG_J
        case 0: case 1: case 2: case 3: syrk_diag16_body<5, 0, KB>(p, lds, wave, lane, bid); break;
        case 4: syrk_diag16_body<4, 3, KB>(p, lds, wave, lane, bid); break;
        case 5: syrk_diag16_body<4, 2, KB>(p, lds, wave, lane, bid); break;
        case 6: syrk_diag16_body<4, 1, KB>(p, lds, wave, lane, bid); break;
        default: syrk_diag16_body<4, 0, KB>(p, lds, wave, lane, bid); break;
    }
}

template <int KB>
__global__ void __launch_bounds__(GR_THREADS, 2)
rr_syrk_f32_diag16_kernel(const SyrkArgs p) {
    __shared__ float lds[2 * KB * GR_TC];  // two [KB][256] tiles (A side only): 64 / 128 KiB
    syrk_diag16_dispatch<KB>(p, lds, blockIdx.x);
}

// ---------------------------------------------------------------------------------------
// Small feature counts over FEW rows (F <= 1024: BASELINE config 1 is F = 512, N = 10 000): 128 x 128 tiles.
// On the 256 x 256 tiles above F = 512 is ONE off-diagonal tile and two diagonal ones: however the rows are split, every
// workgroup pays the big tile's prologue and a 64k-entry atomic epilogue, and the three tiles never fill 256 CUs
// (profiles/r05_c1_latency: 240 + 148 us for 2.6 GFLOP = 17 us of matrix-core time).  Here every upper 128 x 128 tile x
// K-split is a workgroup of 4 waves (64 x 64 per wave: 2 x 2 v_mfma_f32_32x32x2f32 accumulators), k-blocks of 32 rows
// through registers into a double-buffered LDS tile [32][A 128 | B 128]; diagonal tiles compute the whole block and keep
// gr <= gc; the rider column (Phi^T y, column F) and the deterministic slabs go through rr_syrk_out as in the big kernels.
// ---------------------------------------------------------------------------------------
constexpr int GS_TC = 128, GS_KB = 32, GS_LD = 2 * GS_TC, GS_THREADS = 256;

__global__ void __launch_bounds__(GS_THREADS, 2)
rr_syrk_f32_small_kernel(const SyrkArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2][GS_KB * GS_LD];  // 2 x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int nb = (int)(p.ldp / GS_TC);
    const int64_t ks = blockIdx.x / p.ntiles;
    int t = (int)(blockIdx.x % p.ntiles), ta = 0;
    while (t >= nb - ta) {  // upper tiles row by row: (ta, tb), tb >= ta
        t -= nb - ta;
        ++ta;
    }
    const int tb = ta + t;
    const int64_t ca = (int64_t)ta * GS_TC, cb = (int64_t)tb * GS_TC;
    const int64_t r0 = ks * p.rows_per_split;
    int64_t r1 = r0 + p.rows_per_split;
    if (r1 > p.rows) r1 = p.rows;
    const int nkb = (int)((r1 - r0 + GS_KB - 1) / GS_KB);   // (rows % 32 == 0: whole k-blocks)
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
        const float *base = p.P + (r0 + (int64_t)kb * GS_KB + lr) * p.ldp;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = *(const float4 *)(base + (int64_t)8 * u * p.ldp + ca + 4 * lc);
            rb[u] = *(const float4 *)(base + (int64_t)8 * u * p.ldp + cb + 4 * lc);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *(float4 *)&lds[buf][(lr + 8 * u) * GS_LD + 4 * lc] = ra[u];
            *(float4 *)&lds[buf][(lr + 8 * u) * GS_LD + GS_TC + 4 * lc] = rb[u];
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
        for (int k2 = 0; k2 < GS_KB; k2 += 2) {
            const float *row = L + (k2 + kk) * GS_LD;
            const float a0 = row[wr * 64 + i32], a1 = row[wr * 64 + 32 + i32];
            const float b0 = row[GS_TC + wc * 64 + i32], b1 = row[GS_TC + wc * 64 + 32 + i32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kb + 1 < nkb) lstore(cur ^ 1);
        __syncthreads();
    }
    // C/D of the 32x32 forms: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t gc = cb + wc * 64 + 32 * j + i32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (gr <= gc) rr_syrk_out(p, ks, gr, gc, acc[i][j][e]);
            }
        }
}

// Experiment of round 4 (RR_SYRK_MERGE_DIAG=1): ONE launch for the whole Gram -- the off-diagonal tiles'
// workgroups first, the diagonal tiles' (their own K-splits, their own code path) behind them in the same grid.
__global__ void __launch_bounds__(GR_THREADS, 2)
rr_syrk_f32_merged_kernel(const SyrkArgs p, const SyrkArgs pd, const unsigned n_off) {
    __shared__ float lds[2 * GR_KB * GR_LD];
    if (blockIdx.x < n_off) rr_syrk_f32_body<3>(p, lds, blockIdx.x);
    else syrk_diag16_dispatch<32>(pd, lds, blockIdx.x - n_off);
}

// ---------------------------------------------------------------------------------------
// f64 Gram: G(upper) += P^T P with v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s peak).  Same structure
// as the f32 kernel at half the tile: 128x128 block of G per workgroup of 4 waves (64x64 per
// wave = 16 accumulators of 4 f64), k-blocks of 16 rows arriving by LDS-DMA as [16][128 | 128]
// f64 with the row stride padded by 128 B so that the 4 rows of a k-step fall on different
// bank halves (conflict-free ds_read_b64), two workgroups per CU (68 KiB of LDS each).
// ---------------------------------------------------------------------------------------
constexpr int G64_TC = 128;            // columns per tile side
constexpr int G64_KB = 16;             // rows per k-block
constexpr int G64_LDB = 2 * G64_TC * 8 + 128;  // LDS row stride in bytes (2048 + 128 pad)
constexpr int G64_THREADS = 256;

struct Syrk64Args {
    const double *P;  // (rows, ldp) f64 features, zero padded; rows % 16 == 0, ldp % 128 == 0
    int64_t rows, ldp;
    int F, nb, ntiles;
    int64_t rows_per_split;  // multiple of 16
    double *G;
    int offdiag_only;  // 1: tiles ta < tb only (the diagonal tiles run in rr_syrk_f64_diag_kernel)
    int lower_tri = 0;  // 1: P is square and lower triangular (C = Y^T Y of rr_posdef.hip): rows above a tile's last column
                        // block are zero in it, its k-loop starts at that block
    double *part = nullptr;  // deterministic mode: slab ks (part_stride = ldp * ldp doubles) takes K-split ks' partials
    int64_t part_stride = 0;
};

__device__ __forceinline__ void rr_syrk64_out(const Syrk64Args &p, int64_t ks, int64_t gr, int64_t gc, double v) {
    if (p.part != nullptr) p.part[ks * p.part_stride + gr * p.ldp + gc] = v;
    else unsafeAtomicAdd(&p.G[gr * (int64_t)p.F + gc], v);
}

// deterministic mode: G[gr][gc] (gr <= gc < F; gc == F: bcol[gr]) += sum over the K-splits' slabs in ascending order
template <typename T>
__global__ void __launch_bounds__(256)
rr_syrk_det_reduce_kernel(const T *__restrict__ part, int64_t stride, int64_t ldp, int F, int nsplit, double *__restrict__ G,
                          double *__restrict__ bcol) {
    const int64_t gc = (int64_t)blockIdx.x * 256 + threadIdx.x, gr = blockIdx.y;
    if (gc < gr || gc > F || (gc == F && bcol == nullptr)) return;
    const T *q = part + gr * ldp + gc;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += (double)q[(int64_t)k * stride];
    if (gc < F) G[gr * (int64_t)F + gc] += s;
    else bcol[gr] += s;
}

template <int OFF>
__device__ __forceinline__ double lds_read_b64(unsigned addr) {
    double r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}

struct KOps64 {
    double a[4], b[4];
    // k-step T: rows 4T + (lane >> 4) (folded into the bases); block i at + i * 16 columns
    template <int T>
    __device__ __forceinline__ void load(unsigned abase, unsigned bbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = 0;
        a[0] = lds_read_b64<4 * T * G64_LDB + 0 * 128>(abase);
        a[1] = lds_read_b64<4 * T * G64_LDB + 1 * 128>(abase);
        a[2] = lds_read_b64<4 * T * G64_LDB + 2 * 128>(abase);
        a[3] = lds_read_b64<4 * T * G64_LDB + 3 * 128>(abase);
        b[0] = lds_read_b64<4 * T * G64_LDB + 0 * 128>(bbase);
        b[1] = lds_read_b64<4 * T * G64_LDB + 1 * 128>(bbase);
        b[2] = lds_read_b64<4 * T * G64_LDB + 2 * 128>(bbase);
        b[3] = lds_read_b64<4 * T * G64_LDB + 3 * 128>(bbase);
    }
};

template <int FIRST, int LAST>
__device__ __forceinline__ void gram64_mfma(const KOps64 &o, doublex4 (&acc)[4][4]) {
#pragma unroll
    for (int q = FIRST; q < LAST; ++q)
        acc[q >> 2][q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[q >> 2], o.b[q & 3], acc[q >> 2][q & 3], 0, 0, 0);
}

#define RR_STEP64(T, CUR, NXT)                                  \
    lds_wait();                                                 \
    __builtin_amdgcn_sched_barrier(0);                          \
    gram64_mfma<0, 1>(CUR, acc);                                \
    __builtin_amdgcn_sched_barrier(0);                          \
    if ((T) + 1 < 4) NXT.template load<((T) + 1) & 3>(abase, bbase); \
    __builtin_amdgcn_sched_barrier(0);                          \
    gram64_mfma<1, 16>(CUR, acc);                               \
    __builtin_amdgcn_sched_barrier(0);

__global__ void __launch_bounds__(G64_THREADS, 2)
rr_syrk_f64_kernel(const Syrk64Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G64_KB * G64_LDB];  // 68 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tdx = blockIdx.x % p.ntiles;
    const int ks = blockIdx.x / p.ntiles;
    int ta = 0;
    const int od = p.offdiag_only;  // row ta then holds nb - ta - od tiles
    while (tdx >= p.nb - ta - od) {
        tdx -= p.nb - ta - od;
        ++ta;
    }
    const int tb = ta + tdx + od;
    const int ca = ta * G64_TC, cb = tb * G64_TC;

    int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;
    if (p.lower_tri && row_begin < cb) row_begin = cb;
    if (row_begin >= row_end && p.part == nullptr) return;  // nothing to add (deterministic mode still writes its zeros)

    // wave (wr, wc): rows [wr*64, +64) of side A x cols [wc*64, +64) of side B
    const int wr = wave >> 1, wc_ = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned aoff = (lane >> 4) * G64_LDB + 8u * (wr * 64 + (lane & 15));
    const unsigned boff = (lane >> 4) * G64_LDB + 8u * (G64_TC + wc_ * 64 + (lane & 15));
    doublex4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

    // DMA: 16 rows x 2 sides = 32 row segments of 1 KiB; wave w moves rows 4w..4w+3
    // (requests through buffer descriptors: one constant VGPR, scalar row offsets -- see rr_dma_kblock)
    const unsigned voff = 16u * lane, ldp8 = (unsigned)p.ldp * 8u;
    auto dma_tile = [&](unsigned char *buf, int64_t kb0) {
        const rr_rsrc_t ra = rr_make_rsrc(p.P + kb0 * p.ldp + ca, 0x7fffffffu), rb = rr_make_rsrc(p.P + kb0 * p.ldp + cb, 0x7fffffffu);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = 4 * wave + k;
            unsigned char *dst = buf + lr * G64_LDB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)dst, 16, voff, (unsigned)lr * ldp8, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + G64_TC * 8), 16, voff, (unsigned)lr * ldp8, 0, 0);
        }
    };

    const int64_t nkb = (row_end - row_begin) / G64_KB;
    RR_DEV_ASSERT(p.rows % G64_KB == 0 && p.rows_per_split % G64_KB == 0 && cb + G64_TC <= p.ldp && row_end <= p.rows);
    if (nkb > 0) {
        dma_tile(lds, row_begin);
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (G64_KB * G64_LDB), row_begin + (kb + 1) * G64_KB);
            const unsigned abase = lds0 + cbuf * (G64_KB * G64_LDB) + aoff;
            const unsigned bbase = lds0 + cbuf * (G64_KB * G64_LDB) + boff;
            KOps64 o0, o1;
            o0.load<0>(abase, bbase);
            RR_STEP64(0, o0, o1) RR_STEP64(1, o1, o0) RR_STEP64(2, o0, o1) RR_STEP64(3, o1, o0)
            __syncthreads();
        }
    }

    // flush (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg)
    const int64_t F = p.F;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gc = cb + wc_ * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t gr = ca + wr * 64 + i * 16 + (lane >> 4) + 4 * e;
                if (gr <= gc && gc < F) rr_syrk64_out(p, ks, gr, gc, acc[i][j][e]);
            }
        }
}

// ---------------------------------------------------------------------------------------
// Diagonal tiles (ta == tb) of the f64 SYRK.  Of the 64 16x16 blocks of a diagonal 128x128 tile only the 36 with
// block-row <= block-column are needed; the full-tile kernel computes all 64 and throws 28 away at the flush.  Here
// the 36 are dealt 9 per wave (one wave per SIMD) in groups that share operands -- wave w reads the A-side operands
// DA[w] and B-side operands DB[w] (7-8 ds_read_b64 per k-step instead of 8) and issues 9 instead of 16 MFMAs -- and
// only ONE side of the tile is DMA'd (both operands come from the same 128 columns).  Launched after the
// off-diagonal kernel with its own K-split count.
//   wave 0: rows 0-2 x cols 0-3 (upper part)        wave 1: (3,3) + rows 0-1 x cols 4-7
//   wave 2: rows 2-3 x cols 4-7 + (4,4)             wave 3: rows 4-7 x cols 5-7 (upper part)
// ---------------------------------------------------------------------------------------
constexpr int G64D_LDB = G64_TC * 8 + 128;  // LDS row stride of the one-sided tile (1024 + 128 pad: conflict-free as above)

template <int W> struct Diag64 {};
template <> struct Diag64<0> {
    static constexpr int NA = 3, NBO = 4;
    static constexpr int A[4] = {0, 1, 2, 0}, B[5] = {0, 1, 2, 3, 0};
    static constexpr int QA[9] = {0, 0, 0, 0, 1, 1, 1, 2, 2}, QB[9] = {0, 1, 2, 3, 1, 2, 3, 2, 3};  // indices into A / B
};
template <> struct Diag64<1> {
    static constexpr int NA = 3, NBO = 5;
    static constexpr int A[4] = {0, 1, 3, 0}, B[5] = {3, 4, 5, 6, 7};
    static constexpr int QA[9] = {2, 0, 0, 0, 0, 1, 1, 1, 1}, QB[9] = {0, 1, 2, 3, 4, 1, 2, 3, 4};
};
template <> struct Diag64<2> {
    static constexpr int NA = 3, NBO = 4;
    static constexpr int A[4] = {2, 3, 4, 0}, B[5] = {4, 5, 6, 7, 0};
    static constexpr int QA[9] = {2, 0, 0, 0, 0, 1, 1, 1, 1}, QB[9] = {0, 0, 1, 2, 3, 0, 1, 2, 3};
};
template <> struct Diag64<3> {
    static constexpr int NA = 4, NBO = 3;
    static constexpr int A[4] = {4, 5, 6, 7}, B[5] = {5, 6, 7, 0, 0};
    static constexpr int QA[9] = {0, 0, 0, 1, 1, 1, 2, 2, 3}, QB[9] = {0, 1, 2, 0, 1, 2, 1, 2, 2};
};

template <int W>
struct KOps64D {
    typedef Diag64<W> D;
    double a[D::NA], b[D::NBO];
    template <int T>
    __device__ __forceinline__ void load(unsigned base) {
        if constexpr (D::NA > 0) a[0] = lds_read_b64<4 * T * G64D_LDB + D::A[0] * 128>(base);
        if constexpr (D::NA > 1) a[1] = lds_read_b64<4 * T * G64D_LDB + D::A[1] * 128>(base);
        if constexpr (D::NA > 2) a[2] = lds_read_b64<4 * T * G64D_LDB + D::A[2] * 128>(base);
        if constexpr (D::NA > 3) a[3] = lds_read_b64<4 * T * G64D_LDB + D::A[3] * 128>(base);
        if constexpr (D::NBO > 0) b[0] = lds_read_b64<4 * T * G64D_LDB + D::B[0] * 128>(base);
        if constexpr (D::NBO > 1) b[1] = lds_read_b64<4 * T * G64D_LDB + D::B[1] * 128>(base);
        if constexpr (D::NBO > 2) b[2] = lds_read_b64<4 * T * G64D_LDB + D::B[2] * 128>(base);
        if constexpr (D::NBO > 3) b[3] = lds_read_b64<4 * T * G64D_LDB + D::B[3] * 128>(base);
        if constexpr (D::NBO > 4) b[4] = lds_read_b64<4 * T * G64D_LDB + D::B[4] * 128>(base);
    }
};

template <int W, int FIRST, int LAST>
__device__ __forceinline__ void gram64d_mfma(const KOps64D<W> &o, doublex4 (&acc)[9]) {
    typedef Diag64<W> D;
#pragma unroll
    for (int q = FIRST; q < LAST; ++q)
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[D::QA[q]], o.b[D::QB[q]], acc[q], 0, 0, 0);
}

#define RR_STEP64D(T, CUR, NXT)                            \
    lds_wait();                                            \
    __builtin_amdgcn_sched_barrier(0);                     \
    gram64d_mfma<W, 0, 1>(CUR, acc);                       \
    __builtin_amdgcn_sched_barrier(0);                     \
    if ((T) + 1 < 4) NXT.template load<((T) + 1) & 3>(base); \
    __builtin_amdgcn_sched_barrier(0);                     \
    gram64d_mfma<W, 1, 9>(CUR, acc);                       \
    __builtin_amdgcn_sched_barrier(0);

template <int W>
__device__ __forceinline__ void syrk64_diag_body(const Syrk64Args &p, unsigned char *lds, int lane) {
    typedef Diag64<W> D;
    const int ta = blockIdx.x % p.nb;
    const int ks = blockIdx.x / p.nb;
    const int ca = ta * G64_TC;
    int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;
    if (p.lower_tri && row_begin < ca) row_begin = ca;
    if (row_begin >= row_end && p.part == nullptr) return;  // (the same for all four waves)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned off = (lane >> 4) * G64D_LDB + 8u * (lane & 15);
    doublex4 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = 0.0;

    // DMA: 16 row segments of 1 KiB per k-block; wave w moves rows 4w..4w+3
    auto dma_tile = [&](unsigned char *buf, int64_t kb0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = 4 * W + k;
            RR_DEV_ASSERT(kb0 + lr < p.rows && ca + G64_TC <= p.ldp);
            const double *src = p.P + (kb0 + lr) * p.ldp + ca + 2 * lane;  // 16 B per lane
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + lr * G64D_LDB), 16, 0, 0);
        }
    };

    const int64_t nkb = (row_end - row_begin) / G64_KB;
    if (nkb > 0) {
        dma_tile(lds, row_begin);
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cbuf = (int)(kb & 1);
            if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (G64_KB * G64D_LDB), row_begin + (kb + 1) * G64_KB);
            const unsigned base = lds0 + cbuf * (G64_KB * G64D_LDB) + off;
            KOps64D<W> o0, o1;
            o0.template load<0>(base);
            RR_STEP64D(0, o0, o1) RR_STEP64D(1, o1, o0) RR_STEP64D(2, o0, o1) RR_STEP64D(3, o1, o0)
            __syncthreads();
        }
    }

    // flush (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg); the 8 blocks on the diagonal keep gr <= gc
    const int64_t F = p.F;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int64_t gc = ca + 16 * D::B[D::QB[q]] + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t gr = ca + 16 * D::A[D::QA[q]] + (lane >> 4) + 4 * e;
            if (gr <= gc && gc < F) rr_syrk64_out(p, ks, gr, gc, acc[q][e]);
        }
    }
}
#undef RR_STEP64D

__global__ void __launch_bounds__(G64_THREADS, 2)
rr_syrk_f64_diag_kernel(const Syrk64Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G64_KB * G64D_LDB];  // 36 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {  // wave-uniform: all four paths hit the same barriers
        case 0: syrk64_diag_body<0>(p, lds, lane); break;
        case 1: syrk64_diag_body<1>(p, lds, lane); break;
        case 2: syrk64_diag_body<2>(p, lds, lane); break;
        default: syrk64_diag_body<3>(p, lds, lane); break;
    }
}

// D[M][N] = A^T B in f64 (second pass of _elbo / predict_moments in f64 arithmetic): A (K, lda) with M columns,
// B (K, ldb) with N columns, row-major, K % 16 == 0, M % 128 == 0, N % 128 == 0.  The f64 SYRK kernel's tile,
// operand reads and MFMA schedule with two source matrices, the whole K loop in one workgroup, plain stores.
struct Gemm64Args {
    const double *A, *B;
    double *D;
    int64_t lda, ldb, ldd;
    int K, ntb;  // ntb = N / 128
    int subtract;    // 1: D -= A^T B (blocked Cholesky updates) instead of D = A^T B
    int upper_only;  // 1: only tiles with column tile >= row tile (symmetric trailing update, upper triangle kept);
                     // 2: the same without tile (0, 0) (look-ahead Cholesky: that block is updated ahead, rr_posdef.hip)
};

__global__ void __launch_bounds__(G64_THREADS, 2)
rr_gemm_tn_f64_kernel(const Gemm64Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G64_KB * G64_LDB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ca = (int64_t)(blockIdx.x / p.ntb) * G64_TC;
    const int cb = (int)(blockIdx.x % p.ntb) * G64_TC;
    if (p.upper_only && (cb < ca || (p.upper_only == 2 && ca == 0 && cb == 0))) return;  // workgroup-uniform
    const int wr = wave >> 1, wc_ = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned aoff = (lane >> 4) * G64_LDB + 8u * (wr * 64 + (lane & 15));
    const unsigned boff = (lane >> 4) * G64_LDB + 8u * (G64_TC + wc_ * 64 + (lane & 15));
    doublex4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
    // (requests through buffer descriptors: one constant VGPR, scalar row offsets -- see rr_dma_kblock)
    const unsigned voff = 16u * lane, lda8 = (unsigned)p.lda * 8u, ldb8 = (unsigned)p.ldb * 8u;
    auto dma_tile = [&](unsigned char *buf, int64_t kb0) {
        const rr_rsrc_t ra = rr_make_rsrc(p.A + kb0 * p.lda + ca, 0x7fffffffu), rb = rr_make_rsrc(p.B + kb0 * p.ldb + cb, 0x7fffffffu);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = 4 * wave + k;
            unsigned char *dst = buf + lr * G64_LDB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)dst, 16, voff, (unsigned)lr * lda8, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + G64_TC * 8), 16, voff, (unsigned)lr * ldb8, 0, 0);
        }
    };
    const int nkb = p.K / G64_KB;
    dma_tile(lds, 0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cbuf = kb & 1;
        if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (G64_KB * G64_LDB), (int64_t)(kb + 1) * G64_KB);
        const unsigned abase = lds0 + cbuf * (G64_KB * G64_LDB) + aoff;
        const unsigned bbase = lds0 + cbuf * (G64_KB * G64_LDB) + boff;
        KOps64 o0, o1;
        o0.load<0>(abase, bbase);
        RR_STEP64(0, o0, o1) RR_STEP64(1, o1, o0) RR_STEP64(2, o0, o1) RR_STEP64(3, o1, o0)
        __syncthreads();
    }
    // (subtract: the 16 reads of a block row of D are requested together -- written element by element hipcc serialises
    // 64 load -> wait -> store round trips per lane, ~100 us per tile against 7 us of MFMAs at K = 128)
    double *Dp = p.D + (ca + wr * 64 + (lane >> 4)) * p.ldd + cb + wc_ * 64 + (lane & 15);  // (i, j, e): + (16 i + 4 e) ldd + 16 j
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (p.subtract) {
            double dv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[j][e] = Dp[(int64_t)(16 * i + 4 * e) * p.ldd + 16 * j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = dv[j][e] - acc[i][j][e];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) Dp[(int64_t)(16 * i + 4 * e) * p.ldd + 16 * j] = acc[i][j][e];
        __builtin_amdgcn_sched_barrier(0);
    }
}
// K = M = 128 (the blocked Cholesky's "block row <- U_jj^-T block row" and its one-tile updates, rr_posdef.hip): the
// products on the factorisation's dependent chain are ONE tile of the kernel above -- 8 double-buffered k-blocks behind
// each other in a single workgroup, ~19 us for 4 MFLOP.  Here a workgroup owns 32 columns of B / D (N / 32 workgroups),
// wave w the 32 rows 32 w .. of D; every wave loads its operands for the WHOLE K straight into registers in the MFMA
// operand layout (2 + 2 values per k-step, 128 B per 16 lanes; A and B come from L2), no LDS, no k-loop barrier: one
// memory latency, then 128 MFMAs.  In place (D == B) is safe because a workgroup reads and writes whole columns: the
// barrier before the stores is behind every wave's last MFMA, i.e. behind its last load.
__global__ void __launch_bounds__(G64_THREADS)
rr_gemm_tn_f64_k128_kernel(const Gemm64Args p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.x * 32;
    const int lq = lane >> 4, l15 = lane & 15;
    const double *Ap = p.A + (int64_t)lq * p.lda + wave * 32 + l15;  // rows 4 T + lq, columns 32 wave + 16 i + l15
    const double *Bp = p.B + (int64_t)lq * p.ldb + cb + l15;
    double a[32][2], b[32][2];
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[t][i] = Ap[(int64_t)(4 * t) * p.lda + 16 * i];
            b[t][i] = Bp[(int64_t)(4 * t) * p.ldb + 16 * i];
        }
    __builtin_amdgcn_sched_barrier(0);  // every load is requested before the first MFMA (hipcc otherwise keeps ~20 in flight)
    doublex4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][i], b[t][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);  // the barrier stays behind the last MFMA, i.e. behind vmcnt(0) of every wave's operand loads
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    double *Dp = p.D + (int64_t)(32 * wave + lq) * p.ldd + cb + l15;  // element (i, j, e): + (16 i + 4 e) ldd + 16 j
    if (p.subtract) {  // all 16 reads of D before the first store (one round trip, not 16)
        double dv[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[i][j][e] = Dp[(int64_t)(16 * i + 4 * e) * p.ldd + 16 * j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = dv[i][j][e] - acc[i][j][e];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) Dp[(int64_t)(16 * i + 4 * e) * p.ldd + 16 * j] = acc[i][j][e];
}
// ---------------------------------------------------------------------------------------------
// The float64 second pass' U = Phi C contracted with Phi, Err m^T and X block by block in registers (the float64
// counterpart of rr_gemm_gradt_f32_kernel, rr_elbo.hip; slm.py:193-195): the accumulator layout of
// v_mfma_f64_16x16x4_f64 -- element e of lane l is row 4 e + (l >> 4), column l & 15 -- is its B-operand layout for the
// four rows 4 e .. 4 e + 3, so   R[r][c] = +-(U[r][c] - err[r] m[c]) P[r][c +- n]   (formed in place of U) goes back into
// the matrix pipe with A = X[4 e + (l >> 4)][16 ib + (l & 15)]:  T[i][c mod n] += sum_r X[r][i] R[r][c].  A workgroup keeps
// its 128-column block and walks over row tiles g, g + G, ..; T (d <= 32: two 16-row blocks x four column blocks per wave)
// stays in registers until one f64-atomic flush.  Two workgroups per CU cover each other's epilogues.  U is never stored.
// ---------------------------------------------------------------------------------------------
struct Gradt64Args {
    const double *A, *B;  // A = Phi^T (K, lda): feature-major; B = C (K, ldb)
    int64_t lda, ldb;
    int K, ntb, nta;      // ntb = 2n / 128 column tiles, nta row tiles
    const double *P;      // (rows, ldp) row-major features
    int64_t ldp;
    const double *X;      // (rows, ldx), d <= 32 valid columns
    int64_t ldx, rows;
    int n, d;
    const double *err, *mvec;
    double *T;            // (d, n), accumulated into
};

__global__ void __launch_bounds__(G64_THREADS, 2)
rr_gemm_gradt_f64_kernel(const Gradt64Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G64_KB * G64_LDB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % p.ntb));
    const int g0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / p.ntb)), G = __builtin_amdgcn_readfirstlane((int)(gridDim.x / p.ntb));
    const int cb = tb * G64_TC;
    const bool cosblk = cb < p.n;
    const int pcol = cosblk ? cb + p.n : cb - p.n;  // partner column block in P
    const int tcol = cosblk ? cb : cb - p.n;        // column block in T
    const double sgn = cosblk ? 1.0 : -1.0;         // (A = Err m^T - U: see rr_grad_t64_kernel)
    const int wr = wave >> 1, wc_ = wave & 1;
    const int lq = lane >> 4, l15 = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned aoff = (lane >> 4) * G64_LDB + 8u * (wr * 64 + (lane & 15));
    const unsigned boff = (lane >> 4) * G64_LDB + 8u * (G64_TC + wc_ * 64 + (lane & 15));
    const int nkb = p.K / G64_KB;

    doublex4 tacc[2][4];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) tacc[ib][j][e] = 0.0;
    double mcol[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mcol[j] = p.mvec[cb + wc_ * 64 + j * 16 + l15];
    double xm[2];  // 1 for the lanes whose column of X exists
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) xm[ib] = 16 * ib + l15 < p.d ? 1.0 : 0.0;

    auto dma_tile = [&](unsigned char *buf, int64_t ca, int64_t kb0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = 4 * wave + k;
            const double *sa = p.A + (kb0 + lr) * p.lda + ca + 2 * lane;
            const double *sb = p.B + (kb0 + lr) * p.ldb + cb + 2 * lane;
            unsigned char *dst = buf + lr * G64_LDB;
            __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(dst + G64_TC * 8), 16, 0, 0);
        }
    };

    for (int ta = g0; ta < p.nta; ta += G) {
        const int64_t ca = (int64_t)ta * G64_TC;
        doublex4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
        dma_tile(lds, ca, 0);
        __syncthreads();
        for (int kb = 0; kb < nkb; ++kb) {
            const int cbuf = kb & 1;
            if (kb + 1 < nkb) dma_tile(lds + (cbuf ^ 1) * (G64_KB * G64_LDB), ca, (int64_t)(kb + 1) * G64_KB);
            const unsigned abase = lds0 + cbuf * (G64_KB * G64_LDB) + aoff;
            const unsigned bbase = lds0 + cbuf * (G64_KB * G64_LDB) + boff;
            KOps64 o0, o1;
            o0.load<0>(abase, bbase);
            RR_STEP64(0, o0, o1) RR_STEP64(1, o1, o0) RR_STEP64(2, o0, o1) RR_STEP64(3, o1, o0)
            __syncthreads();
        }
        // rows of this tile that exist; a row past the end reads row 0 of the tile instead (its U and Err are masked to 0)
        const int64_t trows = p.rows - ca < G64_TC ? p.rows - ca : G64_TC;
        const double *Pt = p.P + ca * p.ldp + pcol + wc_ * 64 + l15;
        const double *Xt = p.X + ca * p.ldx;
        const double *Et = p.err + ca;
        // R in place of U, one 16-row block at a time (its 16 + 4 loads are issued before their first use)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double pv[4][4], ev[4];
            int64_t ro[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = wr * 64 + i * 16 + 4 * e + lq;
                ro[e] = r < trows ? r : -1;
                const int64_t rc = r < trows ? r : 0;
                ev[e] = Et[rc];
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[e][j] = Pt[rc * p.ldp + j * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double w = ro[e] >= 0 ? sgn : 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][e] = w * fma(-ev[e], mcol[j], acc[i][j][e]) * pv[e][j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // T += X^T R: per 16-row block and 16-column block of X, four products over the lane's column blocks
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double xv[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = wr * 64 + i * 16 + 4 * e + lq;
                const int64_t rc = r < trows ? r : 0;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) xv[ib][e] = Xt[rc * p.ldx + (xm[ib] != 0.0 ? 16 * ib + l15 : 0)] * xm[ib];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        tacc[ib][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[ib][e], acc[i][j][e], tacc[ib][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 16 * ib + 4 * e + lq;
                if (i < p.d) unsafeAtomicAdd(&p.T[(size_t)i * p.n + tcol + wc_ * 64 + j * 16 + l15], tacc[ib][j][e]);
            }
}

int rr_launch_gemm_gradt_f64(rr_ctx *c, const double *Pt, int64_t lda, const double *C, int64_t ldb, int64_t K, int64_t mpad,
                             int64_t Fp, const double *P, int64_t ldp, const double *X, int64_t ldx, int64_t rows, int n, int d,
                             const double *err, const double *mvec, double *T) {
    Gradt64Args g;
    g.A = Pt; g.lda = lda; g.B = C; g.ldb = ldb; g.K = (int)K; g.ntb = (int)(Fp / G64_TC); g.nta = (int)(mpad / G64_TC);
    g.P = P; g.ldp = ldp; g.X = X; g.ldx = ldx; g.rows = rows; g.n = n; g.d = d; g.err = err; g.mvec = mvec; g.T = T;
    int G = 2 * c->num_cu / g.ntb;  // two workgroups per CU, each keeps its column block
    if (G < 1) G = 1;
    if (G > g.nta) G = g.nta;
    hipLaunchKernelGGL(rr_gemm_gradt_f64_kernel, dim3((unsigned)(G * g.ntb)), dim3(G64_THREADS), 0, c->stream, g);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

#undef RR_STEP64

// Host feature matrices of ANY basis (concatenations, LinearBasis, ...) reach the SYRK kernel
// through this repack: (rows, F) f32|f64 with leading dimension lds -> zero-padded f32 (rows_pad, ldp).
template <typename TS, typename TD = float>
__global__ void __launch_bounds__(256) rr_pack_f32_kernel(const TS *__restrict__ src, int64_t rows, int F,
                                                          int64_t lds_, TD *__restrict__ dst, int64_t ldp,
                                                          int64_t rows_pad) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ldp) return;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    for (int64_t r = r0; r < r0 + 64 && r < rows_pad; ++r)
        dst[r * ldp + c] = (r < rows && c < F) ? (TD)src[r * lds_ + c] : (TD)0;
}

// b += P^T y for a packed feature matrix (one column per thread, rows split over blockIdx.y)
template <typename TY, typename TP = float>
__global__ void __launch_bounds__(256) rr_gemv_t_kernel(const TP *__restrict__ P, const TY *__restrict__ y,
                                                        int64_t rows, int F, int64_t ldp, double *__restrict__ bvec,
                                                        int rows_per_block, int64_t bdet = 0) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (c >= F) return;
    double acc = 0.0;
    for (int64_t r = r0; r < r1; ++r) acc += (double)P[r * ldp + c] * (double)y[r];
    rr_acc_out(bvec, bdet, blockIdx.y, c, acc);
}

// y^T y (slm.py:161-162 via sqErr = yty - 2 m.b + m G m)
template <typename TX>
__global__ void __launch_bounds__(256) rr_yty_kernel(const TX *__restrict__ y, int64_t N, double *out, int64_t det = 0) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
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

// lower triangle <- upper triangle, 32x32 tiles through LDS so both sides stay coalesced
__global__ void __launch_bounds__(256) rr_symmetrize_kernel(double *G, int64_t F) {
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;  // tile (bi, bj) of the UPPER part, bi <= bj
    if (bi > bj) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = (int64_t)bi * 32 + k, c = (int64_t)bj * 32 + tx;
        tile[k][tx] = (r < F && c < F) ? G[r * F + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = (int64_t)bj * 32 + k, c = (int64_t)bi * 32 + tx;  // transposed position
        if (r < F && c < F && r > c) G[r * F + c] = tile[tx][k];
    }
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------

int rr_pick_dmax(int d) { return d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : d <= 128 ? 128 : 0; }

// ---------------------------------------------------------------------------------------
// Xdim > 128 (b->large).  A frequency's W column no longer fits in a thread's registers, so the phases
// Z = X Ws are produced as a matrix first -- f32 arithmetic: (X^T)^T Ws on the matrix cores
// (rr_gemm_tn_f32_kernel, rr_elbo.hip; operands X^T (dpad, rows) and Ws (dpad, npad), both zero padded),
// f64 arithmetic: a VALU kernel, 8 rows per thread -- and rr_trig_kernel applies cos / sin / scale (and
// accumulates Phi^T y).  Row sub-chunks of RR_LG_ROWS bound the scratch (X^T: dpad x rows, Z: rows x npad).
// Same arithmetic as the register kernels: f32 products of x and Ws summed in f32 (MFMA order), the phase
// reduced to [-0.5, 0.5] revolutions before v_sin / v_cos.
// ---------------------------------------------------------------------------------------
constexpr int64_t RR_LG_ROWS = 131072;

int rr_launch_gemm_tn_f32(rr_ctx *c, const float *A, int64_t lda, const float *B, int64_t ldb, float *D, int64_t ldd,
                          int64_t K, int64_t M, int64_t N);  // rr_elbo.hip
int rr_launch_gemm_trig_f32(rr_ctx *c, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t K, int64_t M, int64_t N,
                            float *P, int64_t ldp, int n, float scale, int64_t nvalid, int64_t nout, const void *y, int y_f64,
                            double *bvec);  // rr_elbo.hip: the same GEMM with the cos / sin / Phi^T y epilogue

// Xt[c][r] = (float) X[r][c], c < dpad (grid.x = dpad / 64), r < rows256 (grid.y = rows256 / 64); rows >= `rows` -> 0
template <typename TX>
__global__ void __launch_bounds__(256)
rr_xt_kernel(const TX *__restrict__ X, int64_t rows, int64_t ldx, float *__restrict__ Xt, int64_t ldt) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t r = r0 + ty + 4 * k;
        tile[ty + 4 * k][tx] = r < rows ? (float)X[r * ldx + c0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) Xt[(c0 + ty + 4 * k) * ldt + r0 + tx] = tile[tx][ty + 4 * k];
}

// Z[r][f] = sum_i x[r][i] Ws[i][f] in f64: one frequency per thread, 8 rows per block (x through the scalar cache)
template <typename TX>
__global__ void __launch_bounds__(256)
rr_phase_f64_kernel(const TX *__restrict__ X, int64_t rows, int64_t ldx, const double *__restrict__ Ws, int dpad,
                    int npad, double *__restrict__ Z, int64_t ldz) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 8;
    const TX *xr[8];
    double z[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t r = r0 + k < rows ? r0 + k : rows - 1;
        xr[k] = X + r * ldx;
        z[k] = 0.0;
    }
    for (int i = 0; i < dpad; ++i) {
        const double w = Ws[(size_t)i * npad + f];
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = fma((double)xr[k][i], w, z[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (r0 + k < rows) Z[(r0 + k) * ldz + f] = z[k];
}

// P[r][f] = cos(2 pi Z[r][f]) scale, P[r][n + f] = sin(..) scale for r < N, zero rows for N <= r < Npad;
// HAS_Y: bvec += P^T y.
template <typename TC, typename TO, bool HAS_Y, typename TY>
__global__ void __launch_bounds__(256)
rr_trig_kernel(const TC *__restrict__ Z, int64_t ldz, const TY *__restrict__ y, int64_t N, int64_t Npad, int n,
               TO *__restrict__ P, int64_t ldp, double *__restrict__ bvec, TC scale, int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fvalid = f < n;
    const int fc = fvalid ? f : 0;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > Npad) r1 = Npad;
    TC bc = 0, bs = 0;
    for (int64_t r = r0; r < r1; ++r) {
        TC c = 0, s = 0;
        if (r < N) {  // uniform
            sincos_rev(Z[r * ldz + fc], s, c);
            c *= scale;
            s *= scale;
            if (HAS_Y) {
                const TC yv = (TC)y[r];
                bc = fma(c, yv, bc);
                bs = fma(s, yv, bs);
            }
        }
        if (fvalid) {
            P[r * ldp + f] = (TO)c;
            P[r * ldp + n + f] = (TO)s;
        }
    }
    if (HAS_Y && fvalid) {
        unsafeAtomicAdd(&bvec[f], (double)bc);
        unsafeAtomicAdd(&bvec[n + f], (double)bs);
    }
}

static int large_ensure(rr_basis *b, size_t xt_bytes, size_t z_bytes) {
    if (b->lg_xt_bytes >= xt_bytes && b->lg_z_bytes >= z_bytes) return RR_OK;
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    if (b->lg_xt_bytes < xt_bytes) {
        if (b->lg_xt) (void)hipFree(b->lg_xt);
        b->lg_xt = nullptr;
        b->lg_xt_bytes = 0;
        RR_CHECK_HIP(hipMalloc((void **)&b->lg_xt, xt_bytes));
        b->lg_xt_bytes = xt_bytes;
    }
    if (b->lg_z_bytes < z_bytes) {
        if (b->lg_z) (void)hipFree(b->lg_z);
        b->lg_z = nullptr;
        b->lg_z_bytes = 0;
        RR_CHECK_HIP(hipMalloc(&b->lg_z, z_bytes));
        b->lg_z_bytes = z_bytes;
    }
    return RR_OK;
}

// Features of m rows into P (row-major, leading dimension ldp, [cos | sin] at columns [0, n) and [n, 2n)); rows
// [m, mpad) are written as zeros; db (optional, with y) accumulates Phi^T y.
template <typename TX, typename TC, typename TO>
static int large_features(rr_basis *b, const TX *X, const TX *y, int64_t m, int64_t mpad, int64_t ldx, TO *P, int64_t ldp,
                          double *db) {
    rr_ctx *c = b->ctx;
    constexpr bool F32 = sizeof(TC) == 4;
    const TC scale = (TC)(1.0 / sqrt((double)b->n));
    int64_t sub = (mpad + 255) / 256 * 256;
    if (sub > RR_LG_ROWS) sub = RR_LG_ROWS;
    int rc = large_ensure(b, F32 ? (size_t)b->dpad * sub * 4 : 16, (size_t)sub * b->npad * sizeof(TC));
    if (rc != RR_OK) return rc;
    TC *Z = (TC *)b->lg_z;
    const int fblocks = (b->n + 255) / 256;
    for (int64_t s0 = 0; s0 < mpad; s0 += sub) {
        const int64_t rows_out = mpad - s0 < sub ? mpad - s0 : sub;                // rows of P this sub-chunk writes
        const int64_t rows_in = m - s0 < 0 ? 0 : (m - s0 < sub ? m - s0 : sub);   // rows of X behind them
        if (rows_in > 0) {
            const TX *Xs = X + s0 * ldx;
            const int64_t r256 = (rows_in + 255) / 256 * 256;
            if constexpr (F32) {
                hipLaunchKernelGGL(rr_xt_kernel<TX>, dim3((unsigned)(b->dpad / 64), (unsigned)(r256 / 64)), dim3(256), 0,
                                   c->stream, Xs, rows_in, ldx, b->lg_xt, sub);
                if constexpr (std::is_same<TO, float>::value) {
                    // f32 features: cos / sin / Phi^T y in the GEMM's epilogue, the phase matrix never reaches HBM
                    rc = rr_launch_gemm_trig_f32(c, b->lg_xt, sub, b->dWs32, b->npad, b->dpad, r256, b->npad, P + s0 * ldp, ldp,
                                                 b->n, (float)scale, rows_in, rows_out, (y && db) ? (const void *)(y + s0) : nullptr,
                                                 sizeof(TX) == 8, db);
                    if (rc != RR_OK) return rc;
                    continue;
                }
                rc = rr_launch_gemm_tn_f32(c, b->lg_xt, sub, b->dWs32, b->npad, (float *)Z, b->npad, b->dpad, r256, b->npad);
                if (rc != RR_OK) return rc;
            } else {
                hipLaunchKernelGGL(rr_phase_f64_kernel<TX>, dim3((unsigned)(b->npad / 256), (unsigned)((rows_in + 7) / 8)),
                                   dim3(256), 0, c->stream, Xs, rows_in, ldx, b->dWs64, b->dpad, b->npad, (double *)Z,
                                   (int64_t)b->npad);
            }
        }
        int64_t rpb = (rows_out * fblocks + (int64_t)c->num_cu * 16 - 1) / ((int64_t)c->num_cu * 16);
        if (rpb < 16) rpb = 16;
        if (rpb > 1024) rpb = 1024;
        const dim3 grid(fblocks, (unsigned)((rows_out + rpb - 1) / rpb));
        TO *Ps = P + s0 * ldp;
        if (y && db)
            hipLaunchKernelGGL((rr_trig_kernel<TC, TO, true, TX>), grid, dim3(256), 0, c->stream, Z, (int64_t)b->npad, y + s0,
                               rows_in, rows_out, b->n, Ps, ldp, db, scale, (int)rpb);
        else
            hipLaunchKernelGGL((rr_trig_kernel<TC, TO, false, TX>), grid, dim3(256), 0, c->stream, Z, (int64_t)b->npad,
                               (const TX *)nullptr, rows_in, rows_out, b->n, Ps, ldp, (double *)nullptr, scale, (int)rpb);
        RR_CHECK_HIP(hipGetLastError());
    }
    return RR_OK;
}

// d Phi / d l_i from finished features (large Xdim):  dz_i = -x_i Ws[i][f] gfac_i,  d cos = -(P_s) dz_i,  d sin = P_c dz_i
// (P already carries the 1/sqrt(n) scale).  Same output layout as rr_rff_grad_kernel.
template <typename TX, typename TC, typename TO>
__global__ void __launch_bounds__(256)
rr_rff_grad_p_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, const TC *__restrict__ Ws,
                     const TC *__restrict__ gfac, const TC *__restrict__ P, int64_t ldp, int n, int npad, int nout,
                     TO *__restrict__ out, int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int64_t r = r0; r < r1; ++r) {
        const TX *xr = X + r * ldx;
        const TC c = P[r * ldp + f], s = P[r * ldp + n + f];
        TO *oc = out + ((size_t)r * 2 * n + f) * nout;
        TO *os = out + ((size_t)r * 2 * n + n + f) * nout;
        for (int i = 0; i < nout; ++i) {
            const TC dz = -(TC)xr[i] * Ws[(size_t)i * npad + f] * gfac[i];
            oc[i] = (TO)(-s * dz);
            os[i] = (TO)(c * dz);
        }
    }
}

template <typename TX, typename TC, typename TO>
static int launch_transform(rr_basis *b, const void *dX, int64_t N, int64_t ldx, void *dPhi, int64_t ldphi) {
    rr_ctx *c = b->ctx;
    const TC *Ws = (sizeof(TC) == 4) ? (const TC *)b->dWs32 : (const TC *)b->dWs64;
    const TC scale = (TC)(1.0 / sqrt((double)b->n));
    if (b->large) return large_features<TX, TC, TO>(b, (const TX *)dX, nullptr, N, N, ldx, (TO *)dPhi, ldphi, nullptr);
    if constexpr (sizeof(TC) == 4) {  // f32 arithmetic: whole 32-row tiles with the projection on MFMA, the rest below
        const int64_t full = N / 32 * 32;
        if (full > 0 && rr_features_mfma_launch<TX, TO>(b, (const TX *)dX, nullptr, full, full, ldx, (TO *)dPhi, ldphi,
                                                        nullptr, (float)scale)) {
            RR_CHECK_HIP(hipGetLastError());
            if (full == N) return RR_OK;
            dX = (const TX *)dX + full * ldx;
            dPhi = (TO *)dPhi + full * ldphi;
            N -= full;
        }
    } else {  // f64 arithmetic: whole 16-row tiles with the projection on the f64 MFMA
        const int64_t full = N / 16 * 16;
        if (full > 0 && rr_features_mfma64_launch<TX, TO>(b, (const TX *)dX, nullptr, full, full, ldx, (TO *)dPhi, ldphi,
                                                          nullptr, (double)scale)) {
            RR_CHECK_HIP(hipGetLastError());
            if (full == N) return RR_OK;
            dX = (const TX *)dX + full * ldx;
            dPhi = (TO *)dPhi + full * ldphi;
            N -= full;
        }
    }
    const int fblocks = (b->n + 255) / 256;
    // enough row blocks to fill the chip a few times over, >= 16 rows each
    int64_t rpb = (N * fblocks + (int64_t)c->num_cu * 16 - 1) / ((int64_t)c->num_cu * 16);
    if (rpb < 16) rpb = 16;
    if (rpb > 1024) rpb = 1024;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    dim3 grid(fblocks, (unsigned)((N + rpb - 1) / rpb));
#define RR_LT(DM)                                                                                  \
    hipLaunchKernelGGL((rr_rff_transform_kernel<DM, TX, TC, TO>), grid, dim3(256), 0, c->stream,   \
                       (const TX *)dX, N, ldx, Ws, b->n, b->npad, (TO *)dPhi, ldphi, scale, (int)rpb)
    switch (b->dpad) {
        case 8: RR_LT(8); break;
        case 16: RR_LT(16); break;
        case 32: RR_LT(32); break;
        case 64: RR_LT(64); break;
        case 128: RR_LT(128); break;
        default: rr_set_error("transform: d=%d > 128 is not supported yet", b->d); return RR_ERR_UNSUPPORTED;
    }
#undef RR_LT
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

static int ensure_zbuf(rr_basis *b, size_t bytes);

template <typename TX, typename TC, typename TO>
static int launch_grad(rr_basis *b, const void *dX, int64_t N, int64_t ldx, void *dOut, int nout) {
    rr_ctx *c = b->ctx;
    const TC *Ws = (sizeof(TC) == 4) ? (const TC *)b->dWs32 : (const TC *)b->dWs64;
    const TC *gf = (sizeof(TC) == 4) ? (const TC *)b->dgfac32 : (const TC *)b->dgfac64;
    const TC scale = (TC)(1.0 / sqrt((double)b->n));
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = 16;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    dim3 grid(fblocks, (unsigned)((N + rpb - 1) / rpb));
    if (b->large) {  // features first (compute dtype, in the Gram scratch), then the product rule per (row, frequency, i)
        const int64_t ldp = 2 * (int64_t)b->n;
        int rc = ensure_zbuf(b, (size_t)N * ldp * sizeof(TC));
        if (rc == RR_OK) rc = large_features<TX, TC, TC>(b, (const TX *)dX, nullptr, N, N, ldx, (TC *)b->zbuf, ldp, nullptr);
        if (rc != RR_OK) return rc;
        hipLaunchKernelGGL((rr_rff_grad_p_kernel<TX, TC, TO>), grid, dim3(256), 0, c->stream, (const TX *)dX, N, ldx, Ws, gf,
                           (const TC *)b->zbuf, ldp, b->n, b->npad, nout, (TO *)dOut, (int)rpb);
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
#define RR_LG(DM)                                                                                \
    hipLaunchKernelGGL((rr_rff_grad_kernel<DM, TX, TC, TO>), grid, dim3(256), 0, c->stream,       \
                       (const TX *)dX, N, ldx, Ws, gf, b->n, b->npad, nout, (TO *)dOut, scale,    \
                       (int)rpb)
    switch (b->dpad) {
        case 8: RR_LG(8); break;
        case 16: RR_LG(16); break;
        case 32: RR_LG(32); break;
        case 64: RR_LG(64); break;
        case 128: RR_LG(128); break;
        default: rr_set_error("grad: d=%d > 128 is not supported yet", b->d); return RR_ERR_UNSUPPORTED;
    }
#undef RR_LG
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// dispatch on (x dtype, compute dtype, out dtype)
#define RR_DISPATCH3(FN, xdt, cdt, odt, ...)                                              \
    do {                                                                                  \
        const int key = (xdt) * 4 + (cdt) * 2 + (odt);                                    \
        switch (key) {                                                                    \
            case 0: return FN<float, float, float>(__VA_ARGS__);                          \
            case 1: return FN<float, float, double>(__VA_ARGS__);                         \
            case 2: return FN<float, double, float>(__VA_ARGS__);                         \
            case 3: return FN<float, double, double>(__VA_ARGS__);                        \
            case 4: return FN<double, float, float>(__VA_ARGS__);                         \
            case 5: return FN<double, float, double>(__VA_ARGS__);                        \
            case 6: return FN<double, double, float>(__VA_ARGS__);                        \
            case 7: return FN<double, double, double>(__VA_ARGS__);                       \
        }                                                                                 \
        rr_set_error("bad dtype combination");                                            \
        return RR_ERR_INVALID;                                                            \
    } while (0)

static bool dtype_ok(int t) { return t == RR_F32 || t == RR_F64; }
static size_t dtype_size(int t) { return t == RR_F32 ? 4 : 8; }

static int transform_dev_impl(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                              void *dPhi, int out_dtype, int64_t ldphi) {
    RR_DISPATCH3(launch_transform, x_dtype, (b->phase64 ? (int)RR_F64 : b->compute), out_dtype, b, dX, N, ldx, dPhi, ldphi);
}

static int grad_dev_impl(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, void *dOut,
                         int out_dtype, int nout) {
    RR_DISPATCH3(launch_grad, x_dtype, (b->phase64 ? (int)RR_F64 : b->compute), out_dtype, b, dX, N, ldx, dOut, nout);
}

// Feature scratch: grow-only, owned by the basis (freed in rr_basis_destroy).
static int ensure_zbuf(rr_basis *b, size_t bytes) {
    if (b->zbuf_bytes >= bytes) return RR_OK;
    if (b->zbuf) {
        RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
        (void)hipFree(b->zbuf);
        b->zbuf = nullptr;
        b->zbuf_bytes = 0;
    }
    hipError_t e = hipMalloc((void **)&b->zbuf, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("gram: could not allocate %zu bytes of feature scratch", bytes);
        return RR_ERR_OOM;
    }
    b->zbuf_bytes = bytes;
    return RR_OK;
}

// Greedy XCD-aware tile order: dispatch position q goes to XCD q % 8, so XCD x receives positions
// x, x+8, ...; fill each XCD's quota with tiles that add the fewest new column blocks to the set it
// already reads.  Only used when ntiles is a multiple of 8 (equal quotas keep the load balanced).
static void rr_build_tile_map_uncached(int nb, int od, int nxcd, std::vector<int> &map);

void rr_build_tile_map(int nb, int od, int nxcd, std::vector<int> &map) {
    static std::mutex mu;
    static std::map<int, std::vector<int>> cache;  // contexts of one process (device groups, alternating bases) share the maps
    std::lock_guard<std::mutex> lk(mu);
    const int key = (nb * 2 + od) * 64 + nxcd;
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<int> m;
        rr_build_tile_map_uncached(nb, od, nxcd, m);
        it = cache.emplace(key, std::move(m)).first;
    }
    map = it->second;
}

static void rr_build_tile_map_uncached(int nb, int od, int nxcd, std::vector<int> &map) {
    const int ntiles = od ? nb * (nb - 1) / 2 : nb * (nb + 1) / 2;
    map.assign(ntiles, 0);
    std::vector<int> ta(ntiles), tb(ntiles);
    for (int a = 0, t = 0; a < nb; ++a)
        for (int b2 = a + od; b2 < nb; ++b2, ++t) { ta[t] = a; tb[t] = b2; }
    std::vector<char> used(ntiles, 0);
    const int quota = ntiles / nxcd;
    std::vector<int> part(ntiles, 0);  // tile -> XCD
    // greedy start: every XCD in turn takes the tiles that add the fewest new column blocks to what it already reads
    for (int x = 0; x < nxcd; ++x) {
        std::vector<char> have(nb, 0);
        for (int s = 0; s < quota; ++s) {
            int best = -1, bestcost = 1 << 30;
            for (int t = 0; t < ntiles; ++t) {
                if (used[t]) continue;
                const int cost = (have[ta[t]] ? 0 : 1) + ((have[tb[t]] || tb[t] == ta[t]) ? 0 : 1);
                if (cost < bestcost) { bestcost = cost; best = t; }
            }
            used[best] = 1;
            have[ta[best]] = have[tb[best]] = 1;
            part[best] = x;
        }
    }
    // ... improved by annealing over tile SWAPS between XCDs (round 5; RR_GRAM_TILE_ANNEAL=0: the greedy map).  The cost is
    // what the eight L2s read per row split: the number of distinct column blocks of P each XCD's tiles touch, summed --
    // 65 for the greedy partition of F = 4096's 120 off-diagonal tiles, 57 after annealing (a (16, 6, 1) design, 48, does not
    // exist); 141 -> 130 at 32 column blocks.  Deterministic (fixed seed); made once per process and tile count (cache below).
    static const bool anneal = !(getenv("RR_GRAM_TILE_ANNEAL") && atoi(getenv("RR_GRAM_TILE_ANNEAL")) == 0);
    if (anneal && nxcd > 1 && ntiles >= 2 * nxcd && nb <= 128) {
        std::vector<int> cnt((size_t)nxcd * nb, 0);
        auto add = [&](int x, int t, int d) {
            cnt[(size_t)x * nb + ta[t]] += d;
            if (tb[t] != ta[t]) cnt[(size_t)x * nb + tb[t]] += d;
        };
        for (int t = 0; t < ntiles; ++t) add(part[t], t, 1);
        auto cost_of = [&]() {
            int c2 = 0;
            for (size_t i = 0; i < cnt.size(); ++i) c2 += cnt[i] > 0;
            return c2;
        };
        // blocks of XCD x in use if tile `out` leaves and tile `in` joins
        auto delta_one = [&](int x, int out, int in) {
            int before = 0, after = 0;
            const int bl[4] = {ta[out], tb[out], ta[in], tb[in]};
            for (int i = 0; i < 4; ++i) {
                bool seen = false;
                for (int j = 0; j < i; ++j) seen = seen || bl[j] == bl[i];
                if (seen) continue;
                const int b = bl[i];
                const int c0 = cnt[(size_t)x * nb + b];
                const int c1 = c0 - ((ta[out] == b) || (tb[out] == b) ? 1 : 0) + ((ta[in] == b) || (tb[in] == b) ? 1 : 0);
                before += c0 > 0;
                after += c1 > 0;
            }
            return after - before;
        };
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        auto next = [&]() {
            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
            return rng;
        };
        int cur = cost_of(), best = cur;
        std::vector<int> best_part = part;
        const int64_t iters = std::min<int64_t>((int64_t)20000 * ntiles, 6000000);  // 0.1 s at F = 4096, 0.3 s at most
        for (int64_t it = 0; it < iters; ++it) {
            const int t1 = (int)(next() % (uint64_t)ntiles), t2 = (int)(next() % (uint64_t)ntiles);
            const int x = part[t1], y = part[t2];
            if (x == y) continue;
            const int d = delta_one(x, t1, t2) + delta_one(y, t2, t1);
            const double temp = 0.4 * (1.0 - (double)it / (double)iters) + 0.05;
            if (d <= 0 || (double)(next() >> 11) / 9007199254740992.0 < exp(-(double)d / temp)) {
                add(x, t1, -1); add(y, t2, -1); add(x, t2, 1); add(y, t1, 1);
                part[t1] = y; part[t2] = x;
                cur += d;
                if (cur < best) { best = cur; best_part = part; }
            }
        }
        part = best_part;
    }
    // launch order: workgroup s * nxcd + x runs on XCD x; an XCD's tiles in (a, b) order, so that neighbours in time share blocks
    std::vector<int> fill(nxcd, 0);
    for (int t = 0; t < ntiles; ++t) map[(size_t)fill[part[t]]++ * nxcd + part[t]] = t;
}

// G(upper) += P^T P for a zero-padded f32 feature matrix (rows % 32 == 0, ldp % 256 == 0).

int rr_launch_syrk_f32(rr_ctx *c, const float *P, int64_t rows, int64_t ldp, int F, double *dG, hipEvent_t mid,
                       double *bcol = nullptr) {
    if (c->gram_engine != 0) {
        RR_REQUIRE(bcol == nullptr, "gram: the rider column needs the f32 engine");
        RR_REQUIRE(!c->deterministic, "gram: deterministic mode (rr_set_deterministic) needs the f32 engine, not a split 16-bit one");
        return rr_launch_syrk_bf16(c, c->gram_engine, P, nullptr, rows, ldp, F, dG, mid);
    }
    RR_REQUIRE(bcol == nullptr || F < ldp, "gram: no pad column for the rider");
    // small feature counts over few rows: 128 x 128 tiles (rr_syrk_f32_small_kernel); RR_SYRK_SMALL=0: never (A/B runs)
    static const bool small_off = getenv("RR_SYRK_SMALL") != nullptr && atoi(getenv("RR_SYRK_SMALL")) == 0;
    // (where it wins, tools/small_gram_bench.py: while the 256 x 256 tiles x 1024-row splits cannot give every CU a workgroup --
    // F = 512: up to ~85 000 rows (10 000 rows: 0.148 against 0.200 ms per pass), F = 1024: up to ~26 000; above that the big
    // tiles' LDS-DMA pipeline is 2-3x faster per flop than this kernel's register-staged loads)
    const int64_t big_tiles = (ldp / GR_TC) * (ldp / GR_TC + 1) / 2;
    if (!small_off && ldp <= 1024 && big_tiles * ((rows + 1023) / 1024) < c->num_cu && rows % GS_KB == 0 && !getenv("RR_GRAM_ABLATE")) {
        const int nbs = (int)(ldp / GS_TC), nt = nbs * (nbs + 1) / 2;
        // K-splits: enough workgroups for every CU (two fit one), at least 128 rows and at most 32768 each
        int64_t ns = std::max<int64_t>((2 * c->num_cu + nt - 1) / nt, (rows + 32767) / 32768);
        ns = std::min<int64_t>(ns, std::max<int64_t>(rows / 128, 1));
        const int64_t rps_s = ((rows + ns - 1) / ns + GS_KB - 1) / GS_KB * GS_KB;
        ns = (rows + rps_s - 1) / rps_s;
        SyrkArgs a;
        a.P = P; a.rows = rows; a.ldp = ldp; a.F = F; a.nb = nbs; a.ntiles = nt; a.rows_per_split = rps_s; a.G = dG;
        a.bcol = bcol; a.tile_map = nullptr; a.offdiag_only = 0; a.ablate = 0;
        if (c->deterministic) {
            void *slabs = nullptr;
            int rc = rr_det_scratch(c, (size_t)ns * (size_t)ldp * (size_t)ldp * sizeof(float), &slabs);
            if (rc != RR_OK) return rc;
            a.part = (float *)slabs;
            a.part_stride = ldp * ldp;
        }
        hipLaunchKernelGGL(rr_syrk_f32_small_kernel, dim3((unsigned)(ns * nt)), dim3(GS_THREADS), 0, c->stream, a);
        if (mid) RR_CHECK_HIP(hipEventRecord(mid, c->stream));
        if (a.part)
            hipLaunchKernelGGL(rr_syrk_det_reduce_kernel<float>, dim3((unsigned)((F + 1 + 255) / 256), (unsigned)F), dim3(256), 0,
                               c->stream, a.part, a.part_stride, ldp, F, (int)ns, dG, bcol);
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    const int nb_all = (int)(ldp / GR_TC);
    const int od = (nb_all >= 2 && !getenv("RR_SYRK_NO_DIAG_KERNEL")) ? 1 : 0;  // diagonal tiles in their own kernel
    // ragged last block (<= 192 valid columns): its off-diagonal tiles go to rr_syrk_f32_ragged_kernel, and the main
    // kernel enumerates the tiles among the first nb_all - 1 blocks only
    const int w_last = F + (bcol ? 1 : 0) - GR_TC * (nb_all - 1);
    const int rg = (od && nb_all >= 3 && w_last <= 192 && !getenv("RR_SYRK_NO_RAGGED_KERNEL")) ? 1 : 0;
    const int nb = nb_all - rg;
    const int ntiles = od ? nb * (nb - 1) / 2 : nb * (nb + 1) / 2;
    const int nxcd = 8;
    const bool use_map = (ntiles % nxcd == 0) && !getenv("RR_GRAM_NO_TILE_MAP");
    if (use_map && c->tile_map_nb != nb * 2 + od) {
        std::vector<int> map;
        rr_build_tile_map(nb, od, nxcd, map);
        if (c->tile_map) (void)hipFree(c->tile_map);
        c->tile_map = nullptr;
        RR_CHECK_HIP(hipMalloc((void **)&c->tile_map, map.size() * sizeof(int)));
        RR_CHECK_HIP(hipMemcpy(c->tile_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        c->tile_map_nb = nb * 2 + od;
    }
    // K-splits: f32 accumulation over <= 32768 rows per split.  Workgroups of one kernel all cost the same, so
    // each kernel gets the split count that makes ITS workgroup count (close to) a multiple of the CU count:
    // exactly for the off-diagonal kernel (nsplit = k * CUs / gcd(CUs, ntiles), k minimal), by search for the
    // diagonal one (nb workgroups per split; an odd nb would otherwise force CUs splits on both).
    const int64_t total_tiles = (int64_t)nb_all * (nb_all + 1) / 2;
    auto gcd64 = [](int64_t x, int64_t y) { while (y) { const int64_t u = x % y; x = y; y = u; } return x; };
    const int64_t min_splits = (rows + 32767) / 32768;
    auto rows_per = [&](int64_t ns) { return ((rows + ns - 1) / ns + GR_KB - 1) / GR_KB * GR_KB; };
    const int64_t unit = ntiles > 0 ? c->num_cu / gcd64(c->num_cu, ntiles) : 1;
    int64_t nsplit = (min_splits + unit - 1) / unit * unit;
    // small inputs: just cover the rows, >= 1024 per split -- unless that leaves most CUs without a workgroup (few tiles:
    // config 1's F = 512 is ONE off-diagonal tile and two diagonal ones, 10 splits of its 10 000 rows = 10 workgroups of
    // 32 k-blocks at one CU's rate each): then down to 256 rows per split, until the tiles x splits fill the CUs
    // (config 1's `_elbo`: 1.42 -> 1.13 ms)
    auto floor_rows = [&](int64_t tiles_per_split) {
        return tiles_per_split * ((rows + 1023) / 1024) < c->num_cu ? (int64_t)256 : (int64_t)1024;
    };
    auto small_split = [&](int64_t tiles_per_split) {
        const int64_t fr = floor_rows(tiles_per_split);
        if (fr == 1024) return (rows + 1023) / 1024;
        const int64_t want = (c->num_cu + tiles_per_split - 1) / std::max<int64_t>(tiles_per_split, 1);
        return std::min((rows + fr - 1) / fr, std::max((rows + 1023) / 1024, want));
    };
    if (rows / nsplit < 1024) {
        nsplit = small_split(std::max(ntiles, 1));
        // (round 6) ... and among the split counts around it, the one whose workgroups finish soonest: time ~ rounds x (rows per
        // split + a workgroup's prologue and 64k-entry atomic epilogue, ~96 rows' worth).  F = 1024, N = 44 484 (the reference's
        // SARCOS shape): 44 splits x 6 tiles = 264 workgroups were TWO rounds on 256 CUs, the second with 8 workgroups.
        static const bool no_search = getenv("RR_SYRK_SPLIT_SEARCH") != nullptr && atoi(getenv("RR_SYRK_SPLIT_SEARCH")) == 0;
        if (ntiles > 0 && !no_search) {
            double best_cost = 1e300;
            int64_t best_ns = nsplit;
            const int64_t ns_hi = std::max<int64_t>(rows / 256, 1), ns_lo = std::max<int64_t>(rows / 32768, 1);
            for (int64_t ns = ns_lo; ns <= ns_hi && ns <= ns_lo + 4096; ++ns) {
                const int64_t rp = rows_per(ns), n2 = (rows + rp - 1) / rp;
                const int64_t rounds = ((int64_t)ntiles * n2 + c->num_cu - 1) / c->num_cu;
                const double cost = (double)rounds * ((double)rp + 96.0);
                if (cost < best_cost - 1e-9) {
                    best_cost = cost;
                    best_ns = n2;
                }
            }
            nsplit = best_ns;
        }
    }
    if (nsplit < 1) nsplit = 1;
    int64_t rps = rows_per(nsplit);
    int64_t nsplit_d = nsplit, rps_d = rps;
    if (od) {
        double best = -1.0;
        for (int64_t ns = min_splits; ns < min_splits + 96; ++ns) {
            if (rows / ns < floor_rows(nb_all) && ns > 1) break;
            const int64_t wg = ns * nb_all, rounds = (wg + c->num_cu - 1) / c->num_cu;
            const double eff = (double)wg / (double)(rounds * c->num_cu);
            if (eff > best + 1e-9) {
                best = eff;
                nsplit_d = ns;
            }
        }
        rps_d = rows_per(nsplit_d);
        if (rows % 64 == 0 && getenv("RR_SYRK_DIAG_KB") != nullptr && atoi(getenv("RR_SYRK_DIAG_KB")) == 64)
            rps_d = (rps_d + 63) / 64 * 64;  // whole 64-row k-blocks for rr_syrk_f32_diag16_kernel<64>
    }
    int64_t nsplit_r = nsplit, rps_r = rps;
    if (rg) {  // nb_all - 1 equal-cost workgroups per split: the split count whose workgroups fill whole rounds best
        double best = -1.0;
        for (int64_t ns = min_splits; ns < min_splits + 96; ++ns) {
            if (rows / ns < floor_rows(nb_all - 1) && ns > 1) break;
            const int64_t wg = ns * (nb_all - 1), rounds = (wg + c->num_cu - 1) / c->num_cu;
            const double eff = (double)wg / (double)(rounds * c->num_cu);
            if (eff > best + 1e-9) {
                best = eff;
                nsplit_r = ns;
            }
        }
        rps_r = rows_per(nsplit_r);
    }
    const char *renv = getenv("RR_GRAM_ROWS_PER_SPLIT");
    if (renv && atoll(renv) >= GR_KB) rps = rps_d = rps_r = (atoll(renv) / GR_KB) * GR_KB;
    if (c->deterministic) rps_d = rps_r = rps;  // one slab per K-split, shared by the three kernels
    nsplit = (rows + rps - 1) / rps;
    nsplit_d = (rows + rps_d - 1) / rps_d;
    nsplit_r = (rows + rps_r - 1) / rps_r;
    RR_REQUIRE(nsplit * total_tiles < (int64_t)1 << 31 && nsplit_d * nb_all < (int64_t)1 << 31 &&
                   nsplit_r * nb_all < (int64_t)1 << 31, "gram: grid too large");
    SyrkArgs a;
    a.P = P; a.rows = rows; a.ldp = ldp; a.F = F; a.nb = nb; a.ntiles = ntiles; a.rows_per_split = rps; a.G = dG;
    a.bcol = bcol;
    a.tile_map = use_map ? c->tile_map : nullptr;
    a.offdiag_only = od;
    a.ablate = getenv("RR_GRAM_ABLATE") ? atoi(getenv("RR_GRAM_ABLATE")) : 0;
    a.spread = rr_dma_spread_env();
    if (c->deterministic) {
        void *slabs = nullptr;
        int rc = rr_det_scratch(c, (size_t)nsplit * (size_t)ldp * (size_t)ldp * sizeof(float), &slabs);
        if (rc != RR_OK) return rc;
        a.part = (float *)slabs;
        a.part_stride = ldp * ldp;
    }
    // RR_SYRK_STAGGER = 0 / 1 / 2: the flat LDS-DMA of rounds 1-2 / the same staggered / buffer-descriptor requests right after
    // the barrier (A/B runs: 217.8 / 245 / 214.4 ms per 2M-row launch against 211.1 ms for the default, 3)
    static const int syrk_mode = getenv("RR_SYRK_STAGGER") ? atoi(getenv("RR_SYRK_STAGGER")) : 3;
    static const bool merge_diag = getenv("RR_SYRK_MERGE_DIAG") != nullptr && atoi(getenv("RR_SYRK_MERGE_DIAG")) != 0;
    if (merge_diag && od && !rg && ntiles > 0 && syrk_mode == 3 && !a.ablate && !getenv("RR_SYRK_NO_DIAG16")) {
        SyrkArgs ad = a;
        ad.nb = nb_all;
        ad.rows_per_split = rps_d;
        const unsigned n_off = (unsigned)(nsplit * ntiles);
        hipLaunchKernelGGL(rr_syrk_f32_merged_kernel, dim3(n_off + (unsigned)(nsplit_d * nb_all)), dim3(GR_THREADS), 0, c->stream, a, ad, n_off);
        if (mid) RR_CHECK_HIP(hipEventRecord(mid, c->stream));
        if (a.part)
            hipLaunchKernelGGL(rr_syrk_det_reduce_kernel<float>, dim3((unsigned)((F + 1 + 255) / 256), (unsigned)F), dim3(256), 0,
                               c->stream, a.part, a.part_stride, ldp, F, (int)nsplit, dG, bcol);
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    if (ntiles > 0) {
        if (syrk_mode == 1 && !a.ablate)
            hipLaunchKernelGGL(rr_syrk_f32_flatstag_kernel, dim3((unsigned)(nsplit * ntiles)), dim3(GR_THREADS), 0, c->stream, a);
        else if (syrk_mode == 2 && !a.ablate)
            hipLaunchKernelGGL(rr_syrk_f32_buf_kernel, dim3((unsigned)(nsplit * ntiles)), dim3(GR_THREADS), 0, c->stream, a);
        else if (syrk_mode == 0 || a.ablate)  // (the ablation switches live in the flat variant)
            hipLaunchKernelGGL(rr_syrk_f32_flat_kernel, dim3((unsigned)(nsplit * ntiles)), dim3(GR_THREADS), 0, c->stream, a);
        else
            hipLaunchKernelGGL(rr_syrk_f32_kernel, dim3((unsigned)(nsplit * ntiles)), dim3(GR_THREADS), 0, c->stream, a);
    }
    if (rg) {
        SyrkArgs ar = a;
        ar.nb = nb_all;
        ar.rows_per_split = rps_r;
        hipLaunchKernelGGL(rr_syrk_f32_ragged_kernel, dim3((unsigned)(nsplit_r * (nb_all - 1))), dim3(GR_THREADS), 0, c->stream, ar);
    }
    if (mid) RR_CHECK_HIP(hipEventRecord(mid, c->stream));
    if (od) {
        SyrkArgs ad = a;
        ad.nb = nb_all;
        ad.rows_per_split = rps_d;
        // diagonal blocks as 16x16 sub-blocks (round 3); RR_SYRK_NO_DIAG16=1: the whole-block kernel of round 2 (A/B runs)
        static const bool no_diag16 = getenv("RR_SYRK_NO_DIAG16") != nullptr;
        // RR_SYRK_DIAG_KB=64 (A/B runs): 64-row k-blocks when every K-split holds whole ones.  Measured in round 3 and not
        // adopted: 87.7-88.1 vs 86.3-86.9 ms per 10M rows -- half the barriers buy nothing (DESIGN Appendix A.2)
        static const bool kb64 = getenv("RR_SYRK_DIAG_KB") != nullptr && atoi(getenv("RR_SYRK_DIAG_KB")) == 64;
        if (no_diag16)
            hipLaunchKernelGGL(rr_syrk_f32_diag_kernel, dim3((unsigned)(nsplit_d * nb_all)), dim3(GR_THREADS), 0, c->stream, ad);
        else if (kb64 && rows % 64 == 0 && ad.rows_per_split % 64 == 0)
            hipLaunchKernelGGL(rr_syrk_f32_diag16_kernel<64>, dim3((unsigned)(nsplit_d * nb_all)), dim3(GR_THREADS), 0, c->stream, ad);
        else
            hipLaunchKernelGGL(rr_syrk_f32_diag16_kernel<32>, dim3((unsigned)(nsplit_d * nb_all)), dim3(GR_THREADS), 0, c->stream, ad);
    }
    if (a.part)
        hipLaunchKernelGGL(rr_syrk_det_reduce_kernel<float>, dim3((unsigned)((F + 1 + 255) / 256), (unsigned)F), dim3(256), 0,
                           c->stream, a.part, a.part_stride, ldp, F, (int)nsplit, dG, bcol);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_launch_syrk_f64(rr_ctx *c, const double *P, int64_t rows, int64_t ldp, int F, double *dG, int lower_tri = 0) {
    const int nb = (int)(ldp / G64_TC);
    const int od = (nb >= 2 && !getenv("RR_SYRK_NO_DIAG_KERNEL")) ? 1 : 0;  // diagonal tiles in their own kernel
    const int ntiles = od ? nb * (nb - 1) / 2 : nb * (nb + 1) / 2;
    const int64_t slots = (int64_t)c->num_cu * 2;  // two workgroups per CU
    // equal-cost workgroups: make their count a multiple of the slots (no partial last round) while a split keeps
    // >= 2048 rows; otherwise ~4 rounds
    int64_t g = slots, t = ntiles;
    while (t) { const int64_t u = g % t; g = t; t = u; }
    const int64_t unit = slots / g;
    // (the posterior's C = Y^T Y at small F -- a square, triangular operand of <= 2048 rows: a few tiles whose k-loops are the
    // whole launch -- splits down to 128 rows: at F = 1024 two splits of 512 rows left 56 workgroups walking up to 32
    // k-blocks each, 69 + 40 us for a third of a GFLOP)
    static const bool no_fine = getenv("RR_SYRK64_FINE_SPLIT") != nullptr && atoi(getenv("RR_SYRK64_FINE_SPLIT")) == 0;  // A/B runs
    const int64_t min_rows = (lower_tri && rows == ldp && rows <= 2048 && !no_fine) ? 128 : 512;
    int64_t nsplit = unit;
    if (rows / nsplit < 2048) nsplit = (slots * 4 + ntiles - 1) / ntiles;
    if (rows / nsplit < min_rows) nsplit = (rows + min_rows - 1) / min_rows;
    if (nsplit < 1) nsplit = 1;
    const int64_t rps = ((rows + nsplit - 1) / nsplit + G64_KB - 1) / G64_KB * G64_KB;
    nsplit = (rows + rps - 1) / rps;
    RR_REQUIRE(nsplit * ntiles < (int64_t)1 << 31, "gram: grid too large");
    Syrk64Args a;
    a.P = P; a.rows = rows; a.ldp = ldp; a.F = F; a.nb = nb; a.ntiles = ntiles; a.rows_per_split = rps; a.G = dG;
    a.offdiag_only = od;
    static const bool no_tri = getenv("RR_SYRK64_TRI") != nullptr && atoi(getenv("RR_SYRK64_TRI")) == 0;  // A/B runs
    a.lower_tri = lower_tri && !no_tri && rows == ldp ? 1 : 0;
    if (c->deterministic) {
        void *slabs = nullptr;
        int rc = rr_det_scratch(c, (size_t)nsplit * (size_t)ldp * (size_t)ldp * sizeof(double), &slabs);
        if (rc != RR_OK) return rc;
        a.part = (double *)slabs;
        a.part_stride = ldp * ldp;
    }
    if (ntiles > 0)
        hipLaunchKernelGGL(rr_syrk_f64_kernel, dim3((unsigned)(nsplit * ntiles)), dim3(G64_THREADS), 0, c->stream, a);
    if (od) {
        // the diagonal kernel's own K-split: nb equal-cost workgroups per split, up to 4 resident per CU (36 KiB of LDS
        // each); pick the split count whose workgroup count fills whole rounds best while a split keeps >= 512 rows
        const int64_t dslots = (int64_t)c->num_cu * 4;
        int64_t best_ns = 1;
        double best = -1.0;
        for (int64_t ns = 1; ns <= 4 * dslots / nb + 1; ++ns) {
            if (rows / ns < min_rows && ns > 1) break;
            const int64_t wg = ns * nb, rounds = (wg + dslots - 1) / dslots;
            const double eff = (double)wg / (double)(rounds * dslots);
            if (eff > best + 1e-9) {
                best = eff;
                best_ns = ns;
            }
        }
        Syrk64Args ad = a;
        ad.rows_per_split = ((rows + best_ns - 1) / best_ns + G64_KB - 1) / G64_KB * G64_KB;
        if (c->deterministic) ad.rows_per_split = rps;  // one slab per K-split, shared by both kernels
        const int64_t nsd = (rows + ad.rows_per_split - 1) / ad.rows_per_split;
        hipLaunchKernelGGL(rr_syrk_f64_diag_kernel, dim3((unsigned)(nsd * nb)), dim3(G64_THREADS), 0, c->stream, ad);
    }
    if (a.part)
        hipLaunchKernelGGL(rr_syrk_det_reduce_kernel<double>, dim3((unsigned)((F + 255) / 256), (unsigned)F), dim3(256), 0,
                           c->stream, a.part, a.part_stride, ldp, F, (int)nsplit, dG, (double *)nullptr);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_launch_gemm_tn_f64(rr_ctx *c, const double *A, int64_t lda, const double *B, int64_t ldb, double *D, int64_t ldd,
                          int64_t K, int64_t M, int64_t N, int subtract, int upper_only) {
    Gemm64Args g;
    g.A = A; g.B = B; g.D = D; g.lda = lda; g.ldb = ldb; g.ldd = ldd; g.K = (int)K; g.ntb = (int)(N / G64_TC);
    g.subtract = subtract;
    g.upper_only = upper_only;
    // RR_GEMM64_K128=0: the tile kernel for these shapes too (A/B runs)
    static const bool no_k128 = getenv("RR_GEMM64_K128") != nullptr && atoi(getenv("RR_GEMM64_K128")) == 0;
    if (!no_k128 && K == 128 && M == 128 && N % 32 == 0 && (upper_only == 0 || (upper_only == 1 && N == 128))) {
        // (upper_only = 1 on a single tile: the whole tile, as the tile kernel does)
        hipLaunchKernelGGL(rr_gemm_tn_f64_k128_kernel, dim3((unsigned)(N / 32)), dim3(G64_THREADS), 0, c->stream, g);
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    hipLaunchKernelGGL(rr_gemm_tn_f64_kernel, dim3((unsigned)((M / G64_TC) * g.ntb)), dim3(G64_THREADS), 0, c->stream, g);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// TC = float: f32 features + rr_syrk_f32_kernel;  TC = double: f64 features + rr_syrk_f64_kernel
template <typename TX, typename TC>
static int launch_gram(rr_basis *b, const void *dX, const void *dy, int64_t N, int64_t ldx, double *dG,
                       double *db) {
    rr_ctx *c = b->ctx;
    constexpr bool F32 = sizeof(TC) == 4;
    constexpr int TCOLS = F32 ? GR_TC : G64_TC;
    constexpr int KB = F32 ? GR_KB : G64_KB;
    const int F = 2 * b->n;
    const int64_t ldp = ((int64_t)F + TCOLS - 1) / TCOLS * TCOLS;
    const TC scale = (TC)(1.0 / sqrt((double)b->n));
    const TC *Ws = F32 ? (const TC *)b->dWs32 : (const TC *)b->dWs64;
    // row chunks: feature scratch of at most ~32 GiB (or RR_GRAM_CHUNK_ROWS), multiple of KB rows
    int64_t chunk = (int64_t)(((size_t)32 << 30) / ((size_t)ldp * sizeof(TC)));
    const char *cenv = getenv("RR_GRAM_CHUNK_ROWS");
    if (cenv && atoll(cenv) >= KB) chunk = atoll(cenv);
    else if (N > chunk) {
        // several chunks: equal ones (10M rows at F = 4096: 5 x 2 000 000 instead of 4 x 2 097 152 + 1 611 392).  Besides the
        // balance, 2^21-row launches put the K-splits of the SYRK exactly 2^15 rows = 512 MiB apart, and the concurrently
        // running workgroups of different splits then alias in the L2: 107 instead of 83 KB/row of L2->fabric fetch
        // (profiles/r03_headline; no change in time, the kernel is MFMA-bound, but 30 % more fabric traffic for nothing)
        const int64_t nchunks = (N + chunk - 1) / chunk;
        chunk = (N + nchunks - 1) / nchunks;
    }
    if (chunk > N) chunk = N;
    // split-bf16 engine with the MFMA feature kernel: the features are produced directly in the SYRK's K-blocked
    // bf16 hi/lo layout (same 4 bytes per value), whole 64-row groups
    RR_REQUIRE(!(F32 && c->deterministic && c->gram_engine != 0),
               "gram: deterministic mode (rr_set_deterministic) needs the f32 engine, not a split 16-bit one");
    const bool fused_pb = F32 && !b->phase64 && c->gram_engine != 0 && rr_features_mfma_ok<TX>(b, (const TX *)dX, N, ldx);
    const int KBR = fused_pb ? 64 : KB;
    chunk = (chunk + KBR - 1) / KBR * KBR;
    // RR_GRAM_OVERLAP=1 (opt-in; measured in round 3, profiles/r03_overlap): with more than one chunk, chunk k+1's feature
    // kernel runs on a second stream while chunk k's SYRK runs on the context's, into the other half of a double-buffered
    // scratch.  It does not pay on gfx950: the feature kernel's VALU / MFMA issue comes straight out of the co-resident SYRK
    // waves' MFMA issue (no co-execution), so the SYRK kernel slows by what the features save (1095 -> 1120 ms
    // per 10M rows against 31 ms of hidden features).  Off by default; never in deterministic mode (single-stream scratch).
    static const bool overlap_on = getenv("RR_GRAM_OVERLAP") != nullptr && atoi(getenv("RR_GRAM_OVERLAP")) != 0;
    const bool overlap = N > chunk && overlap_on && !c->deterministic;
    const int nbuf = overlap ? 2 : 1;
    int rc = ensure_zbuf(b, (size_t)nbuf * (size_t)chunk * ldp * sizeof(TC));
    if (rc != RR_OK) return rc;
    TC *P = (TC *)b->zbuf;
    if (overlap && !c->stream2) {
        RR_CHECK_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        RR_CHECK_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    }
    hipStream_t fstream = overlap ? c->stream2 : c->stream;  // the stream the feature kernels go to
    const hipStream_t main_stream = c->stream;
    if (fused_pb && ldp > F) {
        const int64_t cnt = (nbuf * chunk / 16) * (ldp - F) * 4;
        hipLaunchKernelGGL(rr_zero_padcols_pb_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream,
                           (uintx4 *)P, nbuf * chunk / 16, ldp, F);
        RR_CHECK_HIP(hipGetLastError());
    } else if (ldp > F) {  // pad columns are never written by the feature kernel: zero them once per call
        const int64_t cnt = nbuf * chunk * (ldp - F);
        hipLaunchKernelGGL(rr_zero_padcols_kernel<TC>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream,
                           P, nbuf * chunk, ldp, F);
        RR_CHECK_HIP(hipGetLastError());
    }

    TC *const Pbase = P;
    struct StreamGuard {  // an error return between the two stream switches must not leave the context on the second stream
        rr_ctx *c;
        hipStream_t s;
        ~StreamGuard() { c->stream = s; }
    } guard{c, main_stream};
    for (int64_t r0 = 0; r0 < N; r0 += chunk) {
        const int64_t m = (N - r0 < chunk) ? N - r0 : chunk;
        const int64_t mpad = (m + KBR - 1) / KBR * KBR;
        const TX *Xc = (const TX *)dX + r0 * ldx;
        const TX *yc = dy ? (const TX *)dy + r0 : nullptr;
        const int64_t kchunk = r0 / chunk;
        P = Pbase + (overlap ? (kchunk & 1) * chunk * ldp : 0);
        // five events per chunk bracket the kernels (read back by rr_rff_gram_timings): features begin / end (on the
        // feature stream), SYRK begin / mid / end (on the context's stream)
        const size_t e0 = (size_t)kchunk * 5;
        while (b->events.size() < e0 + 5) {
            hipEvent_t ev;
            RR_CHECK_HIP(hipEventCreate(&ev));
            b->events.push_back(ev);
        }
        if (overlap) {
            // the feature stream starts behind whatever the context's stream holds at entry (zeroed accumulators, the
            // pad columns), and reuses a buffer only after the SYRK that read it (chunk k - 2) has finished
            if (kchunk == 0) {
                RR_CHECK_HIP(hipEventRecord(c->ev_fork, main_stream));
                RR_CHECK_HIP(hipStreamWaitEvent(fstream, c->ev_fork, 0));
            } else if (kchunk >= 2) {
                RR_CHECK_HIP(hipStreamWaitEvent(fstream, b->events[e0 - 10 + 4], 0));
            }
            c->stream = fstream;  // the feature launchers below take the stream from the context (restored by `guard`)
        }
        RR_CHECK_HIP(hipEventRecord(b->events[e0], c->stream));
        // (A) features (+ Phi^T y): MFMA projection for f32 X, else the VALU kernel
        bool done_a = false;
        if constexpr (F32) {
            if (fused_pb) {
                done_a = c->gram_engine == RR_GRAM_FP16X3
                             ? rr_features_mfma_launch<TX, rr_pf_t>(b, Xc, yc, m, mpad, ldx, (rr_pf_t *)P, ldp, db, (float)scale)
                             : rr_features_mfma_launch<TX, rr_pb_t>(b, Xc, yc, m, mpad, ldx, (rr_pb_t *)P, ldp, db, (float)scale);
                if (!done_a) {
                    rr_set_error("gram: internal: split-bf16 feature launch refused");
                    return RR_ERR_INVALID;
                }
            }
        }
        if (done_a) {
        } else if (F32 && b->phase64) {
            if constexpr (F32)
                done_a = rr_features_mfma64_launch<TX, float>(b, Xc, yc, m, mpad, ldx, (float *)P, ldp, db, 1.0 / sqrt((double)b->n));
            if (!done_a) {
                rr_set_error("gram: a float64-phase basis (RR_F32P64) needs 16-byte aligned rows of X");
                return RR_ERR_UNSUPPORTED;
            }
        } else if (b->large) {
            RR_REQUIRE(!c->deterministic || yc == nullptr, "gram: deterministic mode does not cover Xdim > 128");
            rc = large_features<TX, TC, TC>(b, Xc, yc, m, mpad, ldx, P, ldp, db);
            if (rc != RR_OK) return rc;
            done_a = true;
        } else if constexpr (F32) done_a = rr_features_mfma_launch<TX, float>(b, Xc, yc, m, mpad, ldx, (float *)P, ldp, db, (float)scale);
        else done_a = rr_features_mfma64_launch<TX, double>(b, Xc, yc, m, mpad, ldx, (double *)P, ldp, db, (double)scale);
        if (!done_a) {
            const int fblocks = (b->n + 255) / 256;
            int64_t rpb = 256;
            if ((mpad + rpb - 1) / rpb > 65535) rpb = (mpad + 65534) / 65535;
            const dim3 grid(fblocks, (unsigned)((mpad + rpb - 1) / rpb));
#define RR_LPH(DM)                                                                                               \
    do {                                                                                                         \
        if (yc && c->deterministic) {                                                                            \
            void *part = nullptr;                                                                                \
            rc = rr_det_scratch(c, (size_t)grid.y * F * 8, &part);                                               \
            if (rc != RR_OK) return rc;                                                                          \
            hipLaunchKernelGGL((rr_rff_features_kernel<DM, true, TX, TC>), grid, dim3(256), 0, c->stream, Xc, yc, \
                               m, mpad, ldx, Ws, b->n, b->npad, P, ldp, (double *)part, scale, (int)rpb, (int64_t)F); \
            rc = rr_det_reduce(c, (const double *)part, grid.y, F, F, db);                                        \
            if (rc != RR_OK) return rc;                                                                          \
        } else if (yc) hipLaunchKernelGGL((rr_rff_features_kernel<DM, true, TX, TC>), grid, dim3(256), 0, c->stream, Xc, \
                                   yc, m, mpad, ldx, Ws, b->n, b->npad, P, ldp, db, scale, (int)rpb);            \
        else hipLaunchKernelGGL((rr_rff_features_kernel<DM, false, TX, TC>), grid, dim3(256), 0, c->stream, Xc,  \
                                yc, m, mpad, ldx, Ws, b->n, b->npad, P, ldp, db, scale, (int)rpb);               \
    } while (0)
            switch (b->dpad) {
                case 8: RR_LPH(8); break;
                case 16: RR_LPH(16); break;
                case 32: RR_LPH(32); break;
                case 64: RR_LPH(64); break;
                case 128: RR_LPH(128); break;
                default: rr_set_error("gram: d=%d > 128 is not supported yet", b->d); return RR_ERR_UNSUPPORTED;
            }
#undef RR_LPH
        }
        RR_CHECK_HIP(hipGetLastError());
        RR_CHECK_HIP(hipEventRecord(b->events[e0 + 1], c->stream));
        if (overlap) {
            c->stream = main_stream;
            RR_CHECK_HIP(hipStreamWaitEvent(main_stream, b->events[e0 + 1], 0));
        }
        RR_CHECK_HIP(hipEventRecord(b->events[e0 + 2], c->stream));
        // (B) G += P^T P
        if constexpr (F32) {
            if (fused_pb)
                rc = rr_launch_syrk_bf16(c, c->gram_engine, nullptr, (const void *)P, mpad, ldp, F, dG, b->events[e0 + 3],
                                         c->gram_engine == RR_GRAM_FP16X3 ? f16_store_scale((float)scale) : 0.f);
            else
                rc = rr_launch_syrk_f32(c, P, mpad, ldp, F, dG, b->events[e0 + 3]);
        } else {
            rc = rr_launch_syrk_f64(c, P, mpad, ldp, F, dG);
            if (rc == RR_OK) RR_CHECK_HIP(hipEventRecord(b->events[e0 + 3], c->stream));
        }
        if (rc != RR_OK) return rc;
        RR_CHECK_HIP(hipEventRecord(b->events[e0 + 4], c->stream));
        b->events_used = e0 + 5;
    }
    b->gram_kernel = !F32 ? "rr_syrk_f64_kernel" : c->gram_engine == 0 ? "rr_syrk_f32_kernel" : "rr_syrk_b16w4_kernel";
    return RR_OK;
}

// Row-major f32 features of a row block into a caller-provided scratch (used by the second _elbo pass).
int rr_features_rowmajor_f32(rr_basis *b, const void *dX, int x_dtype, int64_t m, int64_t mpad, int64_t ldx,
                             float *P, int64_t ldp, bool zero_pad_cols, float *Pt, int64_t ldt, bool *pt_written, double scale_mult) {
    // scale_mult: 1, or sqrt(1/2) for one of the two blocks of a spectral-mixture component (1 / sqrt(2 n) over its 4 n columns)
    if (pt_written) *pt_written = false;
    rr_ctx *c = b->ctx;
    const int F = 2 * b->n;
    const float scale = (float)(scale_mult / sqrt((double)b->n));
    if (zero_pad_cols && ldp > F) {
        const int64_t cnt = mpad * (ldp - F);
        hipLaunchKernelGGL(rr_zero_padcols_kernel<float>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, P,
                           mpad, ldp, F);
    }
    if (b->phase64) {  // RR_F32P64: phases on the f64 matrix cores, float32 features out; no direct P^T (consumers transpose)
        const double sc = scale_mult / sqrt((double)b->n);
        const bool done = x_dtype == RR_F32
                              ? rr_features_mfma64_launch<float, float>(b, (const float *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, sc)
                              : rr_features_mfma64_launch<double, float>(b, (const double *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, sc);
        if (!done) {
            rr_set_error("features: a float64-phase basis (RR_F32P64) needs 16-byte aligned rows of X");
            return RR_ERR_UNSUPPORTED;
        }
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    if (b->large)
        return x_dtype == RR_F32 ? large_features<float, float, float>(b, (const float *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr)
                                 : large_features<double, float, float>(b, (const double *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr);
    if (Pt != nullptr && !b->large &&
        (x_dtype == RR_F32 ? rr_features_mfma_launch<float, float>(b, (const float *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, scale, Pt, ldt)
                           : rr_features_mfma_launch<double, float>(b, (const double *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, scale, Pt, ldt))) {
        RR_CHECK_HIP(hipGetLastError());
        if (pt_written) *pt_written = true;
        return RR_OK;
    }
    if (x_dtype == RR_F32 ? rr_features_mfma_launch<float, float>(b, (const float *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, scale)
                          : rr_features_mfma_launch<double, float>(b, (const double *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, scale)) {
        RR_CHECK_HIP(hipGetLastError());
        return RR_OK;
    }
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = 256;
    if ((mpad + rpb - 1) / rpb > 65535) rpb = (mpad + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((mpad + rpb - 1) / rpb));
#define RR_LF(DM)                                                                                                   \
    do {                                                                                                            \
        if (x_dtype == RR_F32)                                                                                      \
            hipLaunchKernelGGL((rr_rff_features_kernel<DM, false, float, float>), grid, dim3(256), 0, c->stream,     \
                               (const float *)dX, (const float *)nullptr, m, mpad, ldx, b->dWs32, b->n, b->npad, P,  \
                               ldp, (double *)nullptr, scale, (int)rpb);                                             \
        else                                                                                                        \
            hipLaunchKernelGGL((rr_rff_features_kernel<DM, false, double, float>), grid, dim3(256), 0, c->stream,    \
                               (const double *)dX, (const double *)nullptr, m, mpad, ldx, b->dWs32, b->n, b->npad, P, \
                               ldp, (double *)nullptr, scale, (int)rpb);                                             \
    } while (0)
    switch (b->dpad) {
        case 8: RR_LF(8); break;
        case 16: RR_LF(16); break;
        case 32: RR_LF(32); break;
        case 64: RR_LF(64); break;
        case 128: RR_LF(128); break;
        default: rr_set_error("features: d=%d is not supported", b->d); return RR_ERR_UNSUPPORTED;
    }
#undef RR_LF
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// Row-major f64 features of a row block (second pass in f64 arithmetic): VALU feature kernel, sincospi.
int rr_features_rowmajor_f64(rr_basis *b, const void *dX, int x_dtype, int64_t m, int64_t mpad, int64_t ldx,
                             double *P, int64_t ldp, bool zero_pad_cols) {
    rr_ctx *c = b->ctx;
    const int F = 2 * b->n;
    const double scale = 1.0 / sqrt((double)b->n);
    if (zero_pad_cols && ldp > F) {
        const int64_t cnt = mpad * (ldp - F);
        hipLaunchKernelGGL(rr_zero_padcols_kernel<double>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, P,
                           mpad, ldp, F);
    }
    if (b->large)
        return x_dtype == RR_F32 ? large_features<float, double, double>(b, (const float *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr)
                                 : large_features<double, double, double>(b, (const double *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr);
    if (mpad % 16 == 0) {  // projection on the f64 MFMA (whole 16-row tiles)
        const bool done = x_dtype == RR_F32
                              ? rr_features_mfma64_launch<float, double>(b, (const float *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, scale)
                              : rr_features_mfma64_launch<double, double>(b, (const double *)dX, nullptr, m, mpad, ldx, P, ldp, nullptr, scale);
        if (done) {
            RR_CHECK_HIP(hipGetLastError());
            return RR_OK;
        }
    }
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = 256;
    if ((mpad + rpb - 1) / rpb > 65535) rpb = (mpad + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((mpad + rpb - 1) / rpb));
#define RR_LF64(DM)                                                                                                 \
    do {                                                                                                            \
        if (x_dtype == RR_F32)                                                                                      \
            hipLaunchKernelGGL((rr_rff_features_kernel<DM, false, float, double>), grid, dim3(256), 0, c->stream,    \
                               (const float *)dX, (const float *)nullptr, m, mpad, ldx, b->dWs64, b->n, b->npad, P,  \
                               ldp, (double *)nullptr, scale, (int)rpb);                                             \
        else                                                                                                        \
            hipLaunchKernelGGL((rr_rff_features_kernel<DM, false, double, double>), grid, dim3(256), 0, c->stream,   \
                               (const double *)dX, (const double *)nullptr, m, mpad, ldx, b->dWs64, b->n, b->npad, P, \
                               ldp, (double *)nullptr, scale, (int)rpb);                                             \
    } while (0)
    switch (b->dpad) {
        case 8: RR_LF64(8); break;
        case 16: RR_LF64(16); break;
        case 32: RR_LF64(32); break;
        case 64: RR_LF64(64); break;
        case 128: RR_LF64(128); break;
        default: rr_set_error("features: d=%d is not supported", b->d); return RR_ERR_UNSUPPORTED;
    }
#undef RR_LF64
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// Device staging buffer for host rows: (chunk, dpad) with the pad columns zeroed once.
struct RowStage {
    void *dX = nullptr;
    int64_t chunk = 0;
};

static int stage_alloc(rr_basis *b, int x_dtype, int64_t N, size_t extra_row_bytes, RowStage *st,
                       size_t budget = (size_t)1 << 30) {
    const size_t xs = dtype_size(x_dtype);
    const size_t row_bytes = (size_t)b->dpad * xs + extra_row_bytes;
    int64_t chunk = (int64_t)(budget / row_bytes);  // device staging per chunk (1 GiB unless the caller pipelines)
    if (chunk < 1) chunk = 1;
    if (chunk > N) chunk = N;
    RR_CHECK_HIP(hipMalloc(&st->dX, (size_t)chunk * b->dpad * xs));
    RR_CHECK_HIP(hipMemsetAsync(st->dX, 0, (size_t)chunk * b->dpad * xs, b->ctx->stream));
    st->chunk = chunk;
    return RR_OK;
}

static hipError_t stage_rows(rr_basis *b, const RowStage &st, const void *X, int x_dtype, int64_t r0, int64_t m,
                             int64_t ldx) {
    const size_t xs = dtype_size(x_dtype);
    return hipMemcpy2DAsync(st.dX, (size_t)b->dpad * xs, (const char *)X + (size_t)r0 * ldx * xs, (size_t)ldx * xs,
                            (size_t)b->d * xs, (size_t)m, hipMemcpyHostToDevice, b->ctx->stream);
}

extern "C" {

int rr_symmetrize_dev(rr_ctx *c, double *dG, int64_t F);

int rr_rff_padded_dim(rr_basis *b) { return b ? b->dpad : 0; }

int rr_upload_matrix(rr_ctx *c, const void *X, int dtype, int64_t N, int64_t d, int64_t ldx, int64_t ld_dev,
                     void **dptr) {
    RR_REQUIRE(c != nullptr && dptr != nullptr, "rr_upload_matrix: null argument");
    *dptr = nullptr;
    RR_REQUIRE(dtype_ok(dtype), "rr_upload_matrix: bad dtype");
    RR_REQUIRE(N >= 0 && d >= 1 && ldx >= d && ld_dev >= d, "rr_upload_matrix: bad shape");
    RR_REQUIRE(N == 0 || X != nullptr, "rr_upload_matrix: null host buffer");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = dtype_size(dtype);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)(N > 0 ? N : 1) * ld_dev * es);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_upload_matrix: hipMalloc(%zu bytes) failed", (size_t)N * ld_dev * es);
        return RR_ERR_OOM;
    }
    if (N > 0) {
        if (ld_dev > d) e = hipMemsetAsync(p, 0, (size_t)N * ld_dev * es, c->stream);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(p, (size_t)ld_dev * es, X, (size_t)ldx * es, (size_t)d * es, (size_t)N,
                                 hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(p);
            rr_set_error("rr_upload_matrix: copy failed: %s", hipGetErrorString(e));
            return RR_ERR_HIP;
        }
    }
    *dptr = p;
    return RR_OK;
}

int rr_upload_rows(rr_ctx *c, void *dptr, int64_t ld_dev, int64_t row0, const void *X, int dtype, int64_t N,
                   int64_t d, int64_t ldx) {
    RR_REQUIRE(c != nullptr && dptr != nullptr, "rr_upload_rows: null argument");
    RR_REQUIRE(dtype_ok(dtype), "rr_upload_rows: bad dtype");
    RR_REQUIRE(N >= 0 && row0 >= 0 && d >= 1 && ldx >= d && ld_dev >= d, "rr_upload_rows: bad shape");
    if (N == 0) return RR_OK;
    RR_REQUIRE(X != nullptr, "rr_upload_rows: null host buffer");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = dtype_size(dtype);
    RR_CHECK_HIP(hipMemcpy2DAsync((char *)dptr + (size_t)row0 * ld_dev * es, (size_t)ld_dev * es, X,
                                  (size_t)ldx * es, (size_t)d * es, (size_t)N, hipMemcpyHostToDevice, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    return RR_OK;
}

int rr_rff_transform_dev(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx,
                         const double *lenscale, int n_ls, void *dPhi, int out_dtype, int64_t ldphi) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "rr_rff_transform_dev: not an RFF basis");
    RR_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype), "rr_rff_transform_dev: bad dtype");
    RR_REQUIRE(N >= 0 && ldphi >= 2 * (int64_t)b->n, "rr_rff_transform_dev: bad shape");
    RR_REQUIRE(ldx >= b->dpad, "rr_rff_transform_dev: device X needs ldx >= rr_rff_padded_dim() = %d "
               "with zero pad columns (got ldx=%lld); use rr_upload_matrix", b->dpad, (long long)ldx);
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    if (N == 0) return RR_OK;
    RR_REQUIRE(dX != nullptr && dPhi != nullptr, "rr_rff_transform_dev: null buffer");
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    return transform_dev_impl(b, dX, x_dtype, N, ldx, dPhi, out_dtype, ldphi);
}

// Host-buffer transform: stream X up / Phi down in row chunks sized to a fixed device budget.
int rr_rff_transform(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx,
                     const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "rr_rff_transform: not an RFF basis");
    RR_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype), "rr_rff_transform: bad dtype");
    RR_REQUIRE(N >= 0 && ldx >= b->d && ldphi >= 2 * (int64_t)b->n, "rr_rff_transform: bad shape");
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    if (N == 0) return RR_OK;
    RR_REQUIRE(X != nullptr && Phi != nullptr, "rr_rff_transform: null buffer");
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t os = dtype_size(out_dtype);
    const int64_t F = 2 * (int64_t)b->n;
    RowStage st;
    rc = stage_alloc(b, x_dtype, N, (size_t)F * os, &st, (size_t)256 << 20);  // 256 MiB chunks: pipelined to the host
    if (rc != RR_OK) return rc;
    void *dP = nullptr;
    hipError_t e = hipMalloc(&dP, (size_t)st.chunk * F * os);
    if (e != hipSuccess) {
        (void)hipFree(st.dX);
        rr_set_error("rr_rff_transform: device allocation failed");
        return RR_ERR_OOM;
    }
    rr_host_sink sink;
    rc = rr_sink_open(c, (size_t)st.chunk * F * os, &sink);
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += st.chunk) {
        const int64_t m = (N - r0 < st.chunk) ? N - r0 : st.chunk;
        e = stage_rows(b, st, X, x_dtype, r0, m, ldx);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_transform: upload failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        rc = transform_dev_impl(b, st.dX, x_dtype, m, b->dpad, dP, out_dtype, F);
        if (rc != RR_OK) break;
        rc = rr_sink_push(&sink, dP, (char *)Phi + (size_t)r0 * ldphi * os, (size_t)m, (size_t)F * os, (size_t)ldphi * os);
    }
    if (rc == RR_OK) rc = rr_sink_close(&sink);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(st.dX);
    (void)hipFree(dP);
    return rc;
}

int rr_rff_grad(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                int n_ls, void *dPhi, int out_dtype) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "rr_rff_grad: not an RFF basis");
    RR_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype), "rr_rff_grad: bad dtype");
    RR_REQUIRE(N >= 0 && ldx >= b->d, "rr_rff_grad: bad shape");
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    if (N == 0) return RR_OK;
    RR_REQUIRE(X != nullptr && dPhi != nullptr, "rr_rff_grad: null buffer");
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const int nout = (n_ls == 1) ? 1 : b->d;  // iso: dimension 0 only (reference quirk)
    const size_t os = dtype_size(out_dtype);
    const size_t out_row = (size_t)2 * b->n * nout;
    RowStage st;
    rc = stage_alloc(b, x_dtype, N, out_row * os, &st, (size_t)256 << 20);
    if (rc != RR_OK) return rc;
    void *dO = nullptr;
    hipError_t e = hipMalloc(&dO, (size_t)st.chunk * out_row * os);
    if (e != hipSuccess) {
        (void)hipFree(st.dX);
        rr_set_error("rr_rff_grad: device allocation failed");
        return RR_ERR_OOM;
    }
    rr_host_sink sink;
    rc = rr_sink_open(c, (size_t)st.chunk * out_row * os, &sink);
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += st.chunk) {
        const int64_t m = (N - r0 < st.chunk) ? N - r0 : st.chunk;
        e = stage_rows(b, st, X, x_dtype, r0, m, ldx);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_grad: upload failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        rc = grad_dev_impl(b, st.dX, x_dtype, m, b->dpad, dO, out_dtype, nout);
        if (rc != RR_OK) break;
        rc = rr_sink_push(&sink, dO, (char *)dPhi + (size_t)r0 * out_row * os, (size_t)m, out_row * os, out_row * os);
    }
    if (rc == RR_OK) rc = rr_sink_close(&sink);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(st.dX);
    (void)hipFree(dO);
    return rc;
}

}  // extern "C"

// upload mean / (2 pi) for the spectral-mixture kernels
static int gm_prepare_mean(rr_basis *b, const double *mean) {
    RR_REQUIRE(mean != nullptr, "mean: null argument");
    const double inv2pi = 0.15915494309189533576888;
    std::vector<double> m64(b->dpad, 0.0);
    std::vector<float> m32(b->dpad, 0.f);
    for (int i = 0; i < b->d; ++i) {
        m64[i] = mean[i] * inv2pi;
        m32[i] = (float)m64[i];
    }
    if (!b->dmu32) {
        RR_CHECK_HIP(hipMalloc((void **)&b->dmu32, (size_t)b->dpad * 4));
        RR_CHECK_HIP(hipMalloc((void **)&b->dmu64, (size_t)b->dpad * 8));
    }
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    RR_CHECK_HIP(hipMemcpy(b->dmu32, m32.data(), m32.size() * 4, hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->dmu64, m64.data(), m64.size() * 8, hipMemcpyHostToDevice));
    return RR_OK;
}

template <typename TX, typename TC, typename TO>
static int launch_gm(rr_basis *b, bool grad, const void *dX, int64_t N, int64_t ldx, void *o0, void *o1, int64_t ldo) {
    rr_ctx *c = b->ctx;
    const bool f32 = sizeof(TC) == 4;
    const TC *Ws = f32 ? (const TC *)b->dWs32 : (const TC *)b->dWs64;
    const TC *mu = f32 ? (const TC *)b->dmu32 : (const TC *)b->dmu64;
    const TC *gf = f32 ? (const TC *)b->dgfac32 : (const TC *)b->dgfac64;
    const TC scale = (TC)(1.0 / sqrt(2.0 * (double)b->n));
    const int fblocks = (b->n + 255) / 256;
    int64_t rpb = grad ? 16 : 64;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    const dim3 grid(fblocks, (unsigned)((N + rpb - 1) / rpb));
#define RR_GM(DM)                                                                                                    \
    do {                                                                                                             \
        if (grad)                                                                                                    \
            hipLaunchKernelGGL((rr_gm_grad_kernel<DM, TX, TC, TO>), grid, dim3(256), 0, c->stream, (const TX *)dX, N, ldx, \
                               Ws, mu, gf, b->n, b->npad, b->d, (TO *)o0, (TO *)o1, scale, (int)rpb);                 \
        else                                                                                                         \
            hipLaunchKernelGGL((rr_gm_transform_kernel<DM, TX, TC, TO>), grid, dim3(256), 0, c->stream, (const TX *)dX,  \
                               N, ldx, Ws, mu, b->n, b->npad, (TO *)o0, ldo, scale, (int)rpb);                         \
    } while (0)
    switch (b->dpad) {
        case 8: RR_GM(8); break;
        case 16: RR_GM(16); break;
        case 32: RR_GM(32); break;
        case 64: RR_GM(64); break;
        case 128: RR_GM(128); break;
        default: rr_set_error("gm: d=%d is not supported", b->d); return RR_ERR_UNSUPPORTED;
    }
#undef RR_GM
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

static int gm_dev_impl(rr_basis *b, bool grad, const void *dX, int x_dtype, int64_t N, int64_t ldx, void *o0, void *o1,
                       int out_dtype, int64_t ldo) {
    RR_DISPATCH3(launch_gm, x_dtype, (b->phase64 ? (int)RR_F64 : b->compute), out_dtype, b, grad, dX, N, ldx, o0, o1, ldo);
}

// host-buffer driver shared by rr_gm_transform / rr_gm_grad
static int gm_host_call(rr_basis *b, bool grad, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                        const double *lenscale, int n_ls, void *h0, void *h1, int out_dtype, int64_t ldo,
                        const char *who) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "%s: not an RFF-type basis", who);
    RR_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype), "%s: bad dtype", who);
    RR_REQUIRE(N >= 0 && ldx >= b->d, "%s: bad shape", who);
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc == RR_OK) rc = gm_prepare_mean(b, mean);
    if (rc != RR_OK || N == 0) return rc;
    RR_REQUIRE(X != nullptr && h0 != nullptr && (!grad || h1 != nullptr), "%s: null buffer", who);
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t os = dtype_size(out_dtype);
    const size_t out_row = grad ? (size_t)4 * b->n * b->d : (size_t)4 * b->n;
    RowStage st;
    rc = stage_alloc(b, x_dtype, N, out_row * os * (grad ? 2 : 1), &st);
    if (rc != RR_OK) return rc;
    void *d0 = nullptr, *d1 = nullptr;
    hipError_t e = hipMalloc(&d0, (size_t)st.chunk * out_row * os);
    if (e == hipSuccess && grad) e = hipMalloc(&d1, (size_t)st.chunk * out_row * os);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(st.dX);
        if (d0) (void)hipFree(d0);
        rr_set_error("%s: device allocation failed", who);
        return RR_ERR_OOM;
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += st.chunk) {
        const int64_t m = (N - r0 < st.chunk) ? N - r0 : st.chunk;
        e = stage_rows(b, st, X, x_dtype, r0, m, ldx);
        if (e == hipSuccess) {
            rc = gm_dev_impl(b, grad, st.dX, x_dtype, m, b->dpad, d0, d1, out_dtype, (int64_t)out_row);
            if (rc != RR_OK) break;
            if (grad) {
                e = hipMemcpyAsync((char *)h0 + (size_t)r0 * out_row * os, d0, (size_t)m * out_row * os, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync((char *)h1 + (size_t)r0 * out_row * os, d1, (size_t)m * out_row * os, hipMemcpyDeviceToHost, c->stream);
            } else {
                e = hipMemcpy2DAsync((char *)h0 + (size_t)r0 * ldo * os, (size_t)ldo * os, d0, out_row * os, out_row * os,
                                     (size_t)m, hipMemcpyDeviceToHost, c->stream);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rr_set_error("%s: copy/launch failed: %s", who, hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(st.dX);
    (void)hipFree(d0);
    if (d1) (void)hipFree(d1);
    return rc;
}

extern "C" {

int rr_gm_transform(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                    const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi) {
    RR_REQUIRE(b == nullptr || ldphi >= 4 * (int64_t)b->n, "rr_gm_transform: bad ldphi");
    return gm_host_call(b, false, X, x_dtype, N, ldx, mean, lenscale, n_ls, Phi, nullptr, out_dtype, ldphi, "rr_gm_transform");
}

int rr_gm_grad(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
               const double *lenscale, int n_ls, void *dmean, void *dlen, int out_dtype) {
    return gm_host_call(b, true, X, x_dtype, N, ldx, mean, lenscale, n_ls, dmean, dlen, out_dtype, 0, "rr_gm_grad");
}

int rr_rff_gram_dev(rr_basis *b, const void *dX, const void *dy, int x_dtype, int64_t N, int64_t ldx,
                    const double *lenscale, int n_ls, double *dG, double *db, double *dyty) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "rr_rff_gram_dev: not an RFF basis");
    RR_REQUIRE(dtype_ok(x_dtype), "rr_rff_gram_dev: bad dtype");
    RR_REQUIRE(N >= 0, "rr_rff_gram_dev: bad shape");
    RR_REQUIRE(ldx >= b->dpad, "rr_rff_gram_dev: device X needs ldx >= rr_rff_padded_dim() = %d with zero "
               "pad columns (got ldx=%lld); use rr_upload_matrix", b->dpad, (long long)ldx);
    RR_REQUIRE(dG != nullptr, "rr_rff_gram_dev: null G");
    RR_REQUIRE((dy == nullptr) == (db == nullptr) && (dy == nullptr) == (dyty == nullptr),
               "rr_rff_gram_dev: y, b and yty must be given together");
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    if (N == 0) return RR_OK;
    RR_REQUIRE(dX != nullptr, "rr_rff_gram_dev: null X");
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    if (b->compute == RR_F32)
        rc = (x_dtype == RR_F32) ? launch_gram<float, float>(b, dX, dy, N, ldx, dG, db)
                                 : launch_gram<double, float>(b, dX, dy, N, ldx, dG, db);
    else
        rc = (x_dtype == RR_F32) ? launch_gram<float, double>(b, dX, dy, N, ldx, dG, db)
                                 : launch_gram<double, double>(b, dX, dy, N, ldx, dG, db);
    if (rc != RR_OK) return rc;
    if (dy) {
        int blocks = (int)((N + 255) / 256);
        if (blocks > c->num_cu * 8) blocks = c->num_cu * 8;
        double *ydst = dyty;
        if (c->deterministic) {
            void *part = nullptr;
            rc = rr_det_scratch(c, (size_t)blocks * 8, &part);
            if (rc != RR_OK) return rc;
            ydst = (double *)part;
        }
        if (x_dtype == RR_F32)
            hipLaunchKernelGGL(rr_yty_kernel<float>, dim3(blocks), dim3(256), 0, c->stream, (const float *)dy, N, ydst,
                               (int64_t)(c->deterministic ? 1 : 0));
        else
            hipLaunchKernelGGL(rr_yty_kernel<double>, dim3(blocks), dim3(256), 0, c->stream, (const double *)dy, N, ydst,
                               (int64_t)(c->deterministic ? 1 : 0));
        RR_CHECK_HIP(hipGetLastError());
        if (c->deterministic) {
            rc = rr_det_reduce(c, ydst, blocks, 1, 1, dyty);
            if (rc != RR_OK) return rc;
        }
    }
    return RR_OK;
}

int rr_rff_gram_timings(rr_basis *b, float *features_ms, float *syrk_ms, float *diag_ms, int *launches) {
    RR_REQUIRE(b != nullptr, "rr_rff_gram_timings: null basis");
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    float pa = 0.f, pg = 0.f, pd = 0.f;
    for (size_t i = 0; i + 5 <= b->events_used; i += 5) {
        float t = 0.f;
        RR_CHECK_HIP(hipEventElapsedTime(&t, b->events[i], b->events[i + 1]));
        pa += t;
        RR_CHECK_HIP(hipEventElapsedTime(&t, b->events[i + 2], b->events[i + 3]));
        pg += t;
        RR_CHECK_HIP(hipEventElapsedTime(&t, b->events[i + 3], b->events[i + 4]));
        pd += t;
    }
    if (features_ms) *features_ms = pa;
    if (syrk_ms) *syrk_ms = pg;
    if (diag_ms) *diag_ms = pd;
    if (launches) *launches = (int)(b->events_used / 5);
    return RR_OK;
}

int rr_dense_gram(rr_ctx *c, const void *Phi, int dtype, int64_t N, int64_t F, int64_t ldphi, const void *y,
                  double *G, double *bvec, double *yty) {
    RR_REQUIRE(c != nullptr && G != nullptr, "rr_dense_gram: null argument");
    RR_REQUIRE(dtype_ok(dtype), "rr_dense_gram: bad dtype");
    RR_REQUIRE(N >= 0 && F >= 1 && ldphi >= F && F < (1 << 30), "rr_dense_gram: bad shape");
    RR_REQUIRE((y == nullptr) == (bvec == nullptr) && (y == nullptr) == (yty == nullptr),
               "rr_dense_gram: y, b and yty must be given together");
    RR_REQUIRE(N == 0 || Phi != nullptr, "rr_dense_gram: null Phi");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = dtype_size(dtype);
    // float64 input keeps float64 arithmetic (f64 features into the f64 MFMA SYRK): the reference's X^T X / Phi^T dPhi
    // are float64, and unscaled linear features with large offsets do not survive a cast to f32
    const bool f64 = dtype == RR_F64;
    const int64_t tc = f64 ? G64_TC : GR_TC, kb = f64 ? G64_KB : GR_KB;
    const size_t ps = f64 ? 8 : 4;
    const int64_t ldp = (F + tc - 1) / tc * tc;
    int64_t chunk = (int64_t)(((size_t)1 << 30) / ((size_t)ldp * ps + (size_t)F * es));
    if (chunk > N) chunk = N;
    chunk = (chunk + kb - 1) / kb * kb;
    if (chunk < kb) chunk = kb;
    void *dRaw = nullptr, *dy = nullptr;
    void *dP = nullptr;
    double *dG = nullptr, *db = nullptr;
    int rc = RR_OK;
    hipError_t e = hipMalloc((void **)&dG, (size_t)F * F * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&db, (size_t)(F + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&dRaw, (size_t)chunk * F * es);
    if (e == hipSuccess) e = hipMalloc(&dP, (size_t)chunk * ldp * ps);
    if (e == hipSuccess) e = hipMalloc(&dy, (size_t)chunk * es);
    if (e == hipSuccess) e = hipMemsetAsync(dG, 0, (size_t)F * F * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(db, 0, (size_t)(F + 1) * sizeof(double), c->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_dense_gram: device allocation failed: %s", hipGetErrorString(e));
        rc = RR_ERR_OOM;
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += chunk) {
        const int64_t m = (N - r0 < chunk) ? N - r0 : chunk;
        const int64_t mpad = (m + kb - 1) / kb * kb;
        e = hipMemcpy2DAsync(dRaw, (size_t)F * es, (const char *)Phi + (size_t)r0 * ldphi * es, (size_t)ldphi * es,
                             (size_t)F * es, (size_t)m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && y)
            e = hipMemcpyAsync(dy, (const char *)y + (size_t)r0 * es, (size_t)m * es, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_dense_gram: upload failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        const dim3 pg((unsigned)((ldp + 255) / 256), (unsigned)((mpad + 63) / 64));
        const int rpb = 512;
        const dim3 gg((unsigned)((F + 255) / 256), (unsigned)((m + rpb - 1) / rpb));
        const int yb = (int)((m + 255) / 256) > c->num_cu * 8 ? c->num_cu * 8 : (int)((m + 255) / 256);
        if (!f64) {
            hipLaunchKernelGGL(rr_pack_f32_kernel<float>, pg, dim3(256), 0, c->stream, (const float *)dRaw, m, (int)F, F,
                               (float *)dP, ldp, mpad);
            if (y && c->deterministic) {  // ordered partial sums (rr_internal.h): each kernel, then its reduction
                void *part = nullptr;
                rc = rr_det_scratch(c, (size_t)(gg.y > (unsigned)yb ? gg.y : (unsigned)yb) * (size_t)(F + 1) * 8, &part);
                if (rc != RR_OK) break;
                hipLaunchKernelGGL(rr_gemv_t_kernel<float>, gg, dim3(256), 0, c->stream, (const float *)dP,
                                   (const float *)dy, m, (int)F, ldp, (double *)part, rpb, (int64_t)F);
                rc = rr_det_reduce(c, (const double *)part, gg.y, F, F, db);
                hipLaunchKernelGGL(rr_yty_kernel<float>, dim3(yb), dim3(256), 0, c->stream, (const float *)dy, m, (double *)part, (int64_t)1);
                if (rc == RR_OK) rc = rr_det_reduce(c, (const double *)part, yb, 1, 1, db + F);
                if (rc != RR_OK) break;
            } else if (y) {
                hipLaunchKernelGGL(rr_gemv_t_kernel<float>, gg, dim3(256), 0, c->stream, (const float *)dP,
                                   (const float *)dy, m, (int)F, ldp, db, rpb);
                hipLaunchKernelGGL(rr_yty_kernel<float>, dim3(yb), dim3(256), 0, c->stream, (const float *)dy, m, db + F);
            }
        } else {
            hipLaunchKernelGGL((rr_pack_f32_kernel<double, double>), pg, dim3(256), 0, c->stream, (const double *)dRaw, m,
                               (int)F, F, (double *)dP, ldp, mpad);
            if (y && c->deterministic) {
                void *part = nullptr;
                rc = rr_det_scratch(c, (size_t)(gg.y > (unsigned)yb ? gg.y : (unsigned)yb) * (size_t)(F + 1) * 8, &part);
                if (rc != RR_OK) break;
                hipLaunchKernelGGL((rr_gemv_t_kernel<double, double>), gg, dim3(256), 0, c->stream, (const double *)dP,
                                   (const double *)dy, m, (int)F, ldp, (double *)part, rpb, (int64_t)F);
                rc = rr_det_reduce(c, (const double *)part, gg.y, F, F, db);
                hipLaunchKernelGGL(rr_yty_kernel<double>, dim3(yb), dim3(256), 0, c->stream, (const double *)dy, m, (double *)part, (int64_t)1);
                if (rc == RR_OK) rc = rr_det_reduce(c, (const double *)part, yb, 1, 1, db + F);
                if (rc != RR_OK) break;
            } else if (y) {
                hipLaunchKernelGGL((rr_gemv_t_kernel<double, double>), gg, dim3(256), 0, c->stream, (const double *)dP,
                                   (const double *)dy, m, (int)F, ldp, db, rpb);
                hipLaunchKernelGGL(rr_yty_kernel<double>, dim3(yb), dim3(256), 0, c->stream, (const double *)dy, m,
                                   db + F);
            }
        }
        if (hipGetLastError() != hipSuccess) {
            rr_set_error("rr_dense_gram: launch failed");
            rc = RR_ERR_HIP;
            break;
        }
        rc = f64 ? rr_launch_syrk_f64(c, (const double *)dP, mpad, ldp, (int)F, dG)
                 : rr_launch_syrk_f32(c, (const float *)dP, mpad, ldp, (int)F, dG, nullptr);
        if (rc == RR_OK && (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            rr_set_error("rr_dense_gram: kernel failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    if (rc == RR_OK) rc = rr_symmetrize_dev(c, dG, F);
    if (rc == RR_OK) {
        e = hipMemcpyAsync(G, dG, (size_t)F * F * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && y) {
            e = hipMemcpyAsync(bvec, db, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(yty, db + F, sizeof(double), hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_dense_gram: download failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    (void)hipStreamSynchronize(c->stream);
    if (dG) (void)hipFree(dG);
    if (db) (void)hipFree(db);
    if (dRaw) (void)hipFree(dRaw);
    if (dP) (void)hipFree(dP);
    if (dy) (void)hipFree(dy);
    return rc;
}

int rr_symmetrize_dev(rr_ctx *c, double *dG, int64_t F) {
    RR_REQUIRE(c != nullptr && dG != nullptr && F >= 0, "rr_symmetrize_dev: bad argument");
    if (F == 0) return RR_OK;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const unsigned t = (unsigned)((F + 31) / 32);
    hipLaunchKernelGGL(rr_symmetrize_kernel, dim3(t, t), dim3(256), 0, c->stream, dG, F);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

int rr_rff_gram(rr_basis *b, const void *X, const void *y, int x_dtype, int64_t N, int64_t ldx,
                const double *lenscale, int n_ls, double *G, double *bvec, double *yty) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_RFF, "rr_rff_gram: not an RFF basis");
    RR_REQUIRE(dtype_ok(x_dtype), "rr_rff_gram: bad dtype");
    RR_REQUIRE(N >= 0 && ldx >= b->d && G != nullptr, "rr_rff_gram: bad argument");
    RR_REQUIRE((y == nullptr) == (bvec == nullptr) && (y == nullptr) == (yty == nullptr),
               "rr_rff_gram: y, b and yty must be given together");
    RR_REQUIRE(N == 0 || X != nullptr, "rr_rff_gram: null X");
    int rc = rr_basis_prepare(b, lenscale, n_ls);
    if (rc != RR_OK) return rc;
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t xs = dtype_size(x_dtype);
    const int64_t F = 2 * (int64_t)b->n;
    double *dG = nullptr, *db = nullptr;
    void *dy = nullptr;
    RowStage st;
    hipError_t e = hipMalloc((void **)&dG, (size_t)F * F * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&db, (size_t)(F + 1) * sizeof(double));
    if (e != hipSuccess) {
        rr_set_error("rr_rff_gram: device allocation failed: %s", hipGetErrorString(e));
        rc = RR_ERR_OOM;
    }
    if (rc == RR_OK) rc = stage_alloc(b, x_dtype, N > 0 ? N : 1, xs, &st);
    if (rc == RR_OK && hipMalloc(&dy, (size_t)st.chunk * xs) != hipSuccess) {
        rr_set_error("rr_rff_gram: device allocation failed");
        rc = RR_ERR_OOM;
    }
    if (rc == RR_OK) {
        e = hipMemsetAsync(dG, 0, (size_t)F * F * sizeof(double), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(db, 0, (size_t)(F + 1) * sizeof(double), c->stream);
        if (e != hipSuccess) rc = RR_ERR_HIP;
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += st.chunk) {
        const int64_t m = (N - r0 < st.chunk) ? N - r0 : st.chunk;
        e = stage_rows(b, st, X, x_dtype, r0, m, ldx);
        if (e == hipSuccess && y)
            e = hipMemcpyAsync(dy, (const char *)y + (size_t)r0 * xs, (size_t)m * xs, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_gram: upload failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        rc = rr_rff_gram_dev(b, st.dX, y ? dy : nullptr, x_dtype, m, b->dpad, lenscale, n_ls, dG,
                             y ? db : nullptr, y ? db + F : nullptr);
        if (rc == RR_OK && (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            rr_set_error("rr_rff_gram: kernel failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    if (rc == RR_OK) rc = rr_symmetrize_dev(c, dG, F);
    if (rc == RR_OK) {
        e = hipMemcpyAsync(G, dG, (size_t)F * F * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && y) {
            e = hipMemcpyAsync(bvec, db, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(yty, db + F, sizeof(double), hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_gram: download failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    (void)hipStreamSynchronize(c->stream);
    if (dG) (void)hipFree(dG);
    if (db) (void)hipFree(db);
    if (st.dX) (void)hipFree(st.dX);
    if (dy) (void)hipFree(dy);
    return rc;
}

}  // extern "C"
