// Random Fourier feature kernels for gfx950 (MI355X, CDNA4): Phi, dPhi/dl and the fused
// Phi -> Phi^T Phi / Phi^T y accumulation.  HIP source, wave64, f32-input MFMA.
//
// Phase convention: the host uploads Ws[i][f] = W[i][f] / (l_i * 2 pi), so the projection
// t = sum_i x_i Ws[i][f] is the phase in REVOLUTIONS; after the exact reduction
// t - rint(t) in [-0.5, 0.5] the hardware v_sin_f32 / v_cos_f32 (which take revolutions)
// give sin/cos directly.
#include "rr_internal.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ void sincos_rev(float t, float &s, float &c) {
    const float f = t - __builtin_rintf(t);
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}

__device__ __forceinline__ void sincos_rev(double t, double &s, double &c) {
    const double f = t - rint(t);
    sincospi(2.0 * f, &s, &c);
}

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
// Phi^T Phi / Phi^T y in two kernels per row chunk.
//
//  (A) rr_rff_phase_kernel:  Z[r][f] = frac(x_r . Ws[:, f])  in [-0.5, 0.5] revolutions, f32,
//      (rows, npad) row-major in HBM scratch -- 4n bytes per row, written once.  It also takes
//      cos/sin of its own phases to accumulate b = Phi^T y (one pass over every (row, f)).
//  (B) rr_rff_gram_phase_kernel: one workgroup (8 waves) owns the 256x256 block of G spanned
//      by frequency blocks (fa <= fb) of 128 frequencies -- local columns
//      [cos_a | sin_a] x [cos_b | sin_b] -- for one K-split of rows.  Per k-block of 32 rows it
//      loads its 2 x 128 phases per row (coalesced float4, prefetched one k-block ahead into
//      registers), takes v_sin/v_cos, writes the [32][512] Phi tile to LDS (double-buffered, one
//      barrier per k-block) and accumulates with v_mfma_f32_32x32x2_f32; operands come straight
//      from LDS with conflict-free ds_read_b32 (lane -> column, lane>>5 -> row of the 2-row
//      k-step, which IS the 32x32x2 A/B operand layout).  f32 accumulation inside a K-split,
//      f64 atomics across K-splits into the upper triangle of G.
//
// Phi itself never exists in HBM; the projection is done once per row (not once per tile), so
// the Gram kernel's MFMA pipe does Gram work only, independent of Xdim.
// ---------------------------------------------------------------------------------------
constexpr int GR_TF = 128;   // frequencies per tile side
constexpr int GR_KB = 32;    // rows per k-block
constexpr int GR_LD = 512;   // LDS tile row length (floats)
constexpr int GR_THREADS = 512;
#ifndef RR_GRAM_NO_PRODUCE
#define RR_GRAM_NO_PRODUCE 0  // build-time ablation: skip the in-loop cos/sin production
#endif

typedef float float4v __attribute__((ext_vector_type(4)));

template <int DMAX, bool HAS_Y, typename TX>
__global__ void __launch_bounds__(256)
rr_rff_phase_kernel(const TX *__restrict__ X, const TX *__restrict__ y, int64_t N, int64_t ldx,
                    const float *__restrict__ Ws, int n, int npad, float *__restrict__ Z,
                    double *__restrict__ bvec, float scale, int rows_per_block) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool fpad = f < npad;
    float w[DMAX];
    load_w<DMAX, float>(w, Ws, npad, fpad ? f : 0);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    float bc = 0.f, bs = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float t = project_row<DMAX, false, TX, float>(X + r * ldx, DMAX, w);
        const float fr = t - __builtin_rintf(t);
        if (fpad) Z[r * npad + f] = fr;
        if (HAS_Y) {
            const float yv = (float)y[r];
            bc = fmaf(__builtin_amdgcn_cosf(fr), yv, bc);
            bs = fmaf(__builtin_amdgcn_sinf(fr), yv, bs);
        }
    }
    if (HAS_Y && f < n) {
        unsafeAtomicAdd(&bvec[f], (double)(bc * scale));
        unsafeAtomicAdd(&bvec[n + f], (double)(bs * scale));
    }
}

struct GramArgs {
    const float *Z;  // (rows, npad) phases in revolutions
    int64_t N;       // rows in this chunk
    int n, npad, nfb;  // nfb = npad / 128 frequency blocks
    int ntiles;        // nfb (nfb + 1) / 2
    int64_t rows_per_split;
    double *G;
    float scale;
};

// 32 rows x (128 + 128) phases = 2048 float4 per k-block, 4 per thread: wave w takes rows
// w, w+8, w+16, w+24; lanes 0-31 the A-side frequencies (512 contiguous bytes), lanes 32-63 the B side.
struct PhaseStage {
    float4v v[4];
    __device__ __forceinline__ void load(const GramArgs &p, int64_t kb0, int64_t row_end, int wave, int fcol) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int64_t r = kb0 + wave + 8 * k;
            if (r >= row_end) r = row_end - 1;  // clamp: loads stay in bounds, zeroed in sincos_store
            v[k] = *(const float4v *)(p.Z + r * p.npad + fcol);
        }
    }
    __device__ __forceinline__ void sincos_store(float *__restrict__ buf, int64_t kb0, int64_t row_end,
                                                 int wave, int lcol, float scale) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = wave + 8 * k;
            const float sc = (kb0 + lr < row_end) ? scale : 0.f;  // wave-uniform
            float4v c, s;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c[e] = __builtin_amdgcn_cosf(v[k][e]) * sc;
                s[e] = __builtin_amdgcn_sinf(v[k][e]) * sc;
            }
            *(float4v *)(buf + lr * GR_LD + lcol) = c;
            *(float4v *)(buf + lr * GR_LD + lcol + GR_TF) = s;
        }
    }
};

// One sixteenth of sincos_store (row group K, element E), meant to be dropped between the
// MFMA groups of gram_consume so the transcendentals issue while the matrix pipe is busy.
template <int K, int E>
__device__ __forceinline__ void sincos_piece(const PhaseStage &st, float4v &c, float4v &s, float *__restrict__ buf,
                                             int64_t kb0, int64_t row_end, int wave, int lcol, float scale) {
    const int lr = wave + 8 * K;
    const float sc = (kb0 + lr < row_end) ? scale : 0.f;  // wave-uniform
    c[E] = __builtin_amdgcn_cosf(st.v[K][E]) * sc;
    s[E] = __builtin_amdgcn_sinf(st.v[K][E]) * sc;
    if (E == 3) {
        *(float4v *)(buf + lr * GR_LD + lcol) = c;
        *(float4v *)(buf + lr * GR_LD + lcol + GR_TF) = s;
    }
}

// MFMA operands of one 2-row k-step: a[i] = Phi[row][A col block i], b[j] = Phi[row][B col block j]
struct KOps {
    float a[4], b[2];
    __device__ __forceinline__ void load(const float *__restrict__ buf, int t, int aoff, int boff) {
        const float *row = buf + (2 * t) * GR_LD;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = row[aoff + i * 32];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = row[boff + j * 32];
    }
};

__device__ __forceinline__ void gram_mfma8(const KOps &o, floatx16 (&acc)[4][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i], o.b[j], acc[i][j], 0, 0, 0);
}

// 16 k-steps (8 MFMAs each) over the current Phi tile.  The operands of k-step T+1 are read
// from LDS BEFORE the MFMAs of k-step T are issued (an MFMA blocks the in-order stream until the
// matrix pipe takes it, so a read placed after them would expose its latency); when PROD, one
// sixteenth of the next tile's cos/sin is computed and stored per k-step in the MFMA shadow.
template <bool PROD>
__device__ __forceinline__ void gram_consume(const float *__restrict__ cur, floatx16 (&acc)[4][2],
                                             int aoff, int boff, const PhaseStage &st, float *__restrict__ nxt,
                                             int64_t kb1, int64_t row_end, int wave, int lcol, float scale) {
    float4v c, s;
    KOps o0, o1;
    o0.load(cur, 0, aoff, boff);
#define RR_STEP2(T)                                                                                 \
    o1.load(cur, (T) + 1, aoff, boff);                                                              \
    gram_mfma8(o0, acc);                                                                            \
    if (PROD) sincos_piece<(T) / 4, (T) % 4>(st, c, s, nxt, kb1, row_end, wave, lcol, scale);       \
    if ((T) + 2 < GR_KB / 2) o0.load(cur, (T) + 2, aoff, boff);                                     \
    gram_mfma8(o1, acc);                                                                            \
    if (PROD) sincos_piece<((T) + 1) / 4, ((T) + 1) % 4>(st, c, s, nxt, kb1, row_end, wave, lcol, scale);
    RR_STEP2(0) RR_STEP2(2) RR_STEP2(4) RR_STEP2(6) RR_STEP2(8) RR_STEP2(10) RR_STEP2(12) RR_STEP2(14)
#undef RR_STEP2
}

__global__ void __launch_bounds__(GR_THREADS, 2)
rr_rff_gram_phase_kernel(const GramArgs p) {
    __shared__ float lds[2 * GR_KB * GR_LD];  // 128 KiB: two [32][512] Phi tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tile (fa <= fb) and K-split of this workgroup
    int tdx = blockIdx.x % p.ntiles;
    const int ks = blockIdx.x / p.ntiles;
    int fa = 0;
    while (tdx >= p.nfb - fa) {
        tdx -= p.nfb - fa;
        ++fa;
    }
    const int fb = fa + tdx;
    const bool diag = (fa == fb);

    const int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.N) row_end = p.N;

    // producer role: lanes 0-31 -> side A, 32-63 -> side B, 4 consecutive frequencies each
    const int side = lane >> 5, f4 = lane & 31;
    const int fcol = (side ? fb : fa) * GR_TF + 4 * f4;  // column of Z  (< npad)
    const int lcol = side * 256 + 4 * f4;               // column of the LDS tile (cos; sin at +128)

    // consumer role: wave (wr, wc) -> rows [wr*128, +128) of side A, cols [wc*64, +64) of side B
    const int wr = wave >> 2, wc_ = wave & 3;
    const int aoff = (lane >> 5) * GR_LD + wr * 128 + (lane & 31);
    const int boff = (lane >> 5) * GR_LD + 256 + wc_ * 64 + (lane & 31);
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int64_t nkb = (row_end - row_begin + GR_KB - 1) / GR_KB;
    if (nkb > 0) {
        PhaseStage st, pre;
        st.load(p, row_begin, row_end, wave, fcol);
        st.sincos_store(lds, row_begin, row_end, wave, lcol, p.scale);  // Phi(0) -> LDS
        st.load(p, row_begin + GR_KB, row_end, wave, fcol);             // Z(1) (clamped if absent)
        __syncthreads();
        for (int64_t kb = 0; kb < nkb; ++kb) {
            const int cb = (int)(kb & 1);
            const float *cur = lds + cb * (GR_KB * GR_LD);
            float *nxt = lds + (cb ^ 1) * (GR_KB * GR_LD);
            const int64_t kb1 = row_begin + (kb + 1) * GR_KB;
            pre.load(p, kb1 + GR_KB, row_end, wave, fcol);  // Z(kb+2) -> regs, consumed next iteration
            // Single consume site (the accumulators must not flow through divergent paths) that
            // ALWAYS produces: in the last iteration it writes a tile nobody reads.
            gram_consume<!RR_GRAM_NO_PRODUCE>(cur, acc, aoff, boff, st, nxt, kb1, row_end, wave, lcol, p.scale);
            st = pre;
            __syncthreads();
        }
    }

    // ---- flush: f32 partial -> f64 G (upper triangle only) ----
    const int64_t F = 2 * (int64_t)p.n;
    const int hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lb = wc_ * 64 + j * 32 + (lane & 31);  // local B column
        const int fbq = fb * GR_TF + (lb & (GR_TF - 1));
        const bool bvalid = fbq < p.n;
        const int64_t gb = (lb < GR_TF) ? fbq : p.n + fbq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int la = wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                const int faq = fa * GR_TF + (la & (GR_TF - 1));
                const int64_t ga = (la < GR_TF) ? faq : p.n + faq;
                const bool lower = ga > gb;
                const bool keep = bvalid && (faq < p.n) && !(lower && diag);
                const int64_t gr = lower ? gb : ga, gc = lower ? ga : gb;
                if (keep) unsafeAtomicAdd(&p.G[gr * F + gc], (double)acc[i][j][e]);
            }
        }
    }
}

// y^T y (slm.py:161-162 via sqErr = yty - 2 m.b + m G m)
template <typename TX>
__global__ void __launch_bounds__(256) rr_yty_kernel(const TX *__restrict__ y, int64_t N, double *out) {
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
    if (threadIdx.x == 0) unsafeAtomicAdd(out, part[0] + part[1] + part[2] + part[3]);
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

template <typename TX, typename TC, typename TO>
static int launch_transform(rr_basis *b, const void *dX, int64_t N, int64_t ldx, void *dPhi, int64_t ldphi) {
    rr_ctx *c = b->ctx;
    const TC *Ws = (sizeof(TC) == 4) ? (const TC *)b->dWs32 : (const TC *)b->dWs64;
    const TC scale = (TC)(1.0 / sqrt((double)b->n));
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
    RR_DISPATCH3(launch_transform, x_dtype, b->compute, out_dtype, b, dX, N, ldx, dPhi, ldphi);
}

static int grad_dev_impl(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, void *dOut,
                         int out_dtype, int nout) {
    RR_DISPATCH3(launch_grad, x_dtype, b->compute, out_dtype, b, dX, N, ldx, dOut, nout);
}

// Z scratch: grow-only, owned by the basis (freed in rr_basis_destroy).
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
        rr_set_error("gram: could not allocate %zu bytes of phase scratch", bytes);
        return RR_ERR_OOM;
    }
    b->zbuf_bytes = bytes;
    return RR_OK;
}

template <typename TX>
static int launch_gram_f32(rr_basis *b, const void *dX, const void *dy, int64_t N, int64_t ldx, double *dG,
                           double *db) {
    rr_ctx *c = b->ctx;
    const int nfb = b->npad / GR_TF;
    const int ntiles = nfb * (nfb + 1) / 2;
    const float scale = (float)(1.0 / sqrt((double)b->n));
    // row chunks: phase scratch of at most ~16 GiB (or RR_GRAM_CHUNK_ROWS)
    int64_t chunk = (int64_t)(((size_t)16 << 30) / ((size_t)b->npad * sizeof(float)));
    const char *cenv = getenv("RR_GRAM_CHUNK_ROWS");
    if (cenv && atoll(cenv) >= GR_KB) chunk = atoll(cenv);
    if (chunk > N) chunk = N;
    int rc = ensure_zbuf(b, (size_t)chunk * b->npad * sizeof(float));
    if (rc != RR_OK) return rc;
    const char *renv = getenv("RR_GRAM_ROWS_PER_SPLIT");

    for (int64_t r0 = 0; r0 < N; r0 += chunk) {
        const int64_t m = (N - r0 < chunk) ? N - r0 : chunk;
        const TX *Xc = (const TX *)dX + r0 * ldx;
        const TX *yc = dy ? (const TX *)dy + r0 : nullptr;
        // three events per chunk bracket the two kernels (read back by rr_rff_gram_timings)
        const size_t e0 = (size_t)(r0 / chunk) * 3;
        while (b->events.size() < e0 + 3) {
            hipEvent_t ev;
            RR_CHECK_HIP(hipEventCreate(&ev));
            b->events.push_back(ev);
        }
        RR_CHECK_HIP(hipEventRecord(b->events[e0], c->stream));
        // (A) phases (+ Phi^T y)
        {
            const int fblocks = (b->npad + 255) / 256;
            int64_t rpb = 256;
            if ((m + rpb - 1) / rpb > 65535) rpb = (m + 65534) / 65535;
            const dim3 grid(fblocks, (unsigned)((m + rpb - 1) / rpb));
#define RR_LPH(DM)                                                                                          \
    do {                                                                                                    \
        if (yc) hipLaunchKernelGGL((rr_rff_phase_kernel<DM, true, TX>), grid, dim3(256), 0, c->stream, Xc, yc, \
                                   m, ldx, b->dWs32, b->n, b->npad, b->zbuf, db, scale, (int)rpb);         \
        else hipLaunchKernelGGL((rr_rff_phase_kernel<DM, false, TX>), grid, dim3(256), 0, c->stream, Xc, yc, \
                                m, ldx, b->dWs32, b->n, b->npad, b->zbuf, db, scale, (int)rpb);            \
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
            RR_CHECK_HIP(hipGetLastError());
        }
        RR_CHECK_HIP(hipEventRecord(b->events[e0 + 1], c->stream));
        // (B) Gram from phases.  K-splits: f32 accumulation is limited to <= 32768 rows per
        // split; use more (smaller) splits when needed to give every CU several workgroups.
        {
            // Every workgroup costs the same, so make their number a multiple of the CU count
            // (no partial last round): nsplit = k * CUs / gcd(CUs, ntiles), k minimal such that
            // a split has <= 32768 rows (the bound on f32 accumulation length).
            int64_t g = c->num_cu, t = ntiles;
            while (t) { const int64_t u = g % t; g = t; t = u; }
            const int64_t unit = c->num_cu / g;  // 32 for 256 CUs and 136 tiles
            int64_t nsplit = ((m + 32767) / 32768 + unit - 1) / unit * unit;
            if (m / nsplit < 1024) nsplit = (m + 1023) / 1024;  // small inputs: just cover the rows
            if (nsplit < 1) nsplit = 1;
            int64_t rps = ((m + nsplit - 1) / nsplit + GR_KB - 1) / GR_KB * GR_KB;
            if (renv && atoll(renv) >= GR_KB) rps = (atoll(renv) / GR_KB) * GR_KB;
            nsplit = (m + rps - 1) / rps;
            RR_REQUIRE(nsplit * ntiles < (int64_t)1 << 31, "gram: grid too large");
            GramArgs a;
            a.Z = b->zbuf; a.N = m; a.n = b->n; a.npad = b->npad; a.nfb = nfb; a.ntiles = ntiles;
            a.rows_per_split = rps; a.G = dG; a.scale = scale;
            hipLaunchKernelGGL(rr_rff_gram_phase_kernel, dim3((unsigned)(nsplit * ntiles)), dim3(GR_THREADS), 0,
                               c->stream, a);
            RR_CHECK_HIP(hipGetLastError());
        }
        RR_CHECK_HIP(hipEventRecord(b->events[e0 + 2], c->stream));
        b->events_used = e0 + 3;
    }
    b->gram_kernel = "rr_rff_gram_phase_kernel";
    return RR_OK;
}

// Device staging buffer for host rows: (chunk, dpad) with the pad columns zeroed once.
struct RowStage {
    void *dX = nullptr;
    int64_t chunk = 0;
};

static int stage_alloc(rr_basis *b, int x_dtype, int64_t N, size_t extra_row_bytes, RowStage *st) {
    const size_t xs = dtype_size(x_dtype);
    const size_t row_bytes = (size_t)b->dpad * xs + extra_row_bytes;
    int64_t chunk = (int64_t)(((size_t)1 << 30) / row_bytes);  // ~1 GiB of device staging
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
    rc = stage_alloc(b, x_dtype, N, (size_t)F * os, &st);
    if (rc != RR_OK) return rc;
    void *dP = nullptr;
    hipError_t e = hipMalloc(&dP, (size_t)st.chunk * F * os);
    if (e != hipSuccess) {
        (void)hipFree(st.dX);
        rr_set_error("rr_rff_transform: device allocation failed");
        return RR_ERR_OOM;
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += st.chunk) {
        const int64_t m = (N - r0 < st.chunk) ? N - r0 : st.chunk;
        e = stage_rows(b, st, X, x_dtype, r0, m, ldx);
        if (e == hipSuccess) {
            rc = transform_dev_impl(b, st.dX, x_dtype, m, b->dpad, dP, out_dtype, F);
            if (rc != RR_OK) break;
            e = hipMemcpy2DAsync((char *)Phi + (size_t)r0 * ldphi * os, (size_t)ldphi * os, dP,
                                 (size_t)F * os, (size_t)F * os, (size_t)m, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_transform: copy/launch failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
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
    rc = stage_alloc(b, x_dtype, N, out_row * os, &st);
    if (rc != RR_OK) return rc;
    void *dO = nullptr;
    hipError_t e = hipMalloc(&dO, (size_t)st.chunk * out_row * os);
    if (e != hipSuccess) {
        (void)hipFree(st.dX);
        rr_set_error("rr_rff_grad: device allocation failed");
        return RR_ERR_OOM;
    }
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += st.chunk) {
        const int64_t m = (N - r0 < st.chunk) ? N - r0 : st.chunk;
        e = stage_rows(b, st, X, x_dtype, r0, m, ldx);
        if (e == hipSuccess) {
            rc = grad_dev_impl(b, st.dX, x_dtype, m, b->dpad, dO, out_dtype, nout);
            if (rc != RR_OK) break;
            e = hipMemcpyAsync((char *)dPhi + (size_t)r0 * out_row * os, dO, (size_t)m * out_row * os,
                               hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_rff_grad: copy/launch failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(st.dX);
    (void)hipFree(dO);
    return rc;
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
    if (b->compute != RR_F32) {
        rr_set_error("rr_rff_gram_dev: f64 Gram kernel not built yet");
        return RR_ERR_UNSUPPORTED;
    }
    rc = (x_dtype == RR_F32) ? launch_gram_f32<float>(b, dX, dy, N, ldx, dG, db)
                             : launch_gram_f32<double>(b, dX, dy, N, ldx, dG, db);
    if (rc != RR_OK) return rc;
    if (dy) {
        int blocks = (int)((N + 255) / 256);
        if (blocks > c->num_cu * 8) blocks = c->num_cu * 8;
        if (x_dtype == RR_F32)
            hipLaunchKernelGGL(rr_yty_kernel<float>, dim3(blocks), dim3(256), 0, c->stream, (const float *)dy, N, dyty);
        else
            hipLaunchKernelGGL(rr_yty_kernel<double>, dim3(blocks), dim3(256), 0, c->stream, (const double *)dy, N, dyty);
        RR_CHECK_HIP(hipGetLastError());
    }
    return RR_OK;
}

int rr_rff_gram_timings(rr_basis *b, float *phase_ms, float *gram_ms, int *launches) {
    RR_REQUIRE(b != nullptr, "rr_rff_gram_timings: null basis");
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    float pa = 0.f, pg = 0.f;
    for (size_t i = 0; i + 3 <= b->events_used; i += 3) {
        float t = 0.f;
        RR_CHECK_HIP(hipEventElapsedTime(&t, b->events[i], b->events[i + 1]));
        pa += t;
        RR_CHECK_HIP(hipEventElapsedTime(&t, b->events[i + 1], b->events[i + 2]));
        pg += t;
    }
    if (phase_ms) *phase_ms = pa;
    if (gram_ms) *gram_ms = pg;
    if (launches) *launches = (int)(b->events_used / 3);
    return RR_OK;
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
