// Split 16-bit Gram / GEMM engine for gfx950 (MI355X, CDNA4): bf16 or fp16 hi + lo operands on the 16-bit matrix pipe,
// f32 accumulation (RR_GRAM_BF16X3 / BF16X4 / FP16X3, include/revrand_hip.h).
#include "rr_syrk_args.h"
#include "rr_mfma_tile.h"  // rr_make_rsrc
#include <cstring>

// ---------------------------------------------------------------------------------------
// Split-bf16 SYRK ("bf16x3" / "bf16x4"): every f32 feature value p is split into hi = bf16(p) and
// lo = bf16(p - hi) (|p - hi - lo| <= 2^-18 |p|) and G accumulates hi.hi + hi.lo + lo.hi (+ lo.lo for x4) in f32 on
// the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate).  x3 drops lo.lo (<= 2^-18 |p_a p_b|,
// same sign on the diagonal: a ~1e-6 relative bias there); x4 keeps it; fp16x3 (rr_pf_t) uses fp16 parts of values
// scaled into [-1, 1].  One workgroup owns a 256x256 block of G for one K-split, f64 atomics across K-splits.
//
// Operands need 8 consecutive k (rows) of one column per lane, so the features are laid out K-blocked:
// Pb[kb][c] = 64 B = four 16-B granules [hi rows 0-7 | hi rows 8-15 | lo rows 0-7 | lo rows 8-15] of the 16
// rows of k-step kb, column c of ldp (rr_split_bf16_kernel converts a row-major f32 chunk).  A 256-column side
// of one k-step is 16 KiB contiguous.  One k-step (16 rows, [A side | B side] = 32 KiB) is one stage of a
// 4-stage LDS ring filled by LDS-DMA three k-steps ahead of its use; in LDS the four granules of a column are
// XOR-swizzled with (c >> 2) & 3 -- the DMA is lane-linear in LDS and applies the permutation on its global
// addresses -- which makes every ds_read_b128 operand fetch conflict-free.  One barrier per k-step.
// ---------------------------------------------------------------------------------------
constexpr int B16_STAGE = 32768;  // bytes per ring stage: [A side 16 KiB | B side 16 KiB]

__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// grid (rows64 / 16, ldp / 256), 256 threads: thread = column, 16 rows of it in registers; rows >= `rows` -> 0
__global__ void __launch_bounds__(256)
rr_split_bf16_kernel(const float *__restrict__ P, int64_t rows, int64_t ldp, uintx4 *__restrict__ Pb) {
    const int64_t c = (int64_t)blockIdx.y * 256 + threadIdx.x, kb = blockIdx.x;  // k-steps on x: rows / 16 can exceed 65535
    const float *src = P + kb * 16 * ldp + c;
    const bool live = kb * 16 < rows;  // rows is a multiple of 32
    unsigned hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x0 = live ? src[(2 * k) * ldp] : 0.f, x1 = live ? src[(2 * k + 1) * ldp] : 0.f;
        const unsigned h0 = bf16_rne(x0), h1 = bf16_rne(x1);
        const unsigned l0 = bf16_rne(x0 - __uint_as_float(h0 << 16)), l1 = bf16_rne(x1 - __uint_as_float(h1 << 16));
        hi[k] = h0 | (h1 << 16);
        lo[k] = l0 | (l1 << 16);
    }
    uintx4 *dst = Pb + (kb * ldp + c) * 4;
    dst[0] = uintx4{hi[0], hi[1], hi[2], hi[3]};
    dst[1] = uintx4{hi[4], hi[5], hi[6], hi[7]};
    dst[2] = uintx4{lo[0], lo[1], lo[2], lo[3]};
    dst[3] = uintx4{lo[4], lo[5], lo[6], lo[7]};
}

template <int OFF>
__device__ __forceinline__ uintx4 lds_read_b128(unsigned addr) {
    uintx4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}

// ---------------------------------------------------------------------------------------
// ONE wave per SIMD (4 waves of 128x128, 256 accumulator registers each in AGPRs): a wave's non-MFMA work -- 8 DMA
// instructions and the 16 operand reads of the NEXT k-step (second register set) -- is placed one instruction at a
// time into the shadows of its own 48 (64) MFMAs (a 32x32x16 MFMA occupies the pipe for 32 cycles = ~5 issue slots).
// An earlier version with 8 waves of 128x64 (two per SIMD, as in the f32 kernel) left the pipe 26 % idle: both waves
// of a SIMD sit at the same barrier, so neither covers the other's DMA issue and reads; it also needed 1/3 more
// operand reads per MFMA.  The k loop is branch-free: past the end of a K-split the DMA re-fetches the last k-step into a free
// buffer and the reads fetch operands that are never used.
// ---------------------------------------------------------------------------------------
struct B16Ops4 {
    uintx4 ah[4], al[4], bh[4], bl[4];
    // read IDX (0..15) of ring stage BUF, in the order the MFMAs need them: (b0, a0, a1, a2, a3, b1, b2, b3) x (hi, lo)
    template <int BUF, int IDX>
    __device__ __forceinline__ void load_one(const unsigned (&a)[2][2], const unsigned (&b)[2][2]) {
        constexpr int O = (BUF & 1) * B16_STAGE;
        constexpr int PART = IDX >> 3, W = IDX & 7;
        if constexpr (W == 0) (PART ? bl[0] : bh[0]) = lds_read_b128<O>(b[BUF >> 1][PART]);
        else if constexpr (W <= 4) (PART ? al[W - 1] : ah[W - 1]) = lds_read_b128<O + (W - 1) * 2048>(a[BUF >> 1][PART]);
        else (PART ? bl[W - 4] : bh[W - 4]) = lds_read_b128<O + (W - 4) * 2048>(b[BUF >> 1][PART]);
    }
};

// MFMA Q of a k-step: products in the order hi.hi (needs only the "hi" reads), hi.lo, lo.hi, lo.lo; tiles (i, j)
template <int Q, bool F16>
__device__ __forceinline__ void b16w4_mfma(const B16Ops4 &o, floatx16 (&acc)[4][4]) {
    constexpr int pr = Q >> 4, i = (Q >> 2) & 3, j = Q & 3;
    const uintx4 a = (pr & 2) ? o.al[i] : o.ah[i];
    const uintx4 b = (pr & 1) ? o.bl[j] : o.bh[j];
    if (F16)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b),
                                                           acc[i][j], 0, 0, 0);
    else
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                            acc[i][j], 0, 0, 0);
}

template <int NPROD, bool GEMM, bool F16>
__global__ void __launch_bounds__(256, 1)
rr_syrk_b16w4_kernel(const SyrkArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[4 * B16_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int tdx = (int)(blockIdx.x % p.ntiles);
    const int ks = (int)(blockIdx.x / p.ntiles);  // GEMM: K-split too when there are few tiles (f32 atomics into D)
    int ta = 0, tb = 0;
    if (GEMM) {
        ta = tdx / p.nb;
        tb = tdx % p.nb;
        if (p.upper_b) tb = (tb + ta) % p.nb;  // tile cost grows with tb, block b runs on XCD b % 8: give every XCD every cost
    } else {
        if (p.tile_map) tdx = p.tile_map[tdx];
        const int od = p.offdiag_only;
        while (tdx >= p.nb - ta - od) {
            tdx -= p.nb - ta - od;
            ++ta;
        }
        tb = ta + tdx + od;
    }
    const int ca = ta * GR_TC, cb = tb * GR_TC;
    const int64_t row_begin = (int64_t)ks * p.rows_per_split;
    int64_t row_end = row_begin + p.rows_per_split;
    if (row_end > p.rows) row_end = p.rows;
    int S = (int)((row_end - row_begin) / 16);  // k-steps, a multiple of 4
    if (GEMM && p.upper_b && S > (tb + 1) * 16) S = (tb + 1) * 16;  // upper-triangular B: nothing below the diagonal

    // DMA role: 32 instructions of 1 KiB per stage; wave w issues t = 8 w + k: waves 0-1 the A side, 2-3 the B side
    const int side = wave >> 1;
    const int tt0 = (wave & 1) * 8;
    const unsigned lane_src = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
    const int64_t ld_side = (GEMM && side) ? p.ldp2 : p.ldp;
    // (requests through a buffer descriptor: the lane part is one constant VGPR, the stage / segment step a scalar offset --
    // no vector instruction per request in the MFMA stream; see rr_dma_kblock, rr_mfma_tile.h)
    // (Gram mode only: a GEMM operand's stages are ld * 64 bytes apart with ld = rows per chunk -- beyond a 32-bit offset)
    const char *src0 = (const char *)((GEMM && side) ? p.P2 : p.P) + ((row_begin / 16) * ld_side + (side ? cb : ca)) * 64 + tt0 * 1024;
    const rr_rsrc_t srs = rr_make_rsrc(src0, 0x7fffffffu);
    const int64_t stage_stride = ld_side * 64;
    char *dst0 = lds + side * 16384 + tt0 * 1024;

    // consumer role: wave (wr, wc) -> columns [wr*128, +128) of side A, [wc*128, +128) of side B
    const int wr = wave >> 1, wc_ = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    unsigned abase[2][2], bbase[2][2];  // [ring half][part]
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const unsigned xs = (unsigned)(((2 * pp + h) ^ ((l31 >> 2) & 3)) * 16);
            abase[hf][pp] = lds0 + hf * 2 * B16_STAGE + (unsigned)((wr * 128 + l31) * 64) + xs;
            bbase[hf][pp] = lds0 + hf * 2 * B16_STAGE + 16384u + (unsigned)((wc_ * 128 + l31) * 64) + xs;
        }
    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int NM = NPROD * 16;
    if (S > 0) {
        auto dma_one = [&](int g, int buf, int k) {  // g is clamped: past the end the last k-step is fetched again
            if constexpr (!GEMM) {
                const unsigned so = (unsigned)(g < S ? g : S - 1) * (unsigned)stage_stride + (unsigned)(k * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lptr_t)(dst0 + buf * B16_STAGE + k * 1024), 16, lane_src, so, 0, 0);
            } else {
                const char *src = src0 + lane_src + (int64_t)(g < S ? g : S - 1) * stage_stride + k * 1024;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst0 + buf * B16_STAGE + k * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int k = 0; k < 8; ++k) dma_one(st, st, k);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!GEMM && ta == tb && wr == 1 && wc_ == 0) {
            // diagonal tile: this wave's 128x128 block lies below the diagonal.  It keeps its DMA duty and the
            // barriers but issues no reads and no MFMAs (the kernel is power-limited: an idle SIMD is not wasted).
            for (int g = 0; g < S; ++g) {
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int k = 0; k < 8; ++k) dma_one(g + 4, g & 3, k);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        B16Ops4 r0, r1;
#define RR_W4_LOADALL(R, BUF)                                                                                    \
    R.template load_one<BUF, 0>(abase, bbase); R.template load_one<BUF, 1>(abase, bbase);                         \
    R.template load_one<BUF, 2>(abase, bbase); R.template load_one<BUF, 3>(abase, bbase);                         \
    R.template load_one<BUF, 4>(abase, bbase); R.template load_one<BUF, 5>(abase, bbase);                         \
    R.template load_one<BUF, 6>(abase, bbase); R.template load_one<BUF, 7>(abase, bbase);                         \
    R.template load_one<BUF, 8>(abase, bbase); R.template load_one<BUF, 9>(abase, bbase);                         \
    R.template load_one<BUF, 10>(abase, bbase); R.template load_one<BUF, 11>(abase, bbase);                       \
    R.template load_one<BUF, 12>(abase, bbase); R.template load_one<BUF, 13>(abase, bbase);                       \
    R.template load_one<BUF, 14>(abase, bbase); R.template load_one<BUF, 15>(abase, bbase);
        RR_W4_LOADALL(r0, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // MFMA Q, then one filler: DMA instruction Q of stage g+4 (Q < 8), operand read Q-8 of stage g+1 (8 <= Q < 24)
#define RR_W4_M(Q, CUR, NXT, BUFN)                                                           \
    if constexpr ((Q) < NM) {                                                                \
        b16w4_mfma<(Q), F16>(CUR, acc);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if constexpr ((Q) < 8) dma_one(g + 4, QB, (Q));                                      \
        else if constexpr ((Q) < 24) NXT.template load_one<BUFN, ((Q) - 8) & 15>(abase, bbase); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    }
#define RR_W4_M8(Q, CUR, NXT, BUFN)                                                                          \
    RR_W4_M((Q), CUR, NXT, BUFN) RR_W4_M((Q) + 1, CUR, NXT, BUFN) RR_W4_M((Q) + 2, CUR, NXT, BUFN)           \
    RR_W4_M((Q) + 3, CUR, NXT, BUFN) RR_W4_M((Q) + 4, CUR, NXT, BUFN) RR_W4_M((Q) + 5, CUR, NXT, BUFN)       \
    RR_W4_M((Q) + 6, CUR, NXT, BUFN) RR_W4_M((Q) + 7, CUR, NXT, BUFN)
#define RR_W4_STEP(QQ, CUR, NXT)                                                             \
    {                                                                                        \
        const int g = g0 + (QQ);                                                             \
        constexpr int QB = (QQ);                                                             \
        constexpr int BUFN = ((QQ) + 1) & 3;                                                 \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        RR_W4_M8(0, CUR, NXT, BUFN) RR_W4_M8(8, CUR, NXT, BUFN) RR_W4_M8(16, CUR, NXT, BUFN) \
        RR_W4_M8(24, CUR, NXT, BUFN) RR_W4_M8(32, CUR, NXT, BUFN) RR_W4_M8(40, CUR, NXT, BUFN) \
        RR_W4_M8(48, CUR, NXT, BUFN) RR_W4_M8(56, CUR, NXT, BUFN)                            \
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    }
        for (int g0 = 0; g0 < S; g0 += 4) {
            RR_W4_STEP(0, r0, r1)
            RR_W4_STEP(1, r1, r0)
            RR_W4_STEP(2, r0, r1)
            RR_W4_STEP(3, r1, r0)
        }
#undef RR_W4_STEP
#undef RR_W4_M8
#undef RR_W4_M
#undef RR_W4_LOADALL
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must land before the LDS is released
    }

    const int64_t F = p.F;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t gc = cb + wc_ * 128 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = ca + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (GEMM) {
                    if (p.offdiag_only)  // GEMM mode reuses this field: 1 = several K-splits accumulate into a zeroed D
                        unsafeAtomicAdd(&p.D[gr * p.ldd + gc], acc[i][j][e]);
                    else
                        p.D[gr * p.ldd + gc] = acc[i][j][e];
                } else if (gr <= gc && gc < F)
                    unsafeAtomicAdd(&p.G[gr * F + gc], (double)(F16 ? acc[i][j][e] * p.out_scale : acc[i][j][e]));
            }
        }
    }
}

// D (M, N) f32 = A^T B with A (K, M), B (K, N) row-major f32 (K % 64 == 0, M, N % 256 == 0, all zero padded): both are
// converted to the K-blocked split-bf16 layout into caller scratch (sa: K*lda*4 bytes, sb: K*ldb*4; sb_ready: B was
// converted by an earlier call and is unchanged) and multiplied on the bf16 matrix pipe with nprod products.
int rr_launch_gemm_tn_bf16(rr_ctx *c, int nprod, const float *A, int64_t lda, const float *B, int64_t ldb, float *D,
                           int64_t ldd, int64_t K, int64_t M, int64_t N, void *sa, void *sb, bool sb_ready, bool upper_b) {
    hipLaunchKernelGGL(rr_split_bf16_kernel, dim3((unsigned)(K / 16), (unsigned)(M / 256)), dim3(256), 0, c->stream, A, K, lda,
                       (uintx4 *)sa);
    if (!sb_ready)
        hipLaunchKernelGGL(rr_split_bf16_kernel, dim3((unsigned)(K / 16), (unsigned)(N / 256)), dim3(256), 0, c->stream, B, K,
                           ldb, (uintx4 *)sb);
    SyrkArgs a;
    a.P = (const float *)sa; a.ldp = lda; a.P2 = (const float *)sb; a.ldp2 = ldb; a.rows = K; a.rows_per_split = K;
    a.F = (int)N; a.nb = (int)(N / 256); a.ntiles = (int)((M / 256) * (N / 256)); a.G = nullptr; a.tile_map = nullptr;
    a.offdiag_only = 0; a.ablate = 0; a.D = D; a.ldd = ldd;
    a.upper_b = (upper_b && K == N) ? 1 : 0;
    RR_REQUIRE((M / 256) * (N / 256) < (int64_t)1 << 24, "gemm: grid too large");
    int64_t nsplit = 1;
    if (!a.upper_b && a.ntiles < 2 * c->num_cu && K >= 2048) {  // too few tiles to fill the chip: split K (>= 512 rows each), f32 atomics
        nsplit = (2 * (int64_t)c->num_cu + a.ntiles - 1) / a.ntiles;
        if (nsplit > K / 512) nsplit = K / 512;
        a.rows_per_split = ((K + nsplit - 1) / nsplit + 63) / 64 * 64;
        nsplit = (K + a.rows_per_split - 1) / a.rows_per_split;
        if (nsplit > 1) {
            a.offdiag_only = 1;
            RR_CHECK_HIP(hipMemsetAsync(D, 0, (size_t)M * ldd * sizeof(float), c->stream));
        }
    }
    const dim3 grid((unsigned)(a.ntiles * nsplit));
    if (nprod == 4)
        hipLaunchKernelGGL((rr_syrk_b16w4_kernel<4, true, false>), grid, dim3(256), 0, c->stream, a);
    else
        hipLaunchKernelGGL((rr_syrk_b16w4_kernel<3, true, false>), grid, dim3(256), 0, c->stream, a);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

// nprod: 3 or 4 products.  Either P (row-major f32, rows % 32 == 0: converted here into the context's Pb scratch) or
// pb (features already in the K-blocked layout, rows % 64 == 0, pad rows / columns zero).
// f16_scale > 0: pb holds fp16 parts of value * f16_scale (rr_pf_t, RR_GRAM_FP16X3); engine code 5 without such a
// producer (conversion path) falls back to the bf16 split with 3 products.
int rr_launch_syrk_bf16(rr_ctx *c, int nprod, const float *P, const void *pb, int64_t rows, int64_t ldp, int F, double *dG,
                        hipEvent_t mid, float f16_scale) {
    const int nb = (int)(ldp / GR_TC);
    const int od = 0;
    const int ntiles = nb * (nb + 1) / 2;
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
    const int64_t rows64 = (rows + 63) / 64 * 64;
    const size_t need = pb ? 0 : (size_t)rows64 * ldp * 4;
    if (c->pb_bytes < need) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (c->pb) (void)hipFree(c->pb);
        c->pb = nullptr;
        c->pb_bytes = 0;
        RR_CHECK_HIP(hipMalloc(&c->pb, need));
        c->pb_bytes = need;
    }
    if (!pb) {
        hipLaunchKernelGGL(rr_split_bf16_kernel, dim3((unsigned)(rows64 / 16), (unsigned)(ldp / 256)), dim3(256), 0, c->stream,
                           P, rows, ldp, (uintx4 *)c->pb);
        pb = c->pb;
    }
    if (mid) RR_CHECK_HIP(hipEventRecord(mid, c->stream));
    auto gcd64 = [](int64_t x, int64_t y) { while (y) { const int64_t u = x % y; x = y; y = u; } return x; };
    // f32 accumulation per K-split: the fp16 engine is accurate enough (max error = the f32 engine's) for the pipe's
    // accumulate bias to show in trace(G) -- 1.0e-6 of N at 32 768 rows per split, 6e-7 at 16 384 (+1.3 % time)
    const int64_t max_rows_split = f16_scale > 0.f ? 16384 : 32768;
    const int64_t min_splits = (rows64 + max_rows_split - 1) / max_rows_split;
    const int64_t unit = c->num_cu / gcd64(c->num_cu, ntiles);
    int64_t nsplit = (min_splits + unit - 1) / unit * unit;
    if (rows64 / nsplit < 1024) nsplit = (rows64 + 1023) / 1024;
    if (nsplit < 1) nsplit = 1;
    int64_t rps = ((rows64 + nsplit - 1) / nsplit + 63) / 64 * 64;
    const char *renv = getenv("RR_GRAM_ROWS_PER_SPLIT");
    if (renv && atoll(renv) >= 64) rps = (atoll(renv) / 64) * 64;
    else
        while (rps > 64 && (rps / 16) * ldp * 64 + 32768 >= (int64_t)1 << 31) rps = (rps / 2 + 63) / 64 * 64;  // (see the check below)
    nsplit = (rows64 + rps - 1) / rps;
    RR_REQUIRE(nsplit * ntiles < (int64_t)1 << 31, "gram: grid too large");
    // the kernel's LDS-DMA addresses a K-split's stages through ONE buffer descriptor with 32-bit offsets: stage g of the
    // split sits g * ldp * 64 bytes behind its first (a wide matrix or a large RR_GRAM_ROWS_PER_SPLIT would wrap
    // silently and the Gram would be wrong)
    RR_REQUIRE((rps / 16) * ldp * 64 + 32768 < (int64_t)1 << 31,
               "gram (split 16-bit engine): %lld rows per K-split of a %lld-column feature matrix exceed the kernel's 32-bit "
               "stage offsets: lower RR_GRAM_ROWS_PER_SPLIT", (long long)rps, (long long)ldp);
    SyrkArgs a;
    a.P = (const float *)pb; a.rows = rows64; a.ldp = ldp; a.F = F; a.nb = nb; a.ntiles = ntiles; a.rows_per_split = rps;
    a.G = dG;
    a.tile_map = use_map ? c->tile_map : nullptr;
    a.offdiag_only = od;
    a.ablate = getenv("RR_GRAM_ABLATE") ? atoi(getenv("RR_GRAM_ABLATE")) : 0;
    const dim3 grid((unsigned)(nsplit * ntiles));
    if (f16_scale > 0.f) {
        a.out_scale = 1.f / (f16_scale * f16_scale);
        hipLaunchKernelGGL((rr_syrk_b16w4_kernel<3, false, true>), grid, dim3(256), 0, c->stream, a);
    } else if (nprod == 4)
        hipLaunchKernelGGL((rr_syrk_b16w4_kernel<4, false, false>), grid, dim3(256), 0, c->stream, a);
    else
        hipLaunchKernelGGL((rr_syrk_b16w4_kernel<3, false, false>), grid, dim3(256), 0, c->stream, a);
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

