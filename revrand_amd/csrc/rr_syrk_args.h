// Shared by rr_rff.hip (feature kernels, f32 / f64 SYRK) and rr_syrk16.hip (split 16-bit SYRK / GEMM engine): the
// K-blocked split layout's output tags and split functions, and the argument block of the SYRK kernels.
#pragma once
#include "rr_internal.h"
#include "rr_mfma_tile.h"
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
// Output tag of the feature kernel: the K-blocked split-bf16 layout of rr_syrk_b16w4_kernel (see there):
// Pb[kstep][column] = 64 B = granules [hi rows 0-7 | hi rows 8-15 | lo rows 0-7 | lo rows 8-15] of a 16-row k-step.
struct rr_pb_t { uintx4 g[4]; };
// The same layout with fp16 parts of the value scaled by a power of two into [-1, 1] (random Fourier features are
// bounded by 1/sqrt(n)): 11 + 11 mantissa bits, |p' - hi - lo| <= 2^-23 of full scale -- f32-grade products from
// three fp16 MFMAs.  lo is below the fp16 normal range; the matrix pipe takes fp16 denormals at full precision.
struct rr_pf_t { uintx4 g[4]; };
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_f16x8(const float *v, float s16, uintx4 &hi, uintx4 &lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = v[2 * q] * s16, x1 = v[2 * q + 1] * s16;
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
        const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
        hi[q] = __builtin_bit_cast(unsigned, halfx2{h0, h1});
        lo[q] = __builtin_bit_cast(unsigned, halfx2{l0, l1});
    }
}

// power of two s with scale * s in [0.5, 1)
__host__ __device__ __forceinline__ float f16_store_scale(float scale) {
    int ex;
    (void)frexpf(scale, &ex);
    return ldexpf(1.f, -ex);
}

// 8 f32 values -> one granule of bf16 hi parts and one of lo parts (hi = bf16(v), lo = bf16(v - hi)); v_cvt_pk_bf16_f32
__device__ __forceinline__ void split_bf16x8(const float *v, uintx4 &hi, uintx4 &lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2v x = {v[2 * q], v[2 * q + 1]};
        const bf16x2 hb = __builtin_convertvector(x, bf16x2);
        const float2v hf = __builtin_convertvector(hb, float2v);
        const bf16x2 lb = __builtin_convertvector(x - hf, bf16x2);
        hi[q] = __builtin_bit_cast(unsigned, hb);
        lo[q] = __builtin_bit_cast(unsigned, lb);
    }
}


struct SyrkArgs {
    const float *P;  // (rows, ldp) f32 features, zero padded; rows % 32 == 0, ldp % 256 == 0
    int64_t rows, ldp;
    int F;       // valid columns
    int nb;      // ldp / 256 column blocks
    int ntiles;  // nb (nb + 1) / 2
    int64_t rows_per_split;  // multiple of 32
    double *G;   // (F, F) f64, upper triangle accumulated
    const int *tile_map;  // optional (ntiles): position in dispatch order -> tile id (XCD-aware), or null
    int offdiag_only;     // 1: enumerate only tiles with ta < tb (the diagonal ones go to the diag kernel)
    int ablate;  // debug (RR_GRAM_ABLATE): bit0 = no in-loop DMA, bit1 = no in-loop barrier
    int spread = 2;  // rr_dma_slot (RR_DMA_SPREAD)
    // GEMM mode of rr_syrk_b16w4_kernel (D = A^T B over K-blocked operands): B side matrix, output
    const float *P2 = nullptr;
    int64_t ldp2 = 0;
    float *D = nullptr;   // (M, ldd) f32, plain stores
    int64_t ldd = 0;
    float out_scale = 1.f;  // fp16 operands: 1 / s^2 of the producer's store scale
    int upper_b = 0;        // GEMM mode: B (K == N) is upper triangular, column tile tb needs only k < 256 (tb + 1)
    // f32 SYRK kernels: column F of P (its first pad column) holds one more vector y; bcol[r] += P[:, r] . y for r < F
    // (Phi^T y rides along with Phi^T Phi instead of costing its own pass over P)
    double *bcol = nullptr;
    // deterministic mode (rr_set_deterministic): K-split ks stores its tile partials into slab ks of `part` (part_stride
    // = ldp * ldp floats each, element [gr][gc] at gr * ldp + gc) instead of adding them to G; rr_syrk_det_reduce_kernel
    // then adds the slabs of every element in ascending ks
    float *part = nullptr;
    int64_t part_stride = 0;
};

#ifdef __HIPCC__
// one accumulator of an f32 SYRK tile (K-split ks) into the upper triangle of G, or -- column F, the rider -- into bcol
__device__ __forceinline__ void rr_syrk_out(const SyrkArgs &p, int64_t ks, int64_t gr, int64_t gc, float v) {
    const int64_t F = p.F;
    if (p.part != nullptr) {
        if (gc < F || (gc == F && p.bcol != nullptr && gr < F)) p.part[ks * p.part_stride + gr * p.ldp + gc] = v;
        return;
    }
    if (gc < F) unsafeAtomicAdd(&p.G[gr * F + gc], (double)v);
    else if (gc == F && p.bcol != nullptr && gr < F) unsafeAtomicAdd(&p.bcol[gr], (double)v);
}
#endif

// Greedy XCD-aware tile order of the SYRK kernels (rr_rff.hip)
void rr_build_tile_map(int nb, int od, int nxcd, std::vector<int> &map);

// rr_syrk16.hip.  nprod: engine code (3, 4: bf16 products; 5: fp16x3 when f16_scale > 0, else bf16x3)
int rr_launch_syrk_bf16(rr_ctx *c, int nprod, const float *P, const void *pb, int64_t rows, int64_t ldp, int F, double *dG,
                        hipEvent_t mid, float f16_scale = 0.f);
int rr_launch_gemm_tn_bf16(rr_ctx *c, int nprod, const float *A, int64_t lda, const float *B, int64_t ldb, float *D,
                           int64_t ldd, int64_t K, int64_t M, int64_t N, void *sa, void *sb, bool sb_ready,
                           bool upper_b = false);
