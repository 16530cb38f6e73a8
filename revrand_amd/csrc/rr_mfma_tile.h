// Device helpers shared by the f32 MFMA tile kernels (SYRK in rr_rff.hip, GEMM-TN in rr_elbo.hip):
// the [32 rows][256 | 256] LDS tile, its conflict-free operand reads and the pinned MFMA schedule.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int GR_TC = 256;   // columns per tile side
constexpr int GR_KB = 32;    // rows per k-block
constexpr int GR_LD = 512;   // LDS tile row length (floats): [A side 256 | B side 256]
constexpr int GR_THREADS = 512;

typedef float float2v __attribute__((ext_vector_type(2)));

typedef __amdgpu_buffer_rsrc_t rr_rsrc_t;
typedef unsigned rr_u4_t __attribute__((ext_vector_type(4)));
// raw buffer descriptor over `bytes` bytes at a WAVE-UNIFORM address (made provably uniform for the compiler;
// readfirstlane returns int: the halves go through unsigned so that the low one is not sign-extended over the high one)
__device__ __forceinline__ rr_rsrc_t rr_make_rsrc(const void *base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | (uint64_t)lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
}

// One wave's share of a k-block tile -- rows 4 w .. 4 w + 3 of the [32][256 | 256] LDS tile, A side from pa + row lda, B side
// from pb + row ldb (pa, pb: the k-block's first row at the tile's first column, wave-uniform) -- as 8 LDS-DMA requests
// through buffer descriptors: the lane part is ONE constant VGPR (16 lane bytes), the row step a scalar offset, so a request
// costs no vector instruction.  (The flat form, global_load_lds with a 64-bit VGPR address, needs a v_lshl_add_u64 each; f32
// MFMA and VALU do not co-execute on gfx950, so between MFMAs every one of them waits for the matrix pipe.)
// lda, ldb < 2^24 floats (31 rows x 4 lda bytes stay below 2^31).
__device__ __forceinline__ void rr_dma_kblock(const float *pa, int64_t lda, const float *pb, int64_t ldb, float *buf, int wave,
                                              unsigned voff) {
    const rr_rsrc_t ra = rr_make_rsrc(pa, 0x7fffffffu), rb = rr_make_rsrc(pb, 0x7fffffffu);
    const unsigned lda4 = (unsigned)lda * 4u, ldb4 = (unsigned)ldb * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lr = 4 * wave + k;
        float *dst = buf + lr * GR_LD;  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)dst, 16, voff, (unsigned)lr * lda4, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + GR_TC), 16, voff, (unsigned)lr * ldb4, 0, 0);
    }
}
// the slot (k-step pair before which the wave issues its requests).  spread 2 (default): the two waves of a SIMD (w, w + 4)
// two pairs apart over pairs 0..3; spread 1: pairs 0 / 1 only (waves 0-3 / 4-7); spread 0: all right after the barrier
__device__ __forceinline__ int rr_dma_slot(int wave, int spread = 2) {
    return spread == 2 ? (wave < 4 ? wave : ((wave + 2) & 3)) : spread == 1 ? (wave >> 2) : 0;
}
inline int rr_dma_spread_env() {  // RR_DMA_SPREAD (A/B runs)
    const char *e = getenv("RR_DMA_SPREAD");
    return e ? atoi(e) : 2;
}

// ds_read2st64_b32: two dwords at byte addresses addr + O0*256 and addr + O1*256.  Written as
// inline asm because hipcc prefers to pair neighbouring columns into ds_read2_b32, whose 8-bit
// dword offsets cannot span rows, and then pays a v_add per row -- VALU time is MFMA time on
// this chip.  The compiler does not count asm loads: lds_wait() + sched_barrier precede every use.
template <int O0, int O1>
__device__ __forceinline__ float2v lds_read2st64(unsigned addr) {
    float2v r;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "i"(O0), "i"(O1));
    return r;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// MFMA operands of k-steps 2P and 2P+1 (rows 4P + h and 4P + 2 + h, h = lane >> 5 folded into the
// base addresses): a[i] = {A col block i of step 2P, of step 2P+1}, b[j] likewise.
struct KOps2 {
    float2v a[4], b[2];
    template <int P>
    __device__ __forceinline__ void load(const unsigned (&abase)[4], const unsigned (&bbase)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = lds_read2st64<32 * P, 32 * P + 16>(abase[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = lds_read2st64<32 * P, 32 * P + 16>(bbase[j]);
    }
};

// MFMAs [FIRST, LAST) of the 16 of a k-step pair, in (s, i, j) order s*8 + i*2 + j
template <int FIRST, int LAST>
__device__ __forceinline__ void gram_mfma(const KOps2 &o, floatx16 (&acc)[4][2]) {
#pragma unroll
    for (int q = FIRST; q < LAST; ++q)
        acc[(q >> 1) & 3][q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[(q >> 1) & 3][q >> 3], o.b[q & 1][q >> 3],
                                                                        acc[(q >> 1) & 3][q & 1], 0, 0, 0);
}

// 16 k-steps (8 MFMAs each) over the current tile, operands double-buffered in registers, two
// k-steps per buffer.  Pinned order per pair: [wait] [first MFMA of pair P] [LDS reads of pair
// P+1] [other 15 MFMAs of P].  The wait for P's operands sits before P+1's reads are issued (so
// it never waits for them), and the reads fly under 15 MFMAs (960 cycles).  If every read sat
// just before its use, the two waves of a SIMD -- which interleave their MFMAs 1:1 and so stay
// in lockstep -- would stall on LDS latency together.
#define RR_PAIR(P, CUR, NXT)                                   \
    lds_wait();                                                \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma<0, 1>(CUR, acc);                                 \
    __builtin_amdgcn_sched_barrier(0);                         \
    if ((P) + 1 < 8) NXT.template load<((P) + 1) & 7>(abase, bbase); \
    __builtin_amdgcn_sched_barrier(0);                         \
    gram_mfma<1, 16>(CUR, acc);                                \
    __builtin_amdgcn_sched_barrier(0);

__device__ __forceinline__ void gram_consume(unsigned cur, floatx16 (&acc)[4][2], unsigned aoff, unsigned boff) {
    unsigned abase[4], bbase[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = cur + aoff + i * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = cur + boff + j * 128;
    KOps2 o0, o1;
    o0.load<0>(abase, bbase);
    RR_PAIR(0, o0, o1) RR_PAIR(1, o1, o0) RR_PAIR(2, o0, o1) RR_PAIR(3, o1, o0)
    RR_PAIR(4, o0, o1) RR_PAIR(5, o1, o0) RR_PAIR(6, o0, o1) RR_PAIR(7, o1, o0)
}

// The same with the wave's request for the NEXT tile (`dma()`, its 8 LDS-DMA instructions) issued before k-step pair
// `slot` (wave-uniform, 0..3) instead of right after the barrier, where all 8 waves' bursts collide in the texture path and
// no wave issues MFMAs until they are through (ablation: the loop without its DMA runs at 0.954 of the peak against 0.915).
// The two waves of a SIMD get slots two pairs apart: one of them always has the matrix pipe.
template <typename DMA>
__device__ __forceinline__ void gram_consume_staggered(unsigned cur, floatx16 (&acc)[4][2], unsigned aoff, unsigned boff, int slot,
                                                       DMA dma) {
    unsigned abase[4], bbase[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = cur + aoff + i * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = cur + boff + j * 128;
    KOps2 o0, o1;
    if (slot == 0) dma();
    o0.load<0>(abase, bbase);
    RR_PAIR(0, o0, o1)
    if (slot == 1) dma();
    RR_PAIR(1, o1, o0)
    if (slot == 2) dma();
    RR_PAIR(2, o0, o1)
    if (slot == 3) dma();
    RR_PAIR(3, o1, o0)
    RR_PAIR(4, o0, o1) RR_PAIR(5, o1, o0) RR_PAIR(6, o0, o1) RR_PAIR(7, o1, o0)
}
#undef RR_PAIR

