// FastFood (Le, Sarlos, Smola) structured projection and the Walsh-Hadamard transform for gfx950.
//
//   VX = [ H( (H(x~ * B_j))[PI_j] * G_j ) * S_j * sqrt(d2) ]_{j<k},   H = natural-order WHT / d2
//   Phi = [cos VX, sin VX] / sqrt(d2 k)
//
// reference: FastFoodRBF.transform / _makeVX  (revrand/basis_functions.py:1263-1289, 1356-1371),
//            mathfun.linalg.hadamard          (revrand/mathfun/linalg.py:182-236).
//
// Kernel shape: ONE WAVE owns one block j (or 64/d2 blocks when d2 < 64) for a chunk of rows.  Its
// diagonals B_j, G_j, S_j and permutation PI_j live in registers for the whole chunk; per row the wave
// loads x~ (coalesced), runs the two length-d2 transforms with cross-lane butterflies
// (__shfl_xor -> DPP / ds_bpermute, no barriers) plus register butterflies when d2 > 64, gathers the
// permutation through a per-wave LDS line, takes cos/sin and streams 2 x d2 outputs.  The transform
// is bound by the HBM write of Phi (4 (d + 2 d2 k) bytes per row in f32).
#include <algorithm>

#include "rr_internal.h"

template <typename TC>
__device__ __forceinline__ void ff_sincos_rev(TC t, TC &s, TC &c);
template <>
__device__ __forceinline__ void ff_sincos_rev<float>(float t, float &s, float &c) {
    const float f = t - __builtin_rintf(t);
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}
template <>
__device__ __forceinline__ void ff_sincos_rev<double>(double t, double &s, double &c) { rr_sincos_rev_f64(t, s, c); }

// Butterfly partner v[lane ^ M] without LDS traffic: DPP quad permutes (M = 1, 2), masked DPP row
// shifts (M = 4, 8: lanes whose bit is clear read lane + M, the others lane - M, selected by bank
// mask) and the gfx950 half-row / half-wave swaps (M = 16, 32).
template <int M>
__device__ __forceinline__ float xor_partner(float v) {
    const int iv = __float_as_int(v);
    int r;
    if constexpr (M == 1) {
        r = __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    } else if constexpr (M == 2) {
        r = __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    } else if constexpr (M == 4) {
        r = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xF, 0x5, false);  // row_shl:4 -> banks 0,2
        r = __builtin_amdgcn_update_dpp(r, iv, 0x114, 0xF, 0xA, false);   // row_shr:4 -> banks 1,3
    } else if constexpr (M == 8) {
        r = __builtin_amdgcn_update_dpp(iv, iv, 0x108, 0xF, 0x3, false);  // row_shl:8 -> banks 0,1
        r = __builtin_amdgcn_update_dpp(r, iv, 0x118, 0xF, 0xC, false);   // row_shr:8 -> banks 2,3
    } else if constexpr (M == 16) {
        const auto p = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
        r = ((threadIdx.x >> 4) & 1) ? p[0] : p[1];
    } else {
        const auto p = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
        r = ((threadIdx.x >> 5) & 1) ? p[0] : p[1];
    }
    return __int_as_float(r);
}
template <int M>
__device__ __forceinline__ double xor_partner(double v) { return __shfl_xor(v, M, 64); }

// In-wave unnormalised natural-order WHT of length L = min(d2, 64) * R: element index of register q in
// lane l is q * 64 + (l % 64) when d2 >= 64, else l % d2 (several blocks side by side in one wave).
// Stage M: v <- partner + sgn_M * v with sgn_M = -1 on lanes whose bit M is set.
template <int M, int R, typename TC>
__device__ __forceinline__ void fwht_stage(TC (&v)[R], int lane) {
    const TC sgn = (lane & M) ? TC(-1) : TC(1);
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = fma(sgn, v[q], xor_partner<M>(v[q]));
}

template <int R, typename TC>
__device__ __forceinline__ void wave_fwht(TC (&v)[R], int lane, int lane_len) {
    if (lane_len > 1) fwht_stage<1, R, TC>(v, lane);  // lane_len is wave-uniform
    if (lane_len > 2) fwht_stage<2, R, TC>(v, lane);
    if (lane_len > 4) fwht_stage<4, R, TC>(v, lane);
    if (lane_len > 8) fwht_stage<8, R, TC>(v, lane);
    if (lane_len > 16) fwht_stage<16, R, TC>(v, lane);
    if (lane_len > 32) fwht_stage<32, R, TC>(v, lane);
#pragma unroll
    for (int s = 1; s < R; s <<= 1) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (!(q & s)) {
                const TC a = v[q], b = v[q | s];
                v[q] = a + b;
                v[q | s] = a - b;
            }
        }
    }
}

// PHI = true : out (N, 2 n) = [cos | sin] * scale          (transform)
// PHI = false: out (N, n)   = VX in radians                 (_makeVX; used to build the dense
//                                                            equivalent matrix and by tests)
template <int R, bool PHI, typename TX, typename TC, typename TO>
__global__ void __launch_bounds__(256)
rr_fastfood_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, int d, int d2, int k,
                   const TC *__restrict__ Bm, const TC *__restrict__ Gm, const int *__restrict__ PIm,
                   const TC *__restrict__ Sm, const TC *__restrict__ invls, TO *__restrict__ out, int64_t ldo,
                   TC scale, int rows_per_block) {
    __shared__ TC perm[4][64 * R];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lane_len = d2 < 64 ? d2 : 64;       // lanes spanned by one block
    const int bpw = 64 / lane_len;                // blocks per wave
    const int sub = lane / lane_len, le = lane % lane_len;
    const int j = (blockIdx.x * 4 + wave) * bpw + sub;  // FastFood block of this lane group
    const bool active = j < k;
    const int n = d2 * k;

    TC Bv[R], Gv[R], Sv[R], Lv[R];
    int Pv[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int e = q * 64 + le;
        const size_t idx = (size_t)(active ? j : 0) * d2 + e;
        Bv[q] = Bm[idx];
        Gv[q] = Gm[idx];
        Sv[q] = Sm[idx];  // S * d2^-1.5, and / (2 pi) when PHI (phase in revolutions)
        Pv[q] = PIm[idx];
        Lv[q] = (e < d) ? invls[e] : TC(0);
    }
    TC *line = perm[wave] + sub * d2;

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    // x of the next row is fetched while this row is transformed (the only HBM read of the loop)
    TX xn[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int e = q * 64 + le;
        xn[q] = (e < d && r0 < r1) ? X[r0 * ldx + e] : TX(0);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) Lv[q] *= Bv[q];  // fold the +-1 diagonal into 1/l
    for (int64_t r = r0; r < r1; ++r) {
        TC v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = (TC)xn[q] * Lv[q];
        if (r + 1 < r1) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int e = q * 64 + le;
                xn[q] = (e < d) ? X[(r + 1) * ldx + e] : TX(0);
            }
        }
        wave_fwht<R, TC>(v, lane, lane_len);
#pragma unroll
        for (int q = 0; q < R; ++q) line[q * 64 + le] = v[q];
        // same wave wrote and reads: LDS operations of a wave execute in order
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = line[Pv[q]] * Gv[q];
        wave_fwht<R, TC>(v, lane, lane_len);
        if (active) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int col = j * d2 + q * 64 + le;
                const TC ph = v[q] * Sv[q];
                if (PHI) {
                    TC s, c;
                    ff_sincos_rev<TC>(ph, s, c);
                    out[r * ldo + col] = (TO)(c * scale);
                    out[r * ldo + n + col] = (TO)(s * scale);
                } else {
                    out[r * ldo + col] = (TO)ph;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// f32 fast path for d2 = 16 R, R in {1, 2, 4, 8, 16}: one block = 16 lanes (one DPP row) x R registers, element
// e = lane16 * R + q (lane-major), four blocks side by side in a wave.  The WHT's low log2(R) bits are register
// butterflies, the next four bits DPP butterflies inside the row (quad_perm for lane bits 0, 1, bank-masked
// row_shl / row_shr for bits 2, 3) -- no permlane swaps, no LDS; pairs of registers go through the packed f32
// ALU (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  x of a row is read once per wave (coalesced) and Phi is written
// 64 x 4 contiguous floats per store instruction; both change layout through per-wave LDS lines (no barriers:
// a wave's LDS operations execute in order).  ~2.7x fewer VALU cycles per (row, block) than the lane-minor
// kernel above (which was issue-bound) -- this one is bound by the HBM write.
// ---------------------------------------------------------------------------------------------
typedef float ff2 __attribute__((ext_vector_type(2)));

// d = v[lane ^ M] + t for lane bit M (1, 2, 4, 8) inside a row of 16 lanes, the partner read through the DPP
// operand of the add itself (v_add_f32_dpp; the builtin route costs a v_mov_b32_dpp plus register copies).
// M = 4, 8: two adds with complementary bank masks, each writing the lanes whose partner lies in its direction.
// Hazard: a DPP read needs two wait states after the VALU write of that register; callers keep >= 2
// instructions between them (NOP2 = true inserts s_nop 1 for the one-register case).
template <int M, bool NOP2>
__device__ __forceinline__ float row_add_partner(float v, float t) {
    float d;
    if constexpr (NOP2) asm volatile("s_nop 1");
    if constexpr (M == 1) {
        asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(v), "v"(t));
    } else if constexpr (M == 2) {
        asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(v), "v"(t));
    } else if constexpr (M == 4) {
        asm volatile("v_add_f32_dpp %0, %1, %2 row_shl:4 row_mask:0xf bank_mask:0x5" : "=&v"(d) : "v"(v), "v"(t));
        asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(d) : "v"(v), "v"(t));
    } else {
        asm volatile("v_add_f32_dpp %0, %1, %2 row_shl:8 row_mask:0xf bank_mask:0x3" : "=&v"(d) : "v"(v), "v"(t));
        asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(d) : "v"(v), "v"(t));
    }
    return d;
}

// Sum of a value over the 16 lanes of its DPP row, in every lane: four butterfly adds.  ONE asm statement, wait states
// included -- as separate statements the compiler is free to sink the instruction that PRODUCES the input between an
// s_nop and the DPP read it protects (it did, in the mixture mode's prologue: a v_fma one instruction ahead of the read).
__device__ __forceinline__ float row_allreduce16(float a) {
    float b;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %1, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %1, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %1, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "s_nop 1"
                 : "+v"(a), "=&v"(b));
    return a;
}

// stage on lane bit M: v <- partner + sgn * v  (sgn = -1 on lanes whose bit is set)
template <int M, int R>
__device__ __forceinline__ void row_stage(float (&v)[R], float sgn) {
    float t[R];
    if constexpr (R >= 2) {
        const ff2 s2 = {sgn, sgn};
#pragma unroll
        for (int q = 0; q < R; q += 2) {
            const ff2 a = {v[q], v[q + 1]};
            const ff2 o = s2 * a;  // v_pk_mul_f32
            t[q] = o.x;
            t[q + 1] = o.y;
        }
    } else {
        t[0] = sgn * v[0];
    }
    float dnew[R];
#pragma unroll
    for (int q = 0; q < R; ++q) dnew[q] = row_add_partner<M, (R < 4)>(v[q], t[q]);
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = dnew[q];
}

// unnormalised natural-order WHT of the 16 R elements of a block (element = lane16 * R + q)
template <int R>
__device__ __forceinline__ void row_fwht(float (&v)[R], const float (&sg)[4]) {
    // register butterflies: element bits below log2(R)
    if constexpr (R >= 2) {
#pragma unroll
        for (int q = 0; q < R; q += 2) {  // bit 0: pairs inside a register pair
            const float a = v[q], b = v[q + 1];
            v[q] = a + b;
            v[q + 1] = a - b;
        }
    }
#pragma unroll
    for (int st = 2; st < R; st <<= 1) {  // bits 1..: whole register pairs, packed
#pragma unroll
        for (int q = 0; q < R; q += 2) {
            if (!(q & st)) {
                const ff2 a = {v[q], v[q + 1]}, b = {v[q | st], v[(q | st) + 1]};
                const ff2 su = a + b, di = a - b;
                v[q] = su.x; v[q + 1] = su.y;
                v[q | st] = di.x; v[(q | st) + 1] = di.y;
            }
        }
    }
    row_stage<1, R>(v, sg[0]);
    row_stage<2, R>(v, sg[1]);
    row_stage<4, R>(v, sg[2]);
    row_stage<8, R>(v, sg[3]);
}

// VEC: rows of X and of the output are aligned for VW-element vector accesses and d is a multiple of VW; FULL: k is a
// multiple of 4 (every DPP row of every wave owns a block).  With both, the row loop is ONE basic block -- loads from
// clamped addresses plus selects instead of guarded loads, unconditional stores -- so the compiler can count the
// stores issued after the next row's x load and waits with vmcnt(stores) instead of vmcnt(0): a wave no longer drains
// its own stores before it starts the next row.
//
// GM (PHI only): one Gaussian spectral-mixture component, FastFoodGM.transform (basis_functions.py:1443-1475) --
//   out (N, 4 n) = [cos(VX + mX) | sin(VX + mX) | cos(VX - mX) | sin(VX - mX)] * scale,   mX = x . mean
// with `mrev` = mean / 2 pi (d2 entries, zero beyond d).  mX of a row is a 16-lane DPP-row reduction of x o mrev on the RAW x
// (not on x / l: a length scale beyond the float32 range must not reach it), carried one row ahead next to v.
template <int R, bool PHI, bool VEC, bool FULL, typename TX, typename TO, bool GM = false>
__global__ void __launch_bounds__(256)
rr_fastfood16_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, int d, int k, const float *__restrict__ Bm,
                     const float *__restrict__ Gm, const int *__restrict__ PIm, const float *__restrict__ Sm,
                     const float *__restrict__ invls, TO *__restrict__ out, int64_t ldo, float scale,
                     int rows_per_block, const float *__restrict__ mrev = nullptr) {
    static_assert(!GM || PHI, "the mixture component has features only");
    constexpr int D2 = 16 * R;
    constexpr int VW = R >= 4 ? 4 : R;         // contiguous elements per lane and group
    constexpr int NG = R / VW;                 // groups: element(l16, q) = (q / VW) * 16 VW + l16 * VW + q % VW
    typedef TX xvec __attribute__((ext_vector_type(VW)));
    typedef TO ovec __attribute__((ext_vector_type(VW)));
    typedef float fvec __attribute__((ext_vector_type(VW)));
    __shared__ __attribute__((aligned(16))) float perm[4][4 * D2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, sub = lane >> 4;
    const int jb = (blockIdx.x * 4 + wave) * 4;        // first FastFood block of this wave
    const int j = jb + sub;                            // block of this DPP row
    if (jb >= k) return;                               // (whole wave beyond the last block; the kernel has no barriers)
    const bool active = FULL || j < k;
    const int n = D2 * k;
    // x rows hold d <= d2 elements; an output row holds n (VX) or 2 n (Phi) columns; every permutation entry stays
    // inside its block
    RR_DEV_ASSERT(d <= D2 && d <= ldx && (GM ? 4 : PHI ? 2 : 1) * (int64_t)n <= ldo && (!FULL || k % 4 == 0) &&
                  (!VEC || (d % VW == 0 && ldx % VW == 0 && ldo % VW == 0)));

    int el[R];  // element of register q
#pragma unroll
    for (int q = 0; q < R; ++q) el[q] = (q / VW) * (16 * VW) + l16 * VW + (q % VW);
    float Lv[R], Gv[R], Sv[R], Mv[GM ? R : 1];
    int Pv[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const size_t idx = (size_t)(active ? j : 0) * D2 + el[q];
        Lv[q] = (el[q] < d ? invls[el[q]] : 0.f) * Bm[idx];  // +-1 diagonal folded into 1/l; 0 beyond d
        Gv[q] = Gm[idx];
        Sv[q] = Sm[idx];  // S * d2^-1.5 (/ 2 pi when PHI: phase in revolutions)
        Pv[q] = PIm[idx];
        if (GM) Mv[q] = el[q] < d ? mrev[el[q]] : 0.f;
        RR_DEV_ASSERT(Pv[q] >= 0 && Pv[q] < D2);
    }
    float sg[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) sg[b] = (l16 >> b) & 1 ? -1.f : 1.f;
    float *line = perm[wave] + sub * D2;

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    if (r0 >= r1) return;
    // x: every lane loads its own elements (16 lanes x VW contiguous values per group, the 4 blocks of the wave the same
    // addresses), one row ahead.  Elements >= d are read from a clamped address of the same row and meet Lv = 0 (no
    // select on the loaded value: its first use stays a whole iteration away from the load; a non-finite x makes the
    // whole output row non-finite either way, as in the reference)
    int xoff[R];
#pragma unroll
    for (int q = 0; q < R; ++q) xoff[q] = el[q] < d ? el[q] : 0;
    auto load_x = [&](int64_t r, float (&xg)[R]) {
        const TX *xr = X + r * ldx;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (VEC) {
                const xvec t = *reinterpret_cast<const xvec *>(xr + xoff[g * VW]);  // d % VW == 0: a group is in or out
#pragma unroll
                for (int w = 0; w < VW; ++w) xg[g * VW + w] = (float)t[w];
            } else {
#pragma unroll
                for (int w = 0; w < VW; ++w) xg[g * VW + w] = (float)xr[xoff[g * VW + w]];
            }
        }
    };
    // x is loaded TWO rows ahead (two register sets, the row loop unrolled by two so that they swap roles without a
    // copy).  On gfx950 loads and stores retire through one in-order counter: waiting for a row's x also waits for every
    // store issued before that load, so the load of row r + 2 goes out at the top of row r and is consumed at the END of
    // row r + 1 -- the stores of rows r and r + 1 stay in flight across it (vmcnt(10)), only row r - 1's must have landed
    float xa[R], xb[R], v[R];
    float mx = 0.f;  // GM: x . mean / 2 pi of the row v belongs to (every lane of the DPP row holds the sum)
    auto row_dot_mean = [&](const float (&xr)[R]) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < R; ++q) t = fmaf(xr[q], Mv[GM ? q : 0], t);
        return row_allreduce16(t);
    };
    const int64_t rl = r1 - 1;
    load_x(r0, xa);
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = xa[q] * Lv[q];
    if (GM) mx = row_dot_mean(xa);
    load_x(r0 + 1 < rl ? r0 + 1 : rl, xa);
    // s_waitcnt vmcnt(0): every table load above has landed before the loop.  The compiler's wait insertion is not path
    // sensitive: a table register first used inside the loop would get a wait there that, on all later iterations,
    // waits for the previous row's stores
    __builtin_amdgcn_s_waitcnt(0x0F70);
    auto one_row = [&](int64_t r, float (&cur)[R], float (&nxt)[R]) {
        load_x(r + 2 < rl ? r + 2 : rl, nxt);
        row_fwht<R>(v, sg);
        // the permutation: through an LDS line of the block (16 lanes x 16 B contiguous per group: no bank conflicts on
        // the way in; the gather reads single words).  Same wave wrote and reads: LDS operations of a wave execute in order
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            fvec t;
#pragma unroll
            for (int w = 0; w < VW; ++w) t[w] = v[g * VW + w];
            *reinterpret_cast<fvec *>(line + g * 16 * VW + l16 * VW) = t;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = line[Pv[q]] * Gv[q];
        row_fwht<R>(v, sg);
        // a group of a lane is VW contiguous outputs, the 16 lanes of a block 16 VW contiguous ones: whole 128-byte lines
        // straight from the registers, as non-temporal stores (a write-once stream far larger than the L2: 3.23 -> 3.00 ms
        // per 262144 x 16384 chunk at config 4's shape)
        float c1[R], s1[R], c2[GM ? R : 1], s2[GM ? R : 1];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (GM) {
                const float ph = v[q] * Sv[q];
                ff_sincos_rev<float>(ph + mx, s1[q], c1[q]);
                ff_sincos_rev<float>(ph - mx, s2[q], c2[q]);
                c1[q] *= scale;
                s1[q] *= scale;
                c2[q] *= scale;
                s2[q] *= scale;
            } else if (PHI) {
                ff_sincos_rev<float>(v[q] * Sv[q], s1[q], c1[q]);
                c1[q] *= scale;
                s1[q] *= scale;
            } else {
                c1[q] = v[q] * Sv[q];
            }
        }
        if (active) {
            TO *orow = out + r * ldo + (int64_t)j * D2 + l16 * VW;
            auto store_block = [&](int half, const auto &src) {  // one [D2]-wide block of column block `half` of the row
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    TO *o = orow + (int64_t)half * n + g * 16 * VW;
                    if (VEC) {
                        ovec t;
#pragma unroll
                        for (int w = 0; w < VW; ++w) t[w] = (TO)src[g * VW + w];
                        __builtin_nontemporal_store(t, reinterpret_cast<ovec *>(o));
                    } else {
#pragma unroll
                        for (int w = 0; w < VW; ++w) __builtin_nontemporal_store((TO)src[g * VW + w], &o[w]);
                    }
                }
            };
            store_block(0, c1);
            if constexpr (PHI) store_block(1, s1);
            if constexpr (GM) {
                store_block(2, c2);
                store_block(3, s2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist the use of cur to just below its load)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[q] = cur[q] * Lv[q];
            asm volatile("" : "+v"(v[q]));  // the PRODUCT is what the next row carries (else the multiply sinks into the
                                            // next iteration and the loaded registers are copied, i.e. waited for, here)
        }
        if (GM) mx = row_dot_mean(cur);
    };
    // an odd row count computes and stores its last row twice (same values) instead of branching inside the loop
    for (int64_t r = r0; r < r1; r += 2) {
        one_row(r, xa, xb);
        one_row(r + 1 < rl ? r + 1 : rl, xb, xa);
    }
}

// ---------------------------------------------------------------------------------------------
// The lane-major layout in FLOAT64 arithmetic (dtype = "f64" bases; round 2): same data flow as rr_fastfood16_kernel --
// one block per DPP row of 16 lanes x R registers, coalesced x through an LDS line, the gather and the output layout
// change through per-wave LDS lines, 1 KiB-contiguous stores -- with float64 registers.  There is no 64-bit DPP ALU
// operand, so a butterfly partner travels as two v_mov_b32_dpp (four for lane bits 2 and 3: masked row shifts both
// ways) followed by one v_fma_f64; the register-level stages are plain adds.  The lane-minor kernel it replaces
// (ds_bpermute butterflies through the LDS pipe) reached 2.66 TB/s of float64 Phi at config 4's shape.
// ---------------------------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ int dpp_partner32(int x) {
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);        // quad_perm:[1,0,3,2]
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
    else if constexpr (M == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);                  // row_shl:4, banks 0 and 2
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);                         // row_shr:4, banks 1 and 3
    } else {
        const int t = __builtin_amdgcn_update_dpp(0, x, 0x108, 0xf, 0x3, false);                  // row_shl:8, banks 0 and 1
        return __builtin_amdgcn_update_dpp(t, x, 0x118, 0xf, 0xc, false);                         // row_shr:8, banks 2 and 3
    }
}

template <int M>
__device__ __forceinline__ double dpp_partner64(double v) {
    const int lo = dpp_partner32<M>(__double2loint(v)), hi = dpp_partner32<M>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// unnormalised natural-order WHT of the 16 R elements of a block (element = lane16 * R + q), float64
template <int R>
__device__ __forceinline__ void row_fwht64(double (&v)[R], const double (&sg)[4]) {
#pragma unroll
    for (int st = 1; st < R; st <<= 1) {  // element bits below log2(R): register butterflies
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (!(q & st)) {
                const double a = v[q], b = v[q | st];
                v[q] = a + b;
                v[q | st] = a - b;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = fma(sg[0], v[q], dpp_partner64<1>(v[q]));  // partner + sgn * v
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = fma(sg[1], v[q], dpp_partner64<2>(v[q]));
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = fma(sg[2], v[q], dpp_partner64<4>(v[q]));
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = fma(sg[3], v[q], dpp_partner64<8>(v[q]));
}

template <int R, bool PHI, bool VEC, bool FULL, typename TX, typename TO, bool GM = false>
__global__ void __launch_bounds__(256)
rr_fastfood16d_kernel(const TX *__restrict__ X, int64_t N, int64_t ldx, int d, int k, const double *__restrict__ Bm,
                      const double *__restrict__ Gm, const int *__restrict__ PIm, const double *__restrict__ Sm,
                      const double *__restrict__ invls, TO *__restrict__ out, int64_t ldo, double scale, int rows_per_block,
                      const double *__restrict__ mrev = nullptr) {
    static_assert(!GM || PHI, "the mixture component has features only");
    // same structure as rr_fastfood16_kernel (element layout in groups of VW contiguous values per lane, x straight from
    // global memory two rows ahead, outputs straight from the registers, one basic block per row), with VW = 2: 16 bytes
    // of float64 per lane and group, 256 contiguous bytes per block and group
    constexpr int D2 = 16 * R;
    constexpr int VW = R >= 2 ? 2 : 1;
    constexpr int NG = R / VW;
    typedef TX xvec __attribute__((ext_vector_type(VW)));
    typedef TO ovec __attribute__((ext_vector_type(VW)));
    typedef double dvec __attribute__((ext_vector_type(VW)));
    __shared__ __attribute__((aligned(16))) double perm[4][4 * D2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, sub = lane >> 4;
    const int jb = (blockIdx.x * 4 + wave) * 4;
    const int j = jb + sub;
    if (jb >= k) return;
    const bool active = FULL || j < k;
    const int n = D2 * k;
    RR_DEV_ASSERT(d <= D2 && d <= ldx && (GM ? 4 : PHI ? 2 : 1) * (int64_t)n <= ldo && (!FULL || k % 4 == 0) &&
                  (!VEC || (d % VW == 0 && ldx % VW == 0 && ldo % VW == 0)));

    int el[R];
#pragma unroll
    for (int q = 0; q < R; ++q) el[q] = (q / VW) * (16 * VW) + l16 * VW + (q % VW);
    double Lv[R], Gv[R], Sv[R], Mv[GM ? R : 1];
    int Pv[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const size_t idx = (size_t)(active ? j : 0) * D2 + el[q];
        Lv[q] = (el[q] < d ? invls[el[q]] : 0.0) * Bm[idx];
        Gv[q] = Gm[idx];
        Sv[q] = Sm[idx];
        Pv[q] = PIm[idx];
        if (GM) Mv[q] = el[q] < d ? mrev[el[q]] : 0.0;
        RR_DEV_ASSERT(Pv[q] >= 0 && Pv[q] < D2);
    }
    double sg[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) sg[b] = (l16 >> b) & 1 ? -1.0 : 1.0;
    double *line = perm[wave] + sub * D2;

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    if (r0 >= r1) return;
    int xoff[R];
#pragma unroll
    for (int q = 0; q < R; ++q) xoff[q] = el[q] < d ? el[q] : 0;
    auto load_x = [&](int64_t r, double (&xg)[R]) {
        const TX *xr = X + r * ldx;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (VEC) {
                const xvec t = *reinterpret_cast<const xvec *>(xr + xoff[g * VW]);
#pragma unroll
                for (int w = 0; w < VW; ++w) xg[g * VW + w] = (double)t[w];
            } else {
#pragma unroll
                for (int w = 0; w < VW; ++w) xg[g * VW + w] = (double)xr[xoff[g * VW + w]];
            }
        }
    };
    double xa[R], xb[R], v[R];
    double mx = 0.0;  // GM: x . mean / 2 pi of the row v belongs to (see rr_fastfood16_kernel)
    auto row_dot_mean = [&](const double (&xr)[R]) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < R; ++q) t = fma(xr[q], Mv[GM ? q : 0], t);
        t += dpp_partner64<1>(t);
        t += dpp_partner64<2>(t);
        t += dpp_partner64<4>(t);
        t += dpp_partner64<8>(t);
        return t;
    };
    const int64_t rl = r1 - 1;
    load_x(r0, xa);
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = xa[q] * Lv[q];
    if (GM) mx = row_dot_mean(xa);
    load_x(r0 + 1 < rl ? r0 + 1 : rl, xa);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see rr_fastfood16_kernel
    auto one_row = [&](int64_t r, double (&cur)[R], double (&nxt)[R]) {
        load_x(r + 2 < rl ? r + 2 : rl, nxt);
        row_fwht64<R>(v, sg);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            dvec t;
#pragma unroll
            for (int w = 0; w < VW; ++w) t[w] = v[g * VW + w];
            *reinterpret_cast<dvec *>(line + g * 16 * VW + l16 * VW) = t;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = line[Pv[q]] * Gv[q];
        row_fwht64<R>(v, sg);
        double c1[R], s1[R], c2[GM ? R : 1], s2[GM ? R : 1];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (GM) {
                const double ph = v[q] * Sv[q];
                rr_sincos_rev_f64(ph + mx, s1[q], c1[q]);
                rr_sincos_rev_f64(ph - mx, s2[q], c2[q]);
                c1[q] *= scale;
                s1[q] *= scale;
                c2[q] *= scale;
                s2[q] *= scale;
            } else if (PHI) {
                rr_sincos_rev_f64(v[q] * Sv[q], s1[q], c1[q]);
                c1[q] *= scale;
                s1[q] *= scale;
            } else {
                c1[q] = v[q] * Sv[q];
            }
        }
        if (active) {
            TO *orow = out + r * ldo + (int64_t)j * D2 + l16 * VW;
            auto store_block = [&](int half, const auto &src) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    TO *o = orow + (int64_t)half * n + g * 16 * VW;
                    if (VEC) {
                        ovec t;
#pragma unroll
                        for (int w = 0; w < VW; ++w) t[w] = (TO)src[g * VW + w];
                        __builtin_nontemporal_store(t, reinterpret_cast<ovec *>(o));
                    } else {
#pragma unroll
                        for (int w = 0; w < VW; ++w) __builtin_nontemporal_store((TO)src[g * VW + w], &o[w]);
                    }
                }
            };
            store_block(0, c1);
            if constexpr (PHI) store_block(1, s1);
            if constexpr (GM) {
                store_block(2, c2);
                store_block(3, s2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[q] = cur[q] * Lv[q];
            asm volatile("" : "+v"(v[q]));
        }
        if (GM) mx = row_dot_mean(cur);
    };
    for (int64_t r = r0; r < r1; r += 2) {
        one_row(r, xa, xb);
        one_row(r + 1 < rl ? r + 1 : rl, xb, xa);
    }
}

// mathfun.linalg.hadamard: rows x n (n = 2^p <= 4096), natural order, normalised by 1/n; optional
// sequency reordering (linalg.py:223-236).  One workgroup per row, butterflies in LDS.
template <typename TC>
__global__ void __launch_bounds__(256)
rr_hadamard_kernel(const TC *__restrict__ Y, int64_t rows, int n, int ordering, TC *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC *buf = reinterpret_cast<TC *>(smem_raw);
    const int64_t r = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) buf[i] = Y[r * n + i];
    __syncthreads();
    for (int h = 1; h < n; h <<= 1) {
        for (int p = threadIdx.x; p < n / 2; p += 256) {
            const int i = ((p / h) * 2 * h) + (p % h);
            const TC a = buf[i], b = buf[i + h];
            buf[i] = a + b;
            buf[i + h] = a - b;
        }
        __syncthreads();
    }
    const TC inv = TC(1) / (TC)n;
    int bits = 0;
    while ((1 << bits) < n) ++bits;
    for (int i = threadIdx.x; i < n; i += 256) {
        int src = i;
        if (ordering) {  // out[i] = natural[bit-reversed Gray code of i]
            const unsigned g = (unsigned)i ^ ((unsigned)i >> 1);
            src = bits ? (int)(__brev(g) >> (32 - bits)) : 0;
        }
        out[r * n + i] = buf[src] * inv;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static size_t ff_dtype_size(int t) { return t == RR_F32 ? 4 : 8; }

template <bool PHI, typename TX, typename TC, typename TO, bool GM = false>
static int ff_launch(rr_basis *b, const void *dX, int64_t N, int64_t ldx, void *dOut, int64_t ldo) {
    rr_ctx *c = b->ctx;
    const int d2 = b->ff_d2, k = b->ff_k;
    const int lane_len = d2 < 64 ? d2 : 64;
    const int bpw = 64 / lane_len;
    const int R = d2 <= 64 ? 1 : d2 / 64;
    const unsigned gx = (unsigned)((k + 4 * bpw - 1) / (4 * bpw));
    int64_t rpb = (N * gx + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
    if (rpb < 8) rpb = 8;
    if (rpb > 512) rpb = 512;
    if ((N + rpb - 1) / rpb > 65535) rpb = (N + 65534) / 65535;
    const dim3 grid(gx, (unsigned)((N + rpb - 1) / rpb));
    const bool f32 = sizeof(TC) == 4;
    const TC *Bm = (const TC *)(f32 ? (void *)b->ffB32 : (void *)b->ffB64);
    const TC *Gm = (const TC *)(f32 ? (void *)b->ffG32 : (void *)b->ffG64);
    const TC *Sm = (const TC *)(f32 ? (void *)(PHI ? b->ffSrev32 : b->ffSrad32) : (void *)(PHI ? b->ffSrev64 : b->ffSrad64));
    const TC *Lm = (const TC *)(f32 ? (void *)b->ffL32 : (void *)b->ffL64);
    const TC *Mm = (const TC *)(f32 ? (void *)b->dmu32 : (void *)b->dmu64);  // GM: mean / 2 pi (ff_prepare_mean)
    const TC scale = (TC)(1.0 / sqrt((GM ? 2.0 : 1.0) * (double)b->n));
    if (GM && (d2 < 16 || d2 > 256)) {
        rr_set_error("fastfood: the mixture-component chain kernel serves 16 <= d2 <= 256 (d2 = %d: use the dense route)", d2);
        return RR_ERR_UNSUPPORTED;
    }
    if constexpr (sizeof(TC) == 8) {  // float64 arithmetic: the lane-major kernel with float64 registers
        static const bool old_kernel64 = getenv("RR_FASTFOOD_OLD") != nullptr;
        if ((GM || !old_kernel64) && d2 >= 16 && d2 <= 256) {
            const unsigned gx16 = (unsigned)((k + 15) / 16);
            const int vw = d2 / 16 >= 2 ? 2 : 1;
            const bool vec = (ldx % vw == 0) && (ldo % vw == 0) && (b->d % vw == 0) && ((uintptr_t)dX % (vw * sizeof(TX)) == 0) &&
                             ((uintptr_t)dOut % (vw * sizeof(TO)) == 0) && !getenv("RR_FF_NO_VEC");
            const bool full = (k % 4 == 0);
            auto rows_per_wg = [&](int occ) {  // whole rounds of equal-cost workgroups, at most 512 rows each
                const int64_t per = std::max<int64_t>(1, (int64_t)c->num_cu * occ / gx16);
                const int64_t m = std::max<int64_t>(1, (N + per * 512 - 1) / (per * 512));
                int64_t rp = (N + per * m - 1) / (per * m);
                if (rp < 4) rp = 4;
                if ((N + rp - 1) / rp > 65535) rp = (N + 65534) / 65535;
                return rp;
            };
#define RR_FF16D_(RR, VEC, FULL)                                                                                     \
    do {                                                                                                             \
        auto kern = rr_fastfood16d_kernel<RR, PHI, VEC, FULL, TX, TO, GM>;                                           \
        static int occ = 0;                                                                                          \
        if (!occ && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0) != hipSuccess || occ < 1)) occ = 2; \
        const int64_t rp = rows_per_wg(occ);                                                                         \
        hipLaunchKernelGGL(kern, dim3(gx16, (unsigned)((N + rp - 1) / rp)), dim3(256), 0, c->stream, (const TX *)dX, N, ldx, \
                           b->d, k, (const double *)Bm, (const double *)Gm, b->ffPI, (const double *)Sm,             \
                           (const double *)Lm, (TO *)dOut, ldo, (double)scale, (int)rp, (const double *)Mm);         \
    } while (0)
#define RR_FF16D(RR)                                \
    do {                                            \
        if (vec && full) RR_FF16D_(RR, true, true); \
        else if (vec) RR_FF16D_(RR, true, false);   \
        else RR_FF16D_(RR, false, false);           \
    } while (0)
            switch (d2 / 16) {
                case 1: RR_FF16D(1); break;
                case 2: RR_FF16D(2); break;
                case 4: RR_FF16D(4); break;
                case 8: RR_FF16D(8); break;
                default: RR_FF16D(16); break;
            }
#undef RR_FF16D
#undef RR_FF16D_
            RR_CHECK_HIP(hipGetLastError());
            return RR_OK;
        }
    }
    if constexpr (sizeof(TC) == 4) {  // lane-major packed kernel: one block per DPP row of 16 lanes
        static const bool old_kernel = getenv("RR_FASTFOOD_OLD") != nullptr;
        if ((GM || !old_kernel) && d2 >= 16 && d2 <= 256) {
            const unsigned gx16 = (unsigned)((k + 15) / 16);
            // vector loads / stores of a lane's VW = min(4, d2 / 16) contiguous elements need that alignment of the rows
            const int vw = d2 / 16 >= 4 ? 4 : d2 / 16;
            const bool vec = (ldx % vw == 0) && (ldo % vw == 0) && (b->d % vw == 0) && ((uintptr_t)dX % (vw * sizeof(TX)) == 0) &&
                             ((uintptr_t)dOut % (vw * sizeof(TO)) == 0) && !getenv("RR_FF_NO_VEC");
            const bool full = (k % 4 == 0);
            // whole rounds of equal-cost workgroups: occ of this instance fit a CU (its registers decide: 5 at d2 = 128);
            // rows per workgroup so that the grid is m x (CUs x occ) with at most 512 rows each
            auto rows_per_wg = [&](int occ) {
                const int64_t per = std::max<int64_t>(1, (int64_t)c->num_cu * occ / gx16);  // row blocks per round
                const int64_t m = std::max<int64_t>(1, (N + per * 512 - 1) / (per * 512));
                int64_t rp = (N + per * m - 1) / (per * m);
                if (rp < 4) rp = 4;
                if (getenv("RR_FF_ROWS_PER_BLOCK")) rp = std::max(1, atoi(getenv("RR_FF_ROWS_PER_BLOCK")));  // measurement only
                if ((N + rp - 1) / rp > 65535) rp = (N + 65534) / 65535;
                return rp;
            };
#define RR_FF16_(RR, VEC, FULL)                                                                                      \
    do {                                                                                                             \
        auto kern = rr_fastfood16_kernel<RR, PHI, VEC, FULL, TX, TO, GM>;                                            \
        static int occ = 0;                                                                                          \
        if (!occ && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0) != hipSuccess || occ < 1)) occ = 4; \
        const int64_t rp = rows_per_wg(occ);                                                                         \
        hipLaunchKernelGGL(kern, dim3(gx16, (unsigned)((N + rp - 1) / rp)), dim3(256), 0, c->stream, (const TX *)dX, N, ldx, \
                           b->d, k, (const float *)Bm, (const float *)Gm, b->ffPI, (const float *)Sm, (const float *)Lm, \
                           (TO *)dOut, ldo, (float)scale, (int)rp, (const float *)Mm);                               \
    } while (0)
#define RR_FF16(RR)                                \
    do {                                           \
        if (vec && full) RR_FF16_(RR, true, true); \
        else if (vec) RR_FF16_(RR, true, false);   \
        else RR_FF16_(RR, false, false);           \
    } while (0)
            switch (d2 / 16) {
                case 1: RR_FF16(1); break;
                case 2: RR_FF16(2); break;
                case 4: RR_FF16(4); break;
                case 8: RR_FF16(8); break;
                default: RR_FF16(16); break;
            }
#undef RR_FF16
#undef RR_FF16_
            RR_CHECK_HIP(hipGetLastError());
            return RR_OK;
        }
    }
#define RR_FF(RR)                                                                                              \
    hipLaunchKernelGGL((rr_fastfood_kernel<RR, PHI, TX, TC, TO>), grid, dim3(256), 0, c->stream, (const TX *)dX, N, \
                       ldx, b->d, d2, k, Bm, Gm, b->ffPI, Sm, Lm, (TO *)dOut, ldo, scale, (int)rpb)
    switch (R) {
        case 1: RR_FF(1); break;
        case 2: RR_FF(2); break;
        case 4: RR_FF(4); break;
        default: rr_set_error("fastfood: d2=%d is not supported", d2); return RR_ERR_UNSUPPORTED;
    }
#undef RR_FF
    RR_CHECK_HIP(hipGetLastError());
    return RR_OK;
}

template <bool PHI, bool GM = false>
static int ff_dispatch(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, void *dOut, int out_dtype,
                       int64_t ldo) {
    const int key = x_dtype * 4 + b->compute * 2 + out_dtype;
    switch (key) {
        case 0: return ff_launch<PHI, float, float, float, GM>(b, dX, N, ldx, dOut, ldo);
        case 1: return ff_launch<PHI, float, float, double, GM>(b, dX, N, ldx, dOut, ldo);
        case 2: return ff_launch<PHI, float, double, float, GM>(b, dX, N, ldx, dOut, ldo);
        case 3: return ff_launch<PHI, float, double, double, GM>(b, dX, N, ldx, dOut, ldo);
        case 4: return ff_launch<PHI, double, float, float, GM>(b, dX, N, ldx, dOut, ldo);
        case 5: return ff_launch<PHI, double, float, double, GM>(b, dX, N, ldx, dOut, ldo);
        case 6: return ff_launch<PHI, double, double, float, GM>(b, dX, N, ldx, dOut, ldo);
        case 7: return ff_launch<PHI, double, double, double, GM>(b, dX, N, ldx, dOut, ldo);
    }
    rr_set_error("fastfood: bad dtype combination");
    return RR_ERR_INVALID;
}

// upload mean / 2 pi (per input dimension, zero beyond d) for the mixture-component mode of the chain kernels
static int ff_prepare_mean(rr_basis *b, const double *mean) {
    RR_REQUIRE(mean != nullptr, "mean: null argument");
    const double inv2pi = 0.15915494309189533576888;
    std::vector<double> m64(b->ff_d2, 0.0);
    std::vector<float> m32(b->ff_d2, 0.f);
    for (int i = 0; i < b->d; ++i) {
        RR_REQUIRE(mean[i] == mean[i], "mean[%d] is NaN", i);
        m64[i] = mean[i] * inv2pi;
        const double lim = 3.0e38;  // (an infinite mean: the reference's features are NaN there; keep float32 finite)
        m32[i] = (float)(m64[i] > lim ? lim : (m64[i] < -lim ? -lim : m64[i]));
    }
    if (!b->dmu32) RR_CHECK_HIP(hipMalloc((void **)&b->dmu32, (size_t)b->ff_d2 * 4));
    if (!b->dmu64) RR_CHECK_HIP(hipMalloc((void **)&b->dmu64, (size_t)b->ff_d2 * 8));
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    RR_CHECK_HIP(hipMemcpy(b->dmu64, m64.data(), m64.size() * 8, hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->dmu32, m32.data(), m32.size() * 4, hipMemcpyHostToDevice));
    return RR_OK;
}

// upload 1/l_i (per input dimension) for the FWHT kernels
static int ff_prepare_lenscale(rr_basis *b, const double *lenscale, int n_ls) {
    RR_REQUIRE(lenscale != nullptr, "lenscale: null argument");
    RR_REQUIRE(n_ls == 1 || n_ls == b->d, "Dimension of input parameter is inconsistent! (n_ls=%d, d=%d)", n_ls, b->d);
    std::vector<double> l64(b->ff_d2, 0.0);
    std::vector<float> l32(b->ff_d2, 0.f);
    for (int i = 0; i < b->d; ++i) {
        const double l = lenscale[n_ls == 1 ? 0 : i];
        RR_REQUIRE(l != 0.0 && l == l, "lenscale[%d] is %g", i, l);
        l64[i] = 1.0 / l;
        // The f32 copy is clamped to a finite range (like the random Fourier kernels' scaled W, rr_api.hip): the optimiser's
        // log-space bounds let a length scale reach 1e-100 (optimize/decorators.py:18), where the reference's float64 phases
        // are finite noise; a float32 phase from x / l beyond 2^24 has no fractional part left either way, but 1 / l = inf
        // would turn the whole chain into inf - inf = NaN.  2^40 leaves the chain's intermediates (x d2 |G| d2 S sqrt(d2))
        // far inside the float32 range.
        const double lim = 1099511627776.0;  // 2^40
        l32[i] = (float)(l64[i] > lim ? lim : (l64[i] < -lim ? -lim : l64[i]));
    }
    RR_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    RR_CHECK_HIP(hipMemcpy(b->ffL64, l64.data(), l64.size() * 8, hipMemcpyHostToDevice));
    RR_CHECK_HIP(hipMemcpy(b->ffL32, l32.data(), l32.size() * 4, hipMemcpyHostToDevice));
    return RR_OK;
}

// common host-buffer driver: stream rows up, run the kernel, stream the output down
template <bool PHI, bool GM = false>
static int ff_host_call(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                        int n_ls, void *out, int out_dtype, int64_t ldo, const char *who, const double *mean = nullptr) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_FASTFOOD, "%s: not a FastFood basis", who);
    RR_REQUIRE((x_dtype == RR_F32 || x_dtype == RR_F64) && (out_dtype == RR_F32 || out_dtype == RR_F64), "%s: bad dtype", who);
    const int64_t width = (GM ? 4 : PHI ? 2 : 1) * (int64_t)b->n;
    RR_REQUIRE(N >= 0 && ldx >= b->d && ldo >= width, "%s: bad shape", who);
    rr_ctx *c = b->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    int rc = ff_prepare_lenscale(b, lenscale, n_ls);
    if (rc == RR_OK && GM) rc = ff_prepare_mean(b, mean);
    if (rc != RR_OK || N == 0) return rc;
    RR_REQUIRE(X != nullptr && out != nullptr, "%s: null buffer", who);
    const size_t xs = ff_dtype_size(x_dtype), os = ff_dtype_size(out_dtype);
    int64_t chunk = (int64_t)(((size_t)256 << 20) / ((size_t)b->d * xs + (size_t)width * os));  // pipelined chunks
    if (chunk < 1) chunk = 1;
    if (chunk > N) chunk = N;
    void *dX = nullptr, *dO = nullptr;
    if (hipMalloc(&dX, (size_t)chunk * b->d * xs) != hipSuccess || hipMalloc(&dO, (size_t)chunk * width * os) != hipSuccess) {
        (void)hipGetLastError();
        if (dX) (void)hipFree(dX);
        rr_set_error("%s: device allocation failed", who);
        return RR_ERR_OOM;
    }
    rr_host_sink sink;
    rc = rr_sink_open(c, (size_t)chunk * width * os, &sink);
    for (int64_t r0 = 0; r0 < N && rc == RR_OK; r0 += chunk) {
        const int64_t m = (N - r0 < chunk) ? N - r0 : chunk;
        hipError_t e = hipMemcpy2DAsync(dX, (size_t)b->d * xs, (const char *)X + (size_t)r0 * ldx * xs, (size_t)ldx * xs,
                                        (size_t)b->d * xs, (size_t)m, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            rr_set_error("%s: upload failed: %s", who, hipGetErrorString(e));
            rc = RR_ERR_HIP;
            break;
        }
        rc = ff_dispatch<PHI, GM>(b, dX, x_dtype, m, b->d, dO, out_dtype, width);
        if (rc != RR_OK) break;
        rc = rr_sink_push(&sink, dO, (char *)out + (size_t)r0 * ldo * os, (size_t)m, (size_t)width * os, (size_t)ldo * os);
    }
    if (rc == RR_OK) rc = rr_sink_close(&sink);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dX);
    (void)hipFree(dO);
    return rc;
}

extern "C" {

int rr_fastfood_create(rr_ctx *ctx, int compute, int d, int d2, int k, const int64_t *B, const double *G,
                       const int64_t *PI, const double *S, rr_basis **out) {
    RR_REQUIRE(ctx != nullptr && out != nullptr && B && G && PI && S, "rr_fastfood_create: null argument");
    *out = nullptr;
    RR_REQUIRE(d >= 1 && k >= 1 && d2 >= d && (d2 & (d2 - 1)) == 0, "rr_fastfood_create: need d2 = 2^p >= d, k >= 1");
    RR_REQUIRE(d2 <= 256, "rr_fastfood_create: d2=%d > 256 is not supported yet", d2);
    RR_REQUIRE(compute == RR_F32 || compute == RR_F64, "rr_fastfood_create: bad compute dtype %d", compute);
    for (size_t i = 0; i < (size_t)k * d2; ++i)
        RR_REQUIRE(PI[i] >= 0 && PI[i] < d2 && (B[i] == 1 || B[i] == -1), "rr_fastfood_create: bad B/PI entry at %zu", i);
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    const size_t cnt = (size_t)k * d2;
    const double norm = pow((double)d2, -1.5);  // (1/d2)(1/d2) sqrt(d2): both WHT normalisations and :1368
    const double inv2pi = 0.15915494309189533576888;
    std::vector<double> b64(cnt), g64(G, G + cnt), srad64(cnt), srev64(cnt);
    std::vector<float> b32(cnt), g32(cnt), srad32(cnt), srev32(cnt);
    std::vector<int> pi(cnt);
    for (size_t i = 0; i < cnt; ++i) {
        b64[i] = (double)B[i];
        srad64[i] = S[i] * norm;
        srev64[i] = srad64[i] * inv2pi;
        b32[i] = (float)b64[i];
        g32[i] = (float)g64[i];
        srad32[i] = (float)srad64[i];
        srev32[i] = (float)srev64[i];
        pi[i] = (int)PI[i];
    }
    rr_basis *b = new rr_basis();
    b->ctx = ctx;
    b->kind = RR_KIND_FASTFOOD;
    b->compute = compute;
    b->d = d;
    b->n = d2 * k;
    b->ff_d2 = d2;
    b->ff_k = k;
    hipError_t e = hipSuccess;
#define RR_UP(dst, vec)                                                                       \
    if (e == hipSuccess) e = hipMalloc((void **)&b->dst, (vec).size() * sizeof((vec)[0]));     \
    if (e == hipSuccess) e = hipMemcpy(b->dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice);
    RR_UP(ffB32, b32) RR_UP(ffB64, b64) RR_UP(ffG32, g32) RR_UP(ffG64, g64) RR_UP(ffSrad32, srad32)
    RR_UP(ffSrad64, srad64) RR_UP(ffSrev32, srev32) RR_UP(ffSrev64, srev64) RR_UP(ffPI, pi)
#undef RR_UP
    if (e == hipSuccess) e = hipMalloc((void **)&b->ffL32, (size_t)d2 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&b->ffL64, (size_t)d2 * 8);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("rr_fastfood_create: device allocation failed: %s", hipGetErrorString(e));
        rr_basis_destroy(b);
        return RR_ERR_OOM;
    }
    *out = b;
    return RR_OK;
}

int rr_fastfood_transform(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                          int n_ls, void *Phi, int out_dtype, int64_t ldphi) {
    return ff_host_call<true>(b, X, x_dtype, N, ldx, lenscale, n_ls, Phi, out_dtype, ldphi, "rr_fastfood_transform");
}

int rr_fastfood_transform_dev(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *lenscale,
                              int n_ls, void *dPhi, int out_dtype, int64_t ldphi) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_FASTFOOD, "rr_fastfood_transform_dev: not a FastFood basis");
    RR_REQUIRE((x_dtype == RR_F32 || x_dtype == RR_F64) && (out_dtype == RR_F32 || out_dtype == RR_F64),
               "rr_fastfood_transform_dev: bad dtype");
    RR_REQUIRE(N >= 0 && ldx >= b->d && ldphi >= 2 * (int64_t)b->n, "rr_fastfood_transform_dev: bad shape");
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    int rc = ff_prepare_lenscale(b, lenscale, n_ls);
    if (rc != RR_OK || N == 0) return rc;
    RR_REQUIRE(dX != nullptr && dPhi != nullptr, "rr_fastfood_transform_dev: null buffer");
    return ff_dispatch<true>(b, dX, x_dtype, N, ldx, dPhi, out_dtype, ldphi);
}

// The chain's Phi straight into a device feature matrix, columns [col0, col0 + 2n) (a FastFood child of a resident fit:
// `_elbo` statistics of FastFoodRBF, basis_functions.py:1263-1289 feeding slm.py:145-157).  The rows of the current
// rr_featmat_begin; f32 arithmetic (the feature matrix is float32).  P^T is not written here: consumers that want it run
// their transposing pass.
int rr_featmat_put_fastfood(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *lenscale,
                            int n_ls, int64_t col0) {
    RR_REQUIRE(fm != nullptr && b != nullptr && b->kind == RR_KIND_FASTFOOD, "rr_featmat_put_fastfood: not a FastFood basis");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_put_fastfood: bad dtype");
    RR_REQUIRE(b->ctx == fm->ctx, "rr_featmat_put_fastfood: basis and feature matrix live on different device contexts");
    RR_REQUIRE(col0 >= 0 && col0 + 2 * (int64_t)b->n <= fm->F, "rr_featmat_put_fastfood: columns out of range");
    RR_REQUIRE(ldx >= b->d, "rr_featmat_put_fastfood: bad shape");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    int rc = ff_prepare_lenscale(b, lenscale, n_ls);
    if (rc != RR_OK || fm->rows == 0) return rc;
    RR_REQUIRE(dX != nullptr, "rr_featmat_put_fastfood: null X");
    rc = rr_fm_claim(fm, col0, 2 * (int64_t)b->n, "rr_featmat_put_fastfood");
    if (rc != RR_OK) return rc;
    return ff_dispatch<true>(b, dX, x_dtype, fm->rows, ldx, fm->P + col0, RR_F32, fm->ld);
}

int rr_fastfood_gm_transform(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                             const double *lenscale, int n_ls, void *Phi, int out_dtype, int64_t ldphi) {
    return ff_host_call<true, true>(b, X, x_dtype, N, ldx, lenscale, n_ls, Phi, out_dtype, ldphi, "rr_fastfood_gm_transform", mean);
}

int rr_fastfood_gm_transform_dev(rr_basis *b, const void *dX, int x_dtype, int64_t N, int64_t ldx, const double *mean,
                                 const double *lenscale, int n_ls, void *dPhi, int out_dtype, int64_t ldphi) {
    RR_REQUIRE(b != nullptr && b->kind == RR_KIND_FASTFOOD, "rr_fastfood_gm_transform_dev: not a FastFood basis");
    RR_REQUIRE((x_dtype == RR_F32 || x_dtype == RR_F64) && (out_dtype == RR_F32 || out_dtype == RR_F64),
               "rr_fastfood_gm_transform_dev: bad dtype");
    RR_REQUIRE(N >= 0 && ldx >= b->d && ldphi >= 4 * (int64_t)b->n, "rr_fastfood_gm_transform_dev: bad shape");
    RR_CHECK_HIP(hipSetDevice(b->ctx->device));
    int rc = ff_prepare_lenscale(b, lenscale, n_ls);
    if (rc == RR_OK) rc = ff_prepare_mean(b, mean);
    if (rc != RR_OK || N == 0) return rc;
    RR_REQUIRE(dX != nullptr && dPhi != nullptr, "rr_fastfood_gm_transform_dev: null buffer");
    return ff_dispatch<true, true>(b, dX, x_dtype, N, ldx, dPhi, out_dtype, ldphi);
}

// The mixture component's four blocks straight into a device feature matrix, columns [col0, col0 + 4n): block pair
// [cos | sin](VX + mX) at col0 and [cos | sin](VX - mX) at col0 + 2n -- to the consumers of the matrix two random-Fourier
// shaped children side by side (rr_featmat_pass2_rff / rr_featmat_glm_rff at either offset).
int rr_featmat_put_fastfood_gm(rr_featmat *fm, rr_basis *b, const void *dX, int x_dtype, int64_t ldx, const double *mean,
                               const double *lenscale, int n_ls, int64_t col0) {
    RR_REQUIRE(fm != nullptr && b != nullptr && b->kind == RR_KIND_FASTFOOD, "rr_featmat_put_fastfood_gm: not a FastFood basis");
    RR_REQUIRE(x_dtype == RR_F32 || x_dtype == RR_F64, "rr_featmat_put_fastfood_gm: bad dtype");
    RR_REQUIRE(b->ctx == fm->ctx, "rr_featmat_put_fastfood_gm: basis and feature matrix live on different device contexts");
    RR_REQUIRE(col0 >= 0 && col0 + 4 * (int64_t)b->n <= fm->F, "rr_featmat_put_fastfood_gm: columns out of range");
    RR_REQUIRE(ldx >= b->d, "rr_featmat_put_fastfood_gm: bad shape");
    RR_CHECK_HIP(hipSetDevice(fm->ctx->device));
    int rc = ff_prepare_lenscale(b, lenscale, n_ls);
    if (rc == RR_OK) rc = ff_prepare_mean(b, mean);
    if (rc != RR_OK || fm->rows == 0) return rc;
    RR_REQUIRE(dX != nullptr, "rr_featmat_put_fastfood_gm: null X");
    rc = rr_fm_claim(fm, col0, 4 * (int64_t)b->n, "rr_featmat_put_fastfood_gm");
    if (rc != RR_OK) return rc;
    return ff_dispatch<true, true>(b, dX, x_dtype, fm->rows, ldx, fm->P + col0, RR_F32, fm->ld);
}

int rr_fastfood_vx(rr_basis *b, const void *X, int x_dtype, int64_t N, int64_t ldx, const double *lenscale, int n_ls,
                   void *VX, int out_dtype, int64_t ldvx) {
    return ff_host_call<false>(b, X, x_dtype, N, ldx, lenscale, n_ls, VX, out_dtype, ldvx, "rr_fastfood_vx");
}

int rr_hadamard(rr_ctx *c, const void *Y, int dtype, int64_t rows, int64_t n, int ordering, void *out) {
    RR_REQUIRE(c != nullptr, "rr_hadamard: null context");
    RR_REQUIRE(dtype == RR_F32 || dtype == RR_F64, "rr_hadamard: bad dtype");
    RR_REQUIRE(rows >= 0 && n >= 1 && (n & (n - 1)) == 0, "rr_hadamard: length must be a power of two");
    RR_REQUIRE(n <= 4096, "rr_hadamard: n=%lld > 4096 is not supported", (long long)n);
    if (rows == 0) return RR_OK;
    RR_REQUIRE(Y != nullptr && out != nullptr, "rr_hadamard: null buffer");
    RR_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = ff_dtype_size(dtype);
    const int64_t chunk = rows < 65535 ? rows : 65535;
    void *dY = nullptr, *dO = nullptr;
    if (hipMalloc(&dY, (size_t)chunk * n * es) != hipSuccess || hipMalloc(&dO, (size_t)chunk * n * es) != hipSuccess) {
        (void)hipGetLastError();
        if (dY) (void)hipFree(dY);
        rr_set_error("rr_hadamard: device allocation failed");
        return RR_ERR_OOM;
    }
    int rc = RR_OK;
    for (int64_t r0 = 0; r0 < rows && rc == RR_OK; r0 += chunk) {
        const int64_t m = rows - r0 < chunk ? rows - r0 : chunk;
        hipError_t e = hipMemcpyAsync(dY, (const char *)Y + (size_t)r0 * n * es, (size_t)m * n * es, hipMemcpyHostToDevice,
                                      c->stream);
        if (e == hipSuccess) {
            if (dtype == RR_F32)
                hipLaunchKernelGGL(rr_hadamard_kernel<float>, dim3((unsigned)m), dim3(256), (size_t)n * 4, c->stream,
                                   (const float *)dY, m, (int)n, ordering, (float *)dO);
            else
                hipLaunchKernelGGL(rr_hadamard_kernel<double>, dim3((unsigned)m), dim3(256), (size_t)n * 8, c->stream,
                                   (const double *)dY, m, (int)n, ordering, (double *)dO);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync((char *)out + (size_t)r0 * n * es, dO, (size_t)m * n * es, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rr_set_error("rr_hadamard: copy/launch failed: %s", hipGetErrorString(e));
            rc = RR_ERR_HIP;
        }
    }
    (void)hipFree(dY);
    (void)hipFree(dO);
    return rc;
}

}  // extern "C"
