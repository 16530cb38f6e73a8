// The small-minibatch regime of GeneralizedLinearModel.fit -- the reference's own defaults are batch_size = 10, K = 10,
// nsamples = 50, maxiter = 3000, nstarts = 500 (glm.py:120-124; "1 million iterations ... batch size of 10", docs/report) --
// as ONE persistent kernel that runs MANY SGD steps per launch.
//
// One step of sgd (optimize/sgd.py:337-425) o logtrick_sgd (decorators.py:329-408) o structured_sgd (:133-252) around `_elbo`
// (glm.py:205-294) at that size is ~1.3 MFLOP: through the general route (rr_glm_sgd_step: ~31 dependent launches of the
// tile kernels) it is dispatch-bound at 140-160 us however little arithmetic the kernels hold.  Here a CLUSTER of K
// workgroups -- workgroup k owns mixture component k: column k of (m, C), its L weight samples, its share of every sum --
// walks through the steps inside the kernel:
//
//   a  x = from_log(z) of ALL coordinates into LDS (every workgroup; the other columns come from what their owners published)
//   b  row k of the mixture's cross terms  q[k][l] = sum_f log(C_fk + C_fl) + (m_fk - m_fl)^2 / (C_fk + C_fl)   glm.py:697-712
//      -> published, barrier B1 ARRIVE (nobody waits yet)
//   c  the minibatch: rows gathered by index from the resident X, Phi (M, F) in float64 (every workgroup, redundantly:
//      M F sincos)                                                                             basis_functions.py:838-864
//   d  e (L, F): the caller's draws (the reference's stream, uploaded for the whole block of steps) or the counter-based
//      device generator of rr_glm_draw_kernel -- same function of (seed, step, sample, feature)
//   e  fs = ws Phi^T for the L samples of component k, ws = m_k + sqrt(C_k) e; df, loglike            glm.py:300-305,321
//   f  Edws = dfs Phi -> Edm, EdC (complete: they are sums over component k's samples only); the component's share of
//      EdPhi = dfs^T ws and of -(EdPhi o dPhi_i).sum() per length scale                            glm.py:308-311,274-275
//   g  barrier B1 WAIT; log N_kl, log z, alpha                                                      glm.py:221-222,244-246
//   h  gradient of column k, chain rule of the log trick, |g|^2, bound truncation, updater, clip   glm.py:252-256, sgd.py:404-420
//      -> new column + the component's scalars published, barrier B2
//   i  the few shared coordinates (regularisers, Gaussian variance, length scales): their gradients are sums over the
//      components' published scalars, formed and applied by EVERY workgroup identically (fixed order) -- no further
//      exchange; workgroup 0 records the step's -ELBO and |g|                                       glm.py:265-292
//
// so a step costs two device-scope barriers (one of them split-phase) instead of thirty launches.  Everything is float64
// (the draws are float32 values, as everywhere on this path).  The K workgroups must be co-resident (K <= 32 of 256 CUs).
//
// rr_glm_svi_starts evaluates the random starts of structured_sgd (decorators.py:541-583) -- nstarts candidates, each on its
// own minibatch with its own draws -- as ONE launch, one workgroup per candidate (objective only).
#include "rr_internal.h"

#include <algorithm>

#define SVI_MAXCHILD 16
#define SVI_MAXK 32
#define SVI_THREADS 512
#define SVI_WAVES (SVI_THREADS / 64)

namespace {

struct SviChild {
    int kind, col0, width, d, n, n_ls, ls0, onescol, x_f64, xoff, woff;  // xoff: first entry of this child's columns in a gathered row; woff: of its W in the LDS copy
    const void *X;
    int64_t ldx;
    const double *W;  // RFF: raw (d, n) row-major on the device
};

struct SviArgs {
    const SviChild *kid;  // the children's table in device memory (uniform loads; not in the kernel arguments, which would
                          // be copied to registers for the dynamic index)
    int nkids, F, Fp, K, L, M, lik, n_lik, n_ls, ns, updater, y_f64, dsum, wtot;  // wtot: entries of all children's W
    int64_t np, N;
    const void *y, *rowarg;
    const double *lconst;  // per row: the f-independent part of loglike (log-factorial terms), or null
    const double *bias;    // per step of this launch: Adam's 1 - beta1^t, 1 - beta2^t (sgd.py:322-323)
    double *z, *s1, *s2;
    const double *lower, *upper;
    const unsigned char *islog;
    double *pubcol, *pubrow, *pubsc;
    unsigned int *bar;
    const float *E;
    const int *idx;
    double *objs, *norms;
    int64_t t0;
    int steps;
    double up[4], bmag;
    uint64_t seed, key0;
    // rr_glm_svi_starts
    const double *cand;
    double *out;
    long long *prof;  // RR_SVI_PROF=1: workgroup 0's time per phase (100 MHz ticks), summed over the steps
};

__device__ __forceinline__ uint64_t svi_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// the draw of rr_glm_draw_kernel (rr_elbo.hip) for (seed, step, counter)
__device__ __forceinline__ float svi_draw(uint64_t stepkey, uint64_t ctr) {
    const uint64_t h = svi_splitmix64(stepkey ^ ctr);
    const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
}

__device__ __forceinline__ double svi_softplus(double f) { return fmax(f, 0.0) + log1p(exp(-fabs(f))); }
__device__ __forceinline__ double svi_expit(double f) {
    const double e = exp(-fabs(f));
    return f >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
}

// d loglike / d f and the f-dependent part of loglike (Gaussian: the squared error)   likelihoods.py:46-104,171-233,298-368,456-521
__device__ __forceinline__ void svi_lik(int lik, double f, double y, double n, double ivar, double &df, double &ll) {
    if (lik == RR_LIK_BERNOULLI) {
        df = y - svi_expit(f);
        ll = y * f - svi_softplus(f);
    } else if (lik == RR_LIK_BINOMIAL) {
        df = y - n * svi_expit(f);
        ll = y * f - n * svi_softplus(f);
    } else if (lik == RR_LIK_GAUSSIAN) {
        const double er = y - f;
        df = er * ivar;
        ll = er * er;
    } else if (lik == RR_LIK_POISSON_EXP) {
        const double g = exp(f);
        df = y - g;
        ll = y * f - g;
    } else {
        const double g = fmax(svi_softplus(f), 1e-100);
        df = svi_expit(f) * (y / g - 1.0);
        ll = y * log(g) - g;
    }
}

typedef __attribute__((address_space(3))) double ldsd;   // LDS pointers keep their address space: ds_read / ds_write, never flat
typedef __attribute__((address_space(3))) float ldsf;
typedef __attribute__((address_space(3))) unsigned char ldsb;

// The workgroup barrier for data exchanged through LDS: orders LDS accesses only (s_waitcnt lgkmcnt(0) + s_barrier), so
// global loads in flight -- the next step's minibatch rows, prefetched a step ahead -- are not waited for at every barrier
// (__syncthreads() waits for vmcnt(0) as well: the prefetch's HBM latency then lands on whatever barrier comes next).
// Nothing one wave of a workgroup writes to HBM is read back by another wave of the same workgroup.
__device__ __forceinline__ void svi_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ double svi_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the block, fixed order (the same bits in every workgroup that sums the same values); all threads get it
__device__ __forceinline__ double svi_block_sum(double v, ldsd *red) {
    const int tid = threadIdx.x;
    v = svi_wave_sum(v);
    svi_sync();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    svi_sync();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < SVI_WAVES; ++w) r += red[w];
    return r;
}

// Device-scope barrier of the K workgroups, split in two.  ARRIVE: what this workgroup publishes is stored to HBM by the
// lanes of wave 0 alone (the callers copy it out of LDS there), so wave 0's release fence -- L2 write-back, the expensive
// part -- covers it; the other waves only reach the workgroup barrier in front.  WAIT: thread 0 spins on the counter with
// acquire loads (the L1 / L2 invalidate that makes the peers' stores visible to this CU is a property of the CU, not of
// the wave that issued it; no wave has loads of published data in flight across the workgroup barrier behind it).
__device__ __forceinline__ void svi_arrive_wave0(unsigned int *ctr) {   // call from wave 0 only, after ITS stores
    __threadfence();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void svi_wait(unsigned int *ctr, unsigned int target) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();   // the full barrier (global memory too): what thread 0 acquired is ordered before every wave's reads
}

__device__ __forceinline__ double svi_xval(const SviChild &c, int64_t row, int i) {
    return c.x_f64 ? ((const double *)c.X)[row * c.ldx + i] : (double)((const float *)c.X)[row * c.ldx + i];
}

#define SVI_GREG 2  // gathered X entries a thread holds in registers between the loads and their use (M dsum <= 2 x 512)

struct SviLds {
    ldsd *xm, *xC, *xs;             // x of all coordinates: (F, K) means, (F, K) covariances, the ns shared coordinates
    ldsd *zc, *s1c, *s2c, *loc, *hic;  // this workgroup's column (2 F: means then covariances): z, updater state, bounds
    ldsd *zs, *s1s, *s2s, *los, *his;  // the shared coordinates (replicated in every workgroup)
    ldsd *Phi, *dfs, *Q, *Xb, *yb, *nb, *Dr, *mk, *sk, *edm, *edc, *q, *logz, *alpha, *gls, *psc, *red, *misc, *Wl, *ils, *Rc, *stage;
    ldsf *Ef;                       // the draws of this component's samples (L, F)
    ldsb *lg;                       // is_log of all np coordinates
};

static __host__ __device__ size_t svi_lds_layout(const SviArgs &a, size_t *off) {
    // offsets in doubles; returns the total (the float / byte arrays are placed behind, rounded to doubles)
    const size_t FK = (size_t)a.F * a.K, F = (size_t)a.F, ns = (size_t)a.ns, M = (size_t)a.M;
    const size_t cnt[] = {FK, FK, ns, 2 * F, 2 * F, 2 * F, 2 * F, 2 * F, ns, ns, ns, ns, ns, M * (size_t)a.Fp, (size_t)a.L * M, M * F,
                          M * (size_t)a.dsum, M, M, M, F, F, F, F, (size_t)a.K * a.K, (size_t)a.K, (size_t)a.K,
                          (size_t)(a.n_ls > 0 ? a.n_ls : 1), (size_t)a.K * (a.n_ls + 4), (size_t)SVI_WAVES, 8,
                          (size_t)a.wtot, (size_t)(a.n_ls > 0 ? a.n_ls : 1), (size_t)a.nkids, (size_t)(a.K + a.n_ls + 4),
                          ((size_t)a.L * F * 4 + 7) / 8, ((size_t)a.np + 7) / 8};
    size_t o = 0;
    for (size_t i = 0; i < sizeof(cnt) / sizeof(cnt[0]); ++i) {
        if (off) off[i] = o;
        o += cnt[i];
    }
    return o;
}

__device__ __forceinline__ SviLds svi_carve(double *sm, const SviArgs &a) {
    size_t off[40];
    svi_lds_layout(a, off);
    ldsd *b = (ldsd *)sm;
    SviLds s;
    int i = 0;
    s.xm = b + off[i++]; s.xC = b + off[i++]; s.xs = b + off[i++];
    s.zc = b + off[i++]; s.s1c = b + off[i++]; s.s2c = b + off[i++]; s.loc = b + off[i++]; s.hic = b + off[i++];
    s.zs = b + off[i++]; s.s1s = b + off[i++]; s.s2s = b + off[i++]; s.los = b + off[i++]; s.his = b + off[i++];
    s.Phi = b + off[i++]; s.dfs = b + off[i++]; s.Q = b + off[i++]; s.Xb = b + off[i++]; s.yb = b + off[i++]; s.nb = b + off[i++];
    s.Dr = b + off[i++]; s.mk = b + off[i++]; s.sk = b + off[i++]; s.edm = b + off[i++]; s.edc = b + off[i++];
    s.q = b + off[i++]; s.logz = b + off[i++]; s.alpha = b + off[i++]; s.gls = b + off[i++]; s.psc = b + off[i++];
    s.red = b + off[i++]; s.misc = b + off[i++];
    s.Wl = b + off[i++]; s.ils = b + off[i++]; s.Rc = b + off[i++]; s.stage = b + off[i++];
    s.Ef = (ldsf *)(b + off[i++]);
    s.lg = (ldsb *)(b + off[i++]);
    return s;
}

static size_t svi_lds_doubles(const SviArgs &a) { return svi_lds_layout(a, nullptr); }

// the minibatch's rows of X (and targets) by index: loads issued now, values parked in registers until svi_features
struct SviGather {
    double x[SVI_GREG];
    double y, n, lc;
};

__device__ __forceinline__ void svi_gather_issue(const SviArgs &a, const int *idx, SviGather &g) {
    const int tid = threadIdx.x, M = a.M, tot = M * a.dsum;
#pragma unroll
    for (int u = 0; u < SVI_GREG; ++u) {
        const int e = tid + u * SVI_THREADS;
        g.x[u] = 0.0;
        if (e < tot) {
            const int r = e / a.dsum, o = e % a.dsum;
            int c = 0;
            while (c + 1 < a.nkids && o >= a.kid[c + 1].xoff) ++c;
            g.x[u] = svi_xval(a.kid[c], (int64_t)idx[r], o - a.kid[c].xoff);
        }
    }
    g.y = g.n = g.lc = 0.0;
    if (tid < M) {
        const int64_t row = idx[tid];
        if (a.lconst) g.lc = a.lconst[row];
        g.y = a.y_f64 ? ((const double *)a.y)[row] : (double)((const float *)a.y)[row];
        if (a.rowarg) g.n = a.y_f64 ? ((const double *)a.rowarg)[row] : (double)((const float *)a.rowarg)[row];
    }
}

// c: Phi (M, F) in float64 from the gathered rows, the batch's loglike constant -> misc[0]
__device__ __forceinline__ void svi_features(const SviArgs &a, const SviLds &s, const SviGather &g) {
    const int tid = threadIdx.x, M = a.M, tot = M * a.dsum;
#pragma unroll
    for (int u = 0; u < SVI_GREG; ++u) {
        const int e = tid + u * SVI_THREADS;
        if (e < tot) s.Xb[e] = g.x[u];
    }
    if (tid < M) {
        s.yb[tid] = g.y;
        s.nb[tid] = g.n;
    }
    svi_sync();
    for (int c = 0; c < a.nkids; ++c) {
        const SviChild &k = a.kid[c];
        if (k.kind == RR_SGD_CHILD_LINEAR) {
            for (int e = tid; e < M * k.width; e += SVI_THREADS) {
                const int r = e / k.width, j = e % k.width;
                s.Phi[r * a.Fp + k.col0 + j] = (k.onescol && j == 0) ? 1.0 : s.Xb[r * a.dsum + k.xoff + j - k.onescol];
            }
        } else {
            const double scale = 1.0 / sqrt((double)k.n);
            const ldsd *il = s.ils + k.ls0, *W = s.Wl + k.woff;
            for (int e = tid; e < M * k.n; e += SVI_THREADS) {
                const int r = e / k.n, j = e % k.n;
                double t = 0.0;
                for (int i = 0; i < k.d; ++i) t = fma(s.Xb[r * a.dsum + k.xoff + i], W[i * k.n + j] * il[k.n_ls == 1 ? 0 : i], t);
                double sn, cs;
                rr_sincos_rev_f64(t, sn, cs);
                s.Phi[r * a.Fp + k.col0 + j] = cs * scale;
                s.Phi[r * a.Fp + k.col0 + k.n + j] = sn * scale;
            }
        }
    }
    // the f-independent part of sum_r loglike per latent sample (likelihoods.py: the log-factorial terms; the Gaussian's
    // follows the variance)
    double lc = tid < M ? g.lc : 0.0;
    lc = svi_block_sum(lc, s.red);
    if (tid == 0) s.misc[0] = lc;
    svi_sync();
}

// d: the draws of samples [kl0, kl0 + L) into LDS: the caller's (E: (L, F) float32 in HBM) or the device generator's
__device__ __forceinline__ void svi_draws(const SviArgs &a, const SviLds &s, const float *E, uint64_t stepkey, int kl0) {
    const int tid = threadIdx.x, tot = a.L * a.F;
    if (E) {
        for (int o = tid; o < tot; o += SVI_THREADS) s.Ef[o] = E[o];
    } else {
        for (int o = tid; o < tot; o += SVI_THREADS) s.Ef[o] = svi_draw(stepkey, (uint64_t)kl0 * (uint64_t)a.F + (uint64_t)o);
    }
}

typedef double svi_d4 __attribute__((ext_vector_type(4)));

// e: fs = ws Phi^T of the L samples in s.Ef against the minibatch (ws = mk + sk e) on the f64 matrix cores -- one
// v_mfma_f64_16x16x4 tile (16 samples x 16 rows) per wave and turn, A[i][k] = ws[l0 + i][j0 + k], B[k][c] = Phi[r0 + c][j0 + k]
// (lane = i + 16 k for A, c + 16 k for B; D[row = k + 4 v][col = c], v < 4) -- then df into dfs (L, M) from the accumulators;
// (sum loglike terms, sum squared errors).  (As scalar dot products out of LDS this pass is bound by LDS reads: 6 us of a step.)
__device__ __forceinline__ void svi_pass1(const SviArgs &a, const SviLds &s, double ivar, double &llsum, double &aux) {
    const int tid = threadIdx.x, M = a.M, F = a.F, L = a.L, wave = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int nrb = (L + 15) / 16, ncb = (M + 15) / 16;
    double ll_acc = 0.0;
    for (int tile = wave; tile < nrb * ncb; tile += SVI_WAVES) {
        const int rb = tile / ncb, cb = tile % ncb;
        const int l = rb * 16 + i, r = cb * 16 + i;
        const bool lv = l < L, rv = r < M;
        const ldsf *er = s.Ef + (lv ? l : 0) * F;
        const ldsd *ph = s.Phi + (rv ? r : 0) * a.Fp;
        svi_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int j0 = 0; j0 < F; j0 += 32) {   // eight k-steps per turn: ALL their LDS reads are issued before the first product
            double sk8[8], mk8[8], ph8[8];
            float e8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 4 * u + kk, jc = j < F ? j : 0;
                sk8[u] = s.sk[jc];
                mk8[u] = s.mk[jc];
                e8[u] = er[jc];
                ph8[u] = ph[jc];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool jv = j0 + 4 * u + kk < F;
                const double av = (lv && jv) ? fma(sk8[u], (double)e8[u], mk8[u]) : 0.0;
                const double bv = (rv && jv) ? ph8[u] : 0.0;
                if (j0 + 4 * u < F) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int lo = rb * 16 + kk + 4 * v, ro = cb * 16 + i;
            if (lo < L && ro < M) {
                double df, ll;
                svi_lik(a.lik, acc[v], s.yb[ro], s.nb[ro], ivar, df, ll);
                s.dfs[lo * M + ro] = df;
                ll_acc += ll;
            }
        }
    }
    const double tot = svi_block_sum(ll_acc, s.red);
    if (a.lik == RR_LIK_GAUSSIAN) {
        aux = tot;
        llsum = -0.5 * tot * ivar;
    } else {
        aux = 0.0;
        llsum = tot;
    }
}

// f (first half): Q[r][j] = sum_l dfs[l][r] e[l][j] and D_r = sum_l dfs[l][r] (the column j = F of the same product, e = 1)
// on the f64 matrix cores: A[i][k] = dfs[l0 + k][r0 + i], B[k][c] = e[l0 + k][j0 + c]
__device__ __forceinline__ void svi_pass2(const SviArgs &a, const SviLds &s) {
    const int tid = threadIdx.x, M = a.M, F = a.F, L = a.L, wave = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int nrb = (M + 15) / 16, ncb = (F + 1 + 15) / 16;
    for (int tile = wave; tile < nrb * ncb; tile += SVI_WAVES) {
        const int rb = tile / ncb, cb = tile % ncb;
        const int r = rb * 16 + i, j = cb * 16 + i;
        const bool rv = r < M, jv = j < F;
        svi_d4 acc = {0.0, 0.0, 0.0, 0.0};
        const int rc = rv ? r : 0, jcl = jv ? j : 0;
        for (int l0 = 0; l0 < L; l0 += 32) {
            double d8[8];
            float e8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int l = l0 + 4 * u + kk, lc = l < L ? l : 0;
                d8[u] = s.dfs[lc * M + rc];
                e8[u] = s.Ef[lc * F + jcl];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool lv = l0 + 4 * u + kk < L;
                const double av = (rv && lv) ? d8[u] : 0.0;
                const double bv = lv ? (jv ? (double)e8[u] : (j == F ? 1.0 : 0.0)) : 0.0;
                if (l0 + 4 * u < L) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ro = rb * 16 + kk + 4 * v, jo = cb * 16 + i;
            if (ro < M) {
                if (jo < F) s.Q[ro * F + jo] = acc[v];
                else if (jo == F) s.Dr[ro] = acc[v];
            }
        }
    }
}

// log N_jl = -(F log 2 pi + q_jl) / 2 and log z_l = logsumexp_j log N_jl from the full q (K, K) in s.q      glm.py:221-222
__device__ __forceinline__ void svi_logz(const SviArgs &a, const SviLds &s) {
    const int tid = threadIdx.x, K = a.K;
    if (tid < K) {
        double mx = -INFINITY;
        for (int j = 0; j < K; ++j) mx = fmax(mx, -0.5 * ((double)a.F * 1.8378770664093453 + s.q[j * K + tid]));
        double sm = 0.0;
        for (int j = 0; j < K; ++j) sm += exp(-0.5 * ((double)a.F * 1.8378770664093453 + s.q[j * K + tid]) - mx);
        s.logz[tid] = log(sm) + mx;
    }
    svi_sync();
}

// q[k][l] for every l: one wave per l (out: LDS or HBM)
template <typename P>
__device__ __forceinline__ void svi_qrow(const SviArgs &a, const SviLds &s, int k, P out) {
    const int tid = threadIdx.x, K = a.K, F = a.F, wave = tid >> 6, lane = tid & 63;
    for (int l = wave; l < K; l += SVI_WAVES) {
        double acc = 0.0;
        for (int f = lane; f < F; f += 64) {
            const double dc = s.xC[f * K + k] + s.xC[f * K + l];
            const double dm = s.xm[f * K + k] - s.xm[f * K + l];
            acc += log(dc) + dm * dm / dc;
        }
        acc = svi_wave_sum(acc);
        if (lane == 0) out[l] = acc;
    }
}

// R_c = sum (m^2 + C) over child c's rows of (m, C) into s.Rc[c]: one wave per child
__device__ __forceinline__ void svi_child_sums(const SviArgs &a, const SviLds &s) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, K = a.K;
    for (int c = wave; c < a.nkids; c += SVI_WAVES) {
        double acc = 0.0;
        const int lo = a.kid[c].col0 * K, hi = (a.kid[c].col0 + a.kid[c].width) * K;
        for (int o = lo + lane; o < hi; o += 64) acc += s.xm[o] * s.xm[o] + s.xC[o];
        acc = svi_wave_sum(acc);
        if (lane == 0) s.Rc[c] = acc;
    }
}

// -ELBO from the sums (glm.py:285-292) by the lanes of ONE wave (R_c in s.Rc, log z in s.logz); every lane gets it
__device__ __forceinline__ double svi_neg_elbo(const SviArgs &a, const SviLds &s, double ell_total, int lane) {
    const int K = a.K, F = a.F;
    double part = 0.0;
    for (int c = lane; c < a.nkids; c += 64) {
        const double reg = s.xs[c];
        part += -0.5 * K * ((double)a.kid[c].width * log(reg)) - 0.5 * (s.Rc[c] / reg);
    }
    for (int k = lane; k < K; k += 64) part -= s.logz[k];
    part = svi_wave_sum(part);
    const double elbo = (ell_total * a.bmag - 0.5 * F * K * 1.8378770664093453 + part + log((double)K)) / K;
    return -elbo;
}

// one coordinate's truncation + updater + clip (sgd.py:404-420, :14-330); g is the gradient in z space
__device__ __forceinline__ double svi_update(const SviArgs &a, double zz, double g, double lo, double hi, double &s1, double &s2,
                                             double b1t, double b2t) {
#pragma clang fp contract(off)
    if (zz <= lo) g = (g <= 0.0 || g != g) ? g : 0.0;
    if (zz >= hi) g = (g >= 0.0 || g != g) ? g : 0.0;
    double zn;
    switch (a.updater) {
        case RR_UPD_SGD: zn = zz - a.up[0] * g; break;
        case RR_UPD_ADADELTA: {
            const double eg2 = a.up[0] * s1 + (1 - a.up[0]) * (g * g);
            const double dx = -g * sqrt(s2 + a.up[1]) / sqrt(eg2 + a.up[1]);
            s1 = eg2;
            s2 = a.up[0] * s2 + (1 - a.up[0]) * (dx * dx);
            zn = zz + dx;
        } break;
        case RR_UPD_ADAGRAD: {
            const double h = s1 + g * g;
            s1 = h;
            zn = zz - a.up[0] * g / (a.up[1] + sqrt(h));
        } break;
        case RR_UPD_MOMENTUM: {
            const double dx = a.up[0] * s1 - a.up[1] * g;
            s1 = dx;
            zn = zz + dx;
        } break;
        default: {
            const double m = a.up[1] * s1 + (1 - a.up[1]) * g;
            const double v = a.up[2] * s2 + (1 - a.up[2]) * (g * g);
            s1 = m;
            s2 = v;
            zn = zz - a.up[0] * (m / b1t) / (sqrt(v / b2t) + a.up[3]);
        }
    }
    return zn < lo ? lo : (zn > hi ? hi : zn);
}

// RR_SVI_PROF=1: workgroup 0's thread 0 accumulates the 100 MHz clock per phase in REGISTERS (a read-modify-write of HBM per
// mark would cost more than the phases it times, and its wait would cover the prefetched loads) and adds them to a.prof at
// the end of the launch
#define SVI_MARK(i)                                                   \
    do {                                                              \
        if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) {          \
            const long long now_ = wall_clock64();                    \
            pacc[i] += now_ - tprev;                                  \
            tprev = now_;                                             \
        }                                                             \
    } while (0)

// a: x = from_log(z) of every coordinate into LDS -- this workgroup's column from its own z (LDS), the others' from `src`
// (their owners' columns: (K, 2 F) as published, or the flat vector z of the launch's start) -- then what depends on x
// alone: component k's means and standard deviations, 1 / (2 pi l) per length scale, R_c = sum (m^2 + C) per child.
// (first half) the raw z of every coordinate into the LDS arrays of x: this workgroup's column from its own z (LDS), the
// others' from `src` (their owners' columns (K, 2 F) as published, or the flat vector z of the launch's start)
__device__ __forceinline__ void svi_load_z(const SviArgs &a, const SviLds &s, int k, const double *src, bool src_is_z) {
    const int tid = threadIdx.x, K = a.K, F = a.F;
    const int fk = F * K;
    for (int p = tid; p < 2 * fk; p += SVI_THREADS) {
        const int cov = p >= fk, q = cov ? p - fk : p, f = q / K, j = q % K;
        double zv;
        if (j == k) zv = s.zc[cov * F + f];
        else zv = src_is_z ? src[p] : src[(size_t)j * 2 * F + cov * F + f];
        (cov ? s.xC : s.xm)[q] = zv;
    }
}

// (second half) x = from_log(z) in place, then what depends on x alone: component k's means and standard deviations,
// 1 / (2 pi l) per length scale, R_c = sum (m^2 + C) per child
__device__ __forceinline__ void svi_form_x(const SviArgs &a, const SviLds &s, int k) {
    const int tid = threadIdx.x, K = a.K, F = a.F, ns = a.ns;
    const int fk = F * K;
    for (int p = tid; p < 2 * fk; p += SVI_THREADS) {
        if (s.lg[p]) {
            ldsd *x = p >= fk ? s.xC + (p - fk) : s.xm + p;
            *x = exp(*x);
        }
    }
    for (int p = tid; p < ns; p += SVI_THREADS) s.xs[p] = s.lg[2 * fk + p] ? exp(s.zs[p]) : s.zs[p];
    svi_sync();
    for (int f = tid; f < F; f += SVI_THREADS) {
        s.mk[f] = s.xm[f * K + k];
        s.sk[f] = sqrt(s.xC[f * K + k]);
    }
    for (int h = tid; h < a.n_ls; h += SVI_THREADS) s.ils[h] = 0.15915494309189533576888 / s.xs[ns - a.n_ls + h];
    svi_child_sums(a, s);   // -> s.Rc
    svi_sync();
}

__global__ void __launch_bounds__(SVI_THREADS) rr_glm_svi_steps_kernel(const SviArgs a) {
#pragma clang fp contract(off)
    extern __shared__ double sm[];
    long long tprev = a.prof ? wall_clock64() : 0;
    long long pacc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const SviLds s = svi_carve(sm, a);
    const int tid = threadIdx.x, k = blockIdx.x, K = a.K, F = a.F, M = a.M, L = a.L, ns = a.ns, nk = a.nkids;
    const int wave = tid >> 6, lane = tid & 63;
    const int64_t fk = (int64_t)F * K;
    const int npub = a.n_ls + 4;
    // ---- launch start: this workgroup's column and the shared coordinates, with their updater state and bounds
    for (int p = tid; p < 2 * F; p += SVI_THREADS) {
        const int64_t g = (p < F ? 0 : fk) + (int64_t)(p < F ? p : p - F) * K + k;
        s.zc[p] = a.z[g];
        s.s1c[p] = a.s1[g];
        s.s2c[p] = a.s2[g];
        s.loc[p] = a.lower[g];
        s.hic[p] = a.upper[g];
    }
    for (int p = tid; p < ns; p += SVI_THREADS) {
        s.zs[p] = a.z[2 * fk + p];
        s.s1s[p] = a.s1[2 * fk + p];
        s.s2s[p] = a.s2[2 * fk + p];
        s.los[p] = a.lower[2 * fk + p];
        s.his[p] = a.upper[2 * fk + p];
    }
    for (int p = tid; p < (int)a.np; p += SVI_THREADS) s.lg[p] = a.islog[p];
    for (int c = 0; c < nk; ++c)
        if (a.kid[c].kind == RR_SGD_CHILD_RFF)
            for (int o = tid; o < a.kid[c].d * a.kid[c].n; o += SVI_THREADS) s.Wl[a.kid[c].woff + o] = a.kid[c].W[o];
    SviGather gth;
    svi_gather_issue(a, a.idx, gth);
    svi_sync();
    svi_load_z(a, s, k, a.z, true);
    svi_sync();
    svi_form_x(a, s, k);
    svi_draws(a, s, a.E ? a.E + (size_t)k * L * F : nullptr, svi_splitmix64(a.seed ^ (a.key0 * 0xD1B54A32D192ED03ull)), k * L);
    svi_sync();
    for (int t = 0; t < a.steps; ++t) {
        const int par = t & 1;
        const int64_t gt = a.t0 + t;
        SVI_MARK(0);
        // ---- b: row k of the mixture's cross terms, published
        svi_qrow(a, s, k, s.stage);
        svi_sync();
        if (wave == 0) {
            for (int l = lane; l < K; l += 64) a.pubrow[((size_t)par * K + k) * K + l] = s.stage[l];
            svi_arrive_wave0(a.bar + 0);
        }
        SVI_MARK(1);
        // ---- c: the minibatch
        svi_features(a, s, gth);
        if (t + 1 < a.steps) svi_gather_issue(a, a.idx + (size_t)(t + 1) * M, gth);  // the next step's rows: in flight from here
        SVI_MARK(2);
        // ---- e: fs, df, loglike (the draws are in LDS since the previous step's barrier)
        const double ivar = a.n_lik ? 1.0 / s.xs[nk] : 0.0;
        double llsum, aux;
        svi_pass1(a, s, ivar, llsum, aux);
        svi_sync();
        SVI_MARK(3);
        // ---- f: with D_r = sum_l dfs[l][r] and Q[r][j] = sum_l dfs[l][r] e[l][j] (the only sums over the samples needed):
        //   Edm[j] = sum_r D_r Phi[r][j] / L                  (Edws = dfs Phi summed over l, glm.py:308-309)
        //   EdC[j] = sum_r Q[r][j] Phi[r][j] / (L sk[j])      (sum_l Edws e / sqrt(C), glm.py:310)
        //   EdPhi[r][j] = (mk[j] D_r + sk[j] Q[r][j]) / (L K) (this component's share of dfs^T ws / L / K, glm.py:311,237)
        svi_pass2(a, s);
        svi_sync();
        for (int j = tid; j < F; j += SVI_THREADS) {
            double am = 0.0, ac = 0.0;
            for (int r = 0; r < M; ++r) {
                const double ph = s.Phi[r * a.Fp + j];
                am = fma(s.Dr[r], ph, am);
                ac = fma(s.Q[r * F + j], ph, ac);
            }
            s.edm[j] = am / L;
            s.edc[j] = ac / (L * s.sk[j]);
        }
        // -(EdPhi o dPhi_i).sum() of this component: W[i, :] . T[i, :] / l_i^2 with T = X^T (E_s o P_c - E_c o P_s); the
        // isotropic parameter takes input dimension 0 only, as the reference does (basis_functions.py:896)
        for (int h = wave; h < a.n_ls; h += SVI_WAVES) {
            int c = 0;
            while (!(a.kid[c].kind == RR_SGD_CHILD_RFF && h >= a.kid[c].ls0 && h < a.kid[c].ls0 + a.kid[c].n_ls)) ++c;
            const SviChild &kd = a.kid[c];
            const int i = h - kd.ls0;
            const double sc = 1.0 / ((double)L * K);
            const ldsd *W = s.Wl + kd.woff + i * kd.n;
            double acc = 0.0;
            for (int o = lane; o < M * kd.n; o += 64) {
                const int r = o / kd.n, j = o % kd.n, jc = kd.col0 + j, js = jc + kd.n;
                const double epc = (s.mk[jc] * s.Dr[r] + s.sk[jc] * s.Q[r * F + jc]) * sc;
                const double eps = (s.mk[js] * s.Dr[r] + s.sk[js] * s.Q[r * F + js]) * sc;
                acc = fma(W[j] * s.Xb[r * a.dsum + kd.xoff + i], eps * s.Phi[r * a.Fp + jc] - epc * s.Phi[r * a.Fp + js], acc);
            }
            acc = svi_wave_sum(acc);
            if (lane == 0) s.gls[h] = acc;
        }
        SVI_MARK(4);
        // ---- g: the other components' rows of q
        svi_wait(a.bar + 0, (unsigned)(t + 1) * (unsigned)K);
        for (int o = tid; o < K * K; o += SVI_THREADS) s.q[o] = a.pubrow[(size_t)par * K * K + o];
        svi_sync();
        SVI_MARK(5);
        svi_logz(a, s);
        if (tid < K) {
            const double lN = -0.5 * ((double)F * 1.8378770664093453 + s.q[tid * K + k]);
            s.alpha[tid] = exp(lN - s.logz[k]) + exp(lN - s.logz[tid]);
        }
        svi_sync();
        SVI_MARK(6);
        // ---- h: column k's gradient and update
        const double b1t = a.bias[2 * t], b2t = a.bias[2 * t + 1];
        double n2 = 0.0;
        for (int p = tid; p < 2 * F; p += SVI_THREADS) {
            const int cov = p >= F, f = cov ? p - F : p;
            int c = 0;
            while (c + 1 < nk && f >= a.kid[c + 1].col0) ++c;
            const double reg = s.xs[c];
            const double mkv = s.xm[f * K + k], Ck = s.xC[f * K + k];
            double mix = 0.0;
            for (int l = 0; l < K; ++l) {
                const double ic = 1.0 / (Ck + s.xC[f * K + l]);
                const double dm = mkv - s.xm[f * K + l];
                const double e = dm * ic;
                mix += (cov ? ic - e * e : ic * dm) * s.alpha[l];
            }
            double g;
            if (!cov) g = -((a.bmag * s.edm[f] - mkv / reg + mix) / K);
            else g = -((a.bmag * s.edc[f] - 1.0 / reg + mix) / (2 * K));
            if (s.lg[(cov ? fk : 0) + (int64_t)f * K + k]) g *= cov ? Ck : mkv;
            n2 += g * g;
            double s1 = s.s1c[p], s2 = s.s2c[p];
            const double zn = svi_update(a, s.zc[p], g, s.loc[p], s.hic[p], s1, s2, b1t, b2t);
            s.s1c[p] = s1;
            s.s2c[p] = s2;
            s.zc[p] = zn;
        }
        n2 = svi_block_sum(n2, s.red);
        SVI_MARK(7);
        if (wave == 0) {   // publish: the new column, this component's scalars
            double *pc = a.pubcol + ((size_t)par * K + k) * 2 * F, *pb = a.pubsc + ((size_t)par * K + k) * npub;
            for (int p = lane; p < 2 * F; p += 64) pc[p] = s.zc[p];
            for (int h = lane; h < a.n_ls; h += 64) pb[h] = s.gls[h];
            if (lane == 0) {
                pb[a.n_ls] = aux;
                pb[a.n_ls + 1] = llsum;
                pb[a.n_ls + 2] = n2;
                pb[a.n_ls + 3] = 0.0;
            }
            svi_arrive_wave0(a.bar + 1);
        }
        // ---- d (of the NEXT step, while the slowest component gets to the barrier): its draws into LDS
        if (t + 1 < a.steps)
            svi_draws(a, s, a.E ? a.E + ((size_t)(t + 1) * K * L + (size_t)k * L) * F : nullptr,
                      svi_splitmix64(a.seed ^ ((a.key0 + (uint64_t)(t + 1)) * 0xD1B54A32D192ED03ull)), k * L);
        SVI_MARK(8);
        svi_wait(a.bar + 1, (unsigned)(t + 1) * (unsigned)K);
        SVI_MARK(9);
        // ---- i: the shared coordinates (every workgroup, identically) and the step's record -- behind ONE round of HBM
        // reads: the components' scalars AND the other components' new columns (raw z into the arrays of x: what follows
        // reads the shared coordinates' x, R_c and log z of THIS step, none of which live there)
        for (int o = tid; o < K * npub; o += SVI_THREADS) s.psc[o] = a.pubsc[(size_t)par * K * npub + o];
        if (t + 1 < a.steps) svi_load_z(a, s, k, a.pubcol + (size_t)par * K * 2 * F, false);
        svi_sync();
        double n2s = 0.0;
        if (tid < ns) {
            const int p = tid;
            double g;
            if (p < nk) {  // dreg of the child's slice (glm.py:265-268)
                const double iL = 1.0 / s.xs[p];
                g = -(0.5 * (s.Rc[p] * (iL * iL) / K - (double)a.kid[p].width * iL));
            } else if (p < nk + a.n_lik) {  // Gaussian variance (likelihoods.py:370-396)
                const double iv = 1.0 / s.xs[p];
                double sm2 = 0.0;
                for (int j = 0; j < K; ++j) sm2 += 0.5 * (s.psc[j * npub + a.n_ls] * iv * iv - iv * (double)M * L) / L;
                g = 0.0 - sm2 / K;
            } else {
                const int h = p - nk - a.n_lik;
                double sm2 = 0.0;
                for (int j = 0; j < K; ++j) sm2 += s.psc[j * npub + h];
                const double l = s.xs[p];
                g = sm2 / (1.0 * (l * l));
            }
            if (s.lg[2 * fk + p]) g *= s.xs[p];
            n2s = g * g;
            double s1 = s.s1s[p], s2 = s.s2s[p];
            s.zs[p] = svi_update(a, s.zs[p], g, s.los[p], s.his[p], s1, s2, b1t, b2t);
            s.s1s[p] = s1;
            s.s2s[p] = s2;
        }
        n2s = svi_block_sum(n2s, s.red);
        if (k == 0 && wave == 0) {  // the step's record: |g| (sgd.py:399) and -ELBO (glm.py:285-292), by one wave
            double tot = 0.0, ell = 0.0;
            for (int j = lane; j < K; j += 64) {
                tot += s.psc[j * npub + a.n_ls + 2];
                ell += s.psc[j * npub + a.n_ls + 1] / L;
            }
            tot = svi_wave_sum(tot) + n2s;
            double llc = s.misc[0];
            if (a.n_lik) llc = -0.5 * log(2.0 * 3.141592653589793 * s.xs[nk]) * (double)M;
            ell = svi_wave_sum(ell) + llc * K;
            const double obj = svi_neg_elbo(a, s, ell, lane);
            if (lane == 0) {
                a.norms[gt] = sqrt(tot);
                a.objs[gt] = obj;
            }
        }
        svi_sync();
        SVI_MARK(10);
        // ---- a (of the next step): x = from_log(z) of everything
        if (t + 1 < a.steps) svi_form_x(a, s, k);
    }
    if (a.prof && k == 0 && tid == 0)
        for (int i = 0; i < 11; ++i) a.prof[i] += pacc[i];
    // ---- launch end: state back to HBM
    for (int p = tid; p < 2 * F; p += SVI_THREADS) {
        const int64_t g = (p < F ? 0 : fk) + (int64_t)(p < F ? p : p - F) * K + k;
        a.z[g] = s.zc[p];
        a.s1[g] = s.s1c[p];
        a.s2[g] = s.s2c[p];
    }
    if (k == 0) {
        for (int p = tid; p < ns; p += SVI_THREADS) {
            a.z[2 * fk + p] = s.zs[p];
            a.s1[2 * fk + p] = s.s1s[p];
            a.s2[2 * fk + p] = s.s2s[p];
        }
    }
}

// One workgroup per candidate: -ELBO of candidate c (x space, np coordinates) on its own minibatch with its own draws.
__global__ void __launch_bounds__(SVI_THREADS) rr_glm_svi_starts_kernel(const SviArgs a) {
#pragma clang fp contract(off)
    extern __shared__ double sm[];
    const SviLds s = svi_carve(sm, a);
    const int tid = threadIdx.x, c = blockIdx.x, K = a.K, F = a.F, M = a.M, L = a.L, ns = a.ns, nk = a.nkids;
    const int lane = tid & 63;
    const int64_t fk = (int64_t)F * K;
    const double *x = a.cand + (size_t)c * a.np;
    SviGather gth;
    svi_gather_issue(a, a.idx + (size_t)c * M, gth);
    for (int p = tid; p < (int)fk; p += SVI_THREADS) {
        s.xm[p] = x[p];
        s.xC[p] = x[fk + p];
    }
    for (int p = tid; p < ns; p += SVI_THREADS) s.xs[p] = x[2 * fk + p];
    for (int ch = 0; ch < nk; ++ch)
        if (a.kid[ch].kind == RR_SGD_CHILD_RFF)
            for (int o = tid; o < a.kid[ch].d * a.kid[ch].n; o += SVI_THREADS) s.Wl[a.kid[ch].woff + o] = a.kid[ch].W[o];
    for (int h = tid; h < a.n_ls; h += SVI_THREADS) s.ils[h] = 0.15915494309189533576888 / x[2 * fk + ns - a.n_ls + h];
    svi_sync();
    svi_features(a, s, gth);
    const double ivar = a.n_lik ? 1.0 / s.xs[nk] : 0.0;
    double ell = 0.0;
    const uint64_t stepkey = svi_splitmix64(a.seed ^ ((a.key0 + (uint64_t)c) * 0xD1B54A32D192ED03ull));
    for (int k = 0; k < K; ++k) {
        svi_sync();
        for (int f = tid; f < F; f += SVI_THREADS) {
            s.mk[f] = s.xm[f * K + k];
            s.sk[f] = sqrt(s.xC[f * K + k]);
        }
        svi_draws(a, s, a.E ? a.E + ((size_t)c * K * L + (size_t)k * L) * F : nullptr, stepkey, k * L);
        svi_sync();
        double llsum, aux;
        svi_pass1(a, s, ivar, llsum, aux);
        ell += llsum / L;
    }
    for (int k = 0; k < K; ++k) svi_qrow(a, s, k, s.q + k * K);
    svi_child_sums(a, s);
    svi_sync();
    svi_logz(a, s);
    if (tid < 64) {
        double llc = s.misc[0];
        if (a.n_lik) llc = -0.5 * log(2.0 * 3.141592653589793 * s.xs[nk]) * (double)M;
        const double obj = svi_neg_elbo(a, s, ell + llc * K, lane);
        if (lane == 0) a.out[c] = obj;
    }
}

}  // namespace

struct rr_glm_svi {
    rr_ctx *ctx = nullptr;
    SviArgs a;
    SviChild hkid[SVI_MAXCHILD];
    SviChild *dkid = nullptr;
    double *dbias = nullptr;
    size_t bias_cap = 0;
    std::vector<double> hbias;
    std::vector<double *> dW;
    unsigned char *islog = nullptr;
    double *lower = nullptr, *upper = nullptr;
    int64_t maxiter = 0, t = 0;
    size_t lds_bytes = 0;
    double *cand = nullptr, *out = nullptr;
    size_t cand_cap = 0;
    long long *prof = nullptr;
};

static void svi_free(rr_glm_svi *o) {
    if (o->prof) (void)hipFree(o->prof);
    if (o->dkid) (void)hipFree(o->dkid);
    if (o->dbias) (void)hipFree(o->dbias);
    void *q[] = {o->a.z, o->a.s1, o->a.s2, o->lower, o->upper, o->islog, o->a.pubcol, o->a.pubrow, o->a.pubsc, o->a.bar,
                 o->a.objs, o->a.norms, o->cand, o->out};
    for (void *v : q)
        if (v) (void)hipFree(v);
    for (double *w : o->dW)
        if (w) (void)hipFree(w);
    delete o;
}

extern "C" {

int rr_glm_svi_supported(int F, int K, int L, int M, int n_children, int dsum, int n_ls) {
    if (F < 1 || K < 1 || K > SVI_MAXK || L < 1 || M < 1 || n_children < 1 || n_children > SVI_MAXCHILD) return 0;
    if (F < n_children || M > SVI_THREADS || n_children + 1 + n_ls > SVI_THREADS) return 0;
    SviArgs a;
    a.F = F; a.Fp = F | 1; a.K = K; a.L = L; a.M = M; a.nkids = n_children; a.dsum = dsum; a.n_ls = n_ls;
    a.ns = n_children + 1 + n_ls;
    // the work one workgroup does per step stays small (this is the dispatch-bound regime, not a GEMM engine), and its
    // state fits the CU's LDS
    if ((int64_t)L * M * F > (int64_t)1 << 20 || (int64_t)M * F > 8192 || (int64_t)M * dsum > SVI_GREG * SVI_THREADS) return 0;
    a.np = 2 * (int64_t)F * K + a.ns;
    a.wtot = dsum * (F / 2 + 1);  // (an upper bound of sum d_c n_c: the children's frequency matrices, kept in LDS)
    return svi_lds_doubles(a) * 8 <= 150 * 1024 ? 1 : 0;
}

int rr_glm_svi_create(rr_ctx *ctx, int n_children, const rr_glm_sgd_child *children, const void *const *dX, const int *x_dtype,
                      const int64_t *ldx, int64_t N, const void *dy, const void *drowarg, const double *dlconst, int dtype, int K, int L,
                      int M, int lik, int n_lik, const double *z0, const double *lower, const double *upper, const unsigned char *is_log,
                      int updater, const double *upd_par, int64_t maxiter, double bmag, rr_glm_svi **out) {
    RR_REQUIRE(ctx != nullptr && children != nullptr && dX != nullptr && x_dtype != nullptr && ldx != nullptr && dy != nullptr &&
               z0 != nullptr && lower != nullptr && upper != nullptr && is_log != nullptr && upd_par != nullptr && out != nullptr,
               "rr_glm_svi_create: null argument");
    *out = nullptr;
    RR_REQUIRE(n_children >= 1 && n_children <= SVI_MAXCHILD, "rr_glm_svi_create: 1 <= children <= %d", SVI_MAXCHILD);
    RR_REQUIRE(K >= 1 && K <= SVI_MAXK && L >= 1 && M >= 1 && N >= 1, "rr_glm_svi_create: bad K, L, minibatch or N");
    RR_REQUIRE(lik >= RR_LIK_BERNOULLI && lik <= RR_LIK_POISSON_SOFTPLUS && (lik == RR_LIK_GAUSSIAN) == (n_lik == 1),
               "rr_glm_svi_create: likelihood %d with %d likelihood parameter(s)", lik, n_lik);
    RR_REQUIRE((lik == RR_LIK_BINOMIAL) == (drowarg != nullptr), "rr_glm_svi_create: the per-row argument goes with the binomial likelihood");
    RR_REQUIRE(updater >= RR_UPD_SGD && updater <= RR_UPD_ADAM, "rr_glm_svi_create: unknown updater %d", updater);
    RR_REQUIRE(maxiter >= 1 && maxiter < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), "rr_glm_svi_create: bad maxiter / N");
    RR_REQUIRE(dtype == RR_F32 || dtype == RR_F64, "rr_glm_svi_create: bad dtype");
    RR_CHECK_HIP(hipSetDevice(ctx->device));
    rr_glm_svi *o = new rr_glm_svi();
    o->ctx = ctx;
    SviArgs &a = o->a;
    memset(&a, 0, sizeof a);
    int col = 0, nls = 0, xoff = 0, woff = 0;
    for (int s = 0; s < n_children; ++s) {
        const rr_glm_sgd_child &k = children[s];
        SviChild &c = o->hkid[s];
        c.kind = k.kind;
        c.col0 = col;
        c.ls0 = nls;
        c.xoff = xoff;
        c.woff = woff;
        c.X = dX[s];
        c.ldx = ldx[s];
        c.x_f64 = x_dtype[s] == RR_F64;
        bool ok = dX[s] != nullptr && (x_dtype[s] == RR_F32 || x_dtype[s] == RR_F64);
        if (ok && k.kind == RR_SGD_CHILD_RFF) {
            rr_basis *b = k.basis;
            ok = b != nullptr && b->kind == RR_KIND_RFF && b->ctx == ctx && (k.n_ls == 1 || k.n_ls == b->d) && (int)b->W.size() == b->d * b->n;
            if (ok) {
                c.d = b->d; c.n = b->n; c.n_ls = k.n_ls; c.width = 2 * b->n; c.onescol = 0;
                double *w = nullptr;
                if (hipMalloc((void **)&w, b->W.size() * 8) != hipSuccess ||
                    hipMemcpy(w, b->W.data(), b->W.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipGetLastError();
                    if (w) (void)hipFree(w);
                    svi_free(o);
                    rr_set_error("rr_glm_svi_create: device allocation failed");
                    return RR_ERR_OOM;
                }
                o->dW.push_back(w);
                c.W = w;
            }
        } else if (ok && k.kind == RR_SGD_CHILD_LINEAR) {
            ok = k.d >= 1 && k.n_ls == 0;
            c.d = k.d; c.n = 0; c.n_ls = 0; c.onescol = k.onescol ? 1 : 0; c.width = k.d + c.onescol;
        } else {
            ok = false;
        }
        if (!ok) {
            svi_free(o);
            rr_set_error("rr_glm_svi_create: child %d: a random Fourier basis of this context with 1 or Xdim length scales, or a "
                         "linear child with d >= 1 columns, and its resident rows", s);
            return RR_ERR_INVALID;
        }
        col += c.width;
        nls += c.n_ls;
        xoff += c.d;
        if (c.kind == RR_SGD_CHILD_RFF) woff += c.d * c.n;
    }
    a.nkids = n_children; a.F = col; a.Fp = col | 1; a.K = K; a.L = L; a.M = M; a.lik = lik; a.n_lik = n_lik; a.n_ls = nls;
    a.ns = n_children + n_lik + nls; a.updater = updater; a.y_f64 = dtype == RR_F64; a.dsum = xoff; a.N = N; a.wtot = woff;
    a.np = 2 * (int64_t)col * K + a.ns;
    a.y = dy; a.rowarg = drowarg; a.lconst = dlconst; a.bmag = bmag;
    for (int i = 0; i < 4; ++i) a.up[i] = upd_par[i];
    if (!rr_glm_svi_supported(a.F, K, L, M, n_children, a.dsum, nls)) {
        svi_free(o);
        rr_set_error("rr_glm_svi_create: F = %d, K = %d, nsamples = %d, minibatch = %d is outside the fused small-batch loop's range "
                     "(rr_glm_svi_supported)", a.F, K, L, M);
        return RR_ERR_UNSUPPORTED;
    }
    o->lds_bytes = svi_lds_doubles(a) * 8;
    o->maxiter = maxiter;
    const size_t nb = (size_t)a.np * 8;
    const int npub = nls + 4;
    hipError_t e = hipMalloc((void **)&a.z, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->dkid, sizeof(o->hkid));
    if (e == hipSuccess) e = hipMemcpy(o->dkid, o->hkid, sizeof(o->hkid), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&a.s1, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&a.s2, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->lower, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->upper, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&o->islog, (size_t)a.np);
    if (e == hipSuccess) e = hipMalloc((void **)&a.pubcol, (size_t)2 * K * 2 * a.F * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&a.pubrow, (size_t)2 * K * K * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&a.pubsc, (size_t)2 * K * npub * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&a.bar, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&a.objs, (size_t)maxiter * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&a.norms, (size_t)maxiter * 8);
    if (e == hipSuccess) e = hipMemcpy(a.z, z0, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->lower, lower, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->upper, upper, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->islog, is_log, (size_t)a.np, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(a.s1, 0, nb);
    if (e == hipSuccess) e = hipMemset(a.s2, 0, nb);
    if (e == hipSuccess) e = hipMemset(a.objs, 0, (size_t)maxiter * 8);
    if (e == hipSuccess) e = hipMemset(a.norms, 0, (size_t)maxiter * 8);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)rr_glm_svi_steps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)rr_glm_svi_starts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        svi_free(o);
        rr_set_error("rr_glm_svi_create: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RR_ERR_OOM : RR_ERR_HIP;
    }
    a.lower = o->lower; a.upper = o->upper; a.islog = o->islog; a.kid = o->dkid;
    if (getenv("RR_SVI_PROF")) {
        if (hipMalloc((void **)&o->prof, 16 * sizeof(long long)) == hipSuccess) (void)hipMemset(o->prof, 0, 16 * sizeof(long long));
        a.prof = o->prof;
    }
    *out = o;
    return RR_OK;
}

int rr_glm_svi_set_start(rr_glm_svi *o, const double *z0, const double *lower, const double *upper, const unsigned char *is_log) {
    RR_REQUIRE(o != nullptr && o->t == 0, "rr_glm_svi_set_start: before the first step only");
    RR_CHECK_HIP(hipSetDevice(o->ctx->device));
    RR_CHECK_HIP(hipStreamSynchronize(o->ctx->stream));
    const size_t nb = (size_t)o->a.np * 8;
    if (z0) RR_CHECK_HIP(hipMemcpy(o->a.z, z0, nb, hipMemcpyHostToDevice));
    if (lower) RR_CHECK_HIP(hipMemcpy(o->lower, lower, nb, hipMemcpyHostToDevice));
    if (upper) RR_CHECK_HIP(hipMemcpy(o->upper, upper, nb, hipMemcpyHostToDevice));
    if (is_log) RR_CHECK_HIP(hipMemcpy(o->islog, is_log, (size_t)o->a.np, hipMemcpyHostToDevice));
    return RR_OK;
}

int rr_glm_svi_run(rr_glm_svi *o, int64_t steps, const int *d_idx, const float *dE, uint64_t seed, uint64_t key0) {
    RR_REQUIRE(o != nullptr && d_idx != nullptr && steps >= 1 && steps <= (1 << 20), "rr_glm_svi_run: 1 <= steps <= 2^20 per launch");
    RR_REQUIRE(o->t + steps <= o->maxiter, "rr_glm_svi_run: %lld steps done, %lld more asked, %lld at most", (long long)o->t,
               (long long)steps, (long long)o->maxiter);
    rr_ctx *c = o->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    SviArgs a = o->a;
    a.idx = d_idx; a.E = dE; a.seed = seed; a.key0 = key0; a.t0 = o->t; a.steps = (int)steps;
    if (o->bias_cap < (size_t)steps) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (o->dbias) (void)hipFree(o->dbias);
        o->dbias = nullptr;
        o->bias_cap = 0;
        RR_CHECK_HIP(hipMalloc((void **)&o->dbias, (size_t)steps * 16));
        o->bias_cap = (size_t)steps;
    }
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));  // (the previous launch reads dbias; hbias is reused)
    o->hbias.resize((size_t)steps * 2);
    for (int64_t t = 0; t < steps; ++t) {
        const double tt = (double)(o->t + t + 1);
        o->hbias[(size_t)2 * t] = 1.0 - pow(a.up[1], tt);
        o->hbias[(size_t)2 * t + 1] = 1.0 - pow(a.up[2], tt);
    }
    RR_CHECK_HIP(hipMemcpyAsync(o->dbias, o->hbias.data(), (size_t)steps * 16, hipMemcpyHostToDevice, c->stream));
    a.bias = o->dbias;
    RR_CHECK_HIP(hipMemsetAsync(a.bar, 0, 64, c->stream));
    hipLaunchKernelGGL(rr_glm_svi_steps_kernel, dim3((unsigned)a.K), dim3(SVI_THREADS), o->lds_bytes, c->stream, a);
    RR_CHECK_HIP(hipGetLastError());
    o->t += steps;
    return RR_OK;
}

int rr_glm_svi_starts(rr_glm_svi *o, int ncand, const int *d_idx, const double *cand_host, const float *dE, uint64_t seed, uint64_t key0,
                      double *objs_host) {
    RR_REQUIRE(o != nullptr && d_idx != nullptr && cand_host != nullptr && objs_host != nullptr && ncand >= 1 && ncand < (1 << 20),
               "rr_glm_svi_starts: bad argument");
    rr_ctx *c = o->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    SviArgs a = o->a;
    const size_t want = (size_t)ncand * a.np;
    if (o->cand_cap < want) {
        RR_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (o->cand) (void)hipFree(o->cand);
        if (o->out) (void)hipFree(o->out);
        o->cand = o->out = nullptr;
        o->cand_cap = 0;
        RR_CHECK_HIP(hipMalloc((void **)&o->cand, want * 8));
        RR_CHECK_HIP(hipMalloc((void **)&o->out, (size_t)ncand * 8));
        o->cand_cap = want;
    }
    RR_CHECK_HIP(hipMemcpyAsync(o->cand, cand_host, want * 8, hipMemcpyHostToDevice, c->stream));
    a.idx = d_idx; a.E = dE; a.seed = seed; a.key0 = key0; a.cand = o->cand; a.out = o->out;
    hipLaunchKernelGGL(rr_glm_svi_starts_kernel, dim3((unsigned)ncand), dim3(SVI_THREADS), o->lds_bytes, c->stream, a);
    RR_CHECK_HIP(hipGetLastError());
    RR_CHECK_HIP(hipMemcpyAsync(objs_host, o->out, (size_t)ncand * 8, hipMemcpyDeviceToHost, c->stream));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    return RR_OK;
}

int rr_glm_svi_read(rr_glm_svi *o, double *z, double *objs, double *norms, int64_t *steps) {
    RR_REQUIRE(o != nullptr, "rr_glm_svi_read: null argument");
    rr_ctx *c = o->ctx;
    RR_CHECK_HIP(hipSetDevice(c->device));
    RR_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (z) RR_CHECK_HIP(hipMemcpy(z, o->a.z, (size_t)o->a.np * 8, hipMemcpyDeviceToHost));
    if (objs && o->t) RR_CHECK_HIP(hipMemcpy(objs, o->a.objs, (size_t)o->t * 8, hipMemcpyDeviceToHost));
    if (norms && o->t) RR_CHECK_HIP(hipMemcpy(norms, o->a.norms, (size_t)o->t * 8, hipMemcpyDeviceToHost));
    if (steps) *steps = o->t;
    if (o->prof && o->t) {
        long long h[16];
        RR_CHECK_HIP(hipMemcpy(h, o->prof, sizeof h, hipMemcpyDeviceToHost));
        static const char *nm[] = {"i(prev)+loop", "a x=from_log", "b qrow+arrive", "c features", "d draws", "e pass1", "f pass2", "g wait B1", "g read q", "g logz/alpha", "h update", "B2"};
        fprintf(stderr, "rr_glm_svi phases (workgroup 0, us per step over %lld steps):", (long long)o->t);
        const char *names[] = {"a(form x)", "b(q row)", "c(features)", "e(pass1)", "f(pass2)", "g(wait B1)", "g(logz)", "h(update)", "publish+draws", "wait B2", "i(shared)"};
        (void)nm;
        for (int i = 0; i < 11; ++i) fprintf(stderr, " %s=%.2f", names[i], 0.01 * (double)h[i] / (double)o->t);
        fprintf(stderr, "\n");
    }
    return RR_OK;
}

void rr_glm_svi_destroy(rr_glm_svi *o) {
    if (!o) return;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    svi_free(o);
}

}  // extern "C"
