// Host-side random streams of the GLM's SVI step (no device code).
//
// The reference draws the step's standard normals from a NumPy legacy RandomState (glm.py:300: `random_.randn(L, D)` per
// mixture component).  Parity runs must consume exactly that stream, and at config 5's size it is 1 024 000 normals per
// step: ~10 ms of single-threaded NumPy against a 5 ms device step.  rr_legacy_randn advances the same generator state
// and returns the same values bit for bit, faster: the part that is inherently sequential (MT19937 words, the polar
// method's accept / reject) runs on the calling thread, the part that is not (sqrt(-2 log(r2) / r2) per accepted pair, the
// expensive part) on worker threads, block by block behind it.
//
// Algorithm restated from NumPy's published sources, which the reference pins through its `numpy` dependency:
//   mt19937_gen / mt19937_next / mt19937_next_double   numpy/random/src/mt19937/mt19937.{c,h}
//   legacy_gauss (polar Box-Muller with one cached value)  numpy/random/src/legacy/legacy-distributions.c
// `log` and `sqrt` are the C library's, as in NumPy's build; floating-point contraction is off so that x1 x1 + x2 x2
// rounds twice as it does there.  RandomState's stream is frozen by NumPy's compatibility policy (NEP 19), so one
// restatement serves every NumPy the reference runs on; tests/test_host_logic.py compares against the installed one
// (values, cached-value parity, interleaving with other draws, final state).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "rr_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t *key;
    int pos;
    uint32_t buf[MT_N];  // tempered outputs of key[pos .. 624)
    void temper_from(int p0) {
        for (int i = p0; i < MT_N; ++i) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            buf[i] = y;
        }
    }
    void gen() {
        uint32_t y;
        int i;
        for (i = 0; i < MT_N - MT_M; i++) {
            y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
            key[i] = key[i + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        for (; i < MT_N - 1; i++) {
            y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
            key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        pos = 0;
        temper_from(0);
    }
    inline uint32_t next() {
        if (__builtin_expect(pos == MT_N, 0)) gen();
        return buf[pos++];
    }
    inline double next_double() {
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

struct Pair {
    double x1, x2;
};

constexpr int64_t BLOCK = 16384;  // accepted pairs per hand-over

template <typename T>
void finish_pairs(const Pair *p, int64_t np, T *out, int64_t nout) {  // out[2 i] = f x2, out[2 i + 1] = f x1
    for (int64_t i = 0; i < np; ++i) {
        const double r2 = p[i].x1 * p[i].x1 + p[i].x2 * p[i].x2;  // the value the accept test saw
        const double f = sqrt(-2.0 * log(r2) / r2);
        out[2 * i] = (T)(f * p[i].x2);
        if (2 * i + 1 < nout) out[2 * i + 1] = (T)(f * p[i].x1);
    }
}

struct PairBuf {  // grow-only scratch of the calling thread (12 MB at config 5's size: not worth re-faulting every step)
    Pair *p = nullptr;
    size_t cap = 0;
    ~PairBuf() { free(p); }
    Pair *get(size_t n) {
        if (n > cap) {
            free(p);
            p = (Pair *)malloc(n * sizeof(Pair));
            cap = p ? n : 0;
        }
        return p;
    }
};

template <typename T>
int legacy_randn(Mt &mt, int32_t *has_gauss, double *gauss, T *out, int64_t n, int threads) {
    int64_t done = 0;
    if (*has_gauss && n > 0) {
        out[done++] = (T)*gauss;
        *has_gauss = 0;
        *gauss = 0.0;
    }
    const int64_t npairs = (n - done + 1) / 2;
    if (npairs == 0) return RR_OK;
    static thread_local PairBuf scratch;
    Pair *pairs = scratch.get((size_t)npairs);
    if (!pairs) {
        rr_set_error("rr_legacy_randn: out of host memory");
        return RR_ERR_OOM;
    }
    const int64_t nblocks = (npairs + BLOCK - 1) / BLOCK;
    std::atomic<int64_t> ready(0), claim(0);
    T *o = out + done;
    const int64_t nout = n - done;
    auto worker = [&]() {
        for (;;) {
            const int64_t b = claim.fetch_add(1);
            if (b >= nblocks) return;
            while (ready.load(std::memory_order_acquire) <= b) std::this_thread::yield();
            const int64_t p0 = b * BLOCK, p1 = p0 + BLOCK < npairs ? p0 + BLOCK : npairs;
            finish_pairs<T>(pairs + p0, p1 - p0, o + 2 * p0, nout - 2 * p0);
        }
    };
    if (threads < 1) threads = 1;
    if ((int64_t)threads > nblocks) threads = (int)nblocks;
    std::vector<std::thread> pool;
    if (npairs >= 4 * BLOCK)
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
    // accept / reject without a data-dependent branch while that cannot overshoot: a batch of c candidates (4 words each,
    // accepted or not) yields at most c pairs, so batches of min(remaining, 2048) candidates are safe; every candidate is
    // written to the next free slot and the slot advances only if it was accepted.  The last < 64 pairs one by one.
    int64_t cnt = 0;
    while (cnt < npairs) {
        const int64_t remaining = npairs - cnt;
        if (remaining >= 64) {
            const int64_t c = remaining < 2048 ? remaining : 2048;
            for (int64_t i = 0; i < c; ++i) {
                const double x1 = 2.0 * mt.next_double() - 1.0;
                const double x2 = 2.0 * mt.next_double() - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                pairs[cnt] = {x1, x2};
                cnt += (int64_t)((r2 < 1.0) & (r2 != 0.0));
            }
        } else {
            double x1, x2, r2;
            do {
                x1 = 2.0 * mt.next_double() - 1.0;
                x2 = 2.0 * mt.next_double() - 1.0;
                r2 = x1 * x1 + x2 * x2;
            } while (r2 >= 1.0 || r2 == 0.0);
            pairs[cnt++] = {x1, x2};
        }
        ready.store(cnt == npairs ? nblocks : cnt / BLOCK, std::memory_order_release);
    }
    if (pool.empty()) worker();
    for (auto &t : pool) t.join();
    if (nout & 1) {  // the last pair's second value stays cached, as in legacy_gauss
        const Pair &l = pairs[npairs - 1];
        const double r2 = l.x1 * l.x1 + l.x2 * l.x2;
        *gauss = sqrt(-2.0 * log(r2) / r2) * l.x1;
        *has_gauss = 1;
    }
    return RR_OK;
}

}  // namespace

extern "C" int rr_legacy_randn(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, void *out, int out_dtype,
                               int64_t n, int threads) {
    RR_REQUIRE(key != nullptr && pos != nullptr && has_gauss != nullptr && gauss != nullptr, "rr_legacy_randn: null state");
    RR_REQUIRE(*pos >= 0 && *pos <= MT_N, "rr_legacy_randn: position %d outside the MT19937 state", (int)*pos);
    RR_REQUIRE(n >= 0 && (n == 0 || out != nullptr), "rr_legacy_randn: bad output");
    RR_REQUIRE(out_dtype == RR_F32 || out_dtype == RR_F64, "rr_legacy_randn: bad dtype");
    Mt mt;
    mt.key = key;
    mt.pos = (int)*pos;
    mt.temper_from(mt.pos);
    const int rc = out_dtype == RR_F32 ? legacy_randn<float>(mt, has_gauss, gauss, (float *)out, n, threads)
                                       : legacy_randn<double>(mt, has_gauss, gauss, (double *)out, n, threads);
    *pos = mt.pos;
    return rc;
}
