// Host-side random streams of the GLM's SVI step (no device code).
//
// The reference draws the step's standard normals from a NumPy legacy RandomState (glm.py:300: `random_.randn(L, D)` per
// mixture component).  Parity runs must consume exactly that stream, and at config 5's size it is 1 024 000 normals per
// step: ~10 ms of single-threaded NumPy against a 5 ms device step.  rr_legacy_randn advances the same generator state
// and returns the same values bit for bit, faster: the calling thread draws the MT19937 words, forms the candidate pairs of
// the polar method and counts the accepted ones; worker threads pick the accepted candidates and compute
// sqrt(-2 log(r2) / r2) per pair, batch by batch behind it.
//
// Algorithm restated from NumPy's published sources, which the reference pins through its `numpy` dependency:
//   mt19937_gen / mt19937_next / mt19937_next_double   numpy/random/src/mt19937/mt19937.{c,h}
//   legacy_gauss (polar Box-Muller with one cached value)  numpy/random/src/legacy/legacy-distributions.c
// `log` and `sqrt` are the C library's, as in NumPy's build; floating-point contraction is off so that x1 x1 + x2 x2
// rounds twice as it does there.  RandomState's stream is frozen by NumPy's compatibility policy (NEP 19), so one
// restatement serves every NumPy the reference runs on; tests/test_host_logic.py compares against the installed one
// (values, cached-value parity, interleaving with other draws, final state).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <unistd.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rr_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t key[MT_N];  // a COPY of the caller's state (copied back at the end): through a pointer the compiler must assume
                         // that key, buf and the output alias, and the refill loops stay scalar
    int pos;
    uint32_t buf[MT_N];  // tempered outputs of key[pos .. 624)
    __attribute__((always_inline)) void temper_from(int p0) {
        for (int i = p0; i < MT_N; ++i) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            buf[i] = y;
        }
    }
    __attribute__((always_inline)) void gen() {  // (inlined into mt_fill's clones: the AVX2 build of the refill)
        twist();
        pos = 0;
        temper_from(0);
    }
    // the refill alone: key <- the next 624 state words (tempering is the consumer's: mt_fill writes it straight to its output)
    __attribute__((always_inline)) void twist() {
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

// Host code only; the refill and the candidate loop are built a second time for AVX2 hosts and picked at load time.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#define RR_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define RR_HOST_CLONES
#endif

constexpr int64_t BATCH = 2048;  // candidates per hand-over

// n tempered words of the stream into w
RR_HOST_CLONES void mt_fill(Mt &mt, uint32_t *w, int64_t n) {
    while (n > 0) {
        if (mt.pos == MT_N) {
            if (n >= MT_N) {  // a whole refill goes out: tempered from the state straight into w (buf is not touched -- nor
                              // read before the next gen(): pos stays at MT_N)
                mt.twist();
                for (int i = 0; i < MT_N; ++i) {
                    uint32_t y = mt.key[i];
                    y ^= (y >> 11);
                    y ^= (y << 7) & 0x9d2c5680u;
                    y ^= (y << 15) & 0xefc60000u;
                    y ^= (y >> 18);
                    w[i] = y;
                }
                w += MT_N;
                n -= MT_N;
                continue;
            }
            mt.gen();
        }
        int64_t c = MT_N - mt.pos;
        if (c > n) c = n;
        for (int64_t i = 0; i < c; ++i) w[i] = mt.buf[mt.pos + i];
        mt.pos += (int)c;
        w += c;
        n -= c;
    }
}

// c candidates from 4 c words: (x1, x2) = (2 u - 1, 2 u' - 1) with u, u' = mt19937_next_double, and whether legacy_gauss's loop
// would accept them; returns how many it would -- straight loops over arrays
RR_HOST_CLONES int64_t candidates(const uint32_t *w, int64_t c, double *x1o, double *x2o, unsigned char *ok) {
    int64_t cnt = 0;
    for (int64_t i = 0; i < c; ++i) {
        const int32_t a1 = (int32_t)(w[4 * i] >> 5), b1 = (int32_t)(w[4 * i + 1] >> 6);
        const int32_t a2 = (int32_t)(w[4 * i + 2] >> 5), b2 = (int32_t)(w[4 * i + 3] >> 6);
        const double x1 = 2.0 * ((a1 * 67108864.0 + b1) / 9007199254740992.0) - 1.0;
        const double x2 = 2.0 * ((a2 * 67108864.0 + b2) / 9007199254740992.0) - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        const unsigned char acc = (unsigned char)((r2 < 1.0) & (r2 != 0.0));
        x1o[i] = x1;
        x2o[i] = x2;
        ok[i] = acc;
        cnt += acc;
    }
    return cnt;
}

// out[2 i] = f x2, out[2 i + 1] = f x1 for accepted pair number i (legacy_gauss returns f x2 first and caches f x1)
template <typename T>
inline void emit_pair(double x1, double x2, int64_t i, T *o, int64_t nout, double *gauss_last) {
    const double r2 = x1 * x1 + x2 * x2;  // the value the accept test saw
    const double f = sqrt(-2.0 * log(r2) / r2);
    o[2 * i] = (T)(f * x2);
    if (2 * i + 1 < nout) o[2 * i + 1] = (T)(f * x1);
    else *gauss_last = f * x1;  // odd count: the last pair's second value stays cached
}

struct CandBuf {  // grow-only scratch of the calling thread (candidates of one call: 17 bytes each)
    double *x1 = nullptr, *x2 = nullptr;
    unsigned char *ok = nullptr;
    size_t cap = 0;
    ~CandBuf() { free(x1); free(x2); free(ok); }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        free(x1); free(x2); free(ok);
        x1 = (double *)malloc(n * 8); x2 = (double *)malloc(n * 8); ok = (unsigned char *)malloc(n);
        cap = (x1 && x2 && ok) ? n : 0;
        return cap != 0;
    }
};

struct Batch {
    int64_t cand0, c, pair0;  // its candidates [cand0, cand0 + c), the number of its first accepted pair
};

// Worker threads that outlive a call.  An SVI fit at the reference's default sizes asks for 41 500 normals per step,
// thousands of times: std::thread per call costs more than the work it takes over (and round 5 therefore ran such calls on
// the calling thread alone: 235 us each on the GPU box's host, 170 of them the logarithms and square roots).  The workers
// spin for a moment after a job -- the next call is ~100 us away in that loop -- and sleep on a condition variable otherwise.
// One job at a time (a mutex around start .. wait); a forked child starts its own pool (threads do not survive a fork).
class WorkerPool {
  public:
    static WorkerPool &get() {
        static WorkerPool *pool = nullptr;
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (!pool || pool->pid_ != getpid()) pool = new WorkerPool();  // (a fork's inheritance is abandoned, not joined)
        return *pool;
    }
    // fn runs on `n` workers; returns at once.  wait() blocks until all of them are done.
    void start(int n, std::function<void()> fn) {
        call_.lock();
        {
            std::lock_guard<std::mutex> lk(mu_);
            while ((int)th_.size() < n) {
                const int id = (int)th_.size();
                th_.emplace_back([this, id] { loop(id); });
                th_.back().detach();
            }
            job_ = std::move(fn);
            want_ = n;
            remaining_.store(n, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
    }
    void wait() {
        for (int spin = 0; remaining_.load(std::memory_order_acquire) > 0; ++spin) {
            if (spin < 4096) std::this_thread::yield();
            else {
                std::unique_lock<std::mutex> lk(mu_);
                done_.wait_for(lk, std::chrono::microseconds(200), [this] { return remaining_.load(std::memory_order_acquire) == 0; });
            }
        }
        call_.unlock();
    }

  private:
    WorkerPool() : pid_(getpid()) {}
    void loop(int id) {
        int64_t seen = 0;
        for (;;) {
            // a job for this worker? spin first (the caller comes back every ~100 us inside a fit), then sleep
            int64_t g = gen_.load(std::memory_order_acquire);
            const auto t0 = std::chrono::steady_clock::now();
            while (g == seen) {
                std::this_thread::yield();
                g = gen_.load(std::memory_order_acquire);
                if (g == seen && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                    g = gen_.load(std::memory_order_acquire);
                }
            }
            seen = g;
            std::function<void()> fn;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (id >= want_) continue;
                fn = job_;
            }
            fn();
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_.notify_all();
            }
        }
    }
    pid_t pid_;
    std::mutex mu_, call_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    std::function<void()> job_;
    int want_ = 0;
    std::atomic<int> remaining_{0};
    std::atomic<int64_t> gen_{0};
};

// Large requests, pipelined (round 6; opt-in, see legacy_randn).  Until round 5 the calling thread drew the words AND formed the candidates AND counted
// the accepted ones (1.5-2 ns per word plus 4 ns per candidate: 160 of the 235 us of a 41 500-value call on the GPU box's
// host); only the logarithms ran elsewhere.  Now the calling thread does nothing but run the generator: it fills batch j's
// 8192 words (a snapshot of the generator's state in front of every batch) and moves on; a worker takes batch j through
//   1. candidates + accept flags + the count a_j of accepted pairs          (needs the words only)
//   2. pair0[j + 1] = pair0[j] + a_j, as soon as pair0[j] is known          (a chain of additions across the workers)
//   3. sqrt(-2 log(r2) / r2) (x2, x1) of its accepted pairs into out[2 (pair0[j] + i)]
// The stream must be consumed up to EXACTLY the candidate that yields pair number npairs - 1: the worker that finds it says
// which batch and which candidate; the calling thread -- which by then has run a few batches ahead -- goes back to that
// batch's snapshot and advances the generator by the words the batch really used.  Values, order and final state are
// those of the one-by-one loop (tests/test_host_logic.py against the installed NumPy).
struct MtSnap {
    uint32_t key[MT_N];
    int pos;
};

template <typename T>
int legacy_randn_pipelined(Mt &mt, T *o, int64_t nout, int64_t npairs, int threads, double *gauss_last) {
    const int64_t expect = (int64_t)((double)npairs / 0.78) + 4 * BATCH;          // candidates, with room
    const int64_t maxb = std::max<int64_t>((2 * npairs + BATCH) / BATCH + 2, expect / BATCH + 2);
    struct Scratch {
        std::vector<uint32_t> W;
        std::vector<MtSnap> snaps;
        std::vector<double> x1, x2;
        std::vector<unsigned char> ok;
    };
    static thread_local Scratch sc;
    try {
        if ((int64_t)sc.snaps.size() < maxb) {
            sc.W.resize((size_t)maxb * 4 * BATCH);
            sc.snaps.resize((size_t)maxb);
            sc.x1.resize((size_t)maxb * BATCH);
            sc.x2.resize((size_t)maxb * BATCH);
            sc.ok.resize((size_t)maxb * BATCH);
        }
    } catch (const std::bad_alloc &) {
        rr_set_error("rr_legacy_randn: out of host memory");
        return RR_ERR_OOM;
    }
    uint32_t *const W = sc.W.data();
    double *const X1 = sc.x1.data(), *const X2 = sc.x2.data();
    unsigned char *const OK = sc.ok.data();
    std::vector<std::atomic<int64_t>> pair0((size_t)maxb + 1);
    for (auto &v : pair0) v.store(-1, std::memory_order_relaxed);
    pair0[0].store(0, std::memory_order_relaxed);
    std::atomic<int64_t> ready(0), claim(0), phase1(0);
    std::atomic<bool> closed(false), done(false);
    int64_t final_batch = -1, final_cand = -1;
    auto worker = [&]() {
        for (;;) {
            const int64_t j = claim.fetch_add(1);
            while (ready.load(std::memory_order_acquire) <= j) {
                if (done.load(std::memory_order_acquire) || closed.load(std::memory_order_acquire)) {
                    if (ready.load(std::memory_order_acquire) <= j) return;
                }
                std::this_thread::yield();
            }
            double *x1 = X1 + j * BATCH, *x2 = X2 + j * BATCH;
            unsigned char *ok = OK + j * BATCH;
            const int64_t a = candidates(W + j * 4 * BATCH, BATCH, x1, x2, ok);
            phase1.fetch_add(1, std::memory_order_release);
            int64_t p0;
            while ((p0 = pair0[(size_t)j].load(std::memory_order_acquire)) < 0) {
                if (done.load(std::memory_order_acquire)) return;   // (a batch behind the last one needed)
                std::this_thread::yield();
            }
            pair0[(size_t)j + 1].store(p0 + a, std::memory_order_release);
            if (p0 >= npairs) {
                done.store(true, std::memory_order_release);
                return;
            }
            const int64_t need = npairs - p0;
            int64_t i = 0;
            for (int64_t q = 0; q < BATCH; ++q) {
                if (!ok[q]) continue;
                emit_pair<T>(x1[q], x2[q], p0 + i, o, nout, gauss_last);
                if (++i == need) {   // the last pair of the request: the stream ends behind this candidate
                    final_batch = j;
                    final_cand = q;
                    done.store(true, std::memory_order_release);
                    return;
                }
            }
        }
    };
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    WorkerPool &pool = WorkerPool::get();
    pool.start(threads, worker);
    int64_t j = 0;
    while (!done.load(std::memory_order_acquire) && j < maxb) {
        // (not more than a few dozen batches ahead of the candidates: what is drawn beyond the end is drawn for nothing)
        while (j - phase1.load(std::memory_order_acquire) > 8 * threads + 8 && !done.load(std::memory_order_acquire)) std::this_thread::yield();
        memcpy(sc.snaps[(size_t)j].key, mt.key, sizeof(mt.key));
        sc.snaps[(size_t)j].pos = mt.pos;
        mt_fill(mt, W + j * 4 * BATCH, 4 * BATCH);
        ++j;
        ready.store(j, std::memory_order_release);
    }
    closed.store(true, std::memory_order_release);
    pool.wait();
    if (final_batch < 0) {
        rr_set_error("rr_legacy_randn: the candidate buffer ran out (%lld batches for %lld pairs)", (long long)maxb, (long long)npairs);
        return RR_ERR_INVALID;
    }
    // back to the state in front of the last batch needed, then the words it really used
    memcpy(mt.key, sc.snaps[(size_t)final_batch].key, sizeof(mt.key));
    mt.pos = sc.snaps[(size_t)final_batch].pos;
    mt.temper_from(mt.pos);
    mt_fill(mt, W, 4 * (final_cand + 1));
    return RR_OK;
}

// The stream is consumed candidate by candidate (four words each) until `npairs` are accepted -- not one word more, or the
// generator's state would differ from NumPy's afterwards.  A batch of c candidates yields at most c pairs, so batches of
// min(pairs still missing, BATCH) candidates can never overshoot; the last < 64 pairs go one by one.  The calling thread
// draws the words, forms the candidates and COUNTS the accepted ones (loops over arrays: no per-candidate dependency);
// worker threads walk the batches behind it, pick the accepted candidates and write sqrt(-2 log(r2) / r2) (x2, x1) straight
// to their places.  (Rounds 2-3: the calling thread also compacted the accepted pairs one by one, a store-to-load chain of
// ~7 ns per candidate: 4.3 of the call's 4.4 ms on the GPU box's host.)
template <typename T>
int legacy_randn(Mt &mt, int32_t *has_gauss, double *gauss, T *out, int64_t n, int threads) {
    int64_t done = 0;
    if (*has_gauss && n > 0) {
        out[done++] = (T)*gauss;
        *has_gauss = 0;
        *gauss = 0.0;
    }
    const int64_t nout = n - done, npairs = (nout + 1) / 2;
    if (npairs == 0) return RR_OK;
    T *o = out + done;
    double gauss_last = 0.0;
    int64_t cnt = 0;  // accepted pairs so far
    // RR_RANDN_PIPELINE=1 (A/B runs): the pipelined form above.  Measured on the GPU box's host at 41 500 values and not
    // adopted: 146 us against 125 us for the batches below on the pool (the generator itself, ~1.2 ns per word on that
    // host, is the bound either way, and the pipeline's extra hand-overs cost more than the candidate loop it moves away)
    static const bool pipe = getenv("RR_RANDN_PIPELINE") != nullptr && atoi(getenv("RR_RANDN_PIPELINE")) != 0;
    if (npairs >= 4 * BATCH && pipe) {
        const int rc = legacy_randn_pipelined<T>(mt, o, nout, npairs, threads, &gauss_last);
        if (rc != RR_OK) return rc;
        cnt = npairs;
    } else if (npairs >= 64) {
        // room for twice the pairs in candidates (78.5 % are accepted); should a stream ever need more, the rest goes one by one
        const size_t cap = (size_t)(2 * npairs + BATCH);
        static thread_local CandBuf scratch;
        if (!scratch.reserve(cap)) {
            rr_set_error("rr_legacy_randn: out of host memory");
            return RR_ERR_OOM;
        }
        // (the workers must see THIS thread's buffers: a thread_local named inside their lambda would be their own, empty one)
        double *const X1 = scratch.x1, *const X2 = scratch.x2;
        unsigned char *const OK = scratch.ok;
        std::vector<Batch> batches;
        batches.reserve((size_t)(cap / 64 + 16));  // (never reallocated while workers read it)
        std::atomic<int64_t> ready(0), claim(0);
        std::atomic<bool> closed(false);
        auto worker = [&]() {
            for (;;) {
                const int64_t j = claim.fetch_add(1);
                while (ready.load(std::memory_order_acquire) <= j) {
                    if (closed.load(std::memory_order_acquire) && ready.load(std::memory_order_acquire) <= j) return;
                    std::this_thread::yield();
                }
                const Batch &b = batches[(size_t)j];
                const double *x1 = X1 + b.cand0, *x2 = X2 + b.cand0;
                const unsigned char *ok = OK + b.cand0;
                int64_t i = b.pair0;
                for (int64_t q = 0; q < b.c; ++q)
                    if (ok[q]) emit_pair<T>(x1[q], x2[q], i++, o, nout, &gauss_last);
            }
        };
        if (threads < 1) threads = 1;
        if (threads > 16) threads = 16;
        // (from 2 batches on the pick runs on the pool: at 41 500 values -- an SVI step of the reference's default sizes --
        // the logarithms and square roots are 170 of the call's 235 us on one thread)
        const bool pooled = npairs >= 2 * BATCH;
        if (pooled) WorkerPool::get().start(threads, worker);
        uint32_t w[4 * BATCH];
        int64_t cand = 0;
        while (npairs - cnt >= 64 && (size_t)(cand + BATCH) <= cap && batches.size() < batches.capacity()) {
            const int64_t c = npairs - cnt < BATCH ? npairs - cnt : BATCH;
            mt_fill(mt, w, 4 * c);
            const int64_t acc = candidates(w, c, X1 + cand, X2 + cand, OK + cand);
            batches.push_back({cand, c, cnt});
            ready.store((int64_t)batches.size(), std::memory_order_release);
            cand += c;
            cnt += acc;
        }
        closed.store(true, std::memory_order_release);
        if (pooled) WorkerPool::get().wait();
        else worker();
    }
    // the last pairs (fewer than 64, or all of a short request) one by one, as legacy_gauss does
    while (cnt < npairs) {
        double x1, x2, r2;
        do {
            x1 = 2.0 * mt.next_double() - 1.0;
            x2 = 2.0 * mt.next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        emit_pair<T>(x1, x2, cnt++, o, nout, &gauss_last);
    }
    if (nout & 1) {
        *gauss = gauss_last;
        *has_gauss = 1;
    }
    return RR_OK;
}

}  // namespace

// RandomState.permutation(n) (numpy/random/mtrand.pyx: `arr = arange(n); shuffle(arr)`; _shuffle_raw: for i = n-1 .. 1:
// j = random_interval(i), swap(arr[i], arr[j]); random_interval, numpy/random/src/distributions/distributions.c: the smallest
// bit mask >= max, 32-bit words masked and rejected while > max).  NumPy runs this loop holding the GIL -- 18 ms at n = 2M, during
// which no other Python thread (the SVI loop's consumer among them) advances; here it runs without it.
extern "C" int rr_legacy_permutation(uint32_t *key, int32_t *pos, int64_t n, int64_t *out) {
    RR_REQUIRE(key != nullptr && pos != nullptr, "rr_legacy_permutation: null state");
    RR_REQUIRE(*pos >= 0 && *pos <= MT_N, "rr_legacy_permutation: position %d outside the MT19937 state", (int)*pos);
    RR_REQUIRE(n >= 0 && n <= ((int64_t)1 << 32) && (n == 0 || out != nullptr), "rr_legacy_permutation: 0 <= n <= 2^32");
    Mt mt;
    memcpy(mt.key, key, sizeof(mt.key));
    mt.pos = (int)*pos;
    mt.temper_from(mt.pos);
    // The swap chain is a walk of random addresses (cache misses, one after the other).  The partners j do not depend on the
    // array, so they are drawn a block ahead and the lines they name are prefetched a few swaps before they are needed; below
    // 2^31 entries the walk is over 32-bit values (half the footprint: 8 MB at n = 2M) in a grow-only scratch of the calling
    // thread, widened into `out` at the end.
    constexpr int JB = 512, AHEAD = 24;
    uint64_t J[JB];
    auto walk = [&](auto *a) {
        for (int64_t i = 0; i < n; ++i) a[i] = (decltype(a[0] + 0))i;
        int64_t i = n - 1;
        while (i >= 1) {
            const int c = (int)(i < JB ? i : JB);
            for (int q = 0; q < c; ++q) {
                const uint64_t ii = (uint64_t)(i - q);
                uint64_t mask = ii;
                mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
                uint64_t j;
                do {
                    j = (uint64_t)mt.next() & mask;
                } while (j > ii);
                J[q] = j;
            }
            for (int q = 0; q < AHEAD && q < c; ++q) __builtin_prefetch(&a[J[q]], 1);
            for (int q = 0; q < c; ++q) {
                if (q + AHEAD < c) __builtin_prefetch(&a[J[q + AHEAD]], 1);
                const int64_t u = i - q;
                const auto t = a[J[q]];
                a[J[q]] = a[u];
                a[u] = t;
            }
            i -= c;
        }
    };
    if (n < ((int64_t)1 << 31) && n >= 65536) {
        static thread_local std::vector<uint32_t> scratch;
        if ((int64_t)scratch.size() < n) scratch.resize((size_t)n);
        uint32_t *a = scratch.data();
        walk(a);
        for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)a[i];
    } else {
        walk(out);
    }
    memcpy(key, mt.key, sizeof(mt.key));
    *pos = mt.pos;
    return RR_OK;
}

extern "C" int rr_legacy_randn(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, void *out, int out_dtype,
                               int64_t n, int threads) {
    RR_REQUIRE(key != nullptr && pos != nullptr && has_gauss != nullptr && gauss != nullptr, "rr_legacy_randn: null state");
    RR_REQUIRE(*pos >= 0 && *pos <= MT_N, "rr_legacy_randn: position %d outside the MT19937 state", (int)*pos);
    RR_REQUIRE(n >= 0 && (n == 0 || out != nullptr), "rr_legacy_randn: bad output");
    RR_REQUIRE(out_dtype == RR_F32 || out_dtype == RR_F64, "rr_legacy_randn: bad dtype");
    Mt mt;
    memcpy(mt.key, key, sizeof(mt.key));
    mt.pos = (int)*pos;
    mt.temper_from(mt.pos);
    const int rc = out_dtype == RR_F32 ? legacy_randn<float>(mt, has_gauss, gauss, (float *)out, n, threads)
                                       : legacy_randn<double>(mt, has_gauss, gauss, (double *)out, n, threads);
    memcpy(key, mt.key, sizeof(mt.key));
    *pos = mt.pos;
    return rc;
}
