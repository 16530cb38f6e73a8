// Host generator micro-benchmark (tools/probes): ns per MT19937 word, per polar-method candidate, per accepted pair's transform.
//   clang++ -O3 -std=c++17 rng_probe.cpp -o rng_probe && ./rng_probe      (the functions are copied from rr_hostrng.hip by tools/probes/make_rng_probe.py)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <chrono>
#include <vector>
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


}
int main(){
  Mt mt; for(int i=0;i<624;i++) mt.key[i]=i*2654435761u+12345; mt.pos=624;
  const int64_t NW=106000*20; std::vector<uint32_t> w(NW);
  auto t0=std::chrono::steady_clock::now();
  for(int rep=0;rep<5;rep++) mt_fill(mt,w.data(),NW);
  auto t1=std::chrono::steady_clock::now();
  printf("mt_fill: %.3f ns/word\n",std::chrono::duration<double,std::nano>(t1-t0).count()/(5.0*NW));
  int64_t c=NW/4; std::vector<double> x1(c),x2(c); std::vector<unsigned char> ok(c);
  t0=std::chrono::steady_clock::now(); int64_t acc=0;
  for(int rep=0;rep<5;rep++) acc+=candidates(w.data(),c,x1.data(),x2.data(),ok.data());
  t1=std::chrono::steady_clock::now();
  printf("candidates: %.3f ns/candidate (accepted %.3f)\n",std::chrono::duration<double,std::nano>(t1-t0).count()/(5.0*c),(double)acc/(5.0*c));
  t0=std::chrono::steady_clock::now(); double sum=0;
  for(int64_t q=0;q<c;q++) if(ok[q]){ double r2=x1[q]*x1[q]+x2[q]*x2[q]; double f=sqrt(-2.0*log(r2)/r2); sum+=f; }
  t1=std::chrono::steady_clock::now();
  printf("transform: %.3f ns/candidate (%g)\n",std::chrono::duration<double,std::nano>(t1-t0).count()/(1.0*c),sum);
}
