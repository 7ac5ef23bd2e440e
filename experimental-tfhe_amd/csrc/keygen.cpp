// keygen.cpp -- host-side harness of the C ABI: synthetic secret keys, bootstrapping /
// key-switch keys and LWE samples, i.e. the engine's counterpart of what the reference's
// PoC does around the hot path (CB/poc_CircuitBootstrapping.cpp:88-134 encrypt/phase,
// :191-227 TLWE/TGSW encryption, :342-423 Globals key generation; CB/lwe_functions.cpp:116-133
// key-switch key).  Pure host C++, no device code.
//
// Randomness is a counter-based PRNG (splitmix64) so that keys are reproducible across
// processes and ranks; the same specification is restated by the test oracle
// (oracle/tfhe_oracle.c), which is how tests pin this file bit for bit:
//   next():  s += 0x9E3779B97F4A7C15; z = s; z = (z^(z>>30))*0xBF58476D1CE4E5B9;
//            z = (z^(z>>27))*0x94D049BB133111EB; return z^(z>>31)
//   init(seed, stream): s = seed; a = next(); s = a ^ (stream*0xD1342543DE82EF95 + 0x632BE59BD9B4E019)
//   torus32 = high 32 bits of next(); torus64 = next(); key bit = top bit of next()
//   gauss = sqrt(-2 ln u1) cos(2 pi u2), u1 = ((next()>>11)+1) 2^-53, u2 = (next()>>11) 2^-53
//   noise = truncation of gauss*stdev*2^W (as generic_utils.h:172-185)
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/tfhe_amd.h"

namespace {

struct Rng {
    uint64_t s;
    Rng(uint64_t seed, uint64_t stream) : s(seed) {
        const uint64_t a = next();
        s = a ^ (stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull);
    }
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    int32_t torus32() { return (int32_t)(uint32_t)(next() >> 32); }
    int64_t torus64() { return (int64_t)next(); }
    double gauss() {
        const double u1 = (double)((next() >> 11) + 1) * 0x1p-53;
        const double u2 = (double)(next() >> 11) * 0x1p-53;
        return sqrt(-2. * log(u1)) * cos(2. * M_PI * u2);
    }
    int32_t noise32(double stdev) { return (int32_t)(int64_t)(gauss() * stdev * 0x1p32); }
    int64_t noise64(double stdev) { return (int64_t)(gauss() * stdev * 0x1p64); }
};

template <typename T>
struct Wide;
template <>
struct Wide<int32_t> {
    using U = uint32_t;
    static int32_t uniform(Rng &r) { return r.torus32(); }
    static int32_t noise(Rng &r, double sd) { return r.noise32(sd); }
    static constexpr int BITS = 32;
};
template <>
struct Wide<int64_t> {
    using U = uint64_t;
    static int64_t uniform(Rng &r) { return r.torus64(); }
    static int64_t noise(Rng &r, double sd) { return r.noise64(sd); }
    static constexpr int BITS = 64;
};

// b += a * s over Z[X]/(X^N+1) for a 0/1 secret s: one shifted add per set key bit
template <typename T>
void add_a_times_key(T *b, const T *a, const int32_t *key, int N) {
    using U = typename Wide<T>::U;
    for (int sft = 0; sft < N; sft++) {
        if (!key[sft]) continue;
        for (int i = 0; i + sft < N; i++) b[i + sft] = (T)((U)b[i + sft] + (U)a[i]);
        for (int i = N - sft; i < N; i++) b[i + sft - N] = (T)((U)b[i + sft - N] - (U)a[i]);
    }
}

// TLWE encryption of zero (poc:143-152): noise into b, uniform a, b += a*s
template <typename T>
void tlwe_zero(T *ct, double stdev, const int32_t *key, int N, Rng &r) {
    for (int j = 0; j < N; j++) ct[N + j] = Wide<T>::noise(r, stdev);
    for (int j = 0; j < N; j++) ct[j] = Wide<T>::uniform(r);
    add_a_times_key<T>(ct + N, ct, key, N);
}

// TGSW encryption of `mess` (poc:215-227): rows bloc*l+i, message mess*2^(W-(i+1)Bgbit) on a[bloc][0]
template <typename T>
void tgsw_encrypt(T *gsw, int32_t mess, double stdev, const int32_t *key, int N, int l, int Bgbit, Rng &r) {
    using U = typename Wide<T>::U;
    for (int bloc = 0; bloc <= 1; bloc++)
        for (int i = 0; i < l; i++) {
            T *row = gsw + (size_t)(bloc * l + i) * 2 * N;
            tlwe_zero<T>(row, stdev, key, N, r);
            const U h = (U)1 << (Wide<T>::BITS - (i + 1) * Bgbit);
            row[bloc * N] = (T)((U)row[bloc * N] + (U)(T)mess * h);
        }
}

template <typename T>
int keygen_bk(T *bk, const int32_t *lwe_key, int n, const int32_t *tkey, int N, int l, int Bgbit, double stdev,
              uint64_t seed, uint64_t stream) {
    if (!bk || !lwe_key || !tkey || n < 1 || N < 2 || l < 1) return TFHE_AMD_ERR_PARAM;
    const size_t sample = (size_t)2 * l * 2 * N;
    for (int i = 0; i < n; i++) {
        Rng r(seed, stream + (uint64_t)i);  // one stream per key element
        tgsw_encrypt<T>(bk + sample * i, lwe_key[i], stdev, tkey, N, l, Bgbit, r);
    }
    return TFHE_AMD_OK;
}

void lwe_encrypt(int32_t *ct, int32_t mess, double stdev, const int32_t *key, int n, Rng &r) {
    uint32_t b = (uint32_t)mess + (uint32_t)r.noise32(stdev);
    for (int i = 0; i < n; i++) {
        ct[i] = r.torus32();
        b += (uint32_t)ct[i] * (uint32_t)key[i];
    }
    ct[n] = (int32_t)b;
}

}  // namespace

extern "C" {

int tfhe_amd_keygen_binary(int32_t *key, int n, uint64_t seed, uint64_t stream) {
    if (!key || n < 0) return TFHE_AMD_ERR_PARAM;
    Rng r(seed, stream);
    for (int i = 0; i < n; i++) key[i] = (int32_t)(r.next() >> 63);
    return TFHE_AMD_OK;
}

int tfhe_amd_lwe_encrypt32(int32_t *ct, int32_t mess, double stdev, const int32_t *key, int n, uint64_t seed,
                           uint64_t stream) {
    if (!ct || !key || n < 1) return TFHE_AMD_ERR_PARAM;
    Rng r(seed, stream);
    lwe_encrypt(ct, mess, stdev, key, n, r);
    return TFHE_AMD_OK;
}

int32_t tfhe_amd_lwe_phase32(const int32_t *ct, const int32_t *key, int n) {
    uint32_t res = (uint32_t)ct[n];
    for (int i = 0; i < n; i++) res -= (uint32_t)ct[i] * (uint32_t)key[i];
    return (int32_t)res;
}

int tfhe_amd_keygen_bk_torus32(int32_t *bk, const int32_t *lwe_key, int n, const int32_t *tlwe_key, int N, int l,
                               int Bgbit, double stdev, uint64_t seed, uint64_t stream) {
    return keygen_bk<int32_t>(bk, lwe_key, n, tlwe_key, N, l, Bgbit, stdev, seed, stream);
}
int tfhe_amd_keygen_bk_torus64(int64_t *bk, const int32_t *lwe_key, int n, const int32_t *tlwe_key, int N, int l,
                               int Bgbit, double stdev, uint64_t seed, uint64_t stream) {
    return keygen_bk<int64_t>(bk, lwe_key, n, tlwe_key, N, l, Bgbit, stdev, seed, stream);
}

int tfhe_amd_keygen_ks32(int32_t *ks, const int32_t *in_key, int n_in, const int32_t *out_key, int n_out, int t,
                         int basebit, double stdev, uint64_t seed, uint64_t stream) {
    if (!ks || !in_key || !out_key || n_in < 1 || n_out < 1 || t < 1 || basebit < 1) return TFHE_AMD_ERR_PARAM;
    const int base = 1 << basebit;
    const size_t row = (size_t)n_out + 1;
    for (int i = 0; i < n_in; i++) {
        Rng r(seed, stream + (uint64_t)i);  // one stream per input-key element
        for (int j = 0; j < t; j++)
            for (int u = 0; u < base; u++) {
                // poc:379 / lwe_functions.cpp:128: (key_i * u) * 2^(32-(j+1)basebit)
                const int32_t mess = (int32_t)(((uint32_t)in_key[i] << (32 - (j + 1) * basebit)) * (uint32_t)u);
                lwe_encrypt(ks + (((size_t)i * t + j) * base + u) * row, mess, stdev, out_key, n_out, r);
            }
    }
    return TFHE_AMD_OK;
}

}  // extern "C"
