/*
 * tfhe_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See tfhe_oracle.h for scope, conventions and parity status.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma (oracle/Makefile).  Every fused
 * multiply-add of the reference's FMA assembly is written as an explicit fma();
 * everything else must stay un-contracted, hence -ffp-contract=off.
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/* ------------------------------------------------------------------ tables */

/* accurate_cos / accurate_sin, CB/spqlios/spqlios-fft-impl.cpp:99-113:
 * cos/sin(2*pi*i/n) with the angle folded into the first quadrant. */
static double fold_cos(int i, int n) {
    i = ((i % n) + n) % n;
    if (i >= 3 * n / 4) return cos(2. * M_PI * (n - i) / (double)n);
    if (i >= 2 * n / 4) return -cos(2. * M_PI * (i - n / 2) / (double)n);
    if (i >= 1 * n / 4) return -cos(2. * M_PI * (n / 2 - i) / (double)n);
    return cos(2. * M_PI * (i) / (double)n);
}
static double fold_sin(int i, int n) {
    i = ((i % n) + n) % n;
    if (i >= 3 * n / 4) return -sin(2. * M_PI * (n - i) / (double)n);
    if (i >= 2 * n / 4) return -sin(2. * M_PI * (i - n / 2) / (double)n);
    if (i >= 1 * n / 4) return sin(2. * M_PI * (n / 2 - i) / (double)n);
    return sin(2. * M_PI * (i) / (double)n);
}

/* Tables keep the reference's packing, groups of [4 cos | 4 sin]
 * (spqlios-fft-impl.cpp:62-66).  Entry e of the block starting at `blk`: */
static inline double TC(const double *blk, int e) { return blk[8 * (e >> 2) + (e & 3)]; }
static inline double TS(const double *blk, int e) { return blk[8 * (e >> 2) + 4 + (e & 3)]; }

static double *emit_block(double *p, int count, int mult, int n) {
    for (int i = 0; i < count; i += 4) {
        for (int k = 0; k < 4; k++) *(p++) = fold_cos(mult * (i + k), n);
        for (int k = 0; k < 4; k++) *(p++) = fold_sin(mult * (i + k), n);
    }
    return p;
}

orc_tables *orc_tables_new(int N) {
    if (N < 16 || (N & (N - 1))) return NULL;
    orc_tables *t = (orc_tables *)calloc(1, sizeof(*t));
    const int n = 2 * N, ns4 = N / 2;
    t->N = N;
    t->ns4 = ns4;
    t->tab_len = 4 * ns4 - 8;
    t->ifft_trig = (double *)malloc(sizeof(double) * (size_t)t->tab_len);
    t->fft_trig = (double *)malloc(sizeof(double) * (size_t)t->tab_len);
    /* new_ifft_table, spqlios-fft-impl.cpp:400-437: twist, then nn = ns4 .. 8 */
    double *p = emit_block(t->ifft_trig, ns4, 1, n);
    for (int nn = ns4; nn >= 8; nn /= 2) p = emit_block(p, nn / 2, n / nn, n);
    /* new_fft_table, spqlios-fft-impl.cpp:158-193: halfnn = 4 .. ns4/2, then twist */
    p = t->fft_trig;
    for (int halfnn = 4; halfnn < ns4; halfnn *= 2) p = emit_block(p, halfnn, -(n / (2 * halfnn)), n);
    p = emit_block(p, ns4, -1, n);
    return t;
}
void orc_tables_free(orc_tables *t) {
    if (!t) return;
    free(t->ifft_trig);
    free(t->fft_trig);
    free(t);
}
const double *orc_tables_ifft_trig(const orc_tables *t) { return t->ifft_trig; }
const double *orc_tables_fft_trig(const orc_tables *t) { return t->fft_trig; }
int orc_tables_len(const orc_tables *t) { return t->tab_len; }

/* -------------------------------------------------------------------- ifft */
/* spqlios-ifft-fma.s; operation DAG per SURVEY App. A.2 */
void orc_ifft(const orc_tables *t, double *data) {
    const int ns4 = t->ns4;
    double *re = data, *im = data + ns4;
    const double *tt = t->ifft_trig;
    /* twist by omega^j (:63-78): products re*c, re*s rounded, then one fma each */
    for (int j = 0; j < ns4; j++) {
        const double c = TC(tt, j), s = TS(tt, j);
        const double r = re[j], i = im[j];
        re[j] = FMA(-i, s, r * c);
        im[j] = FMA(i, c, r * s);
    }
    /* DIF stages nn = ns4 .. 8 (:113-157) */
    const double *blk = tt;
    for (int nn = ns4; nn >= 8; nn /= 2) {
        const int h = nn / 2;
        blk += 2 * nn;
        for (int b = 0; b < ns4; b += nn) {
            for (int o = 0; o < h; o++) {
                const int i0 = b + o, i1 = i0 + h;
                const double c = TC(blk, o), s = TS(blk, o);
                const double sr = re[i0] + re[i1], si = im[i0] + im[i1];
                const double dr = re[i0] - re[i1], di = im[i0] - im[i1];
                re[i0] = sr;
                im[i0] = si;
                re[i1] = FMA(-di, s, dr * c);
                im[i1] = FMA(di, c, dr * s);
            }
        }
    }
    /* size 4 (:194-213) */
    for (int b = 0; b < ns4; b += 4) {
        const double r0 = re[b], r1 = re[b + 1], r2 = re[b + 2], r3 = re[b + 3];
        const double i0 = im[b], i1 = im[b + 1], i2 = im[b + 2], i3 = im[b + 3];
        re[b] = r0 + r2;
        re[b + 1] = r1 + r3;
        re[b + 2] = r0 - r2;
        re[b + 3] = i3 - i1;
        im[b] = i0 + i2;
        im[b + 1] = i1 + i3;
        im[b + 2] = i0 - i2;
        im[b + 3] = r1 - r3;
    }
    /* size 2 (:247-263) */
    for (int b = 0; b < ns4; b += 2) {
        const double r0 = re[b], r1 = re[b + 1], i0 = im[b], i1 = im[b + 1];
        re[b] = r0 + r1;
        re[b + 1] = r0 - r1;
        im[b] = i0 + i1;
        im[b + 1] = i0 - i1;
    }
}

/* --------------------------------------------------------------------- fft */
/* spqlios-fft-fma.s; operation DAG per SURVEY App. A.3 */
void orc_fft(const orc_tables *t, double *data) {
    const int ns4 = t->ns4;
    double *re = data, *im = data + ns4;
    /* size 2 (:79-95) */
    for (int b = 0; b < ns4; b += 2) {
        const double r0 = re[b], r1 = re[b + 1], i0 = im[b], i1 = im[b + 1];
        re[b] = r0 + r1;
        re[b + 1] = r0 - r1;
        im[b] = i0 + i1;
        im[b + 1] = i0 - i1;
    }
    /* size 4 (:134-152) */
    for (int b = 0; b < ns4; b += 4) {
        const double r0 = re[b], r1 = re[b + 1], r2 = re[b + 2], r3 = re[b + 3];
        const double i0 = im[b], i1 = im[b + 1], i2 = im[b + 2], i3 = im[b + 3];
        re[b] = r0 + r2;
        re[b + 1] = r1 + i3;
        re[b + 2] = r0 - r2;
        re[b + 3] = r1 - i3;
        im[b] = i0 + i2;
        im[b + 1] = i1 - r3;
        im[b + 2] = i0 - i2;
        im[b + 3] = i1 + r3;
    }
    /* DIT stages halfnn = 4 .. ns4/2 (:189-234) */
    const double *blk = t->fft_trig;
    for (int h = 4; h < ns4; h *= 2) {
        const int nn = 2 * h;
        for (int b = 0; b < ns4; b += nn) {
            for (int o = 0; o < h; o++) {
                const int i0 = b + o, i1 = i0 + h;
                const double c = TC(blk, o), s = TS(blk, o);
                const double tr = FMA(-im[i1], s, re[i1] * c);
                const double ti = FMA(im[i1], c, re[i1] * s);
                const double r0 = re[i0], j0 = im[i0];
                re[i1] = r0 - tr;
                im[i1] = j0 - ti;
                re[i0] = r0 + tr;
                im[i0] = j0 + ti;
            }
        }
        blk += nn;
    }
    /* final twist (:255-274), four rounded products, no fma */
    for (int j = 0; j < ns4; j++) {
        const double c = TC(blk, j), s = TS(blk, j);
        const double r = re[j], i = im[j];
        const double rc = r * c, rs = r * s, ic = i * c, is = i * s;
        re[j] = rc - is;
        im[j] = rs + ic;
    }
}

/* ------------------------------------------------------- plugin conversions */

void orc_execute_reverse_int(const orc_tables *t, double *res, const int32_t *a) {
    for (int i = 0; i < t->N; i++) res[i] = (double)a[i]; /* vcvtdq2pd, exact */
    orc_ifft(t, res);
}
void orc_execute_reverse_torus32(const orc_tables *t, double *res, const int32_t *a) {
    orc_execute_reverse_int(t, res, a);
}
void orc_execute_direct_torus32(const orc_tables *t, int32_t *res, const double *a) {
    const int N = t->N;
    const double s = 2. / (double)N;
    double *buf = (double *)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; i++) buf[i] = a[i] * s;
    orc_fft(t, buf);
    /* fft_processor_spqlios.cpp:102 int32_t(int64_t(x)): undefined beyond the int64 range; the reference as
       compiled (cvttsd2si -> 0x8000000000000000) yields 0 there, pinned by tests/test_oracle_golden.py */
    for (int i = 0; i < N; i++) res[i] = (fabs(buf[i]) < 0x1p63) ? (int32_t)(int64_t)buf[i] : 0;
    free(buf);
}
void orc_execute_reverse_torus64(const orc_tables *t, double *res, const int64_t *a) {
    for (int i = 0; i < t->N; i++) res[i] = (double)a[i]; /* round to nearest even */
    orc_ifft(t, res);
}
/* double -> Torus64 exactly as fft_processor_spqlios.cpp:131-142: 53-bit mantissa
 * shifted by (exponent-1075), truncation toward zero, modulo 2^64.  A right shift of
 * 64 or more (|x| < 2^-11, undefined behaviour in the reference) is DEFINED as 0. */
static int64_t dtot64(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const uint64_t mant = (bits & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
    const int expo = (int)((bits >> 52) & 0x7FF);
    const int trans = expo - 1075;
    uint64_t v;
    if (trans > 0)
        v = trans >= 64 ? 0 : (mant << trans);
    else
        v = (-trans) >= 64 ? 0 : (mant >> (-trans));
    return (int64_t)((bits >> 63) ? (0 - v) : v);
}
void orc_execute_direct_torus64(const orc_tables *t, int64_t *res, const double *a) {
    const int N = t->N;
    const double s = 2. / (double)N;
    double *buf = (double *)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; i++) buf[i] = a[i] * s;
    orc_fft(t, buf);
    for (int i = 0; i < N; i++) res[i] = dtot64(buf[i]);
    free(buf);
}

/* lagrangehalfc_impl_fma.s:96-107: two chained fmas per component */
void orc_lagrange_addmul(double *res, const double *a, const double *b, long Ns2) {
    for (long i = 0; i < Ns2; i++) {
        const double ar = a[i], ai = a[Ns2 + i], br = b[i], bi = b[Ns2 + i];
        const double tneg = FMA(ai, bi, -res[i]); /* vfmsub231pd: ai*bi - rr        */
        res[i] = FMA(ar, br, -tneg);               /* vfmsub231pd: ar*br - (above)  */
        const double u = FMA(ar, bi, res[Ns2 + i]);
        res[Ns2 + i] = FMA(ai, br, u);
    }
}

/* ------------------------------------------------------ exact integer ring */

void orc_negacyclic_mul32(int32_t *res, const int32_t *ipoly, const int32_t *tpoly, int N) {
    for (int i = 0; i < N; i++) {
        uint32_t ri = 0;
        for (int j = 0; j <= i; j++) ri += (uint32_t)ipoly[j] * (uint32_t)tpoly[i - j];
        for (int j = i + 1; j < N; j++) ri -= (uint32_t)ipoly[j] * (uint32_t)tpoly[N + i - j];
        res[i] = (int32_t)ri;
    }
}
void orc_negacyclic_mul64(int64_t *res, const int32_t *ipoly, const int64_t *tpoly, int N) {
    for (int i = 0; i < N; i++) {
        uint64_t ri = 0;
        for (int j = 0; j <= i; j++) ri += (uint64_t)(int64_t)ipoly[j] * (uint64_t)tpoly[i - j];
        for (int j = i + 1; j < N; j++) ri -= (uint64_t)(int64_t)ipoly[j] * (uint64_t)tpoly[N + i - j];
        res[i] = (int64_t)ri;
    }
}

/* ----------------------------------------------------------- decomposition */

void orc_decomp32(int32_t *out, const int32_t *in, int N, int l, int Bgbit) {
    const uint32_t Bg = 1u << Bgbit, mask = Bg - 1, halfBg = Bg / 2;
    uint32_t offset = 0; /* TGswParams ctor, tgsw_functions.cpp:29-35 (no rounding bit) */
    for (int i = 0; i < l; i++) offset += 1u << (32 - (i + 1) * Bgbit);
    offset *= halfBg;
    for (int p = 0; p < l; p++) {
        const int decal = 32 - (p + 1) * Bgbit;
        for (int j = 0; j < N; j++) {
            const uint32_t v = (uint32_t)in[j] + offset;
            out[p * N + j] = (int32_t)(((v >> decal) & mask) - halfBg);
        }
    }
}
void orc_decomp64(int32_t *out, const int64_t *in, int N, int l, int Bgbit) {
    const uint64_t Bg = 1ull << Bgbit, mask = Bg - 1;
    const int32_t halfBg = (int32_t)(Bg / 2);
    uint64_t offset = 0; /* poc:349-350: i = 0..l inclusive => includes the rounding bit */
    for (int i = 0; i <= l; i++) offset |= 1ull << (63 - i * Bgbit);
    for (int p = 0; p < l; p++) {
        const int decal = 64 - (p + 1) * Bgbit;
        for (int j = 0; j < N; j++) {
            const uint64_t v = (uint64_t)in[j] + offset;
            out[p * N + j] = (int32_t)(uint32_t)((v >> decal) & mask) - halfBg;
        }
    }
}

/* ------------------------------------------------------- X^a multiplications */

#define DEF_XAI(SUF, T, U)                                                              \
    void orc_mul_xai_minus_one##SUF(T *out, int a, const T *in, int N) {                \
        if (a < N) {                                                                    \
            for (int i = 0; i < a; i++) out[i] = (T)(-(U)in[i - a + N] - (U)in[i]);     \
            for (int i = a; i < N; i++) out[i] = (T)((U)in[i - a] - (U)in[i]);          \
        } else {                                                                        \
            const int aa = a - N;                                                       \
            for (int i = 0; i < aa; i++) out[i] = (T)((U)in[i - aa + N] - (U)in[i]);    \
            for (int i = aa; i < N; i++) out[i] = (T)(-(U)in[i - aa] - (U)in[i]);       \
        }                                                                               \
    }                                                                                   \
    void orc_mul_xai##SUF(T *out, int a, const T *in, int N) {                          \
        if (a < N) {                                                                    \
            for (int i = 0; i < a; i++) out[i] = (T)(-(U)in[i - a + N]);                \
            for (int i = a; i < N; i++) out[i] = in[i - a];                             \
        } else {                                                                        \
            const int aa = a - N;                                                       \
            for (int i = 0; i < aa; i++) out[i] = in[i - aa + N];                       \
            for (int i = aa; i < N; i++) out[i] = (T)(-(U)in[i - aa]);                  \
        }                                                                               \
    }
DEF_XAI(32, int32_t, uint32_t)
DEF_XAI(64, int64_t, uint64_t)

/* --------------------------------------------------------- external product */

void orc_extprod32(const orc_tables *t, int32_t *acc, const double *gsw, int l, int Bgbit) {
    const int N = t->N, kpl = 2 * l;
    int32_t *deca = (int32_t *)malloc(sizeof(int32_t) * (size_t)kpl * N);
    double *decaFFT = (double *)malloc(sizeof(double) * (size_t)kpl * N);
    double *tmpa = (double *)calloc((size_t)2 * N, sizeof(double)); /* tLweFFTClear */
    for (int i = 0; i <= 1; i++) orc_decomp32(deca + i * l * N, acc + i * N, N, l, Bgbit);
    for (int p = 0; p < kpl; p++) orc_execute_reverse_int(t, decaFFT + p * N, deca + p * N);
    for (int p = 0; p < kpl; p++)          /* tgsw_functions.cpp:441-443: p outer ... */
        for (int q = 0; q <= 1; q++)       /* ... tLweFFTAddMulRTo: q inner (tlwe :318-325) */
            orc_lagrange_addmul(tmpa + q * N, decaFFT + p * N, gsw + ((size_t)p * 2 + q) * N, N / 2);
    for (int q = 0; q <= 1; q++) orc_execute_direct_torus32(t, acc + q * N, tmpa + q * N);
    free(deca);
    free(decaFFT);
    free(tmpa);
}
void orc_extprod64(const orc_tables *t, int64_t *acc, const double *gsw, int l, int Bgbit) {
    const int N = t->N, kpl = 2 * l;
    int32_t *deca = (int32_t *)malloc(sizeof(int32_t) * (size_t)kpl * N);
    double *decaFFT = (double *)malloc(sizeof(double) * (size_t)kpl * N);
    double *tmpa = (double *)calloc((size_t)2 * N, sizeof(double));
    for (int i = 0; i <= 1; i++) orc_decomp64(deca + i * l * N, acc + i * N, N, l, Bgbit);
    for (int p = 0; p < kpl; p++) orc_execute_reverse_int(t, decaFFT + p * N, deca + p * N);
    for (int p = 0; p < kpl; p++)
        for (int q = 0; q <= 1; q++)
            orc_lagrange_addmul(tmpa + q * N, decaFFT + p * N, gsw + ((size_t)p * 2 + q) * N, N / 2);
    for (int q = 0; q <= 1; q++) orc_execute_direct_torus64(t, acc + q * N, tmpa + q * N);
    free(deca);
    free(decaFFT);
    free(tmpa);
}

void orc_mux_rotate32(const orc_tables *t, int32_t *out, const int32_t *acc, const double *bki,
                      int barai, int l, int Bgbit) {
    const int N = t->N;
    for (int q = 0; q <= 1; q++) orc_mul_xai_minus_one32(out + q * N, barai, acc + q * N, N);
    orc_extprod32(t, out, bki, l, Bgbit);
    for (int j = 0; j < 2 * N; j++) out[j] = (int32_t)((uint32_t)out[j] + (uint32_t)acc[j]);
}

void orc_blind_rotate32(const orc_tables *t, int32_t *acc, const double *bkfft, const int32_t *bara,
                        int n, int l, int Bgbit) {
    const int N = t->N;
    const size_t row = (size_t)2 * l * 2 * N;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)2 * N);
    for (int i = 0; i < n; i++) {
        if (bara[i] == 0) continue; /* lwe_functions.cpp:348-350 */
        orc_mux_rotate32(t, tmp, acc, bkfft + row * i, bara[i], l, Bgbit);
        memcpy(acc, tmp, sizeof(int32_t) * (size_t)2 * N);
    }
    free(tmp);
}
void orc_blind_rotate64(const orc_tables *t, int64_t *acc, const double *bkfft, const int32_t *bara,
                        int n, int l, int Bgbit) {
    const int N = t->N;
    const size_t row = (size_t)2 * l * 2 * N;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)2 * N);
    for (int i = 0; i < n; i++) {
        if (bara[i] == 0) continue;
        for (int q = 0; q <= 1; q++) orc_mul_xai_minus_one64(tmp + q * N, bara[i], acc + q * N, N);
        orc_extprod64(t, tmp, bkfft + row * i, l, Bgbit);
        for (int j = 0; j < 2 * N; j++) acc[j] = (int64_t)((uint64_t)acc[j] + (uint64_t)tmp[j]);
    }
    free(tmp);
}

void orc_sample_extract32(int32_t *lwe, const int32_t *acc, int N) {
    lwe[0] = acc[0];
    for (int j = 1; j < N; j++) lwe[j] = (int32_t)(0u - (uint32_t)acc[N - j]);
    lwe[N] = acc[N]; /* b = b-polynomial coefficient 0 */
}
void orc_sample_extract64(int64_t *lwe, const int64_t *acc, int N) {
    lwe[0] = acc[0];
    for (int j = 1; j < N; j++) lwe[j] = (int64_t)(0ull - (uint64_t)acc[N - j]);
    lwe[N] = acc[N];
}

void orc_blind_rotate_extract32(const orc_tables *t, int32_t *lwe, const int32_t *v,
                                const double *bkfft, int barb, const int32_t *bara, int n,
                                int l, int Bgbit) {
    const int N = t->N;
    int32_t *acc = (int32_t *)calloc((size_t)2 * N, sizeof(int32_t)); /* a = 0 (noiseless trivial) */
    if (barb != 0)
        orc_mul_xai32(acc + N, 2 * N - barb, v, N);
    else
        memcpy(acc + N, v, sizeof(int32_t) * (size_t)N);
    orc_blind_rotate32(t, acc, bkfft, bara, n, l, Bgbit);
    orc_sample_extract32(lwe, acc, N);
    free(acc);
}

/* exact external product: the reference's FFT-free backend (`#ifndef USE_FFT`, poc:285-316): the
 * accumulation loops of tgsw_functions.cpp:441-443 with torus{32,64}PolynomialMultAddKaratsuba
 * (CB/poc_karatsuba.cpp:80-95,188-203) as the product -- restated with the plain negacyclic
 * convolution, which is the same ring element (pinned against the reference's Karatsuba objects
 * by tests/test_oracle_golden.py).  gsw: coefficient form [2l][2][N]. */
void orc_extprod_exact32(int32_t *acc, const int32_t *gsw, int N, int l, int Bgbit) {
    const int kpl = 2 * l;
    int32_t *deca = (int32_t *)malloc(sizeof(int32_t) * (size_t)kpl * N);
    int32_t *prod = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (int i = 0; i <= 1; i++) orc_decomp32(deca + i * l * N, acc + i * N, N, l, Bgbit);
    memset(acc, 0, sizeof(int32_t) * (size_t)2 * N);
    for (int p = 0; p < kpl; p++)
        for (int q = 0; q <= 1; q++) {
            orc_negacyclic_mul32(prod, deca + p * N, gsw + ((size_t)p * 2 + q) * N, N);
            for (int j = 0; j < N; j++) acc[q * N + j] = (int32_t)((uint32_t)acc[q * N + j] + (uint32_t)prod[j]);
        }
    free(deca);
    free(prod);
}
void orc_extprod_exact64(int64_t *acc, const int64_t *gsw, int N, int l, int Bgbit) {
    const int kpl = 2 * l;
    int32_t *deca = (int32_t *)malloc(sizeof(int32_t) * (size_t)kpl * N);
    int64_t *prod = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    for (int i = 0; i <= 1; i++) orc_decomp64(deca + i * l * N, acc + i * N, N, l, Bgbit);
    memset(acc, 0, sizeof(int64_t) * (size_t)2 * N);
    for (int p = 0; p < kpl; p++)
        for (int q = 0; q <= 1; q++) {
            orc_negacyclic_mul64(prod, deca + p * N, gsw + ((size_t)p * 2 + q) * N, N);
            for (int j = 0; j < N; j++) acc[q * N + j] = (int64_t)((uint64_t)acc[q * N + j] + (uint64_t)prod[j]);
        }
    free(deca);
    free(prod);
}

/* ---- CMux on data and LUT evaluation by vertical packing --------------------------------------
 * The reference ends at the stub `CMux(out, c, in0, in1, env)` (poc:877-879).  Restated from the two
 * reference operations it would be built from: tGswFFTExternMulToTLwe (tgsw_functions.cpp:424-449)
 * and tfhe_MuxRotate_FFT (lwe_functions.cpp:328-333); the tree/rotate/extract composition is the
 * published vertical-packing algorithm (no reference code: "parity unpinned" beyond the primitives). */
void orc_cmux32(const orc_tables *t, int32_t *out, const double *gsw, const int32_t *d0, const int32_t *d1,
                int l, int Bgbit) {
    const int N = t->N;
    for (int j = 0; j < 2 * N; j++) out[j] = (int32_t)((uint32_t)d1[j] - (uint32_t)d0[j]);
    orc_extprod32(t, out, gsw, l, Bgbit);
    for (int j = 0; j < 2 * N; j++) out[j] = (int32_t)((uint32_t)out[j] + (uint32_t)d0[j]);
}

void orc_lut_eval32(const orc_tables *t, int32_t *lwe, const double *bits /* [d][2l][2][N] */, int d,
                    const int32_t *lut /* [max(1, 2^(d-logN))][N] */, int l, int Bgbit) {
    const int N = t->N;
    int logn = 0;
    while ((1 << logn) < N) logn++;
    const int low = d < logn ? d : logn, levels = d - low;
    const size_t row = (size_t)2 * l * 2 * N, S = (size_t)2 * N;
    size_t cnt = (size_t)1 << levels;
    int32_t *cur = (int32_t *)calloc(cnt * S, sizeof(int32_t));
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * S);
    for (size_t p = 0; p < cnt; p++) /* tLweNoiselessTrivial (tlwe_functions.cpp:146-152): a = 0, b = table */
        memcpy(cur + p * S + N, lut + p * N, sizeof(int32_t) * (size_t)N);
    for (int j = 0; j < levels; j++) { /* bit logN + j halves the table */
        cnt >>= 1;
        for (size_t p = 0; p < cnt; p++) {
            orc_cmux32(t, tmp, bits + row * (size_t)(low + j), cur + (2 * p) * S, cur + (2 * p + 1) * S, l, Bgbit);
            memcpy(cur + p * S, tmp, sizeof(int32_t) * S);
        }
    }
    for (int i = 0; i < low; i++) { /* X^{-2^i} when bit i is set */
        orc_mux_rotate32(t, tmp, cur, bits + row * (size_t)i, 2 * N - (1 << i), l, Bgbit);
        memcpy(cur, tmp, sizeof(int32_t) * S);
    }
    orc_sample_extract32(lwe, cur, N);
    free(cur);
    free(tmp);
}

int32_t orc_modswitch32(int32_t phase, int Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    const uint64_t half = interv / 2;
    const uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + half;
    return (int32_t)(phase64 / interv);
}

void orc_bootstrap_woks32(const orc_tables *t, int32_t *lwe_out, const double *bkfft, int32_t mu,
                          const int32_t *x, int n, int l, int Bgbit) {
    const int N = t->N;
    int32_t *bara = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *v = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    const int barb = orc_modswitch32(x[n], 2 * N);
    for (int i = 0; i < n; i++) bara[i] = orc_modswitch32(x[i], 2 * N);
    for (int i = 0; i < N; i++) v[i] = mu;
    orc_blind_rotate_extract32(t, lwe_out, v, bkfft, barb, bara, n, l, Bgbit);
    free(bara);
    free(v);
}

void orc_keyswitch32(int32_t *out, const int32_t *ks, const int32_t *in, int n_in, int n_out, int t,
                     int basebit) {
    const int base = 1 << basebit;
    const uint32_t mask = (uint32_t)base - 1;
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    const size_t row = (size_t)n_out + 1;
    for (int h = 0; h < n_out; h++) out[h] = 0; /* lweNoiselessTrivial(result, sample->b) */
    out[n_out] = in[n_in];
    for (int i = 0; i < n_in; i++) {
        const uint32_t aibar = (uint32_t)in[i] + prec_offset;
        for (int j = 0; j < t; j++) {
            const uint32_t aij = (aibar >> (32 - (j + 1) * basebit)) & mask;
            if (aij != 0) {
                const int32_t *r = ks + (((size_t)i * t + j) * base + aij) * row;
                for (int h = 0; h <= n_out; h++) out[h] = (int32_t)((uint32_t)out[h] - (uint32_t)r[h]);
            }
        }
    }
}

void orc_bootstrap32(const orc_tables *t, int32_t *out, const double *bkfft, const int32_t *ks,
                     int32_t mu, const int32_t *x, int n, int l, int Bgbit, int ks_t, int ks_basebit) {
    const int N = t->N;
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
    orc_bootstrap_woks32(t, u, bkfft, mu, x, n, l, Bgbit);
    orc_keyswitch32(out, ks, u, N, n, ks_t, ks_basebit);
    free(u);
}

/* ---------------------------------------------------------- circuit bootstrap */

void orc_pre_modswitch(int32_t *out, const int32_t *x, int n0, int N2) {
    for (int i = 0; i <= n0; i++) out[i] = orc_modswitch32(x[i], 2 * N2);
}

static void cb_testvector(int64_t *tv, int64_t mu2, int N) {
    /* poc:551-553: (1+X+...+X^{N-1}) * X^{N/2} * mu2 */
    for (int j = 0; j < N / 2; j++) tv[j] = -mu2;
    for (int j = N / 2; j < N; j++) tv[j] = mu2;
}

void orc_cb_bootstrap_woks64(const orc_tables *t, int64_t *lwe, int64_t mu, const int32_t *abar,
                             const double *bkfft, int n0, int l, int Bgbit) {
    const int N = t->N;
    const int64_t mu2 = mu / 2;
    int64_t *tv = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *acc = (int64_t *)calloc((size_t)2 * N, sizeof(int64_t));
    cb_testvector(tv, mu2, N);
    const int bbar = abar[n0];
    if (bbar != 0) /* library form: X^{2N-bbar} * v (lwe_functions.cpp:385-386) */
        orc_mul_xai64(acc + N, 2 * N - bbar, tv, N);
    else
        memcpy(acc + N, tv, sizeof(int64_t) * (size_t)N);
    orc_blind_rotate64(t, acc, bkfft, abar, n0, l, Bgbit);
    orc_sample_extract64(lwe, acc, N);
    lwe[N] = (int64_t)((uint64_t)lwe[N] + (uint64_t)mu2); /* poc:648 */
    free(tv);
    free(acc);
}

int orc_cb_bootstrap_woks64_poc_quirks(const orc_tables *t, int64_t *lwe, int64_t mu,
                                       const int32_t *abar, const double *bkfft0, int n0, int l,
                                       int Bgbit) {
    const int N = t->N;
    const int64_t mu2 = mu / 2;
    for (int i = 0; i < n0; i++)
        if (abar[i] >= N) return -1; /* PoC reads out of bounds there (poc:596-597) */
    int64_t *tv = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *acc = (int64_t *)calloc((size_t)2 * N, sizeof(int64_t));
    int64_t *acc2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)2 * N);
    cb_testvector(tv, mu2, N);
    const int bbar = abar[n0];
    /* quirk (c): test vector times X^{+bbar} (poc:554-562) */
    if (bbar == 0)
        memcpy(acc + N, tv, sizeof(int64_t) * (size_t)N);
    else
        orc_mul_xai64(acc + N, bbar, tv, N);
    for (int i = 0; i < n0; i++) {
        const int a = abar[i];
        if (a == 0) continue;
        /* quirk (b), a < N branch only (poc:593-594): missing negation for j < a */
        for (int q = 0; q <= 1; q++) {
            const int64_t *s = acc + q * N;
            int64_t *d = acc2 + q * N;
            for (int j = 0; j < a; j++) d[j] = (int64_t)((uint64_t)s[j - a + N] - (uint64_t)s[j]);
            for (int j = a; j < N; j++) d[j] = (int64_t)((uint64_t)s[j - a] - (uint64_t)s[j]);
        }
        orc_extprod64(t, acc2, bkfft0, l, Bgbit); /* quirk (a): always bkFFT[0] (poc:548,618) */
        for (int j = 0; j < 2 * N; j++) acc[j] = (int64_t)((uint64_t)acc[j] + (uint64_t)acc2[j]);
    }
    orc_sample_extract64(lwe, acc, N);
    lwe[N] = (int64_t)((uint64_t)lwe[N] + (uint64_t)mu2);
    free(tv);
    free(acc);
    free(acc2);
    return 0;
}

void orc_privks(int32_t *out, const int32_t *privks_u, const int64_t *x, int n2, int N1, int t,
                int basebit) {
    const int base = 1 << basebit;
    const uint64_t mask = (uint64_t)base - 1;
    const uint64_t prec_offset = UINT64_C(1) << (64 - (1 + basebit * t));
    const size_t row = (size_t)2 * N1;
    for (size_t p = 0; p < row; p++) out[p] = 0;
    for (int i = 0; i <= n2; i++) {
        const uint64_t aibar = (uint64_t)x[i] + prec_offset;
        for (int j = 0; j < t; j++) {
            const uint64_t aij = (aibar >> (64 - (j + 1) * basebit)) & mask;
            if (aij != 0) {
                const int32_t *r = privks_u + (((size_t)i * t + j) * base + aij) * row;
                for (size_t p = 0; p < row; p++) out[p] = (int32_t)((uint32_t)out[p] - (uint32_t)r[p]);
            }
        }
    }
}

void orc_circuit_bootstrap(const orc_tables *t2, int32_t *out, const int32_t *x, const int32_t *preks,
                           const double *bkfft, const int32_t *privks, int n0, int N1, int N2, int l1,
                           int Bgbit1, int l2, int Bgbit2, int t10, int bb10, int t21, int bb21) {
    int32_t *pre = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n0 + 1));
    int32_t *abar = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n0 + 1));
    int64_t *boot = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N2 + 1));
    const size_t tlwe = (size_t)2 * N1;
    const size_t privks_u = (size_t)(N2 + 1) * t21 * (1u << bb21) * tlwe;
    orc_keyswitch32(pre, preks, x, N1, n0, t10, bb10); /* preKeySwitch, poc:437-465 */
    orc_pre_modswitch(abar, pre, n0, N2);
    for (int w = 0; w < l1; w++) {
        const int64_t mu1 = (int64_t)(UINT64_C(1) << (64 - (w + 1) * Bgbit1));
        orc_cb_bootstrap_woks64(t2, boot, mu1, abar, bkfft, n0, l2, Bgbit2);
        for (int u = 0; u <= 1; u++) /* result->samples[u][w] */
            orc_privks(out + ((size_t)u * l1 + w) * tlwe, privks + privks_u * u, boot, N2, N1, t21, bb21);
    }
    free(pre);
    free(abar);
    free(boot);
}

/* ------------------------------------------------------------ PRNG + keygen */
/* SPEC (shared with the shipped key generator, experimental-tfhe_amd/csrc/keygen.cpp):
 *   next():  s += 0x9E3779B97F4A7C15; z = s; z = (z^(z>>30))*0xBF58476D1CE4E5B9;
 *            z = (z^(z>>27))*0x94D049BB133111EB; return z^(z>>31)          (splitmix64)
 *   init(seed, stream): s = seed; a = next(); s = a ^ (stream*0xD1342543DE82EF95 + 0x632BE59BD9B4E019)
 *   torus32 = (int32)(next()>>32); torus64 = (int64)next(); bit = next()>>63
 *   gauss   = sqrt(-2 ln u1) * cos(2 pi u2),  u1 = ((next()>>11)+1)*2^-53, u2 = (next()>>11)*2^-53
 *   noise32 = (int32)(int64)(gauss*stdev*2^32)  (truncation, as generic_utils.h:172-177)
 *   noise64 = (int64)(gauss*stdev*2^64)         (generic_utils.h:179-185)
 */
static inline uint64_t sm64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void orc_rng_init(orc_rng *r, uint64_t seed, uint64_t stream) {
    r->s = seed;
    const uint64_t a = sm64(&r->s);
    r->s = a ^ (stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull);
}
uint64_t orc_rng_next(orc_rng *r) { return sm64(&r->s); }
int32_t orc_rng_torus32(orc_rng *r) { return (int32_t)(uint32_t)(sm64(&r->s) >> 32); }
int64_t orc_rng_torus64(orc_rng *r) { return (int64_t)sm64(&r->s); }
double orc_rng_gauss(orc_rng *r) {
    const double u1 = (double)((sm64(&r->s) >> 11) + 1) * 0x1p-53;
    const double u2 = (double)(sm64(&r->s) >> 11) * 0x1p-53;
    return sqrt(-2. * log(u1)) * cos(2. * M_PI * u2);
}
static int32_t noise32(orc_rng *r, double stdev) {
    return (int32_t)(int64_t)(orc_rng_gauss(r) * stdev * 0x1p32);
}
static int64_t noise64(orc_rng *r, double stdev) { return (int64_t)(orc_rng_gauss(r) * stdev * 0x1p64); }

/* synthetic tables: consecutive splitmix64 outputs from state `seed`, high halves */
void orc_fill32(int32_t *out, uint64_t seed, size_t count) {
    uint64_t s = seed;
    for (size_t i = 0; i < count; i++) out[i] = (int32_t)(uint32_t)(sm64(&s) >> 32);
}

void orc_keygen_binary(int32_t *key, int n, uint64_t seed, uint64_t stream) {
    orc_rng r;
    orc_rng_init(&r, seed, stream);
    for (int i = 0; i < n; i++) key[i] = (int32_t)(sm64(&r.s) >> 63);
}

void orc_lwe_encrypt32(int32_t *ct, int32_t mess, double stdev, const int32_t *key, int n, orc_rng *r) {
    uint32_t b = (uint32_t)mess + (uint32_t)noise32(r, stdev);
    for (int i = 0; i < n; i++) {
        ct[i] = orc_rng_torus32(r);
        b += (uint32_t)ct[i] * (uint32_t)key[i];
    }
    ct[n] = (int32_t)b;
}
int32_t orc_lwe_phase32(const int32_t *ct, const int32_t *key, int n) {
    uint32_t res = (uint32_t)ct[n];
    for (int i = 0; i < n; i++) res -= (uint32_t)ct[i] * (uint32_t)key[i];
    return (int32_t)res;
}
int64_t orc_lwe_phase64(const int64_t *ct, const int32_t *key, int n) {
    uint64_t res = (uint64_t)ct[n];
    for (int i = 0; i < n; i++) res -= (uint64_t)ct[i] * (uint64_t)(int64_t)key[i];
    return (int64_t)res;
}

/* b += a * key for a binary (0/1) key: exact negacyclic accumulate */
static void addmul_binkey32(uint32_t *b, const int32_t *a, const int32_t *key, int N) {
    for (int j = 0; j < N; j++) {
        if (!key[j]) continue;
        for (int i = 0; i < N - j; i++) b[i + j] += (uint32_t)a[i];
        for (int i = N - j; i < N; i++) b[i + j - N] -= (uint32_t)a[i];
    }
}
static void addmul_binkey64(uint64_t *b, const int64_t *a, const int32_t *key, int N) {
    for (int j = 0; j < N; j++) {
        if (!key[j]) continue;
        for (int i = 0; i < N - j; i++) b[i + j] += (uint64_t)a[i];
        for (int i = N - j; i < N; i++) b[i + j - N] -= (uint64_t)a[i];
    }
}
/* TLWE zero encryption (poc:143-152): b = gaussian, a uniform, b += a*s.  Draw order: all N
 * noise values of b, then the N coefficients of a. */
static void tlwe_encrypt_zero32(int32_t *ct, double stdev, const int32_t *tkey, int N, orc_rng *r) {
    for (int j = 0; j < N; j++) ct[N + j] = noise32(r, stdev);
    for (int j = 0; j < N; j++) ct[j] = orc_rng_torus32(r);
    addmul_binkey32((uint32_t *)(ct + N), ct, tkey, N);
}
static void tlwe_encrypt_zero64(int64_t *ct, double stdev, const int32_t *tkey, int N, orc_rng *r) {
    for (int j = 0; j < N; j++) ct[N + j] = noise64(r, stdev);
    for (int j = 0; j < N; j++) ct[j] = orc_rng_torus64(r);
    addmul_binkey64((uint64_t *)(ct + N), ct, tkey, N);
}
void orc_tgsw_encrypt32(int32_t *gsw, int32_t mess, double stdev, const int32_t *tkey, int N, int l,
                        int Bgbit, orc_rng *r) {
    for (int bloc = 0; bloc <= 1; bloc++)
        for (int i = 0; i < l; i++) {
            int32_t *row = gsw + ((size_t)(bloc * l + i)) * 2 * N;
            tlwe_encrypt_zero32(row, stdev, tkey, N, r);
            row[bloc * N] = (int32_t)((uint32_t)row[bloc * N] + (uint32_t)mess * (1u << (32 - (i + 1) * Bgbit)));
        }
}
void orc_tgsw_encrypt64(int64_t *gsw, int32_t mess, double stdev, const int32_t *tkey, int N, int l,
                        int Bgbit, orc_rng *r) {
    for (int bloc = 0; bloc <= 1; bloc++)
        for (int i = 0; i < l; i++) {
            int64_t *row = gsw + ((size_t)(bloc * l + i)) * 2 * N;
            tlwe_encrypt_zero64(row, stdev, tkey, N, r);
            row[bloc * N] = (int64_t)((uint64_t)row[bloc * N] +
                                      (uint64_t)(int64_t)mess * (UINT64_C(1) << (64 - (i + 1) * Bgbit)));
        }
}
void orc_bk_create32(const orc_tables *t, double *bkfft, const int32_t *lwe_key, int n,
                     const int32_t *tkey, int l, int Bgbit, double stdev, uint64_t seed, uint64_t stream) {
    const int N = t->N;
    const size_t rowlen = (size_t)2 * l * 2 * N;
    int32_t *gsw = (int32_t *)malloc(sizeof(int32_t) * rowlen);
    for (int i = 0; i < n; i++) {
        orc_rng r;
        orc_rng_init(&r, seed, stream + (uint64_t)i); /* one stream per key element */
        orc_tgsw_encrypt32(gsw, lwe_key[i], stdev, tkey, N, l, Bgbit, &r);
        for (int pq = 0; pq < 2 * l * 2; pq++)
            orc_execute_reverse_torus32(t, bkfft + rowlen * i + (size_t)pq * N, gsw + (size_t)pq * N);
    }
    free(gsw);
}
void orc_bk_create64(const orc_tables *t, double *bkfft, const int32_t *lwe_key, int n,
                     const int32_t *tkey, int l, int Bgbit, double stdev, uint64_t seed, uint64_t stream) {
    const int N = t->N;
    const size_t rowlen = (size_t)2 * l * 2 * N;
    int64_t *gsw = (int64_t *)malloc(sizeof(int64_t) * rowlen);
    for (int i = 0; i < n; i++) {
        orc_rng r;
        orc_rng_init(&r, seed, stream + (uint64_t)i);
        orc_tgsw_encrypt64(gsw, lwe_key[i], stdev, tkey, N, l, Bgbit, &r);
        for (int pq = 0; pq < 2 * l * 2; pq++)
            orc_execute_reverse_torus64(t, bkfft + rowlen * i + (size_t)pq * N, gsw + (size_t)pq * N);
    }
    free(gsw);
}
void orc_ks_create32(int32_t *ks, const int32_t *in_key, int n_in, const int32_t *out_key, int n_out,
                     int t, int basebit, double stdev, uint64_t seed, uint64_t stream) {
    const int base = 1 << basebit;
    const size_t row = (size_t)n_out + 1;
    for (int i = 0; i < n_in; i++) {
        orc_rng r;
        orc_rng_init(&r, seed, stream + (uint64_t)i); /* one stream per input-key element */
        for (int j = 0; j < t; j++)
            for (int u = 0; u < base; u++) {
                /* poc:379 / lwe_functions.cpp:128 */
                const int32_t mess = (int32_t)(((uint32_t)in_key[i] << (32 - (j + 1) * basebit)) * (uint32_t)u);
                orc_lwe_encrypt32(ks + (((size_t)i * t + j) * base + u) * row, mess, stdev, out_key, n_out, &r);
            }
    }
}
void orc_tlwe_phase32(int32_t *phase, const int32_t *ct, const int32_t *tkey, int N) {
    uint32_t *as = (uint32_t *)calloc((size_t)N, sizeof(uint32_t));
    addmul_binkey32(as, ct, tkey, N);
    for (int j = 0; j < N; j++) phase[j] = (int32_t)((uint32_t)ct[N + j] - as[j]);
    free(as);
}
void orc_tlwe_phase64(int64_t *phase, const int64_t *ct, const int32_t *tkey, int N) {
    uint64_t *as = (uint64_t *)calloc((size_t)N, sizeof(uint64_t));
    addmul_binkey64(as, ct, tkey, N);
    for (int j = 0; j < N; j++) phase[j] = (int64_t)((uint64_t)ct[N + j] - as[j]);
    free(as);
}
void orc_privks_create(int32_t *privks, const int32_t *key2, int n2, const int32_t *tkey1, int N1, int t,
                       int basebit, double stdev, uint64_t seed, uint64_t stream) {
    const int base = 1 << basebit;
    const size_t row = (size_t)2 * N1;
    for (int z = 0; z <= 1; z++)
        for (int i = 0; i <= n2; i++) {
            const int32_t ki = (i == n2) ? -1 : key2[i]; /* poc:367 */
            orc_rng r;
            orc_rng_init(&r, seed, stream + (uint64_t)z * (uint64_t)(n2 + 1) + (uint64_t)i);
            for (int j = 0; j < t; j++)
                for (int u = 0; u < base; u++) {
                    int32_t *ct = privks + ((((size_t)z * (n2 + 1) + i) * t + j) * base + u) * row;
                    const int32_t mess = (int32_t)(((uint32_t)ki << (32 - (j + 1) * basebit)) * (uint32_t)u);
                    tlwe_encrypt_zero32(ct, stdev, tkey1, N1, &r);
                    ct[z * N1] = (int32_t)((uint32_t)ct[z * N1] + (uint32_t)mess);
                }
        }
}

/* ---- Real96 high-precision anticyclic FFT (high-precision-anticyclic-fft/src/code.cpp = "HP") ----
 * 128-bit fixed point: a Real96 is a two's-complement integer v standing for v / 2^64 (HP:17-40).
 * PARITY UNPINNED by reference object code: HP/code.cpp needs NTL (absent here), so this restatement
 * is checked by properties only (round trip, unit circle, agreement of two independent twiddle
 * builders); see tests/test_hp_fft.py. */
typedef unsigned __int128 u128;

/* intmul_best == intmul_ref, HP:79-95,148-169: (a * b) >> 64 for a twiddle b in [-1, 1) whose high
 * word is its sign extension */
static u128 hp_intmul(u128 a, u128 b) {
    /* A = a as a signed 128-bit integer = ahi_s * 2^64 + alo (ahi_s signed, alo unsigned);
     * B = the twiddle, |B| < 2^64, sign-extended: B = blo - 2^64 * bneg with blo its low word.
     *   floor(A * blo / 2^64) = ahi_s * blo + floor(alo * blo / 2^64)
     *                         = ahi_u * blo - 2^64 * [ahi_s < 0] * blo + hi64(alo * blo)      (ahi_s = ahi_u - 2^64 [ahi_s < 0])
     * and modulo 2^128 the middle term is ((0 - blo) mod 2^64) << 64.  The bneg part is exact:
     * (2^64 * bneg * A) / 2^64 = bneg * A.  Hence, modulo 2^128:                                          */
    const uint64_t alo = (uint64_t)a, ahi_u = (uint64_t)(a >> 64), blo = (uint64_t)b;
    const int a_negative = (int)(ahi_u >> 63), b_negative = (int)((uint64_t)(b >> 64) >> 63);
    u128 r = (u128)blo * ahi_u;                  /* ahi_u * blo                  */
    r += ((u128)blo * alo) >> 64;                /* + hi64(alo * blo)            */
    if (a_negative) r += (u128)(0 - blo) << 64;  /* - 2^64 * blo  (mod 2^128)    */
    if (b_negative) r -= a;                      /* - bneg * A                   */
    return r;
}
/* std::complex<Real96> product as libstdc++ instantiates it for a class type:
 * re = a.re*b.re - a.im*b.im; im = a.re*b.im + a.im*b.re (data on the left, twiddle on the right) */
static void hp_cmul(u128 *re, u128 *im, u128 are, u128 aim, u128 bre, u128 bim) {
    const u128 r = hp_intmul(are, bre) - hp_intmul(aim, bim);
    const u128 i = hp_intmul(are, bim) + hp_intmul(aim, bre);
    *re = r;
    *im = i;
}

/* accurate_cos / accurate_sin, HP:246-278 (NTL RR there; libquadmath here), tables HP:378-389.
 * out: [n][2] (re, im) */
#include <quadmath.h>
static u128 hp_round64(__float128 x) { /* RoundToZZ(x * 2^64) in Real96 encoding */
    const __float128 r = rintq(ldexpq(x, 64));
    return (u128)(__int128)r;
}
void orc_hp_twiddles(int n, unsigned __int128 *powomega, unsigned __int128 *powombar) {
    for (int i = 0; i < n; i++) {
        const __float128 ang = 2 * M_PIq * (__float128)i / (__float128)n;
        const int ib = (n - i) % n;
        const __float128 angb = 2 * M_PIq * (__float128)ib / (__float128)n;
        const u128 c = (i == 0) ? (u128)UINT64_MAX : hp_round64(cosq(ang));
        const u128 s = (i == n / 4) ? (u128)UINT64_MAX : hp_round64(sinq(ang));
        const u128 sb = (ib == n / 4) ? (u128)UINT64_MAX : hp_round64(sinq(angb));
        if (powomega) {
            powomega[2 * i] = c;
            powomega[2 * i + 1] = s;
        }
        if (powombar) {
            powombar[2 * i] = c;
            powombar[2 * i + 1] = sb;
        }
    }
}

/* iFFT, HP:391-444: P -> P(omega).  in: N Torus64, out: [N/2][2], n = 2N */
void orc_hp_ifft(unsigned __int128 *out, const int64_t *in, int N, const unsigned __int128 *powomega) {
    const int n = 2 * N, ns4 = n / 4;
    for (int j = 0; j < ns4; j++)
        hp_cmul(&out[2 * j], &out[2 * j + 1], (u128)(__int128)in[j], (u128)(__int128)in[j + ns4], powomega[2 * j],
                powomega[2 * j + 1]);
    for (int nn = ns4; nn >= 2; nn /= 2) {
        const int halfnn = nn / 2;
        for (int block = 0; block < ns4; block += nn)
            for (int off = 0; off < halfnn; off++) {
                u128 *p1 = out + 2 * (block + off), *p2 = out + 2 * (block + off + halfnn);
                const u128 t1r = p1[0], t1i = p1[1], t2r = p2[0], t2i = p2[1];
                const int w = (2 * (ns4 / halfnn) * off) % n;
                p1[0] = t1r + t2r;
                p1[1] = t1i + t2i;
                hp_cmul(&p2[0], &p2[1], t1r - t2r, t1i - t2i, powomega[2 * w], powomega[2 * w + 1]);
            }
    }
}
/* FFT, HP:446-512: P(omega) -> P, destroys `in`; the final ">> 10" is the reference's hard-coded
 * division by N/2 for N = 2048 (HP:499-500), kept as log2(N/2) */
void orc_hp_fft(int64_t *out, unsigned __int128 *in, int N, const unsigned __int128 *powombar) {
    const int n = 2 * N, ns4 = n / 4;
    int shift = 0;
    while ((1 << shift) < ns4) shift++;
    for (int nn = 2; nn <= ns4; nn *= 2) {
        const int halfnn = nn / 2;
        for (int block = 0; block < ns4; block += nn)
            for (int off = 0; off < halfnn; off++) {
                u128 *p1 = in + 2 * (block + off), *p2 = in + 2 * (block + off + halfnn);
                const int w = (2 * (ns4 / halfnn) * off) % n;
                u128 t2r, t2i;
                hp_cmul(&t2r, &t2i, p2[0], p2[1], powombar[2 * w], powombar[2 * w + 1]);
                const u128 t1r = p1[0], t1i = p1[1];
                p1[0] = t1r + t2r;
                p1[1] = t1i + t2i;
                p2[0] = t1r - t2r;
                p2[1] = t1i - t2i;
            }
    }
    for (int j = 0; j < ns4; j++) {
        hp_cmul(&in[2 * j], &in[2 * j + 1], in[2 * j], in[2 * j + 1], powombar[2 * j], powombar[2 * j + 1]);
        out[j] = (int64_t)(uint64_t)(in[2 * j] >> shift);
        out[j + ns4] = (int64_t)(uint64_t)(in[2 * j + 1] >> shift);
    }
}
