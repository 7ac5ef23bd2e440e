// dropin_driver_poc.cpp -- a PoC-style driver for the LITERAL drop-in test: it includes "poc_types.h"
// (the reference's own header in the build container, tests/compat/poc_stub/ on the GPU box), declares
// the five entry points exactly as CB/poc_CircuitBootstrapping.cpp:437,472,530,667,823 defines them --
// global scope, `const Globals* env` last, no namespace -- and links against dropin_poc.cpp compiled next
// to the same header.  Same input / output files as `compat_driver poc` (tests/test_compat.py checks them
// against the oracle).  Parameters are compile-time (-DP_N0=... as the PoC's are constants, poc:70-85).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "poc_types.h"

void preKeySwitch(LweSample32* result, const LweSample32* x, const Globals* env);
void preModSwitch(int* result, const LweSample32* x, const Globals* env);
void circuitBootstrapWoKS(LweSample64* result, const Torus64 mu, const int* abar, const Globals* env);
void circuitPrivKS(TLweSample32* result, const int u, const LweSample64* x, const Globals* env);
void tfhe_CircuitBootstrapFFT(TGswSample32* result, const LweSample32* sample, const Globals* env);
void tfhe_CircuitBootstrapFFT_array(TGswSample32* const* results, const LweSample32* const* samples, const Globals* env, int count);
void CMux(TLweSample32* out, const TGswSample32* c, const TLweSample32* in0, const TLweSample32* in1, const Globals* env);

const int Globals::n_lvl0 = P_N0;
const int Globals::n_lvl1 = P_N1;
const int Globals::n_lvl2 = P_N2;
const int Globals::bgbit_lvl1 = P_BG1;
const int Globals::ell_lvl1 = P_L1;
const int Globals::bgbit_lvl2 = P_BG2;
const int Globals::ell_lvl2 = P_L2;
const int Globals::kslength_lvl10 = P_T10;
const int Globals::ksbasebit_lvl10 = P_BB10;
const int Globals::kslength_lvl21 = P_T21;
const int Globals::ksbasebit_lvl21 = P_BB21;
#ifdef DROPIN_REAL_HEADER  // members of the real Globals this test does not use
const double Globals::bkstdev_lvl2 = 0, Globals::ksstdev_lvl10 = 0, Globals::ksstdev_lvl21 = 0;
#endif
Globals::Globals() {}  // keys are attached by main() below (the PoC's constructor generates them, poc:342-423)

template <class T, class... A>
static T *make_array(size_t count, A... args) {  // objects without default constructors, built in place
    T *p = static_cast<T *>(operator new[](sizeof(T) * count));
    for (size_t i = 0; i < count; i++) new (p + i) T(args...);
    return p;
}
static std::vector<uint8_t> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)sz);
    if (sz && fread(b.data(), 1, (size_t)sz, f) != (size_t)sz) { perror("read"); exit(2); }
    fclose(f);
    return b;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    auto blob = slurp(argv[1]);
    const uint8_t *p = blob.data();
    const int32_t *h = (const int32_t *)p;  // n0 N1 N2 l1 bg1 l2 bg2 t10 bb10 t21 bb21 count
    p += 12 * 4;
    const int n0 = P_N0, N1 = P_N1, N2 = P_N2, l1 = P_L1, l2 = P_L2, t10 = P_T10, b10 = 1 << P_BB10, t21 = P_T21, b21 = 1 << P_BB21;
    const int want[11] = {P_N0, P_N1, P_N2, P_L1, P_BG1, P_L2, P_BG2, P_T10, P_BB10, P_T21, P_BB21};
    for (int i = 0; i < 11; i++)
        if (h[i] != want[i]) { fprintf(stderr, "input file was made for other parameters\n"); return 2; }
    const int count = h[11];
    const int32_t *preks = (const int32_t *)p; p += (size_t)N1 * t10 * b10 * (n0 + 1) * 4;
    const double *bk = (const double *)p; p += (size_t)n0 * 2 * l2 * 2 * N2 * 8;
    const int32_t *priv = (const int32_t *)p; p += (size_t)2 * (N2 + 1) * t21 * b21 * 2 * N1 * 4;
    const int32_t *xs = (const int32_t *)p; p += (size_t)count * (N1 + 1) * 4;
    const int64_t *x64 = (const int64_t *)p;

    Globals *env = new Globals();
    // preKS[N1][t10][base] of LweSample32(n0)
    env->preKS = new LweSample32 **[N1];
    for (int i = 0; i < N1; i++) {
        env->preKS[i] = new LweSample32 *[t10];
        for (int j = 0; j < t10; j++) {
            env->preKS[i][j] = make_array<LweSample32>(b10, n0);
            for (int u = 0; u < b10; u++)
                memcpy(env->preKS[i][j][u].a, preks + (((size_t)i * t10 + j) * b10 + u) * (n0 + 1), 4 * (size_t)(n0 + 1));
        }
    }
    // bkFFT[n0] of TGswSampleFFT(l2, N2)
    env->bkFFT = make_array<TGswSampleFFT>(n0, l2, N2);
    for (int i = 0; i < n0; i++)
        for (int r = 0; r < 2 * l2; r++)
            for (int q = 0; q < 2; q++)
                memcpy(env->bkFFT[i].allsamples[r].a[q].values, bk + (((size_t)i * 2 * l2 + r) * 2 + q) * N2, 8 * (size_t)N2);
    // privKS[2][N2+1][t21][base] of TLweSample32(N1)
    env->privKS = new TLweSample32 ***[2];
    for (int u = 0; u < 2; u++) {
        env->privKS[u] = new TLweSample32 **[N2 + 1];
        for (int i = 0; i <= N2; i++) {
            env->privKS[u][i] = new TLweSample32 *[t21];
            for (int j = 0; j < t21; j++) {
                env->privKS[u][i][j] = make_array<TLweSample32>(b21, N1);
                for (int d = 0; d < b21; d++)
                    for (int q = 0; q < 2; q++)
                        memcpy(env->privKS[u][i][j][d].a[q].coefs,
                               priv + ((((((size_t)u * (N2 + 1) + i) * t21 + j) * b21 + d) * 2) + q) * N1, 4 * (size_t)N1);
            }
        }
    }
    std::vector<uint8_t> out;
    auto put = [&](const void *q, size_t n) { out.insert(out.end(), (const uint8_t *)q, (const uint8_t *)q + n); };
    std::vector<int32_t> loop_cb;  // the one-by-one circuit bootstraps, for the array form below
    for (int c = 0; c < count; c++) {
        LweSample32 x(N1), pre(n0);
        memcpy(x.a, xs + (size_t)c * (N1 + 1), 4 * (size_t)(N1 + 1));
        std::vector<int> abar((size_t)n0 + 1);
        preKeySwitch(&pre, &x, env);
        preModSwitch(abar.data(), &pre, env);
        put(pre.a, 4 * (size_t)(n0 + 1));
        put(abar.data(), 4 * abar.size());
        LweSample64 boot(N2), xin(N2);
        circuitBootstrapWoKS(&boot, (Torus64)1 << 56, abar.data(), env);
        put(boot.a, 8 * (size_t)(N2 + 1));
        memcpy(xin.a, x64 + (size_t)c * (N2 + 1), 8 * (size_t)(N2 + 1));
        TLweSample32 tl(N1);
        circuitPrivKS(&tl, 1, &xin, env);
        for (int q = 0; q < 2; q++) put(tl.a[q].coefs, 4 * (size_t)N1);
        TGswSample32 tg(l1, N1);
        tfhe_CircuitBootstrapFFT(&tg, &x, env);
        for (int u = 0; u < 2; u++)
            for (int w = 0; w < l1; w++)
                for (int q = 0; q < 2; q++) {
                    put(tg.samples[u][w].a[q].coefs, 4 * (size_t)N1);
                    loop_cb.insert(loop_cb.end(), tg.samples[u][w].a[q].coefs, tg.samples[u][w].a[q].coefs + N1);
                }
        TLweSample32 mux(N1);
        CMux(&mux, &tg, &tg.allsamples[0], &tg.allsamples[2 * l1 - 1], env);
        for (int q = 0; q < 2; q++) put(mux.a[q].coefs, 4 * (size_t)N1);
    }
    {   // the driver loop above as ONE launch: tfhe_CircuitBootstrapFFT_array must give the loop's results
        std::vector<LweSample32 *> xin((size_t)count);
        std::vector<TGswSample32 *> res((size_t)count);
        for (int c = 0; c < count; c++) {
            xin[c] = new LweSample32(N1);
            memcpy(xin[c]->a, xs + (size_t)c * (N1 + 1), 4 * (size_t)(N1 + 1));
            res[c] = new TGswSample32(l1, N1);
        }
        tfhe_CircuitBootstrapFFT_array(res.data(), xin.data(), env, count);
        std::vector<int32_t> arr_cb;
        for (int c = 0; c < count; c++)
            for (int u = 0; u < 2; u++)
                for (int w = 0; w < l1; w++)
                    for (int q = 0; q < 2; q++)
                        arr_cb.insert(arr_cb.end(), res[c]->samples[u][w].a[q].coefs, res[c]->samples[u][w].a[q].coefs + N1);
        if (arr_cb != loop_cb) {
            fprintf(stderr, "tfhe_CircuitBootstrapFFT_array differs from the one-by-one loop\n");
            return 3;
        }
    }
    FILE *f = fopen(argv[2], "wb");
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return 0;
}
