// ref_driver.cpp -- ORACLE TOOLING (test infrastructure, not product code).
//
// A command-line harness that is LINKED AGAINST THE REFERENCE'S OWN OBJECT FILES
// (compiled in place from /root/reference by oracle/Makefile into oracle/_ref/; no
// reference source is copied into this repository).  It exposes the reference's
// compilable primitives on binary files so that tests/golden/make_golden.py can
// produce golden vectors and tests can pin oracle/tfhe_oracle.c against the real
// thing:
//
//   spqlios core       fft / ifft / new_*_table      CB/spqlios/spqlios-fft.h:46-53
//   FFT plugin         FFT_Processor_Spqlios::*      CB/spqlios/lagrangehalfc_impl.h:8-36
//   AddMul             LagrangeHalfCPolynomialAddMulASM   lagrangehalfc_impl.h:36
//   PoC functions      tGswTorus64PolynomialDecompH, preKeySwitch, preModSwitch,
//                      circuitPrivKS, circuitBootstrapWoKS
//                      CB/poc_CircuitBootstrapping.cpp:437-698
//   Karatsuba          torus32/64PolynomialMultKaratsuba_lvl1/2   CB/poc_karatsuba.cpp:60-76,168-185
//
// usage: ref_driver <op> <in.bin> <out.bin> [int args...]
// Synthetic key tables are filled from a seed with consecutive splitmix64 outputs
// (high 32 bits), the same rule tests/oracle_py.py:fill32() implements in numpy.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <string>
#include <vector>

#include "poc_types.h"  // reference header (defines `k` as a macro!)
#include "spqlios/lagrangehalfc_impl.h"
#include "spqlios/spqlios-fft.h"

// functions defined (without a header) in CB/poc_CircuitBootstrapping.cpp / poc_karatsuba.cpp
void preKeySwitch(LweSample32 *result, const LweSample32 *x, const Globals *env);
void preModSwitch(int *result, const LweSample32 *x, const Globals *env);
void tGswTorus64PolynomialDecompH(IntPolynomial *result, const Torus64Polynomial *sample, const Globals *env);
void circuitBootstrapWoKS(LweSample64 *result, const Torus64 mu, const int *abar, const Globals *env);
void circuitPrivKS(TLweSample32 *result, const int u, const LweSample64 *x, const Globals *env);
void torus32PolynomialMultKaratsuba_lvl1(Torus32Polynomial *result, const IntPolynomial *poly1,
                                         const Torus32Polynomial *poly2, const Globals *env);
void torus64PolynomialMultKaratsuba_lvl2(Torus64Polynomial *result, const IntPolynomial *poly1,
                                         const Torus64Polynomial *poly2, const Globals *env);

namespace {

std::vector<uint8_t> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)sz);
    if (sz && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { perror("fread"); exit(2); }
    fclose(f);
    return buf;
}
void spill(const char *path, const void *p, size_t bytes) {
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); exit(2); }
    if (bytes && fwrite(p, 1, bytes, f) != bytes) { perror("fwrite"); exit(2); }
    fclose(f);
}

struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    int32_t next32() { return (int32_t)(uint32_t)(next() >> 32); }
};

// mirror of the (anonymous-namespace) FFT_PRECOMP of CB/spqlios/spqlios-fft-impl.cpp:48-53
struct PrecompView {
    uint64_t n;
    double *trig;
    double *data;
    void *buf;
};

// the reference's plugin object of ring degree N: its two global instances (fft_processor_spqlios.cpp:163-164), or a new
// one -- FFT_Processor_Spqlios(N) is constructible for every power of two >= 16 (:18-25, spqlios-fft-impl.cpp:157-160)
FFT_Processor_Spqlios &processor(int N) {
    if (N == 1024) return fftp1024;
    if (N == 2048) return fftp2048;
    return *new FFT_Processor_Spqlios(N);  // lives as long as the process, like the globals
}

// A Globals object WITHOUT running Globals::Globals() (which would spend ~75 s generating
// 2.7 GB of keys, poc:342-423): raw storage, fields filled by hand.
Globals *bare_globals() {
    Globals *env = (Globals *)calloc(1, sizeof(Globals));
    env->N_lvl1 = Globals::n_lvl1;
    env->N_lvl2 = Globals::n_lvl2;
    env->t_lvl0 = Globals::kslength_lvl10 * Globals::ksbasebit_lvl10;
    env->t_lvl1 = Globals::kslength_lvl21 * Globals::ksbasebit_lvl21;
    env->torusDecompOffset = 0;  // poc:349-350
    for (int i = 0; i <= Globals::ell_lvl2; ++i)
        env->torusDecompOffset |= (UINT64_C(1) << (63 - i * Globals::bgbit_lvl2));
    env->torusDecompBuf = new uint64_t[env->N_lvl2];
    return env;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <op> <in.bin> <out.bin> [args]\n", argv[0]);
        return 2;
    }
    const std::string op = argv[1];
    const char *inp = argv[2], *outp = argv[3];
    auto arg = [&](int i) -> long { return (argc > 4 + i) ? atol(argv[4 + i]) : 0; };

    if (op == "params") {  // the PoC's compiled-in parameter block (poc:70-85)
        int32_t p[11] = {Globals::n_lvl0, Globals::n_lvl1, Globals::n_lvl2, Globals::bgbit_lvl1,
                         Globals::ell_lvl1, Globals::bgbit_lvl2, Globals::ell_lvl2,
                         Globals::kslength_lvl10, Globals::ksbasebit_lvl10,
                         Globals::kslength_lvl21, Globals::ksbasebit_lvl21};
        spill(outp, p, sizeof(p));
        return 0;
    }
    if (op == "tables") {  // args: N -> [fft_trig | ifft_trig], each 2N-8 doubles
        const int N = (int)arg(0);
        const size_t len = (size_t)2 * N - 8;
        PrecompView *f = (PrecompView *)new_fft_table(N);
        PrecompView *r = (PrecompView *)new_ifft_table(N);
        std::vector<double> out(2 * len);
        memcpy(out.data(), f->trig, len * 8);
        memcpy(out.data() + len, r->trig, len * 8);
        spill(outp, out.data(), out.size() * 8);
        return 0;
    }
    if (op == "ifft" || op == "fft") {  // args: N ; in/out: count*N doubles, raw core transform
        const int N = (int)arg(0);
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / N;
        void *tab = (op == "ifft") ? new_ifft_table(N) : new_fft_table(N);
        double *buf = (op == "ifft") ? ifft_table_get_buffer(tab) : fft_table_get_buffer(tab);
        std::vector<double> out(cnt * N);
        for (size_t c = 0; c < cnt; c++) {
            memcpy(buf, in.data() + c * N * 8, (size_t)N * 8);
            if (op == "ifft") ifft(tab, buf); else fft(tab, buf);
            memcpy(out.data() + c * N, buf, (size_t)N * 8);
        }
        spill(outp, out.data(), out.size() * 8);
        return 0;
    }
    if (op == "rev_int" || op == "rev_t32") {  // int32 -> Lagrange
        const int N = (int)arg(0);
        FFT_Processor_Spqlios &P = processor(N);
        auto in = slurp(inp);
        const size_t cnt = in.size() / 4 / N;
        std::vector<double> out(cnt * N);
        for (size_t c = 0; c < cnt; c++) {
            if (op == "rev_int") P.execute_reverse_int(out.data() + c * N, (const int *)in.data() + c * N);
            else P.execute_reverse_torus32(out.data() + c * N, (const int32_t *)in.data() + c * N);
        }
        spill(outp, out.data(), out.size() * 8);
        return 0;
    }
    if (op == "rev_t64") {  // int64 -> Lagrange
        const int N = (int)arg(0);
        FFT_Processor_Spqlios &P = processor(N);
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / N;
        std::vector<double> out(cnt * N);
        for (size_t c = 0; c < cnt; c++) P.execute_reverse_torus64(out.data() + c * N, (const int64_t *)in.data() + c * N);
        spill(outp, out.data(), out.size() * 8);
        return 0;
    }
    if (op == "dir_t32") {  // Lagrange -> Torus32.  ONE N per process (static _2sN, fft_processor_spqlios.cpp:78)
        const int N = (int)arg(0);
        FFT_Processor_Spqlios &P = processor(N);
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / N;
        std::vector<int32_t> out(cnt * N);
        for (size_t c = 0; c < cnt; c++) P.execute_direct_torus32(out.data() + c * N, (const double *)in.data() + c * N);
        spill(outp, out.data(), out.size() * 4);
        return 0;
    }
    if (op == "dir_t64") {  // Lagrange -> Torus64.  ONE N per process (static _2sN, :106)
        const int N = (int)arg(0);
        FFT_Processor_Spqlios &P = processor(N);
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / N;
        std::vector<int64_t> out(cnt * N);
        for (size_t c = 0; c < cnt; c++) P.execute_direct_torus64(out.data() + c * N, (const double *)in.data() + c * N);
        spill(outp, out.data(), out.size() * 8);
        return 0;
    }
    if (op == "addmul") {  // args: N ; in: count * [res|a|b] (3N doubles) ; out: count*N doubles
        const int N = (int)arg(0);
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / (3 * (size_t)N);
        std::vector<double> out(cnt * N);
        for (size_t c = 0; c < cnt; c++) {
            double *base = (double *)in.data() + c * 3 * N;
            LagrangeHalfCPolynomialAddMulASM(base, base + N, base + 2 * N, N / 2);
            memcpy(out.data() + c * N, base, (size_t)N * 8);
        }
        spill(outp, out.data(), out.size() * 8);
        return 0;
    }
    if (op == "karat32" || op == "karat64") {  // in: count*[int32 poly1 (N) | torus poly2 (N)]
        Globals *env = bare_globals();
        auto in = slurp(inp);
        if (op == "karat32") {
            const int N = env->N_lvl1;
            const size_t cnt = in.size() / (8 * (size_t)N);
            std::vector<int32_t> out(cnt * N);
            IntPolynomial a(N);
            Torus32Polynomial b(N), r(N);
            for (size_t c = 0; c < cnt; c++) {
                memcpy(a.coefs, in.data() + c * 8 * N, 4 * (size_t)N);
                memcpy(b.coefs, in.data() + c * 8 * N + 4 * N, 4 * (size_t)N);
                torus32PolynomialMultKaratsuba_lvl1(&r, &a, &b, env);
                memcpy(out.data() + c * N, r.coefs, 4 * (size_t)N);
            }
            spill(outp, out.data(), out.size() * 4);
        } else {
            const int N = env->N_lvl2;
            const size_t cnt = in.size() / (12 * (size_t)N);
            std::vector<int64_t> out(cnt * N);
            IntPolynomial a(N);
            Torus64Polynomial b(N), r(N);
            for (size_t c = 0; c < cnt; c++) {
                memcpy(a.coefs, in.data() + c * 12 * N, 4 * (size_t)N);
                memcpy(b.coefs, in.data() + c * 12 * N + 4 * N, 8 * (size_t)N);
                torus64PolynomialMultKaratsuba_lvl2(&r, &a, &b, env);
                memcpy(out.data() + c * N, r.coefs, 8 * (size_t)N);
            }
            spill(outp, out.data(), out.size() * 8);
        }
        return 0;
    }
    if (op == "decomp64") {  // in: count*N2 int64 ; out: count*[l2][N2] int32
        Globals *env = bare_globals();
        const int N = env->N_lvl2, l = Globals::ell_lvl2;
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / N;
        std::vector<int32_t> out(cnt * l * N);
        IntPolynomial *res = new_array1<IntPolynomial>(l, N);
        Torus64Polynomial s(N);
        for (size_t c = 0; c < cnt; c++) {
            memcpy(s.coefs, in.data() + c * N * 8, (size_t)N * 8);
            tGswTorus64PolynomialDecompH(res, &s, env);
            for (int p = 0; p < l; p++) memcpy(out.data() + (c * l + p) * N, res[p].coefs, (size_t)N * 4);
        }
        spill(outp, out.data(), out.size() * 4);
        return 0;
    }
    if (op == "premodswitch") {  // in: count*(n0+1) int32 ; out: count*(n0+1) int32
        Globals *env = bare_globals();
        const int n0 = Globals::n_lvl0;
        auto in = slurp(inp);
        const size_t cnt = in.size() / 4 / (n0 + 1);
        std::vector<int32_t> out(cnt * (n0 + 1));
        LweSample32 x(n0);
        for (size_t c = 0; c < cnt; c++) {
            memcpy(x.a, in.data() + c * (n0 + 1) * 4, (size_t)(n0 + 1) * 4);
            preModSwitch((int *)out.data() + c * (n0 + 1), &x, env);
        }
        spill(outp, out.data(), out.size() * 4);
        return 0;
    }
    if (op == "preks") {  // args: seed ; in: count*(N1+1) int32 ; out: count*(n0+1) int32
        Globals *env = bare_globals();
        const int n0 = Globals::n_lvl0, n1 = Globals::n_lvl1, t = Globals::kslength_lvl10;
        const int base = 1 << Globals::ksbasebit_lvl10;
        env->preKS = new_array3<LweSample32>(n1, t, base, n0);
        SplitMix rng((uint64_t)arg(0));
        for (int i = 0; i < n1; i++)
            for (int j = 0; j < t; j++)
                for (int u = 0; u < base; u++)
                    for (int h = 0; h <= n0; h++) env->preKS[i][j][u].a[h] = rng.next32();
        auto in = slurp(inp);
        const size_t cnt = in.size() / 4 / (n1 + 1);
        std::vector<int32_t> out(cnt * (n0 + 1));
        LweSample32 x(n1), r(n0);
        for (size_t c = 0; c < cnt; c++) {
            memcpy(x.a, in.data() + c * (n1 + 1) * 4, (size_t)(n1 + 1) * 4);
            preKeySwitch(&r, &x, env);
            memcpy(out.data() + c * (n0 + 1), r.a, (size_t)(n0 + 1) * 4);
        }
        spill(outp, out.data(), out.size() * 4);
        return 0;
    }
    if (op == "privks") {  // args: seed, u ; in: count*(n2+1) int64 ; out: count*[2][N1] int32
        Globals *env = bare_globals();
        const int n2 = Globals::n_lvl2, N1 = Globals::n_lvl1, t = Globals::kslength_lvl21;
        const int base = 1 << Globals::ksbasebit_lvl21;
        const int u = (int)arg(1);
        // only plane u is touched by circuitPrivKS(.., u, ..): build that plane, alias the other
        TLweSample32 ***plane = new_array3<TLweSample32>(n2 + 1, t, base, N1);
        env->privKS = new TLweSample32 ***[2];
        env->privKS[0] = env->privKS[1] = plane;
        SplitMix rng((uint64_t)arg(0));
        for (int i = 0; i <= n2; i++)
            for (int j = 0; j < t; j++)
                for (int d = 0; d < base; d++)
                    for (int q = 0; q <= 1; q++)
                        for (int p = 0; p < N1; p++) plane[i][j][d].a[q].coefs[p] = rng.next32();
        auto in = slurp(inp);
        const size_t cnt = in.size() / 8 / (n2 + 1);
        std::vector<int32_t> out(cnt * 2 * N1);
        LweSample64 x(n2);
        TLweSample32 r(N1);
        for (size_t c = 0; c < cnt; c++) {
            memcpy(x.a, in.data() + c * (n2 + 1) * 8, (size_t)(n2 + 1) * 8);
            circuitPrivKS(&r, u, &x, env);
            for (int q = 0; q <= 1; q++) memcpy(out.data() + (c * 2 + q) * N1, r.a[q].coefs, (size_t)N1 * 4);
        }
        spill(outp, out.data(), out.size() * 4);
        return 0;
    }
    if (op == "cbwoks") {
        // The PoC blind rotation as written (always bkFFT[0], wrong sign, X^{+bbar});
        // defined behaviour only while every abar[i] < N2.
        // in: [int64 mu][int32 abar (n0+1)][pad to 8][double bkFFT0: [2*l2][2][N2]] ; out: int64 (N2+1)
        Globals *env = bare_globals();
        const int n0 = Globals::n_lvl0, N = Globals::n_lvl2, l = Globals::ell_lvl2;
        auto in = slurp(inp);
        const uint8_t *p = in.data();
        int64_t mu;
        memcpy(&mu, p, 8);
        p += 8;
        std::vector<int> abar(n0 + 1);
        memcpy(abar.data(), p, (size_t)(n0 + 1) * 4);
        p += (((size_t)(n0 + 1) * 4 + 7) / 8) * 8;
        for (int i = 0; i < n0; i++)
            if (abar[i] >= N) { fprintf(stderr, "cbwoks: abar[%d] >= N2 is out of bounds in the PoC\n", i); return 3; }
        env->bkFFT = new_array1<TGswSampleFFT>(1, l, N);
        for (int r = 0; r < 2 * l; r++)
            for (int q = 0; q <= 1; q++) {
                memcpy(env->bkFFT[0].allsamples[r].a[q].values, p, (size_t)N * 8);
                p += (size_t)N * 8;
            }
        LweSample64 res(N);
        circuitBootstrapWoKS(&res, mu, abar.data(), env);
        spill(outp, res.a, (size_t)(N + 1) * 8);
        return 0;
    }
    if (op == "boot32" || op == "bench32") {
        // Gate bootstrap (tfhe_bootstrap_FFT, CB/lwe_functions.cpp:434-446) COMPOSED FROM THE
        // REFERENCE'S OWN FFT/MAC OBJECT CODE: fftp1024.execute_reverse_int, the AddMul assembly and
        // fftp1024.execute_direct_torus32 do the arithmetic; the integer glue around them
        // (decomposition tgsw_functions.cpp:224-337, rotation numeric_functions.cpp:304-347,
        // extraction tlwe_functions.cpp:351-363, mod switch numeric_functions.cpp:54-60, key switch
        // lwe_functions.cpp:136-171) is written here because those reference files do not compile.
        //   boot32  args: n l Bgbit ks_t ks_bb count [first [N]] ; in: [mu i32][pad i32][bkfft][ks][x rows] ; out: count*(n+1) i32
        //           (rows first .. first+count-1 of x: one input file serves several processes)
        //   bench32 args: n l Bgbit ks_t ks_bb seconds ; synthetic keys/samples; prints "<count> <seconds>"
        const int n = (int)arg(0), l = (int)arg(1), Bgbit = (int)arg(2), t = (int)arg(3), bb = (int)arg(4);
        const int N = arg(7) ? (int)arg(7) : 1024;  // boot32 only: ring degree (default 1024)
        FFT_Processor_Spqlios &fftpN = processor(N);
        const int logn = __builtin_ctz((unsigned)N);
        const int kpl = 2 * l, base = 1 << bb;
        const size_t bk_len = (size_t)n * kpl * 2 * N, ks_len = (size_t)N * t * base * (n + 1);
        std::vector<double> bk(bk_len);
        std::vector<int32_t> ks(ks_len), xs;
        int32_t mu = 1 << 29;
        size_t count = 0;
        double budget = 0;
        if (op == "boot32") {
            auto in = slurp(inp);
            const uint8_t *p = in.data();
            memcpy(&mu, p, 4);
            p += 8;
            memcpy(bk.data(), p, bk_len * 8);
            p += bk_len * 8;
            memcpy(ks.data(), p, ks_len * 4);
            p += ks_len * 4;
            count = (size_t)arg(5);
            const size_t first = (size_t)arg(6);
            const size_t have = (in.size() - (size_t)(p - in.data())) / 4 / (size_t)(n + 1);
            if (first + count > have) {
                fprintf(stderr, "boot32: rows %zu..%zu requested, the file holds %zu\n", first, first + count, have);
                return 2;
            }
            xs.resize(count * (n + 1));
            memcpy(xs.data(), p + first * (size_t)(n + 1) * 4, xs.size() * 4);
        } else {
            budget = (double)arg(5);
            SplitMix rng(42);
            std::vector<int32_t> tor(N);
            for (size_t r = 0; r < bk_len / N; r++) {
                for (int j = 0; j < N; j++) tor[j] = rng.next32();
                fftpN.execute_reverse_torus32(bk.data() + r * N, tor.data());
            }
            for (auto &v : ks) v = rng.next32();
            count = 1u << 20;  // upper bound; the time budget stops the loop
            xs.resize((size_t)64 * (n + 1));
            for (auto &v : xs) v = rng.next32();
        }
        uint32_t offset = 0;
        for (int i = 0; i < l; i++) offset += 1u << (32 - (i + 1) * Bgbit);
        offset *= 1u << (Bgbit - 1);
        const uint32_t mask = (1u << Bgbit) - 1;
        const int32_t halfBg = 1 << (Bgbit - 1);
        // bench32 keeps one output row (its `count` is only a loop bound: never size a buffer by it)
        std::vector<int32_t> acc(2 * N), tmp(2 * N), deca((size_t)kpl * N), u(N + 1),
            out((op == "boot32" ? count : 1) * (size_t)(n + 1));
        std::vector<double> decaF((size_t)kpl * N), tmpa(2 * N);
        auto modsw = [&](int32_t ph) { return (int)((((uint64_t)(uint32_t)ph << 32) + (1ull << (62 - logn))) >> (63 - logn)); };
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        size_t done = 0;
        for (size_t c = 0; c < count; c++) {
            const int32_t *x = xs.data() + (op == "boot32" ? c : (c % 64)) * (n + 1);
            const int barb = modsw(x[n]);
            // acc = (0, X^{2N-barb} * mu)
            for (int j = 0; j < N; j++) {
                acc[j] = 0;
                const int idx = (j - (2 * N - barb)) & (2 * N - 1);
                acc[N + j] = (idx & N) ? -mu : mu;
            }
            for (int i = 0; i < n; i++) {
                const int a = modsw(x[i]);
                if (a == 0) continue;
                for (int q = 0; q < 2; q++)
                    for (int j = 0; j < N; j++) {
                        const int idx = (j - a) & (2 * N - 1);
                        const uint32_t s = (uint32_t)acc[q * N + (idx & (N - 1))];
                        tmp[q * N + j] = (int32_t)(((idx & N) ? (0u - s) : s) - (uint32_t)acc[q * N + j]);
                    }
                for (int q = 0; q < 2; q++)
                    for (int d = 0; d < l; d++) {
                        const int decal = 32 - (d + 1) * Bgbit;
                        int32_t *o = deca.data() + (size_t)(q * l + d) * N;
                        for (int j = 0; j < N; j++)
                            o[j] = (int32_t)((((uint32_t)tmp[q * N + j] + offset) >> decal) & mask) - halfBg;
                    }
                for (int p = 0; p < kpl; p++) fftpN.execute_reverse_int(decaF.data() + (size_t)p * N, deca.data() + (size_t)p * N);
                std::fill(tmpa.begin(), tmpa.end(), 0.0);
                const double *row = bk.data() + (size_t)i * kpl * 2 * N;
                for (int p = 0; p < kpl; p++)
                    for (int q = 0; q < 2; q++)
                        LagrangeHalfCPolynomialAddMulASM(tmpa.data() + q * N, decaF.data() + (size_t)p * N,
                                                         (double *)row + ((size_t)p * 2 + q) * N, N / 2);
                for (int q = 0; q < 2; q++) fftpN.execute_direct_torus32(tmp.data() + q * N, tmpa.data() + q * N);
                for (int j = 0; j < 2 * N; j++) acc[j] = (int32_t)((uint32_t)acc[j] + (uint32_t)tmp[j]);
            }
            u[0] = acc[0];
            for (int j = 1; j < N; j++) u[j] = (int32_t)(0u - (uint32_t)acc[N - j]);
            u[N] = acc[N];
            int32_t *r = out.data() + (op == "boot32" ? c : 0) * (n + 1);
            for (int h = 0; h < n; h++) r[h] = 0;
            r[n] = u[N];
            const uint32_t prec = 1u << (32 - (1 + bb * t));
            for (int i = 0; i < N; i++) {
                const uint32_t aibar = (uint32_t)u[i] + prec;
                for (int j = 0; j < t; j++) {
                    const uint32_t aij = (aibar >> (32 - (j + 1) * bb)) & (uint32_t)(base - 1);
                    if (!aij) continue;
                    const int32_t *kr = ks.data() + (((size_t)i * t + j) * base + aij) * (n + 1);
                    for (int h = 0; h <= n; h++) r[h] = (int32_t)((uint32_t)r[h] - (uint32_t)kr[h]);
                }
            }
            done++;
            if (op == "bench32") {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) >= budget) break;
            }
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        if (op == "boot32") spill(outp, out.data(), out.size() * 4);
        printf("%zu %.6f\n", done, secs);
        return 0;
    }
    fprintf(stderr, "unknown op %s\n", op.c_str());
    return 2;
}
