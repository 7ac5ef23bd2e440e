// compat_driver.cpp -- test program for include/tfhe_amd_compat.hpp: builds the reference's
// pointer-rich structs from flat arrays, calls the reference-named entry points, and dumps the
// results for tests/test_compat.py to compare with the oracle.  Links against whichever library
// implements the C ABI (the HIP library on a GPU box, the tests/emu build on the CPU).
//
//   compat_driver lib   <in.bin> <out.bin>     library-form functions + FFT plugin look-alike
//   compat_driver poc   <in.bin> <out.bin>     PoC-form functions (PocEngine<Globals>)
//   compat_driver arr   <in.bin> <out.bin>     ARRAY forms (tfhe_bootstrap_FFT_array, tfhe_bootstrap_woKS_FFT_array,
//                                              lweKeySwitch_array) against the one-by-one loop over the same samples:
//                                              exits 3 when they differ; prints count / seconds of both on stdout
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

// -DDROPIN: the LITERAL drop-in form -- only the global-scope declarations of include/tfhe_amd_dropin.h
// (what a driver written against the reference declares), no namespace, linked against
// libtfhe_amd_dropin.so; the library-form section below is otherwise the same source.
#ifdef DROPIN
#include "tfhe_amd_dropin.h"
#else
#include "tfhe_amd_compat.hpp"

using namespace tfhe_amd_compat;
#endif

// TFHE_COMPAT_DEVICES="1,4,6": the array forms are ALSO run over a pool of these devices (set_devices / tfhe_amd_dropin_set_devices)
static std::vector<int> env_devices() {
    std::vector<int> d;
    const char *e = getenv("TFHE_COMPAT_DEVICES");
    if (!e) return d;
    for (const char *p = e; *p;) {
        d.push_back(atoi(p));
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
    }
    return d;
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
struct Out {
    std::vector<uint8_t> buf;
    void put(const void *p, size_t n) { buf.insert(buf.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
    void save(const char *path) {
        FILE *f = fopen(path, "wb");
        fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
    }
};
struct In {
    const uint8_t *p;
    template <class T> const T *take(size_t n) {
        const T *r = (const T *)p;
        p += sizeof(T) * n;
        return r;
    }
};

// ---------------------------------------------------------------- library form
static int run_lib(const char *inp, const char *outp) {
    auto blob = slurp(inp);
    In in{blob.data()};
    const int32_t *hdr = in.take<int32_t>(8);  // n N l Bgbit ks_t ks_bb count mu
    const int n = hdr[0], N = hdr[1], l = hdr[2], Bgbit = hdr[3], t = hdr[4], bb = hdr[5], count = hdr[6];
    const Torus32 mu = hdr[7];
    const int kpl = 2 * l, base = 1 << bb;
    const double *bkflat = in.take<double>((size_t)n * kpl * 2 * N);
    const int32_t *ksflat = in.take<int32_t>((size_t)N * t * base * (n + 1));
    const int32_t *xs = in.take<int32_t>((size_t)count * (n + 1));
    const int32_t *rot = in.take<int32_t>((size_t)count * (n + 1));   // bara..., barb
    const int32_t *accs = in.take<int32_t>((size_t)count * 2 * N);
    const int32_t *v = in.take<int32_t>((size_t)N);

    LweParams in_out{n, 0, 0};
    TLweParams tlwe{N, 1, 0, 0, LweParams{N, 0, 0}};
    TGswParams gp;
    memset(&gp, 0, sizeof(gp));
    gp.l = l;
    gp.Bgbit = Bgbit;
    gp.Bg = 1 << Bgbit;
    gp.tlwe_params = &tlwe;
    gp.kpl = kpl;
    // TGswSampleFFT array of n samples, each kpl TLweSampleFFT rows of 2 LagrangeHalfC polynomials
    std::vector<LagrangeHalfCPolynomial> polys((size_t)n * kpl * 2);
    std::vector<TLweSampleFFT> rows((size_t)n * kpl);
    std::vector<TGswSampleFFT> gsw((size_t)n);
    for (int i = 0; i < n; i++) {
        for (int r = 0; r < kpl; r++) {
            for (int q = 0; q < 2; q++)
                polys[((size_t)i * kpl + r) * 2 + q].values = (double *)bkflat + (((size_t)i * kpl + r) * 2 + q) * N;
            TLweSampleFFT &row = rows[(size_t)i * kpl + r];
            row.a = &polys[((size_t)i * kpl + r) * 2];
            row.b = row.a + 1;
            row.k = 1;
        }
        gsw[i].all_samples = &rows[(size_t)i * kpl];
        gsw[i].sample = nullptr;
        gsw[i].k = 1;
        gsw[i].l = l;
    }
    // LweKeySwitchKey ks[N][t][base] of LweSample (n)
    std::vector<LweSample> ks0((size_t)N * t * base);
    std::vector<LweSample *> ks1((size_t)N * t);
    std::vector<LweSample **> ks2((size_t)N);
    for (size_t e = 0; e < ks0.size(); e++) {
        ks0[e].a = (Torus32 *)ksflat + e * (n + 1);
        ks0[e].b = ksflat[e * (n + 1) + n];
    }
    for (size_t e = 0; e < ks1.size(); e++) ks1[e] = &ks0[e * base];
    for (int i = 0; i < N; i++) ks2[i] = &ks1[(size_t)i * t];
    LweKeySwitchKey ksk{N, t, bb, base, &in_out, ks0.data(), ks1.data(), ks2.data()};
    LweBootstrappingKeyFFT bk{&in_out, &gp, &tlwe, &tlwe.extracted_lweparams, gsw.data(), &ksk};

    Out out;
    std::vector<Torus32> ra((size_t)N + 1), rb((size_t)n + 1);
    for (int c = 0; c < count; c++) {
        LweSample x{(Torus32 *)xs + (size_t)c * (n + 1), xs[(size_t)c * (n + 1) + n], 0};
        LweSample u{ra.data(), 0, 0}, r{rb.data(), 0, 0};
        tfhe_bootstrap_woKS_FFT(&u, &bk, mu, &x);
        out.put(u.a, 4 * (size_t)N);
        out.put(&u.b, 4);
        tfhe_bootstrap_FFT(&r, &bk, mu, &x);
        out.put(r.a, 4 * (size_t)n);
        out.put(&r.b, 4);
        lweKeySwitch(&r, &ksk, &u);
        out.put(r.a, 4 * (size_t)n);
        out.put(&r.b, 4);
        // blind rotate + extract with an explicit test vector
        TorusPolynomial tv{N, (Torus32 *)v};
        const int32_t *rt = rot + (size_t)c * (n + 1);
        tfhe_blindRotateAndExtract_FFT(&u, &tv, gsw.data(), rt[n], (const int *)rt, n, &gp);
        out.put(u.a, 4 * (size_t)N);
        out.put(&u.b, 4);
        // in-place blind rotation and external product on a TLWE sample
        std::vector<Torus32> acc(accs + (size_t)c * 2 * N, accs + (size_t)(c + 1) * 2 * N);
        TorusPolynomial ap[2] = {{N, acc.data()}, {N, acc.data() + N}};
        TLweSample as{ap, ap + 1, 0, 1};
        tfhe_blindRotate_FFT(&as, gsw.data(), (const int *)rt, n, &gp);
        out.put(acc.data(), 4 * (size_t)2 * N);
        std::vector<Torus32> acc2(accs + (size_t)c * 2 * N, accs + (size_t)(c + 1) * 2 * N);
        TorusPolynomial ap2[2] = {{N, acc2.data()}, {N, acc2.data() + N}};
        TLweSample as2{ap2, ap2 + 1, 0, 1};
        tGswFFTExternMulToTLwe(&as2, &gsw[n - 1], &gp);
        out.put(acc2.data(), 4 * (size_t)2 * N);
        // one CMux step on its own (tfhe_MuxRotate_FFT, CB/lwe_functions.cpp:328-333): into a second sample, then in place;
        // rotation amounts from the sample's own rotation list (any of [0, 2N): 0 must give the accumulator back)
        std::vector<Torus32> acc3(accs + (size_t)c * 2 * N, accs + (size_t)(c + 1) * 2 * N), res3((size_t)2 * N, 0);
        TorusPolynomial ap3[2] = {{N, acc3.data()}, {N, acc3.data() + N}}, rp3[2] = {{N, res3.data()}, {N, res3.data() + N}};
        TLweSample as3{ap3, ap3 + 1, 0, 1}, rs3{rp3, rp3 + 1, 0, 1};
        tfhe_MuxRotate_FFT(&rs3, &as3, &gsw[0], rt[0], &gp);
        out.put(res3.data(), 4 * (size_t)2 * N);
        tfhe_MuxRotate_FFT(&as3, &as3, &gsw[n - 1], c == 0 ? 0 : rt[n - 1], &gp);
        out.put(acc3.data(), 4 * (size_t)2 * N);
    }
    // release(bk), then a DIFFERENT key rebuilt at the same addresses (the TGSW samples in reverse order: same
    // structs, other polynomial pointers): the next call must upload it again, not reuse the stale GPU copy
    {
#ifdef DROPIN
        tfhe_amd_dropin_release(&bk);
#else
        release(&bk);
#endif
        for (int i = 0; i < n; i++)
            for (int r = 0; r < kpl; r++)
                for (int q = 0; q < 2; q++)
                    polys[((size_t)i * kpl + r) * 2 + q].values = (double *)bkflat + (((size_t)(n - 1 - i) * kpl + r) * 2 + q) * N;
        LweSample x{(Torus32 *)xs, xs[n], 0};
        LweSample u{ra.data(), 0, 0}, r{rb.data(), 0, 0};
        tfhe_bootstrap_woKS_FFT(&u, &bk, mu, &x);
        out.put(u.a, 4 * (size_t)N);
        out.put(&u.b, 4);
        tfhe_bootstrap_FFT(&r, &bk, mu, &x);
        out.put(r.a, 4 * (size_t)n);
        out.put(&r.b, 4);
    }
    // ... and the ORIGINAL key put back at the same addresses WITHOUT a release: the content sample taken at every lookup
    // (tfhe_amd_compat::key_fingerprint) must notice, and the outputs must be those of the original key again
    {
        for (int i = 0; i < n; i++)
            for (int r = 0; r < kpl; r++)
                for (int q = 0; q < 2; q++)
                    polys[((size_t)i * kpl + r) * 2 + q].values = (double *)bkflat + (((size_t)i * kpl + r) * 2 + q) * N;
        LweSample x{(Torus32 *)xs, xs[n], 0};
        LweSample r{rb.data(), 0, 0};
        tfhe_bootstrap_FFT(&r, &bk, mu, &x);
        out.put(r.a, 4 * (size_t)n);
        out.put(&r.b, 4);
    }
    // ONE interior TGSW sample replaced in place (sample 1 now points at sample 2's polynomials), again without a release: the
    // content sample touches every TGSW sample of the key, so this too must be served from a fresh upload -- and the original
    // put back the same way for what follows
    if (n >= 5) {
        for (int pass = 0; pass < 2; pass++) {
            for (int r = 0; r < kpl; r++)
                for (int q = 0; q < 2; q++)
                    polys[((size_t)1 * kpl + r) * 2 + q].values = (double *)bkflat + (((size_t)(pass == 0 ? 2 : 1) * kpl + r) * 2 + q) * N;
            LweSample x{(Torus32 *)xs, xs[n], 0};
            LweSample r{rb.data(), 0, 0};
            tfhe_bootstrap_FFT(&r, &bk, mu, &x);
            out.put(r.a, 4 * (size_t)n);
            out.put(&r.b, 4);
        }
    }
#ifdef DROPIN
    tfhe_amd_dropin_release(nullptr);
    out.save(outp);
    return 0;
#else
    // FFT plugin look-alike on the first accumulator polynomial
    {
        FFT_Processor_AMD P(N);
        std::vector<double> lag((size_t)N), lag2((size_t)N);
        std::vector<int32_t> back((size_t)N);
        P.execute_reverse_int(lag.data(), (const int *)accs);
        out.put(lag.data(), 8 * (size_t)N);
        P.execute_direct_torus32(back.data(), lag.data());
        out.put(back.data(), 4 * (size_t)N);
        P.execute_reverse_torus32(lag2.data(), accs + N);
        P.AddMul(lag2.data(), lag.data(), bkflat);
        out.put(lag2.data(), 8 * (size_t)N);
        std::vector<int64_t> a64((size_t)N), b64((size_t)N);
        for (int j = 0; j < N; j++) a64[j] = ((int64_t)accs[j] << 32) ^ (int64_t)accs[N + j];
        P.execute_reverse_torus64(lag.data(), a64.data());
        out.put(lag.data(), 8 * (size_t)N);
        P.execute_direct_torus64(b64.data(), lag.data());
        out.put(b64.data(), 8 * (size_t)N);
    }
    release_all();
    out.save(outp);
    return 0;
#endif
}

// ---------------------------------------------------------------- array forms
static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static int run_arr(const char *inp, const char *outp) {
    auto blob = slurp(inp);
    In in{blob.data()};
    const int32_t *hdr = in.take<int32_t>(8);  // n N l Bgbit ks_t ks_bb count mu   (same file as `lib`)
    const int n = hdr[0], N = hdr[1], l = hdr[2], Bgbit = hdr[3], t = hdr[4], bb = hdr[5], count = hdr[6];
    const Torus32 mu = hdr[7];
    const int kpl = 2 * l, base = 1 << bb;
    const double *bkflat = in.take<double>((size_t)n * kpl * 2 * N);
    const int32_t *ksflat = in.take<int32_t>((size_t)N * t * base * (n + 1));
    const int32_t *xs = in.take<int32_t>((size_t)count * (n + 1));
    LweParams in_out{n, 0, 0};
    TLweParams tlwe{N, 1, 0, 0, LweParams{N, 0, 0}};
    TGswParams gp;
    memset(&gp, 0, sizeof(gp));
    gp.l = l;
    gp.Bgbit = Bgbit;
    gp.Bg = 1 << Bgbit;
    gp.tlwe_params = &tlwe;
    gp.kpl = kpl;
    std::vector<LagrangeHalfCPolynomial> polys((size_t)n * kpl * 2);
    std::vector<TLweSampleFFT> rows((size_t)n * kpl);
    std::vector<TGswSampleFFT> gsw((size_t)n);
    for (int i = 0; i < n; i++) {
        for (int r = 0; r < kpl; r++) {
            for (int q = 0; q < 2; q++)
                polys[((size_t)i * kpl + r) * 2 + q].values = (double *)bkflat + (((size_t)i * kpl + r) * 2 + q) * N;
            TLweSampleFFT &row = rows[(size_t)i * kpl + r];
            row.a = &polys[((size_t)i * kpl + r) * 2];
            row.b = row.a + 1;
            row.k = 1;
        }
        gsw[i].all_samples = &rows[(size_t)i * kpl];
        gsw[i].sample = nullptr;
        gsw[i].k = 1;
        gsw[i].l = l;
    }
    std::vector<LweSample> ks0((size_t)N * t * base);
    std::vector<LweSample *> ks1((size_t)N * t);
    std::vector<LweSample **> ks2((size_t)N);
    for (size_t e = 0; e < ks0.size(); e++) {
        ks0[e].a = (Torus32 *)ksflat + e * (n + 1);
        ks0[e].b = ksflat[e * (n + 1) + n];
    }
    for (size_t e = 0; e < ks1.size(); e++) ks1[e] = &ks0[e * base];
    for (int i = 0; i < N; i++) ks2[i] = &ks1[(size_t)i * t];
    LweKeySwitchKey ksk{N, t, bb, base, &in_out, ks0.data(), ks1.data(), ks2.data()};
    LweBootstrappingKeyFFT bk{&in_out, &gp, &tlwe, &tlwe.extracted_lweparams, gsw.data(), &ksk};

    // the caller's loop objects: count inputs, count outputs of each kind
    std::vector<LweSample> x((size_t)count), ra((size_t)count), rl((size_t)count), ru((size_t)count), rk((size_t)count);
    std::vector<const LweSample *> xp((size_t)count), up((size_t)count);
    std::vector<LweSample *> rap((size_t)count), rup((size_t)count), rkp((size_t)count);
    std::vector<Torus32> a_arr((size_t)count * n), a_loop((size_t)count * n), a_woks((size_t)count * N), a_ks((size_t)count * n);
    for (int c = 0; c < count; c++) {
        x[c] = LweSample{(Torus32 *)xs + (size_t)c * (n + 1), xs[(size_t)c * (n + 1) + n], 0};
        ra[c] = LweSample{a_arr.data() + (size_t)c * n, 0, 0};
        rl[c] = LweSample{a_loop.data() + (size_t)c * n, 0, 0};
        ru[c] = LweSample{a_woks.data() + (size_t)c * N, 0, 0};
        rk[c] = LweSample{a_ks.data() + (size_t)c * n, 0, 0};
        xp[c] = &x[c];
        rap[c] = &ra[c];
        rup[c] = &ru[c];
        up[c] = &ru[c];
        rkp[c] = &rk[c];
    }
    // an empty array is a no-op (no key upload, no launch, nothing dereferenced) in every array form
    tfhe_bootstrap_FFT_array(nullptr, &bk, mu, nullptr, 0);
    tfhe_bootstrap_woKS_FFT_array(nullptr, &bk, mu, nullptr, 0);
    lweKeySwitch_array(nullptr, &ksk, nullptr, 0);
    // first call: uploads the keys (once per key object) and sizes the staging buffers
    tfhe_bootstrap_FFT_array(rap.data(), &bk, mu, xp.data(), count);
    const double t0 = now_s();
    tfhe_bootstrap_FFT_array(rap.data(), &bk, mu, xp.data(), count);  // gather + PCIe + one launch + PCIe + scatter
    const double t_arr = now_s() - t0;
    // the two halves as array forms, chained: woKS then the key switch on its outputs == the full bootstrap
    tfhe_bootstrap_woKS_FFT_array(rup.data(), &bk, mu, xp.data(), count);
    lweKeySwitch_array(rkp.data(), &ksk, up.data(), count);
    // the reference's own schedule: one sample per call
    const double t1 = now_s();
    for (int c = 0; c < count; c++) tfhe_bootstrap_FFT(&rl[c], &bk, mu, &x[c]);
    const double t_loop = now_s() - t1;
    // The reference's OWN parallel construct -- `#pragma omp parallel for` over independent one-item calls
    // (parallel/src/test_parallel_multiplications.cpp:62) -- on the unmodified one-sample entry point: the calls of the
    // threads are coalesced by the shim into array launches (same results, a multiple of the serial loop's rate)
    const int tcount = count;
    std::vector<Torus32> a_thr((size_t)tcount * n);
    std::vector<LweSample> rt((size_t)tcount);
    for (int c = 0; c < tcount; c++) rt[c] = LweSample{a_thr.data() + (size_t)c * n, 0, 0};
    const int want_threads = getenv("TFHE_COMPAT_THREADS") ? atoi(getenv("TFHE_COMPAT_THREADS")) : 32;
    const int nthreads = count < want_threads ? (count < 2 ? 2 : count) : want_threads;
    unsigned long co_b0 = 0, co_q0 = 0, co_b1 = 0, co_q1 = 0, co_w0 = 0, co_t0 = 0, co_w1 = 0, co_t1 = 0;  // launches / requests of the coalescer around the loop
#ifndef DROPIN
    lwe_coalescer_totals(&co_b0, &co_q0, &co_w0, &co_t0);
#endif
    const double t2 = now_s();
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int c = 0; c < tcount; c++) tfhe_bootstrap_FFT(&rt[c], &bk, mu, &x[c]);
#else
    {
        std::vector<std::thread> pool;
        for (int w = 0; w < nthreads; w++)
            pool.emplace_back([&, w]() {
                for (int c = w; c < tcount; c += nthreads) tfhe_bootstrap_FFT(&rt[c], &bk, mu, &x[c]);
            });
        for (auto &th : pool) th.join();
    }
#endif
    const double t_par = now_s() - t2;
#ifndef DROPIN
    lwe_coalescer_totals(&co_b1, &co_q1, &co_w1, &co_t1);
#endif
    // the other coalesced entry points from threads: woKS and the key switch on its outputs == the full bootstrap
    std::vector<Torus32> a_thr_woks((size_t)tcount * N), a_thr_ks((size_t)tcount * n);
    std::vector<LweSample> rtu((size_t)tcount), rtk((size_t)tcount);
    for (int c = 0; c < tcount; c++) {
        rtu[c] = LweSample{a_thr_woks.data() + (size_t)c * N, 0, 0};
        rtk[c] = LweSample{a_thr_ks.data() + (size_t)c * n, 0, 0};
    }
    {
        const int nt = tcount < 8 ? tcount : 8, upto = tcount < 64 ? tcount : 64;
        std::vector<std::thread> pool;
        for (int w = 0; w < nt; w++)
            pool.emplace_back([&, w]() {
                for (int c = w; c < upto; c += nt) {
                    tfhe_bootstrap_woKS_FFT(&rtu[c], &bk, mu, &x[c]);
                    lweKeySwitch(&rtk[c], &ksk, &rtu[c]);
                }
            });
        for (auto &th : pool) th.join();
        for (int c = 0; c < upto; c++)
            if (rtk[c].b != rl[c].b || memcmp(rtk[c].a, rl[c].a, 4 * (size_t)n) != 0) {
                fprintf(stderr, "coalesced woKS + key switch differs from the loop at %d\n", c);
                return 3;
            }
    }
    bool same = a_arr == a_loop && a_ks == a_loop;
    // the array forms over SEVERAL devices: the same loop cut into contiguous slices, one pool member per device
    const std::vector<int> devs = env_devices();
    bool pool_same = true;
    double t_pool = 0;
    if (devs.size() > 1) {
        std::vector<Torus32> p_arr((size_t)count * n), p_woks((size_t)count * N), p_ks((size_t)count * n);
        std::vector<LweSample> pa((size_t)count), pu((size_t)count), pk((size_t)count);
        std::vector<LweSample *> pap((size_t)count), pup((size_t)count), pkp((size_t)count);
        std::vector<const LweSample *> puc((size_t)count);
        for (int c = 0; c < count; c++) {
            pa[c] = LweSample{p_arr.data() + (size_t)c * n, 0, 0};
            pu[c] = LweSample{p_woks.data() + (size_t)c * N, 0, 0};
            pk[c] = LweSample{p_ks.data() + (size_t)c * n, 0, 0};
            pap[c] = &pa[c];
            pup[c] = &pu[c];
            puc[c] = &pu[c];
            pkp[c] = &pk[c];
        }
#ifdef DROPIN
        tfhe_amd_dropin_set_devices(devs.data(), (int)devs.size());
#else
        set_devices(devs.data(), (int)devs.size());
#endif
        tfhe_bootstrap_FFT_array(nullptr, &bk, mu, nullptr, 0);
        tfhe_bootstrap_FFT_array(pap.data(), &bk, mu, xp.data(), count);  // builds the pool: one key upload per device
        const double tp = now_s();
        tfhe_bootstrap_FFT_array(pap.data(), &bk, mu, xp.data(), count);
        t_pool = now_s() - tp;
        tfhe_bootstrap_woKS_FFT_array(pup.data(), &bk, mu, xp.data(), count);
        lweKeySwitch_array(pkp.data(), &ksk, puc.data(), count);
        pool_same = p_arr == a_loop && p_ks == a_loop && p_woks == a_woks;
        for (int c = 0; c < count; c++) pool_same = pool_same && pa[c].b == rl[c].b && pk[c].b == rl[c].b && pu[c].b == ru[c].b;
        // a one-sample call still works (on devices[0]) while several devices are named
        LweSample one{a_thr.data(), 0, 0};
        tfhe_bootstrap_FFT(&one, &bk, mu, &x[0]);
        pool_same = pool_same && one.b == rl[0].b && memcmp(one.a, rl[0].a, 4 * (size_t)n) == 0;
        const int zero = 0;
#ifdef DROPIN
        tfhe_amd_dropin_set_devices(&zero, 1);
#else
        set_devices(&zero, 1);
#endif
        same = same && pool_same;
    }
    for (int c = 0; c < tcount; c++)
        same = same && rt[c].b == rl[c].b && memcmp(rt[c].a, rl[c].a, 4 * (size_t)n) == 0;
    for (int c = 0; c < count; c++) same = same && ra[c].b == rl[c].b && rk[c].b == rl[c].b;
    Out out;
    for (int c = 0; c < count; c++) {
        out.put(ra[c].a, 4 * (size_t)n);
        out.put(&ra[c].b, 4);
    }
    out.save(outp);
    printf("{\"count\": %d, \"array_seconds\": %.6f, \"array_bootstraps_per_s\": %.1f, \"loop_seconds\": %.6f, "
           "\"loop_bootstraps_per_s\": %.1f, \"array_identical_to_loop\": %s, \"pool_devices\": %d, \"pool_seconds\": %.6f, "
           "\"pool_identical_to_loop\": %s, \"parallel_for_threads\": %d, \"parallel_for_seconds\": %.6f, "
           "\"parallel_for_bootstraps_per_s\": %.1f, \"parallel_for_launches\": %lu, \"parallel_for_mean_launch\": %.1f, \"parallel_for_leader_waits\": %lu, \"parallel_for_waits_run_out\": %lu, \"openmp\": %s}\n",
           count, t_arr, count / t_arr, t_loop, count / t_loop, same ? "true" : "false", (int)devs.size(), t_pool,
           devs.size() > 1 ? (pool_same ? "true" : "false") : "null", nthreads, t_par, count / t_par, co_b1 - co_b0,
           co_b1 > co_b0 ? (double)(co_q1 - co_q0) / (double)(co_b1 - co_b0) : 0.0, co_w1 - co_w0, co_t1 - co_t0,
#ifdef _OPENMP
           "true"
#else
           "false"
#endif
    );
#ifdef DROPIN
    tfhe_amd_dropin_release(nullptr);
#else
    release_all();
#endif
    return same ? 0 : 3;
}

#ifndef DROPIN
// ---------------------------------------------------------------- PoC form
// mirrors of the member names of CB/poc_types.h (the real header works the same way)
struct PLwe32 { int32_t *a; int32_t *b; };
struct PLwe64 { int64_t *a; int64_t *b; };
struct PPoly32 { int32_t *coefs; };
struct PTLwe32 { PPoly32 *a; PPoly32 *b; };
struct PTGsw32 { PTLwe32 **samples; PTLwe32 *allsamples; };
struct PLag { double *values; };
struct PTLweFFT { PLag *a; PLag *b; };
struct PTGswFFT { PTLweFFT **samples; PTLweFFT *allsamples; };
struct PGlobals {
    int n_lvl0, n_lvl1, n_lvl2, bgbit_lvl1, ell_lvl1, bgbit_lvl2, ell_lvl2;
    int kslength_lvl10, ksbasebit_lvl10, kslength_lvl21, ksbasebit_lvl21;
    PLwe32 ***preKS;
    PTGswFFT *bkFFT;
    PTLwe32 ****privKS;
};

static int run_poc(const char *inp, const char *outp) {
    auto blob = slurp(inp);
    In in{blob.data()};
    const int32_t *h = in.take<int32_t>(12);  // n0 N1 N2 l1 bg1 l2 bg2 t10 bb10 t21 bb21 count
    PGlobals g;
    g.n_lvl0 = h[0]; g.n_lvl1 = h[1]; g.n_lvl2 = h[2]; g.ell_lvl1 = h[3]; g.bgbit_lvl1 = h[4];
    g.ell_lvl2 = h[5]; g.bgbit_lvl2 = h[6]; g.kslength_lvl10 = h[7]; g.ksbasebit_lvl10 = h[8];
    g.kslength_lvl21 = h[9]; g.ksbasebit_lvl21 = h[10];
    const int count = h[11];
    const int n0 = h[0], N1 = h[1], N2 = h[2], l1 = h[3], l2 = h[5], t10 = h[7], b10 = 1 << h[8], t21 = h[9], b21 = 1 << h[10];
    const int32_t *preks = in.take<int32_t>((size_t)N1 * t10 * b10 * (n0 + 1));
    const double *bk = in.take<double>((size_t)n0 * 2 * l2 * 2 * N2);
    const int32_t *priv = in.take<int32_t>((size_t)2 * (N2 + 1) * t21 * b21 * 2 * N1);
    const int32_t *xs = in.take<int32_t>((size_t)count * (N1 + 1));
    const int64_t *x64 = in.take<int64_t>((size_t)count * (N2 + 1));
    // pointer structures
    std::vector<PLwe32> pk0((size_t)N1 * t10 * b10);
    std::vector<PLwe32 *> pk1((size_t)N1 * t10);
    std::vector<PLwe32 **> pk2((size_t)N1);
    for (size_t e = 0; e < pk0.size(); e++) { pk0[e].a = (int32_t *)preks + e * (n0 + 1); pk0[e].b = pk0[e].a + n0; }
    for (size_t e = 0; e < pk1.size(); e++) pk1[e] = &pk0[e * b10];
    for (int i = 0; i < N1; i++) pk2[i] = &pk1[(size_t)i * t10];
    g.preKS = pk2.data();
    std::vector<PLag> lag((size_t)n0 * 2 * l2 * 2);
    std::vector<PTLweFFT> frow((size_t)n0 * 2 * l2);
    std::vector<PTGswFFT> fg((size_t)n0);
    for (size_t e = 0; e < lag.size(); e++) lag[e].values = (double *)bk + e * N2;
    for (size_t e = 0; e < frow.size(); e++) { frow[e].a = &lag[e * 2]; frow[e].b = frow[e].a + 1; }
    for (int i = 0; i < n0; i++) { fg[i].allsamples = &frow[(size_t)i * 2 * l2]; fg[i].samples = nullptr; }
    g.bkFFT = fg.data();
    const size_t nrow = (size_t)2 * (N2 + 1) * t21 * b21;
    std::vector<PPoly32> pp(nrow * 2);
    std::vector<PTLwe32> pr(nrow);
    for (size_t e = 0; e < pp.size(); e++) pp[e].coefs = (int32_t *)priv + e * N1;
    for (size_t e = 0; e < nrow; e++) { pr[e].a = &pp[e * 2]; pr[e].b = pr[e].a + 1; }
    std::vector<PTLwe32 *> pr1((size_t)2 * (N2 + 1) * t21);
    std::vector<PTLwe32 **> pr2((size_t)2 * (N2 + 1));
    std::vector<PTLwe32 ***> pr3(2);
    for (size_t e = 0; e < pr1.size(); e++) pr1[e] = &pr[e * b21];
    for (size_t e = 0; e < pr2.size(); e++) pr2[e] = &pr1[e * t21];
    for (int u = 0; u < 2; u++) pr3[u] = &pr2[(size_t)u * (N2 + 1)];
    g.privKS = pr3.data();

    PocEngine<PGlobals> eng(&g);
    Out out;
    std::vector<int32_t> all_res;  // the one-sample tfhe_CircuitBootstrapFFT results, for the array form below
    for (int c = 0; c < count; c++) {
        PLwe32 x{(int32_t *)xs + (size_t)c * (N1 + 1), nullptr};
        // preKeySwitch + preModSwitch
        std::vector<int32_t> pre((size_t)n0 + 1);
        std::vector<int> abar((size_t)n0 + 1);
        PLwe32 pres{pre.data(), pre.data() + n0};
        eng.preKeySwitch(&pres, &x);
        eng.preModSwitch(abar.data(), &pres);
        out.put(pre.data(), 4 * pre.size());
        out.put(abar.data(), 4 * abar.size());
        // circuitBootstrapWoKS + circuitPrivKS
        std::vector<int64_t> boot((size_t)N2 + 1);
        PLwe64 bs{boot.data(), boot.data() + N2};
        eng.circuitBootstrapWoKS(&bs, (int64_t)1 << 56, abar.data());
        out.put(boot.data(), 8 * boot.size());
        PLwe64 xin{(int64_t *)x64 + (size_t)c * (N2 + 1), nullptr};
        std::vector<int32_t> tl((size_t)2 * N1);
        PPoly32 tp[2] = {{tl.data()}, {tl.data() + N1}};
        PTLwe32 tls{tp, tp + 1};
        eng.circuitPrivKS(&tls, 1, &xin);
        out.put(tl.data(), 4 * tl.size());
        // the whole circuit bootstrap
        std::vector<int32_t> res((size_t)2 * l1 * 2 * N1);
        std::vector<PPoly32> rp((size_t)2 * l1 * 2);
        std::vector<PTLwe32> rr((size_t)2 * l1);
        std::vector<PTLwe32 *> rs(2);
        for (size_t e = 0; e < rp.size(); e++) rp[e].coefs = res.data() + e * N1;
        for (size_t e = 0; e < rr.size(); e++) { rr[e].a = &rp[e * 2]; rr[e].b = rr[e].a + 1; }
        for (int u = 0; u < 2; u++) rs[u] = &rr[(size_t)u * l1];
        PTGsw32 tg{rs.data(), rr.data()};
        eng.tfhe_CircuitBootstrapFFT(&tg, &x);
        out.put(res.data(), 4 * res.size());
        all_res.insert(all_res.end(), res.begin(), res.end());
        // CMux (stub at poc:877-879) with the circuit bootstrap's TGSW32 output as the selector and
        // two of its own TLWE rows as data
        std::vector<int32_t> mx((size_t)2 * N1);
        PPoly32 mp[2] = {{mx.data()}, {mx.data() + N1}};
        PTLwe32 mux{mp, mp + 1};
        eng.CMux(&mux, &tg, &rr[0], &rr[2 * l1 - 1]);
        out.put(mx.data(), 4 * mx.size());
    }
    out.save(outp);
    // the driver loop as ONE call (tfhe_CircuitBootstrapFFT_array), on one device and -- with TFHE_COMPAT_DEVICES -- over a pool
    const size_t rout = (size_t)2 * l1 * 2 * N1;
    std::vector<int32_t> arr_res(rout * count);
    std::vector<PPoly32> ap((size_t)count * 2 * l1 * 2);
    std::vector<PTLwe32> ar((size_t)count * 2 * l1);
    std::vector<PTLwe32 *> as((size_t)count * 2);
    std::vector<PTGsw32> ag((size_t)count);
    std::vector<PTGsw32 *> agp((size_t)count);
    std::vector<PLwe32> ax((size_t)count);
    std::vector<const PLwe32 *> axp((size_t)count);
    for (size_t e = 0; e < ap.size(); e++) ap[e].coefs = arr_res.data() + e * N1;
    for (size_t e = 0; e < ar.size(); e++) { ar[e].a = &ap[e * 2]; ar[e].b = ar[e].a + 1; }
    for (size_t e = 0; e < as.size(); e++) as[e] = &ar[e * l1];
    for (int c = 0; c < count; c++) {
        ag[c] = PTGsw32{&as[(size_t)c * 2], &ar[(size_t)c * 2 * l1]};
        agp[c] = &ag[c];
        ax[c] = PLwe32{(int32_t *)xs + (size_t)c * (N1 + 1), nullptr};
        axp[c] = &ax[c];
    }
    eng.tfhe_CircuitBootstrapFFT_array(agp.data(), axp.data(), count);
    bool same = arr_res == all_res;
    // the one-sample call from several host threads at once: coalesced into array launches, same results
    {
        std::fill(arr_res.begin(), arr_res.end(), 0);
        std::vector<std::thread> pool;
        for (int c = 0; c < count; c++) pool.emplace_back([&, c]() { eng.tfhe_CircuitBootstrapFFT(agp[c], axp[c]); });
        for (auto &th : pool) th.join();
        same = same && arr_res == all_res;
    }
    const std::vector<int> devs = env_devices();
    if (devs.size() > 1) {
        std::fill(arr_res.begin(), arr_res.end(), 0);
        set_devices(devs.data(), (int)devs.size());
        eng.tfhe_CircuitBootstrapFFT_array(agp.data(), axp.data(), count);
        const int zero = 0;
        set_devices(&zero, 1);
        same = same && arr_res == all_res;
    }
    printf("{\"count\": %d, \"array_identical_to_loop\": %s, \"pool_devices\": %d}\n", count, same ? "true" : "false", (int)devs.size());
    return same ? 0 : 3;
}

#endif

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s lib|poc in.bin out.bin\n", argv[0]);
        return 2;
    }
    if (std::string(argv[1]) == "arr") return run_arr(argv[2], argv[3]);
#ifdef DROPIN
    return run_lib(argv[2], argv[3]);
#else
    return std::string(argv[1]) == "lib" ? run_lib(argv[2], argv[3]) : run_poc(argv[2], argv[3]);
#endif
}
