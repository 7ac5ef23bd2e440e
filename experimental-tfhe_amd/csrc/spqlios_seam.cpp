// spqlios_seam.cpp -- libtfhe_amd_spqlios.so: the reference's FFT plugin symbols (include/tfhe_amd_spqlios.h) as
// forwarders onto the engine's C ABI (include/tfhe_amd.h).  Host C++ only (g++); links against libtfhe_amd.so.
//
//   FFT_Processor_Spqlios::execute_*      CB/spqlios/fft_processor_spqlios.cpp:27-170  -> tfhe_amd_ifft_* / tfhe_amd_fft_*
//   LagrangeHalfCPolynomialAddMulASM      CB/spqlios/lagrangehalfc_impl_fma.s:78-135   -> tfhe_amd_lagrange_addmul
//   new_*_table / *_get_buffer / fft / ifft / *_model   CB/spqlios/spqlios-fft-impl.cpp:158-203,400-447 and the
//                                         assembly cores spqlios-{i,}fft-fma.s          -> tfhe_amd_{i,}fft_f64
// One polynomial per call, synchronous (copy in, one kernel, copy out), like the functions replaced.
// Calls from several host threads are serialised per ring degree (one staging area per engine): the reference's class has
// the same restriction by construction (its objects transform in their own scratch buffers, fft_processor_spqlios.cpp:21-24),
// its AddMul is re-entrant -- here both are safe to call concurrently, one at a time inside.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "tfhe_amd.h"
#include "tfhe_amd_spqlios.h"

namespace {

// one engine per ring degree, shared by the class objects, AddMul and the C core; created at first use
struct Engine {
    tfhe_amd_ctx *ctx;
    int N;
    void *in_d, *out_d, *aux_d;  // one-polynomial staging, N doubles each
    std::mutex mu;               // one call at a time on the staging area
};

[[noreturn]] void die(const char *what, int rc, tfhe_amd_ctx *c) {
    std::fprintf(stderr, "tfhe_amd_spqlios: %s failed (%d): %s\n", what, rc, c ? tfhe_amd_last_error(c) : "");
    std::abort();
}
void chk(int rc, Engine *e, const char *what) {
    if (rc != TFHE_AMD_OK) die(what, rc, e ? e->ctx : nullptr);
}

// every ring degree the reference's plugin accepts: a power of two >= 16 (require() in new_fft_table / new_ifft_table,
// spqlios-fft-impl.cpp:157-160,400-403); like the reference, anything else aborts with a message
bool served(int N) { return N >= 16 && N <= (1 << 20) && (N & (N - 1)) == 0; }
Engine *engine(int N) {
    static Engine *slots[32] = {nullptr};  // by log2(N)
    static std::mutex create_mu;
    if (!served(N)) {
        std::fprintf(stderr, "tfhe_amd_spqlios: ring degree %d is not served (n must be a power of 2, 16 <= n <= 2^20)\n", N);
        std::abort();
    }
    std::lock_guard<std::mutex> guard(create_mu);
    Engine *&e = slots[__builtin_ctz((unsigned)N)];
    if (e) return e;
    const char *dev = std::getenv("TFHE_AMD_DEVICE");
    tfhe_amd_params p;
    std::memset(&p, 0, sizeof(p));
    p.torus_bits = 64;  // the width only selects gadget defaults; every conversion is available
    p.n = 1;
    p.N = N;
    p.k = 1;
    p.l = 1;
    p.Bgbit = 1;
    Engine *n = new Engine();
    n->N = N;
    const int rc = tfhe_amd_ctx_create(&p, dev ? std::atoi(dev) : 0, &n->ctx);
    if (rc != TFHE_AMD_OK) die("tfhe_amd_ctx_create (no usable GPU? the engine has no CPU path)", rc, nullptr);
    chk(tfhe_amd_malloc(n->ctx, &n->in_d, 8 * (size_t)N), n, "malloc");
    chk(tfhe_amd_malloc(n->ctx, &n->out_d, 8 * (size_t)N), n, "malloc");
    chk(tfhe_amd_malloc(n->ctx, &n->aux_d, 8 * (size_t)N), n, "malloc");
    e = n;
    return e;
}

// copy `in_bytes` in, run one batch-1 call, copy `out_bytes` out
template <class Call>
void roundtrip(Engine *e, void *res, size_t out_bytes, const void *src, size_t in_bytes, Call call, const char *what) {
    std::lock_guard<std::mutex> guard(e->mu);
    chk(tfhe_amd_memcpy_h2d(e->ctx, e->in_d, src, in_bytes), e, "h2d");
    chk(call(e), e, what);
    chk(tfhe_amd_memcpy_d2h(e->ctx, res, e->out_d, out_bytes), e, "d2h");
}

// the table object of the C core: the first four fields are the reference's FFT_PRECOMP / IFFT_PRECOMP
// (spqlios-fft-impl.cpp:48-60: n = 2 * ring degree, the trig table, the data buffer, the allocation)
struct Table {
    uint64_t n;
    double *trig;
    double *data;
    void *buf;
    int inverse;
};
Table *new_table(int nn, int inverse) {
    if (!served(nn)) {  // the reference: require(nn >= 16), require(power of 2) -> abort
        std::fprintf(stderr, "tfhe_amd_spqlios: new_%sfft_table(%d): n must be a power of 2, 16 <= n <= 2^20\n", inverse ? "i" : "", nn);
        std::abort();
    }
    Table *t = new Table();
    t->n = 2 * (uint64_t)nn;
    const size_t trig_len = (size_t)2 * nn - 8;  // 4 * (nn / 2) - 8 doubles
    t->buf = std::malloc(32 + (trig_len + (size_t)nn) * 8 + 32);
    const uintptr_t base = ((uintptr_t)t->buf + 31) & ~(uintptr_t)31;
    t->trig = (double *)base;
    t->data = (double *)((base + trig_len * 8 + 31) & ~(uintptr_t)31);
    t->inverse = inverse;
    // host libm only: no context, no device
    if (tfhe_amd_build_tables(nn, inverse ? nullptr : t->trig, inverse ? t->trig : nullptr) != TFHE_AMD_OK) die("tfhe_amd_build_tables", 1, nullptr);
    return t;
}
void core(const Table *t, double *data) {
    Engine *e = engine((int)(t->n / 2));
    const size_t bytes = 8 * (size_t)e->N;
    if (t->inverse)
        roundtrip(e, data, bytes, data, bytes,
                  [](Engine *g) { return tfhe_amd_ifft_f64(g->ctx, (double *)g->out_d, (const double *)g->in_d, 1); }, "ifft");
    else
        roundtrip(e, data, bytes, data, bytes,
                  [](Engine *g) { return tfhe_amd_fft_f64(g->ctx, (double *)g->out_d, (const double *)g->in_d, 1); }, "fft");
}

}  // namespace

// ---- class FFT_Processor_Spqlios (lagrangehalfc_impl.h:8-31)
FFT_Processor_Spqlios::FFT_Processor_Spqlios(const int N_)
    : _2N(2 * N_), N(N_), Ns2(N_ / 2), real_inout_direct(nullptr), imag_inout_direct(nullptr), real_inout_rev(nullptr),
      imag_inout_rev(nullptr), tables_direct(nullptr), tables_reverse(nullptr) {}  // no GPU work here: see the header
FFT_Processor_Spqlios::~FFT_Processor_Spqlios() {}  // the engines live as long as the process (the reference never frees its tables either)

#define SEAM_ENGINE() engine(N)  // created at the first call (under a lock), then a table look-up

void FFT_Processor_Spqlios::execute_reverse_int(double *res, const int *a) {
    roundtrip(SEAM_ENGINE(), res, 8 * (size_t)N, a, 4 * (size_t)N,
              [](Engine *g) { return tfhe_amd_ifft_int32(g->ctx, (double *)g->out_d, (const int32_t *)g->in_d, 1); }, "execute_reverse_int");
}
void FFT_Processor_Spqlios::execute_reverse_torus32(double *res, const int32_t *a) { execute_reverse_int(res, (const int *)a); }
void FFT_Processor_Spqlios::execute_direct_torus32(int32_t *res, const double *a) {
    roundtrip(SEAM_ENGINE(), res, 4 * (size_t)N, a, 8 * (size_t)N,
              [](Engine *g) { return tfhe_amd_fft_torus32(g->ctx, (int32_t *)g->out_d, (const double *)g->in_d, 1); }, "execute_direct_torus32");
}
void FFT_Processor_Spqlios::execute_reverse_torus64(double *res, const int64_t *a) {
    roundtrip(SEAM_ENGINE(), res, 8 * (size_t)N, a, 8 * (size_t)N,
              [](Engine *g) { return tfhe_amd_ifft_torus64(g->ctx, (double *)g->out_d, (const int64_t *)g->in_d, 1); }, "execute_reverse_torus64");
}
void FFT_Processor_Spqlios::execute_direct_torus64(int64_t *res, const double *a) {
    roundtrip(SEAM_ENGINE(), res, 8 * (size_t)N, a, 8 * (size_t)N,
              [](Engine *g) { return tfhe_amd_fft_torus64(g->ctx, (int64_t *)g->out_d, (const double *)g->in_d, 1); }, "execute_direct_torus64");
}

FFT_Processor_Spqlios fftp1024(1024);
FFT_Processor_Spqlios fftp2048(2048);

// ---- LagrangeHalfCPolynomialAddMulASM (lagrangehalfc_impl.h:36): res += a * b, Ns2 complex values
extern "C" void LagrangeHalfCPolynomialAddMulASM(double *res, double *a, double *b, long Ns2) {
    Engine *e = engine((int)(2 * Ns2));
    const size_t bytes = 8 * (size_t)e->N;
    std::lock_guard<std::mutex> guard(e->mu);
    chk(tfhe_amd_memcpy_h2d(e->ctx, e->out_d, res, bytes), e, "h2d");
    chk(tfhe_amd_memcpy_h2d(e->ctx, e->in_d, a, bytes), e, "h2d");
    chk(tfhe_amd_memcpy_h2d(e->ctx, e->aux_d, b, bytes), e, "h2d");
    chk(tfhe_amd_lagrange_addmul(e->ctx, (double *)e->out_d, (const double *)e->in_d, (const double *)e->aux_d, 1, 0), e, "addmul");
    chk(tfhe_amd_memcpy_d2h(e->ctx, res, e->out_d, bytes), e, "d2h");
}

// ---- C core (spqlios-fft.h:46-53)
extern "C" {
void *new_fft_table(int nn) { return new_table(nn, 0); }
void *new_ifft_table(int nn) { return new_table(nn, 1); }
double *fft_table_get_buffer(const void *tables) { return ((const Table *)tables)->data; }
double *ifft_table_get_buffer(const void *tables) { return ((const Table *)tables)->data; }
void fft(const void *tables, double *data) { core((const Table *)tables, data); }
void ifft(const void *tables, double *data) { core((const Table *)tables, data); }
// the reference's scalar models transform the table's own buffer and are specified to equal the assembly cores
void fft_model(const void *tables) { core((const Table *)tables, ((const Table *)tables)->data); }
void ifft_model(void *tables) { core((const Table *)tables, ((const Table *)tables)->data); }
}
