// pool_mock_engine.cpp -- TEST ONLY: a host-memory stand-in for the part of the C ABI (include/tfhe_amd.h) that
// experimental-tfhe_amd/csrc/pool.cpp is written on, so that the pool's THREADING (one worker thread per member, the call lock, the
// pipelined member's bookkeeping) can run under ThreadSanitizer -- the kernel emulator's fibers cannot.  "Device memory" is malloc,
// copies are memcpy at the time of the call, streams and events are tokens, and the three operations are cheap row functions:
//   bootstrap_woks: out[r][j] = 3 * in[r][j % (n+1)] + j + mu      keyswitch: out[r][j] = in[r][j] ^ 0x5a5a5a5a
// tests/compat/pool_tsan_test.cpp checks the pool's outputs against the same formulas.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>

#include "tfhe_amd.h"

struct tfhe_amd_ctx {
    tfhe_amd_params p;
    int device;
    std::string err;
    bool has_bk = false, has_ks = false;
    std::atomic<int> in_call{0};  // a context is not thread-safe: two threads inside one context at once is the pool's bug
};
struct tfhe_amd_gsw {
    tfhe_amd_ctx *ctx;
};
// failure injection: key loads fail on contexts of this device (-1: never)
extern "C" int pool_mock_fail_key_load_on_device;
int pool_mock_fail_key_load_on_device = -1;
namespace {
struct Enter {
    tfhe_amd_ctx *c;
    explicit Enter(tfhe_amd_ctx *c_) : c(c_) {
        if (c->in_call.fetch_add(1) != 0) abort();
    }
    ~Enter() { c->in_call.fetch_sub(1); }
};
}  // namespace
extern "C" {
int tfhe_amd_ctx_create(const tfhe_amd_params *p, int device, tfhe_amd_ctx **out) {
    if (device < 0 || device >= 8) return TFHE_AMD_ERR_DEVICE;
    tfhe_amd_ctx *c = new tfhe_amd_ctx();
    c->p = *p;
    c->device = device;
    *out = c;
    return TFHE_AMD_OK;
}
void tfhe_amd_ctx_destroy(tfhe_amd_ctx *c) { delete c; }
const char *tfhe_amd_last_error(const tfhe_amd_ctx *c) { return c->err.c_str(); }
int tfhe_amd_sync(tfhe_amd_ctx *c) { Enter e(c); return TFHE_AMD_OK; }
int tfhe_amd_set_stream(tfhe_amd_ctx *c, void *) { Enter e(c); return TFHE_AMD_OK; }
int tfhe_amd_malloc(tfhe_amd_ctx *c, void **p, size_t n) { Enter e(c); *p = malloc(n ? n : 1); return TFHE_AMD_OK; }
int tfhe_amd_free(tfhe_amd_ctx *c, void *p) { Enter e(c); free(p); return TFHE_AMD_OK; }
int tfhe_amd_host_alloc(tfhe_amd_ctx *c, void **p, size_t n) { Enter e(c); *p = malloc(n ? n : 1); return TFHE_AMD_OK; }
int tfhe_amd_host_free(tfhe_amd_ctx *c, void *p) { Enter e(c); free(p); return TFHE_AMD_OK; }
int tfhe_amd_memcpy_h2d(tfhe_amd_ctx *c, void *d, const void *s, size_t n) { Enter e(c); memcpy(d, s, n); return TFHE_AMD_OK; }
int tfhe_amd_memcpy_d2h(tfhe_amd_ctx *c, void *d, const void *s, size_t n) { Enter e(c); memcpy(d, s, n); return TFHE_AMD_OK; }
int tfhe_amd_memcpy_h2d_async(tfhe_amd_ctx *c, void *d, const void *s, size_t n) { Enter e(c); memcpy(d, s, n); return TFHE_AMD_OK; }
int tfhe_amd_memcpy_d2h_async(tfhe_amd_ctx *c, void *d, const void *s, size_t n) { Enter e(c); memcpy(d, s, n); return TFHE_AMD_OK; }
int tfhe_amd_stream_create(tfhe_amd_ctx *c, void **s) { Enter e(c); *s = malloc(1); return TFHE_AMD_OK; }
int tfhe_amd_stream_sync(tfhe_amd_ctx *c, void *) { Enter e(c); return TFHE_AMD_OK; }
int tfhe_amd_stream_destroy(tfhe_amd_ctx *c, void *s) { Enter e(c); free(s); return TFHE_AMD_OK; }
int tfhe_amd_event_create(tfhe_amd_ctx *c, void **ev) { Enter e(c); *ev = malloc(1); return TFHE_AMD_OK; }
int tfhe_amd_event_record(tfhe_amd_ctx *c, void *) { Enter e(c); return TFHE_AMD_OK; }
int tfhe_amd_event_sync(tfhe_amd_ctx *c, void *) { Enter e(c); return TFHE_AMD_OK; }
int tfhe_amd_stream_wait_event(tfhe_amd_ctx *c, void *) { Enter e(c); return TFHE_AMD_OK; }
int tfhe_amd_event_destroy(tfhe_amd_ctx *c, void *ev) { Enter e(c); free(ev); return TFHE_AMD_OK; }
int tfhe_amd_gsw_from_fft(tfhe_amd_ctx *c, const double *, int, tfhe_amd_gsw **out) {
    Enter e(c);
    if (c->device == pool_mock_fail_key_load_on_device) { c->err = "injected key-load failure"; return TFHE_AMD_ERR_DEVICE; }
    *out = new tfhe_amd_gsw{c};
    return TFHE_AMD_OK;
}
int tfhe_amd_gsw_from_torus(tfhe_amd_ctx *c, const void *, int, tfhe_amd_gsw **out) { Enter e(c); *out = new tfhe_amd_gsw{c}; return TFHE_AMD_OK; }
void tfhe_amd_gsw_free(tfhe_amd_gsw *g) { delete g; }
int tfhe_amd_set_bootstrap_key(tfhe_amd_ctx *c, const tfhe_amd_gsw *g) { Enter e(c); c->has_bk = g != nullptr; return TFHE_AMD_OK; }
int tfhe_amd_load_keyswitch_key(tfhe_amd_ctx *c, const int32_t *) { Enter e(c); c->has_ks = true; return TFHE_AMD_OK; }
int tfhe_amd_bootstrap_woks(tfhe_amd_ctx *c, int32_t *out, int32_t mu, const int32_t *x, int batch) {
    Enter e(c);
    if (!c->has_bk) { c->err = "no bootstrapping key"; return TFHE_AMD_ERR_STATE; }
    const int n1 = c->p.n + 1, N1 = c->p.N + 1;
    for (int r = 0; r < batch; r++)
        for (int j = 0; j < N1; j++) out[(size_t)r * N1 + j] = (int32_t)(3u * (uint32_t)x[(size_t)r * n1 + j % n1] + (uint32_t)j + (uint32_t)mu);
    return TFHE_AMD_OK;
}
int tfhe_amd_keyswitch(tfhe_amd_ctx *c, int32_t *out, const int32_t *in, int batch) {
    Enter e(c);
    if (!c->has_ks) { c->err = "no key-switch key"; return TFHE_AMD_ERR_STATE; }
    const int N1 = c->p.N + 1, o1 = c->p.ks_n_out + 1;
    for (int r = 0; r < batch; r++)
        for (int j = 0; j < o1; j++) out[(size_t)r * o1 + j] = in[(size_t)r * N1 + j] ^ 0x5a5a5a5a;
    return TFHE_AMD_OK;
}
// the circuit-bootstrap handle is not exercised by the TSan test: link stubs only
struct tfhe_amd_cb { int unused; };
int tfhe_amd_cb_create(const tfhe_amd_cb_params *, int, tfhe_amd_cb **) { return TFHE_AMD_ERR_DEVICE; }
void tfhe_amd_cb_destroy(tfhe_amd_cb *) {}
const char *tfhe_amd_cb_last_error(const tfhe_amd_cb *) { return ""; }
tfhe_amd_ctx *tfhe_amd_cb_ctx_lvl2(tfhe_amd_cb *) { return nullptr; }
int tfhe_amd_cb_load_preks(tfhe_amd_cb *, const int32_t *) { return TFHE_AMD_ERR_STATE; }
int tfhe_amd_cb_load_bk_fft(tfhe_amd_cb *, const double *) { return TFHE_AMD_ERR_STATE; }
int tfhe_amd_cb_load_bk_torus(tfhe_amd_cb *, const int64_t *) { return TFHE_AMD_ERR_STATE; }
int tfhe_amd_cb_load_privks_plane(tfhe_amd_cb *, int, const int32_t *) { return TFHE_AMD_ERR_STATE; }
int tfhe_amd_circuit_bootstrap(tfhe_amd_cb *, int32_t *, const int32_t *, int) { return TFHE_AMD_ERR_STATE; }
}
