// pool_tsan_test.cpp -- experimental-tfhe_amd/csrc/pool.cpp linked against tests/compat/pool_mock_engine.cpp: several host
// threads call one pool of four members at once (sharded host calls of ragged sizes, the pipelined form with a small chunk, the
// callback form, key reloads and option changes in between); every output is checked against the mock's row formulas.  Built with
// -fsanitize=thread by tests/test_coalescer.py: the pool's worker hand-off, call lock and per-member state under ThreadSanitizer.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "tfhe_amd.h"

extern "C" int pool_mock_fail_key_load_on_device;  // tests/compat/pool_mock_engine.cpp: key loads fail on this device

static const int n = 6, N = 20, MU = 77;
static int32_t expect(int32_t x0, const int32_t *row, int j) { (void)x0; return (int32_t)((3u * (uint32_t)row[j % (n + 1)] + (uint32_t)j + (uint32_t)MU)) ^ 0x5a5a5a5a; }
struct Rows {
    const std::vector<int32_t> *x;
    std::vector<int32_t> *out;
};
static void get(void *u, int first, int rows, int32_t *dst) {
    const Rows *r = (const Rows *)u;
    memcpy(dst, r->x->data() + (size_t)first * (n + 1), sizeof(int32_t) * (size_t)rows * (n + 1));
}
static void put(void *u, int first, int rows, const int32_t *src) {
    Rows *r = (Rows *)u;
    memcpy(r->out->data() + (size_t)first * (n + 1), src, sizeof(int32_t) * (size_t)rows * (n + 1));
}
int main() {
    tfhe_amd_params p = {32, n, N, 1, 2, 10, 8, 2, n};
    const int devices[4] = {0, 1, 1, 5};
    tfhe_amd_pool *pool = nullptr;
    if (tfhe_amd_pool_create(&p, devices, 4, &pool)) return 2;
    int32_t x1 = 0;
    if (tfhe_amd_pool_bootstrap_host(pool, &x1, MU, &x1, 0) != TFHE_AMD_OK) return 2;      // empty call
    std::vector<int32_t> probe((size_t)(n + 1)), pout((size_t)(n + 1));
    if (tfhe_amd_pool_bootstrap_host(pool, pout.data(), MU, probe.data(), 1) != TFHE_AMD_ERR_STATE) return 2;  // no keys yet
    double bk = 0;
    int32_t ks = 0;
    if (tfhe_amd_pool_load_keys(pool, &bk, &ks)) return 2;
    tfhe_amd_pool_set_option(pool, TFHE_AMD_POOL_OPT_CHUNK_ROWS, 3);
    int bad = 0;
    auto work = [&](int t) {
        for (int rep = 0; rep < 40; rep++) {
            const int count = 1 + (t * 13 + rep * 7) % 61;
            std::vector<int32_t> x((size_t)count * (n + 1)), out((size_t)count * (n + 1), -1);
            for (size_t i = 0; i < x.size(); i++) x[i] = (int32_t)(i * 2654435761u + (unsigned)t);
            int rc;
            if (rep % 3 == 0) {
                Rows r = {&x, &out};
                rc = tfhe_amd_pool_bootstrap_rows(pool, put, get, &r, MU, count);
            } else {
                rc = tfhe_amd_pool_bootstrap_host(pool, out.data(), MU, x.data(), count);
            }
            if (rc) { __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED); continue; }
            for (int r = 0; r < count; r++)
                for (int j = 0; j <= n; j++)
                    if (out[(size_t)r * (n + 1) + j] != expect(0, &x[(size_t)r * (n + 1)], j)) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            if (rep % 10 == 9) tfhe_amd_pool_set_option(pool, TFHE_AMD_POOL_OPT_CHUNK_ROWS, 1 + (rep + t) % 5);
            if (rep % 17 == 16) tfhe_amd_pool_load_keys(pool, &bk, &ks);
            int counts[4];
            double secs[4];
            tfhe_amd_pool_last_split(pool, counts, secs);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < 6; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    // a key load that fails on ONE member (device 5 = member 3): reported with the member and its device, and the pool then
    // refuses to run -- its members hold different keys -- until a load has succeeded everywhere
    {
        pool_mock_fail_key_load_on_device = 5;
        if (tfhe_amd_pool_load_keys(pool, &bk, &ks) != TFHE_AMD_ERR_DEVICE) bad++;
        if (!strstr(tfhe_amd_pool_last_error(pool), "member 3 (device 5)")) bad++;
        std::vector<int32_t> x((size_t)8 * (n + 1), 1), out((size_t)8 * (n + 1), -1);
        if (tfhe_amd_pool_bootstrap_host(pool, out.data(), MU, x.data(), 8) != TFHE_AMD_ERR_STATE) bad++;
        if (!strstr(tfhe_amd_pool_last_error(pool), "different keys")) bad++;
        if (out[0] != -1) bad++;  // nothing ran
        if (tfhe_amd_pool_load_keys(pool, nullptr, &ks) != TFHE_AMD_OK) bad++;  // the other key loads, the failed one is still owed
        if (tfhe_amd_pool_bootstrap_host(pool, out.data(), MU, x.data(), 8) != TFHE_AMD_ERR_STATE) bad++;
        // ... while the operation that reads only the key every member now holds does run (each checks the keys IT uses)
        std::vector<int32_t> kin((size_t)8 * (N + 1), 1), kout((size_t)8 * (n + 1), -1);
        if (tfhe_amd_pool_keyswitch_host(pool, kout.data(), kin.data(), 8) != TFHE_AMD_OK) bad++;
        if (tfhe_amd_pool_bootstrap_woks_host(pool, kin.data(), MU, x.data(), 8) != TFHE_AMD_ERR_STATE) bad++;  // (uses the mixed one)
        pool_mock_fail_key_load_on_device = -1;
        if (tfhe_amd_pool_load_keys(pool, &bk, nullptr) != TFHE_AMD_OK) bad++;
        if (tfhe_amd_pool_bootstrap_host(pool, out.data(), MU, x.data(), 8) != TFHE_AMD_OK) bad++;
        for (int j = 0; j <= n; j++)
            if (out[(size_t)7 * (n + 1) + j] != expect(0, &x[(size_t)7 * (n + 1)], j)) bad++;
    }
    tfhe_amd_pool_destroy(pool);
    printf("pool_tsan_test: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}
