// seam_threads.cpp -- several host threads call the spqlios seam library (include/tfhe_amd_spqlios.h) at once:
// fftp1024.execute_reverse_int -> execute_direct_torus32 and LagrangeHalfCPolynomialAddMulASM on per-thread inputs;
// every result must equal the one the same calls give from a single thread.  (The reference's class is not re-entrant --
// its objects transform in their own scratch buffers -- but its AddMul is, and the PoC's drivers may be threaded around
// it; the seam serialises its calls per ring degree: csrc/spqlios_seam.cpp.)
//   g++ -std=c++11 -O2 -Iinclude seam_threads.cpp -L<dir> -l:libtfhe_amd_spqlios[_emu].so -Wl,-rpath,<dir> -lpthread
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "tfhe_amd_spqlios.h"

static const int N = 1024;
struct Work {
    std::vector<int> a;
    std::vector<double> lag, b, acc;
    std::vector<int32_t> back;
};
static void fill(Work &w, uint64_t seed) {
    w.a.resize(N); w.lag.resize(N); w.b.resize(N); w.acc.assign(N, 0.25); w.back.resize(N);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (int i = 0; i < N; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        w.a[i] = (int)(s % 1024) - 512;
        w.b[i] = (double)((int64_t)(s >> 20) % 4096) / 64.0;
    }
}
static void run(Work &w, int reps) {
    for (int r = 0; r < reps; r++) {
        fftp1024.execute_reverse_int(w.lag.data(), w.a.data());
        LagrangeHalfCPolynomialAddMulASM(w.acc.data(), w.lag.data(), w.b.data(), N / 2);
        std::vector<double> scaled(w.acc);
        fftp1024.execute_direct_torus32(w.back.data(), scaled.data());
    }
}
int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 4, reps = argc > 2 ? atoi(argv[2]) : 3;
    std::vector<Work> serial(threads), par(threads);
    for (int t = 0; t < threads; t++) { fill(serial[t], 100 + t); fill(par[t], 100 + t); }
    for (int t = 0; t < threads; t++) run(serial[t], reps);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([&, t]() { run(par[t], reps); });
    for (auto &th : pool) th.join();
    for (int t = 0; t < threads; t++) {
        if (memcmp(serial[t].lag.data(), par[t].lag.data(), 8 * N) || memcmp(serial[t].acc.data(), par[t].acc.data(), 8 * N) ||
            memcmp(serial[t].back.data(), par[t].back.data(), 4 * N)) {
            printf("thread %d: results differ from the serial run\n", t);
            return 1;
        }
    }
    printf("seam_threads ok: %d threads x %d rounds identical to the serial run\n", threads, reps);
    return 0;
}
