// ubench_acc_roundtrip.hip -- what would it cost the Torus64 blind rotation (k_blind_rotate<long,11,4,1>: one wave per
// ciphertext, one wave per SIMD, 32 KB accumulator in 128 registers) to keep its accumulator in global memory between
// CMux steps instead?  (docs/experiments.md, round 4.  Diagnostic tool, not part of the product.)
//
// 1024 waves (256 workgroups x 4 waves, one workgroup per CU pinned by LDS), each owning one 32 KB accumulator of a
// 32 MB buffer; `steps` dependent steps, each streaming a 256 KB key row (shared by all waves, as the kernel does:
// 256 coalesced 16-byte loads per lane) under `fill` dependent fp64 FMAs per key load, then updating the accumulator:
//   mode 0  accumulator in registers (what the kernel does today)
//   mode 1  step = load 32 KB .. work .. load 32 KB again, add, store 32 KB, wait for the stores
//   mode 2  step = load 32 KB .. work .. 32 KB of no-return 64-bit atomic adds, wait for them
// Prints microseconds per step per mode (HIP events / steps), modes interleaved over `rounds` rounds.
//   hipcc --offload-arch=gfx950 -O3 ubench_acc_roundtrip.hip -o ubench_acc_roundtrip
//   ./ubench_acc_roundtrip [steps=500] [fill=12] [rounds=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k_steps(u64 *acc, const double2 *key, int steps, int fill, double *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ct = blockIdx.x * 4 + wave;
    u64x2 *mine = reinterpret_cast<u64x2 *>(acc + (size_t)ct * 4096);
    u64x2 r[32];
    if (MODE == 0) {
#pragma unroll
        for (int m = 0; m < 32; m++) r[m] = mine[m * 64 + lane];
    }
    double f[8];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = 1.0 + lane * 1e-9 + k;
    for (int s = 0; s < steps; s++) {
        const double2 *row = key + (size_t)s * 16384;
        u64 mix = 0;
        if (MODE != 0) {
#pragma unroll
            for (int m = 0; m < 32; m++) r[m] = mine[m * 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < 32; m++) mix ^= r[m].x ^ r[m].y;  // the step's work depends on the whole accumulator
        f[0] += (double)(mix & 1);
#pragma unroll 8
        for (int k = 0; k < 256; k++) {
            const double2 kv = row[k * 64 + lane];
            double x = f[k & 7];
            for (int e = 0; e < fill; e++) x = __builtin_fma(x, 0.9999999, 1e-9);
            f[k & 7] = __builtin_fma(x, kv.x, kv.y);
        }
        const u64 delta = (u64)(long long)(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]);
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 32; m++) {
                r[m].x += delta;
                r[m].y += delta + m;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < 32; m++) {
                u64x2 o = mine[m * 64 + lane];
                o.x += delta;
                o.y += delta + m;
                mine[m * 64 + lane] = o;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u64 *p = reinterpret_cast<u64 *>(mine);
#pragma unroll
            for (int m = 0; m < 32; m++) {
                __hip_atomic_fetch_add(&p[(m * 64 + lane) * 2], delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&p[(m * 64 + lane) * 2 + 1], delta + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int m = 0; m < 32; m++) mine[m * 64 + lane] = r[m];
    }
    if (f[0] == 12345.678) sink[0] = f[0];
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 500;
    const int fill = argc > 2 ? atoi(argv[2]) : 12;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    u64 *acc;
    double2 *key;
    double *sink;
    const size_t acc_bytes = (size_t)1024 * 32768, key_bytes = (size_t)steps * 262144;
    CHECK(hipMalloc(&acc, acc_bytes));
    CHECK(hipMalloc(&key, key_bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(acc, 1, acc_bytes));
    {
        std::vector<double> h(key_bytes / 8);
        for (size_t i = 0; i < h.size(); i++) h[i] = 1.0 + 1e-12 * (double)(i % 1000);
        CHECK(hipMemcpy(key, h.data(), key_bytes, hipMemcpyHostToDevice));
    }
    const size_t lds = 100 * 1024;  // one workgroup per CU
    CHECK(hipFuncSetAttribute((const void *)k_steps<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k_steps<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k_steps<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> t[3];
    for (int r = 0; r <= rounds; r++)
        for (int mode = 0; mode < 3; mode++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_steps<0>, dim3(256), dim3(256), lds, 0, acc, key, steps, fill, sink);
            if (mode == 1) hipLaunchKernelGGL(k_steps<1>, dim3(256), dim3(256), lds, 0, acc, key, steps, fill, sink);
            if (mode == 2) hipLaunchKernelGGL(k_steps<2>, dim3(256), dim3(256), lds, 0, acc, key, steps, fill, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r) t[mode].push_back(ms);
        }
    const char *names[3] = {"registers", "reload + store", "atomic add"};
    for (int mode = 0; mode < 3; mode++) {
        std::sort(t[mode].begin(), t[mode].end());
        printf("mode %d (%-14s)  steps %d fill %d: median %.3f ms per launch = %.2f us per step (min %.2f)\n", mode, names[mode], steps, fill,
               t[mode][t[mode].size() / 2], t[mode][t[mode].size() / 2] * 1e3 / steps, t[mode][0] * 1e3 / steps);
    }
    return 0;
}
