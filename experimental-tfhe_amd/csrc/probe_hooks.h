// probe_hooks.h -- PROBE BUILD ONLY (-DTFHE_PROBE, tools/wave_probe.py): what the four TFHE_PROBE_* hooks of
// k_blind_rotate expand to in that build.  The shipped library is built without TFHE_PROBE: devport.h then defines the
// hooks as nothing and this file is not even included, so the kernel source has one code path.
// Records: shader clock and in-loop lifetime of every wave, start / end / placement of every workgroup.
#pragma once
namespace tfhe {
__device__ unsigned long long tfhe_dbg[32];  // 0: sum of shader cycles, 1: sum of 100 MHz ticks, 2: waves, [16 + wave]: ticks per wave index
__device__ unsigned long long tfhe_dbg_wg[1024 * 4];  // per workgroup: start, end (100 MHz ticks), HW_ID, XCC_ID
// the stamps are moved to VECTOR registers at once: held in scalar registers across the CMux loop they are spilled by
// v_writelane in the high-pressure instantiations, which hipcc (ROCm 7.2) then rejects ("Operand has incorrect register class")
__device__ __forceinline__ unsigned long long tfhe_vec(unsigned long long c) {
    unsigned lo = (unsigned)c, hi = (unsigned)(c >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long tfhe_clk() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c = tfhe_vec(__builtin_amdgcn_s_memtime());
    __builtin_amdgcn_sched_barrier(0);
    return c;
}
}  // namespace tfhe
#define TFHE_PROBE_KERNEL_BEGIN()                                                            \
    do {                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                         \
            tfhe_dbg_wg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();              \
            tfhe_dbg_wg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     \
            tfhe_dbg_wg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    \
        }                                                                                    \
    } while (0)
#define TFHE_PROBE_LOOP_BEGIN() const unsigned long long dbg_c0 = tfhe_clk(), dbg_r0 = tfhe_vec(__builtin_amdgcn_s_memrealtime())
#define TFHE_PROBE_LOOP_END(wave, t)                                                         \
    do {                                                                                     \
        const unsigned long long c1 = tfhe_clk(), r1 = __builtin_amdgcn_s_memrealtime();     \
        if ((t) == 0) {                                                                      \
            atomicAdd(&tfhe_dbg[0], c1 - dbg_c0);                                            \
            atomicAdd(&tfhe_dbg[1], r1 - dbg_r0);                                            \
            atomicAdd(&tfhe_dbg[2], 1ull);                                                   \
            atomicAdd(&tfhe_dbg[16 + ((wave) & 7)], r1 - dbg_r0);                            \
        }                                                                                    \
    } while (0)
#define TFHE_PROBE_KERNEL_END(t)                                                             \
    do {                                                                                     \
        if ((t) == 0 && blockIdx.x < 1024)                                                   \
            atomicMax(&tfhe_dbg_wg[blockIdx.x * 4 + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); \
    } while (0)
// host side of the probe build: read and reset the counters (tools/wave_probe.py binds these two by name)
extern "C" int tfhe_amd_dbg_read(unsigned long long *out) {  // experiment build only: read and reset the phase totals
    unsigned long long z[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tfhe::tfhe_dbg), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(tfhe::tfhe_dbg), z, sizeof(z)) != hipSuccess;
}
extern "C" int tfhe_amd_dbg_read_wg(unsigned long long *out) {  // [1024][4], read and reset
    static unsigned long long z[1024 * 4];
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tfhe::tfhe_dbg_wg), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(tfhe::tfhe_dbg_wg), z, sizeof(z)) != hipSuccess;
}
