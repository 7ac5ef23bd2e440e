// devport.h -- the few device idioms the kernels use, named once.
//
// The shipped library is built by hipcc for gfx950 only.  The same kernel source can
// also be compiled by tests/emu/ (TFHE_EMU defined) against a tiny fiber-based
// emulation of a HIP workgroup, so that kernel logic can be checked against the
// oracle -- and run under ASan/UBSan -- on a machine without a GPU.  That build lives
// under tests/ and is never loaded by the product.
#pragma once

#ifdef TFHE_EMU
#include "emu_runtime.h"  // tests/emu/emu_runtime.h
#else
#include <hip/hip_runtime.h>

#define TFHE_DEVICE __device__ __forceinline__
#define TFHE_GLOBAL __global__
#define TFHE_HOST_DEVICE __host__ __device__ __forceinline__

// LDS hand-off between lanes of ONE wave.  A wave's DS instructions execute in issue
// order, so nothing is emitted; the fences only stop the compiler from moving a lane's
// LDS reads above another lane's (program-earlier) writes.
#define TFHE_WAVE_FENCE()                                         \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

#define TFHE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// the SIMD (0..3) this wave runs on: HW_REG_HW_ID bits [5:4]
#define TFHE_SIMD_ID() ((int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3))
// issue priority of this wave among the waves of its SIMD (0 lowest .. 3), s_setprio
#define TFHE_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
// signed bit-field extract: bits [off, off + width) of x, sign-extended (v_bfe_i32; off, width may be scalar registers)
#define TFHE_SBFE(x, off, width) ((int32_t)__builtin_amdgcn_sbfe((int32_t)(x), (uint32_t)(off), (uint32_t)(width)))
// low 32 bits of ((hi:lo) >> sh), 0 <= sh < 32 (v_alignbit_b32)
#define TFHE_ALIGNBIT(hi, lo, sh) __builtin_amdgcn_alignbit((hi), (lo), (uint32_t)(sh))
// next `size` instructions of the classes in `mask` (LLVM sched_group_barrier: 0x2 VALU, 0x8 MFMA, 0x100 DS read, ...)
#define TFHE_SCHED_GROUP(mask, size) __builtin_amdgcn_sched_group_barrier((mask), (size), 0)
// no instruction is scheduled across this point
#define TFHE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// nothing moves across this point: neither the IR optimiser's memory operations (empty asm with a memory clobber)
// nor the machine scheduler's instructions.  Used where the ORDER of independent loads decides the register
// pressure (hipcc otherwise hoists every load of a phase to its top).
#define TFHE_ORDER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
// true in every lane iff `cond` holds in at least one active lane of the wave
#define TFHE_WAVE_ANY(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)
// placed at the top of a wave-uniform `if` body: keeps it a real scalar branch (hipcc otherwise
// if-converts short bodies into per-lane selects, which costs VALU work on the skipped paths)
#define TFHE_KEEP_BRANCH() asm volatile("" ::: "memory")
// makes a register value opaque at this point: arithmetic on it cannot be hoisted above (used to
// keep a rarely taken fallback from being computed speculatively on the hot path)
#define TFHE_OPAQUE(x) asm volatile("" : "+v"(x))
// the same for a wave-uniform value held in scalar registers
#define TFHE_OPAQUE_SCALAR(x) asm volatile("" : "+s"(x))
// *p += v on an LDS word owned by this lane, as ONE DS instruction with no result (ds_add_u32 /
// ds_add_u64): no read, no VALU add, no wait.  Ordered with the wave's other DS operations.
#define TFHE_LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
// Three-operand integer forms the compiler does not select by itself here (it re-associates the
// source expression into longer sequences); each is ONE VALU instruction.
//   tfhe_and_or(x, m, o)  = (x & m) | o          v_and_or_b32   (m wave-uniform)
//   tfhe_sign_mask(x)     = bit BIT of x as 0/-1 v_bfe_i32
//   tfhe_xad(a, b, c)     = (a ^ b) + c          v_xad_u32
__device__ __forceinline__ uint32_t tfhe_and_or(uint32_t x, uint32_t m, uint32_t o) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(m), "v"(o));
    return r;
}
template <int BIT>
__device__ __forceinline__ uint32_t tfhe_sign_mask(uint32_t x) {
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(x), "n"(BIT));
    return r;
}
__device__ __forceinline__ uint32_t tfhe_xad(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// byte offset of an LDS object inside the workgroup's LDS segment (generic -> LDS address-space cast)
__device__ __forceinline__ uint32_t tfhe_lds_offset(const void *p) {
    typedef __attribute__((address_space(3))) const unsigned char lds_byte;
    return (uint32_t)(uintptr_t)(lds_byte *)p;
}
// 32-bit load from LDS byte offset `off` (as returned by tfhe_lds_offset, plus arithmetic)
__device__ __forceinline__ uint32_t tfhe_lds_load32(const void *, uint32_t off) {
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    return *(lds_u32 *)(uintptr_t)off;
}
// p[idx] for a wave-uniform address of data no kernel writes while this one runs: a scalar load
// (constant address space), the result lives in a scalar register
__device__ __forceinline__ int32_t tfhe_uniform_load32(const int32_t *p, int idx) {
    typedef __attribute__((address_space(4))) const int32_t const_i32;
    return ((const_i32 *)(uintptr_t)p)[idx];
}
// 32-bit LDS word shared between waves of the workgroup (progress counters): plain DS instructions on an
// LDS byte offset, re-read on every call (a generic `volatile int *` compiles to FLAT accesses here)
__device__ __forceinline__ uint32_t tfhe_lds_peek32(uint32_t off) {
    typedef __attribute__((address_space(3))) volatile const uint32_t lds_vu32;
    return *(lds_vu32 *)(uintptr_t)off;
}
__device__ __forceinline__ void tfhe_lds_poke32(uint32_t off, uint32_t v) {
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
    *(lds_vu32 *)(uintptr_t)off = v;
}
__device__ __forceinline__ double tfhe_uniform_load_f64(const double *p, int idx) {
    typedef __attribute__((address_space(4))) const double const_f64;
    return ((const_f64 *)(uintptr_t)p)[idx];
}
// Buffer addressing for rows of read-only data: descriptor (base, no stride, no bounds in practice) in scalar
// registers; a load = descriptor + 32-bit lane offset + scalar offset (4 KB window) + 12-bit immediate.
typedef __amdgpu_buffer_rsrc_t TFHE_BUFFER_RSRC;
#define TFHE_MAKE_BUFFER_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, 0x7fffffff, 0x00020000)
__device__ __forceinline__ double2 tfhe_buffer_load_d2(TFHE_BUFFER_RSRC rsrc, uint32_t lane_off, uint32_t off) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off + (off & 4095u)), (int)(off & ~4095u), 0);
    return __builtin_bit_cast(double2, v);
}
// loads / stores of data the launch touches once and that should not displace cached data (the nt cache policy)
template <typename V>
__device__ __forceinline__ V tfhe_nontemporal_load(const V *p) { return __builtin_nontemporal_load(p); }
template <typename V>
__device__ __forceinline__ void tfhe_nontemporal_store(V v, V *p) { __builtin_nontemporal_store(v, p); }
// 32-bit load / store through a pointer KNOWN to be global memory (a generic pointer compiles to flat_* instructions, which
// count against the LDS wait counter too)
__device__ __forceinline__ uint32_t tfhe_global_load32(const void *p, int idx) {
    typedef __attribute__((address_space(1))) const uint32_t glob_u32;
    return ((glob_u32 *)(uintptr_t)p)[idx];
}
__device__ __forceinline__ void tfhe_global_store32(void *p, int idx, uint32_t v) {
    typedef __attribute__((address_space(1))) uint32_t glob_u32;
    ((glob_u32 *)(uintptr_t)p)[idx] = v;
}
#define TFHE_TRAP() __builtin_trap()
// counters of the clock probe: shader cycles (s_memtime), the constant 100 MHz reference (s_memrealtime); a short sleep
#define TFHE_SHADER_CYCLES() __builtin_amdgcn_s_memtime()
#define TFHE_REF_TICKS() __builtin_amdgcn_s_memrealtime()
#define TFHE_SLEEP() __builtin_amdgcn_s_sleep(64)
// the workgroup's dynamic LDS block
#define TFHE_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// D = A(32x32 int8) * B(32x32 int8) + C(32x32 int32) on the matrix cores, one wave.  Lane l holds
// A[row l & 31][k = 16 * (l >> 5) + 0..15] and B[k = 16 * (l >> 5) + 0..15][col l & 31] as 16 bytes each;
// C/D: column l & 31, row (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
#define TFHE_MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)
// value of `v` held by lane `lane` (wave-uniform lane index) -> scalar register
#define TFHE_READLANE(v, lane) __builtin_amdgcn_readlane((v), (lane))
#define TFHE_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
// same launch; the name tells the tests/emu build the kernel never synchronises work-items
#define TFHE_LAUNCH_FLAT(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#endif

// instrumentation hooks of k_blind_rotate: nothing in the shipped library; the probe build of tools/wave_probe.py
// (-DTFHE_PROBE, never shipped) fills them from probe_hooks.h
#if defined(TFHE_PROBE) && !defined(TFHE_EMU)
#include "probe_hooks.h"
#else
#define TFHE_PROBE_KERNEL_BEGIN() ((void)0)
#define TFHE_PROBE_LOOP_BEGIN() ((void)0)
#define TFHE_PROBE_LOOP_END(wave, t) ((void)0)
#define TFHE_PROBE_KERNEL_END(t) ((void)0)
#endif

#include <stdint.h>
