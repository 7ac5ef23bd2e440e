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
// true in every lane iff `cond` holds in at least one active lane of the wave
#define TFHE_WAVE_ANY(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)
// placed at the top of a wave-uniform `if` body: keeps it a real scalar branch (hipcc otherwise
// if-converts short bodies into per-lane selects, which costs VALU work on the skipped paths)
#define TFHE_KEEP_BRANCH() asm volatile("" ::: "memory")
// makes a register value opaque at this point: arithmetic on it cannot be hoisted above (used to
// keep a rarely taken fallback from being computed speculatively on the hot path)
#define TFHE_OPAQUE(x) asm volatile("" : "+v"(x))
// value of `v` held by lane `lane` (wave-uniform lane index) -> scalar register
#define TFHE_READLANE(v, lane) __builtin_amdgcn_readlane((v), (lane))
#define TFHE_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
// same launch; the name tells the tests/emu build the kernel never synchronises work-items
#define TFHE_LAUNCH_FLAT(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#endif

#include <stdint.h>
