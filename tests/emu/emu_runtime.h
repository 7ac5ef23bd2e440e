// emu_runtime.h -- TEST-ONLY emulation of the slice of HIP the kernels and the host
// library use, so the very same sources can be compiled by g++ and checked against the
// oracle (and run under sanitizers) where no GPU exists.  Nothing under
// experimental-tfhe_amd/ ever loads a library built from this; see tests/emu/README.md.
//
// Model: a workgroup is a set of cooperative fibers (ucontext) on one OS thread, one
// fiber per work-item, scheduled round-robin.  __syncthreads() and the wave-level LDS
// fence are rendezvous points; between two rendezvous a work-item runs alone, which is
// a legal interleaving of the lock-step hardware.  Workgroups of one launch are spread
// over a few OS threads.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

namespace emu {
struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local Dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
// (while a stream capture is open on the calling thread, launches are recorded into the graph instead of run)
void launch(const std::function<void()> &body, Dim3 grid, Dim3 block, size_t smem_bytes);
// kernels that never synchronise (no barrier, no LDS hand-off): work-items run as a plain loop
void launch_flat(const std::function<void()> &body, Dim3 grid, Dim3 block);
void syncthreads();
void wave_fence();
unsigned char *dyn_smem();
int readlane(int v, int lane);
bool wave_any(bool cond);
}  // namespace emu

using dim3 = emu::Dim3;
#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)
#define __syncthreads() emu::syncthreads()
#define __launch_bounds__(...)

#define TFHE_DEVICE inline
#define TFHE_GLOBAL
#define TFHE_HOST_DEVICE inline
#define TFHE_WAVE_FENCE() emu::wave_fence()
#define TFHE_UNIFORM(x) (x)
#define TFHE_SCHED_BARRIER() ((void)0)
#define TFHE_ORDER() ((void)0)
#define TFHE_SCHED_GROUP(mask, size) ((void)0)
static inline int32_t tfhe_emu_sbfe(uint32_t x, int off, int width) {
    return (int32_t)((uint32_t)(x >> off) << (32 - width)) >> (32 - width);
}
#define TFHE_SBFE(x, off, width) tfhe_emu_sbfe((uint32_t)(x), (int)(off), (int)(width))
#define TFHE_ALIGNBIT(hi, lo, sh) ((uint32_t)((((uint64_t)(hi) << 32) | (uint32_t)(lo)) >> (sh)))
#define TFHE_SIMD_ID() ((int)((threadIdx.x >> 6) & 3))
#define TFHE_SETPRIO(p) ((void)0)
#define TFHE_WAVE_ANY(cond) emu::wave_any(cond)
#define TFHE_KEEP_BRANCH() ((void)0)
#define TFHE_OPAQUE(x) ((void)0)
#define TFHE_OPAQUE_SCALAR(x) ((void)0)
#define TFHE_READLANE(v, lane) emu::readlane((v), (lane))
#define TFHE_LDS_ADD(p, v) ((void)(*(p) += (v)))
static inline uint32_t tfhe_and_or(uint32_t x, uint32_t m, uint32_t o) { return (x & m) | o; }
template <int BIT>
static inline uint32_t tfhe_sign_mask(uint32_t x) { return (uint32_t)0 - ((x >> BIT) & 1u); }
static inline uint32_t tfhe_xad(uint32_t a, uint32_t b, uint32_t c) { return (a ^ b) + c; }
// "LDS offsets" of the emulation are offsets from the workgroup's dynamic block
static inline uint32_t tfhe_lds_offset(const void *p) { return (uint32_t)((const unsigned char *)p - emu::dyn_smem()); }
static inline uint32_t tfhe_lds_load32(const void *, uint32_t off) {
    uint32_t v;
    memcpy(&v, emu::dyn_smem() + off, 4);
    return v;
}
static inline int32_t tfhe_uniform_load32(const int32_t *p, int idx) { return p[idx]; }
static inline uint32_t tfhe_lds_peek32(uint32_t off) { return *(volatile const uint32_t *)(emu::dyn_smem() + off); }
static inline void tfhe_lds_poke32(uint32_t off, uint32_t v) { *(volatile uint32_t *)(emu::dyn_smem() + off) = v; }
static inline double tfhe_uniform_load_f64(const double *p, int idx) { return p[idx]; }
template <typename V>
static inline V tfhe_nontemporal_load(const V *p) { return *p; }
template <typename V>
static inline void tfhe_nontemporal_store(V v, V *p) { *p = v; }
static inline uint32_t tfhe_global_load32(const void *p, int idx) { return ((const uint32_t *)p)[idx]; }
static inline void tfhe_global_store32(void *p, int idx, uint32_t v) { ((uint32_t *)p)[idx] = v; }
#define TFHE_TRAP() abort()
// clock probe: the reference counter is the wall clock at 100 MHz, the "shader" counter 24 cycles per tick (a nominal 2.4 GHz)
unsigned long long emu_ref_ticks();
#define TFHE_REF_TICKS() emu_ref_ticks()
#define TFHE_SHADER_CYCLES() (24ull * emu_ref_ticks())
#define TFHE_SLEEP() ((void)0)
namespace emu {
typedef int v4i_t __attribute__((vector_size(16)));
typedef int v16i_t __attribute__((vector_size(64)));
v16i_t mfma_i32_32x32x32_i8(const v4i_t &a, const v4i_t &b, const v16i_t &c);  // wave-collective, same operand maps as the hardware's
}
#define TFHE_MFMA_I8(a, b, c) emu::mfma_i32_32x32x32_i8((a), (b), (c))
// Several emulated devices (TFHE_EMU_DEVICES=N): every launch is checked before it runs -- the stream, the kernel's
// dynamic-LDS attribute and every device pointer among the arguments must belong to the calling thread's CURRENT
// device, or the process aborts with a message naming the operand (see "devices" below).
#define TFHE_LAUNCH(kernel, grid, block, smem, stream, ...)                                                    \
    (emu::check_launch(emu::fn_key(kernel), (size_t)(smem), (stream), #kernel), emu::check_args(#kernel, __VA_ARGS__), \
     emu::launch([=]() { kernel(__VA_ARGS__); }, grid, block, smem))
#define TFHE_LAUNCH_FLAT(kernel, grid, block, stream, ...)                                           \
    (emu::check_launch(nullptr, 0, (stream), #kernel), emu::check_args(#kernel, __VA_ARGS__), \
     emu::launch_flat([=]() { kernel(__VA_ARGS__); }, grid, block))

// workgroups of one launch run on several OS threads: global-memory atomics must be real ones
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

struct double2 {
    double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
typedef const unsigned char *TFHE_BUFFER_RSRC;
#define TFHE_MAKE_BUFFER_RSRC(ptr) ((const unsigned char *)(ptr))
static inline double2 tfhe_buffer_load_d2(TFHE_BUFFER_RSRC rsrc, uint32_t lane_off, uint32_t off) {
    return *(const double2 *)(rsrc + lane_off + off);
}

// ---- host runtime subset -------------------------------------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
#define TFHE_DYN_LDS(name) unsigned char *name = emu::dyn_smem()
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorInvalidDevice = 101 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }

// ---- devices ---------------------------------------------------------------------------------------------------
// TFHE_EMU_DEVICES=N (default 1) emulated devices.  The current device is per host thread, as in HIP.  Every allocation,
// stream, event, captured graph and per-kernel LDS attribute is tagged with the device that was current when it was made;
// a launch, copy, memset, free, event record or graph launch whose operands belong to ANOTHER device than the calling
// thread's current one ABORTS the process (message on stderr).  Stricter than HIP in one place on purpose: hipFree and
// hipStreamDestroy / hipEventDestroy of another device's object abort too (HIP accepts them), so that every path of the
// host library is seen to select its device first.
namespace emu {
int device_count();
int current_device();
struct Stream { int device; };
void check_launch(const void *fn, size_t smem_bytes, hipStream_t stream, const char *kernel);  // aborts on a foreign stream / missing LDS attribute
void check_words(const char *kernel, const void *obj, size_t bytes);  // aborts if any 8-byte word of obj points into another device's memory
template <class F>
static inline const void *fn_key(F f) { return reinterpret_cast<const void *>(f); }
static inline void check_args(const char *) {}
template <class A, class... Rest>
static inline void check_args(const char *kernel, const A &a, const Rest &...rest) {
    check_words(kernel, &a, sizeof(A));
    check_args(kernel, rest...);
}
}  // namespace emu
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *d);
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int device);  // "0000:<e0 + device>:00.0"
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
enum { hipHostMallocDefault = 0 };
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned);  // "pinned" = plain host memory here, usable from every device
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t);  // recorded while a capture is open
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t);  // recorded while a capture is open
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// events: wall-clock stamps (the emulator runs launches synchronously)
struct EmuEvent { double t; int device; };
typedef EmuEvent *hipEvent_t;
double emu_now_ms();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t);
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);  // (launches run synchronously: only the device checks remain)
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t emu_func_set_attribute(const void *fn, int bytes);
template <class F>
static inline hipError_t hipFuncSetAttribute(F f, hipFuncAttribute, int bytes) { return emu_func_set_attribute(emu::fn_key(f), bytes); }
// streams: launches run synchronously, a stream is a token that remembers its device
enum { hipStreamNonBlocking = 1 };
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned);
hipError_t hipStreamDestroy(hipStream_t s);
// device queries: every emulated device gives persistent-wave kernels a small fixed grid (3 workgroups), so
// that the batch loop of every wave is exercised at test sizes
struct hipDeviceProp_t {
    char name[256], gcnArchName[256];
    int multiProcessorCount, clockRate, l2CacheSize, warpSize, pciDomainID, pciBusID, pciDeviceID;
    size_t totalGlobalMem, sharedMemPerBlock, sharedMemPerBlockOptin;
};
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int device);
static inline hipError_t hipRuntimeGetVersion(int *v) { *v = 0; return hipSuccess; }
static inline hipError_t hipDriverGetVersion(int *v) { *v = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int device);
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 1; return hipSuccess; }
// stream capture / graphs: a graph is the list of launches recorded between Begin and EndCapture
struct EmuGraph;
typedef EmuGraph *hipGraph_t;
typedef EmuGraph *hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeThreadLocal };
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *graph);
hipError_t hipGraphInstantiate(hipGraphExec_t *exec, hipGraph_t graph, void *, void *, unsigned long long);
hipError_t hipGraphDestroy(hipGraph_t graph);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t);
