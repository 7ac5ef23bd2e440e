// emu_runtime.cpp -- TEST-ONLY fiber scheduler behind tests/emu/emu_runtime.h.
#include "emu_runtime.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <time.h>
#include <ucontext.h>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#endif

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

// Fiber switch.  swapcontext() saves and restores the signal mask with a system call on every switch -- a quarter of the CPU
// suite's time went there -- so on x86-64 the switch is 20 instructions of our own: callee-saved registers, MXCSR / x87 control
// word and the stack pointer.  The sanitizer build keeps ucontext (AddressSanitizer intercepts swapcontext to follow the stacks).
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__)
#define EMU_FAST_SWITCH 1
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch,.-emu_switch
)");
#endif

namespace emu {

thread_local Dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {
// (ASan clears the shadow of a fiber's whole stack at every swapcontext: the sanitizer build keeps it as small as the kernels allow)
#if defined(__SANITIZE_ADDRESS__)
constexpr size_t kStack = 192 * 1024;
#else
constexpr size_t kStack = 512 * 1024;
#endif

struct Rendezvous {
    int expected = 0, count = 0;
    unsigned gen = 0;
};
struct Fiber {
#ifdef EMU_FAST_SWITCH
    void *sp = nullptr;  // saved stack pointer while the fiber is not running
#else
    ucontext_t ctx;
#endif
    void *stack = nullptr;
    bool done = false;
    unsigned tid = 0;
    const Rendezvous *wait_on = nullptr;  // blocked until wait_on->gen != wait_gen: the scheduler skips it
    unsigned wait_gen = 0;
};
struct Block {
    std::vector<int> lane_xchg;  // readlane: two slots per work-item, used alternately (one fence per call)
    std::vector<unsigned char> rl_phase;
    std::vector<int> any_xchg;   // wave_any
    std::vector<int> mfma_xchg;  // mfma: two alternating sets of 8 ints (A then B fragment) per work-item
    std::vector<unsigned char> mfma_phase;
    std::vector<Fiber> fibers;
    Rendezvous all;
    std::vector<Rendezvous> waves;
#ifdef EMU_FAST_SWITCH
    void *sched_sp = nullptr;
#else
    ucontext_t sched;
#endif
    const std::function<void()> *body = nullptr;
    unsigned char *smem = nullptr;
    int current = -1;
};
thread_local Block *t_blk = nullptr;

// Fiber stacks are reused across blocks and launches (a pool shared by the worker threads): mapping and unmapping 256-512
// stacks per emulated workgroup, and faulting their pages in again, was ~40 % of the CPU suite's time (and nearly all of a
// sanitizer run's).  Under AddressSanitizer a reused stack is unpoisoned first, so that no redzone poisoning left by an earlier
// fiber (whose frames never unwound past the trampoline) can outlive it.
#if defined(__SANITIZE_ADDRESS__)
#define EMU_UNPOISON_STACK(p, n) ASAN_UNPOISON_MEMORY_REGION((p), (n))
#else
#define EMU_UNPOISON_STACK(p, n) ((void)0)
#endif
constexpr bool kPoolStacks = true;
std::mutex g_stack_mu;
std::vector<void *> g_stack_pool;
void *stack_acquire() {
    if (kPoolStacks) {
        std::lock_guard<std::mutex> lk(g_stack_mu);
        if (!g_stack_pool.empty()) {
            void *p = g_stack_pool.back();
            g_stack_pool.pop_back();
            EMU_UNPOISON_STACK(p, kStack);
            return p;
        }
    }
    void *p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); abort(); }
    return p;
}
void stack_release(void *p) {
    if (kPoolStacks) {
        std::lock_guard<std::mutex> lk(g_stack_mu);
        if (g_stack_pool.size() < 4096) {  // 8 worker threads x 512 work-items: bounds the mapped (mostly untouched) memory
            g_stack_pool.push_back(p);
            return;
        }
    }
    munmap(p, kStack);
}

#ifdef EMU_FAST_SWITCH
void yield() { emu_switch(&t_blk->fibers[t_blk->current].sp, t_blk->sched_sp); }
#else
void yield() { swapcontext(&t_blk->fibers[t_blk->current].ctx, &t_blk->sched); }
#endif

void arrive(Rendezvous &r) {
    const unsigned g = r.gen;
    if (++r.count >= r.expected) {
        r.count = 0;
        r.gen++;
        return;
    }
    Fiber &f = t_blk->fibers[t_blk->current];
    f.wait_on = &r;
    f.wait_gen = g;
    while (r.gen == g) yield();
    f.wait_on = nullptr;
}
void retire(Rendezvous &r) {  // a work-item that returned no longer takes part
    r.expected--;
    if (r.count > 0 && r.count >= r.expected) {
        r.count = 0;
        r.gen++;
    }
}
void trampoline() {
    Block *b = t_blk;
    Fiber &f = b->fibers[b->current];
    (*b->body)();
    f.done = true;
    retire(b->all);
    retire(b->waves[f.tid / 64]);
#ifdef EMU_FAST_SWITCH
    emu_switch(&f.sp, b->sched_sp);  // never resumed
    abort();
#else
    swapcontext(&f.ctx, &b->sched);
#endif
}

void run_block(const std::function<void()> &body, Dim3 grid, Dim3 block, unsigned bx, size_t smem_bytes) {
    Block b;
    t_blk = &b;
    b.body = &body;
    const unsigned nt = block.x;
    b.fibers.resize(nt);
    b.lane_xchg.assign((size_t)2 * nt, 0);
    b.rl_phase.assign(nt, 0);
    b.any_xchg.assign(nt, 0);
    b.mfma_xchg.assign((size_t)16 * nt, 0);
    b.mfma_phase.assign(nt, 0);
    b.all.expected = (int)nt;
    b.waves.resize((nt + 63) / 64);
    for (unsigned w = 0; w < b.waves.size(); w++) b.waves[w].expected = (int)std::min(64u, nt - 64 * w);
    // exact size (rounded to the 16-byte LDS granule) so ASan sees any out-of-bounds LDS index
    unsigned char *smem_raw = (unsigned char *)malloc(((smem_bytes + 15) / 16) * 16 + 16);
    b.smem = smem_raw;
    b.smem += 16 - ((uintptr_t)b.smem & 15);  // 16-byte aligned like the hardware carve-out
    t_blockIdx = Dim3(bx % grid.x, bx / grid.x);  // bx is the flattened (x, y) block index
    t_blockDim = block;
    t_gridDim = grid;
    for (unsigned t = 0; t < nt; t++) {
        Fiber &f = b.fibers[t];
        f.tid = t;
        f.stack = stack_acquire();
#ifdef EMU_FAST_SWITCH
        // the frame emu_switch pops: [MXCSR | x87 CW] r15 r14 r13 r12 rbx rbp, then `ret` into trampoline with the stack as
        // after a call (return-address slot 16-byte aligned; above it a null "caller")
        void **sp = reinterpret_cast<void **>(static_cast<char *>(f.stack) + kStack);
        *--sp = nullptr;
        *--sp = reinterpret_cast<void *>(&trampoline);
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        unsigned csr[2] = {0, 0};
        asm volatile("stmxcsr %0\n\tfnstcw %1" : "=m"(csr[0]), "=m"(csr[1]));
        uint64_t word = (uint64_t)csr[0] | ((uint64_t)(csr[1] & 0xffffu) << 32);
        *--sp = reinterpret_cast<void *>(word);
        f.sp = sp;
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
#endif
    }
    unsigned alive = nt;
    while (alive) {
        alive = 0;
        for (unsigned t = 0; t < nt; t++) {
            Fiber &f = b.fibers[t];
            if (f.done) continue;
            if (f.wait_on && f.wait_on->gen == f.wait_gen) {  // still blocked: no context switch
                alive++;
                continue;
            }
            b.current = (int)t;
            t_threadIdx = Dim3(t);
#ifdef EMU_FAST_SWITCH
            emu_switch(&b.sched_sp, f.sp);
#else
            swapcontext(&b.sched, &f.ctx);
#endif
            if (!f.done) alive++;
        }
    }
    for (auto &f : b.fibers) stack_release(f.stack);
    free(smem_raw);
    t_blk = nullptr;
}
}  // namespace

void syncthreads() { arrive(t_blk->all); }
void wave_fence() { arrive(t_blk->waves[t_threadIdx.x / 64]); }
unsigned char *dyn_smem() { return t_blk->smem; }
bool wave_any(bool cond) {
    const unsigned tid = t_threadIdx.x, base = tid & ~63u;
    t_blk->any_xchg[tid] = cond ? 1 : 0;
    wave_fence();
    bool any = false;
    const unsigned end = std::min<unsigned>(base + 64, (unsigned)t_blk->any_xchg.size());
    for (unsigned i = base; i < end; i++) any |= (t_blk->fibers[i].done ? false : t_blk->any_xchg[i] != 0);
    wave_fence();
    return any;
}
// Two slot arrays used alternately: a lane can reach call k+2 (same slots as call k) only after the
// fence of call k+1, which every lane passes only once it has finished reading the slots of call k.
int readlane(int v, int lane) {
    const unsigned tid = t_threadIdx.x;
    const size_t nt = t_blk->rl_phase.size(), buf = (t_blk->rl_phase[tid]++ & 1u) * nt;
    t_blk->lane_xchg[buf + tid] = v;
    wave_fence();
    return t_blk->lane_xchg[buf + (tid & ~63u) + (unsigned)lane];
}

// 16-element int8 dot product (the inner loop of the matrix instruction below: 2.4 M wave-level calls per emulated private key
// switch, 32 k multiply-adds each): AVX2 where the host has it (checked at run time), the plain loop otherwise
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static int dot16_avx2(const signed char *a, const signed char *b) {
    const __m256i va = _mm256_cvtepi8_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i *>(a)));
    const __m256i vb = _mm256_cvtepi8_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i *>(b)));
    const __m256i p = _mm256_madd_epi16(va, vb);  // 8 x int32: no intermediate overflow (|a b| <= 2^14)
    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(p), _mm256_extracti128_si256(p, 1));
    s = _mm_hadd_epi32(s, s);
    s = _mm_hadd_epi32(s, s);
    return _mm_cvtsi128_si32(s);
}
static const bool g_have_avx2 = __builtin_cpu_supports("avx2");
#else
static const bool g_have_avx2 = false;
static int dot16_avx2(const signed char *, const signed char *) { return 0; }
#endif
static inline int dot16(const signed char *a, const signed char *b) {
    if (g_have_avx2) return dot16_avx2(a, b);
    int sum = 0;
    for (int j = 0; j < 16; j++) sum += (int)a[j] * (int)b[j];
    return sum;
}

// v_mfma_i32_32x32x32_i8: every lane publishes its A and B fragments, then computes its 16 results
// (column lane & 31, rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).  Two slot sets used alternately,
// one fence per call (same argument as readlane).
v16i_t mfma_i32_32x32x32_i8(const v4i_t &a, const v4i_t &b, const v16i_t &c_in) {
    v16i_t c = c_in;
    const unsigned tid = t_threadIdx.x, base = tid & ~63u, lane = tid & 63u;
    const size_t nt = t_blk->mfma_phase.size(), set = (t_blk->mfma_phase[tid]++ & 1u) * 8 * nt;
    int *slot = &t_blk->mfma_xchg[set + (size_t)8 * tid];
    for (int e = 0; e < 4; e++) {
        slot[e] = a[e];
        slot[4 + e] = b[e];
    }
    wave_fence();
    const unsigned col = lane & 31u;
    for (int e = 0; e < 16; e++) {
        const unsigned row = (unsigned)((e & 3) + 8 * (e >> 2)) + 4u * (lane >> 5);
        int sum = 0;
        for (unsigned kh = 0; kh < 2; kh++) {
            const signed char *ap = reinterpret_cast<const signed char *>(&t_blk->mfma_xchg[set + (size_t)8 * (base + row + 32 * kh)]);
            const signed char *bp = reinterpret_cast<const signed char *>(&t_blk->mfma_xchg[set + (size_t)8 * (base + col + 32 * kh) + 4]);
            sum += dot16(ap, bp);
        }
        c[e] += sum;
    }
    return c;
}

}  // namespace emu
struct EmuGraph {
    std::vector<std::function<void()>> nodes;
    int device = 0;  // the device of the stream the capture was opened on
};
namespace emu {
static thread_local EmuGraph *t_capture = nullptr;  // hipStreamCaptureModeThreadLocal

static void launch_now(const std::function<void()> &body, Dim3 grid, Dim3 block, size_t smem_bytes);
void launch(const std::function<void()> &body, Dim3 grid, Dim3 block, size_t smem_bytes) {
    if (t_capture) {
        t_capture->nodes.push_back([body, grid, block, smem_bytes]() { launch_now(body, grid, block, smem_bytes); });
        return;
    }
    launch_now(body, grid, block, smem_bytes);
}
static void launch_now(const std::function<void()> &body, Dim3 grid, Dim3 block, size_t smem_bytes) {
    const unsigned nb = grid.x * grid.y;
    unsigned nthreads = std::min<unsigned>(nb, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("TFHE_EMU_THREADS")) nthreads = std::max(1, atoi(e));
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        for (;;) {
            const unsigned bx = next.fetch_add(1);
            if (bx >= nb) break;
            run_block(body, grid, block, bx, smem_bytes);
        }
    };
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < nthreads; i++) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
}

static void launch_flat_now(const std::function<void()> &body, Dim3 grid, Dim3 block);
void launch_flat(const std::function<void()> &body, Dim3 grid, Dim3 block) {
    if (t_capture) {
        t_capture->nodes.push_back([body, grid, block]() { launch_flat_now(body, grid, block); });
        return;
    }
    launch_flat_now(body, grid, block);
}
static void launch_flat_now(const std::function<void()> &body, Dim3 grid, Dim3 block) {
    const unsigned nb = grid.x;
    unsigned nthreads = std::min<unsigned>(std::max(1u, nb / 64), std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("TFHE_EMU_THREADS")) nthreads = std::max(1, atoi(e));
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        t_blockDim = block;
        t_gridDim = grid;
        for (;;) {
            const unsigned b0 = next.fetch_add(64);
            if (b0 >= nb) break;
            for (unsigned bx = b0; bx < std::min(nb, b0 + 64); bx++) {
                t_blockIdx = Dim3(bx);
                for (unsigned t = 0; t < block.x; t++) {
                    t_threadIdx = Dim3(t);
                    body();
                }
            }
        }
    };
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < nthreads; i++) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
}

}  // namespace emu

// ---- devices: registry of allocations / streams / attributes, and the cross-device checks ----------------------
namespace emu {
namespace {
struct Alloc {
    size_t bytes;
    int device;
    bool host;  // hipHostMalloc: visible from every device
};
std::mutex g_dev_mu;
std::map<uintptr_t, Alloc> g_allocs;                    // base address -> allocation
std::map<std::pair<const void *, int>, int> g_lds_attr;  // (kernel, device) -> allowed dynamic LDS bytes
thread_local int t_device = 0;

[[noreturn]] void die(const char *what, const char *detail, int other_device) {
    fprintf(stderr, "emu: %s: %s belongs to device %d, the calling thread's current device is %d (missing hipSetDevice?)\n",
            what, detail, other_device, t_device);
    abort();
}
// the allocation p points into (base <= p < base + bytes), or nullptr: host memory the emulator does not know
const Alloc *find_alloc(const void *p) {
    const uintptr_t a = (uintptr_t)p;
    auto it = g_allocs.upper_bound(a);
    if (it == g_allocs.begin()) return nullptr;
    --it;
    return a < it->first + it->second.bytes ? &it->second : nullptr;
}
// a pointer that must be device memory of the current device (must_be_device) or, if it is device memory at all, of the current device
void check_ptr(const void *p, bool must_be_device, const char *what, const char *role) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    const Alloc *a = find_alloc(p);
    if (!a || a->host) {
        if (must_be_device && !a) {
            fprintf(stderr, "emu: %s: %s %p is not device memory\n", what, role, p);
            abort();
        }
        return;
    }
    if (a->device != t_device) die(what, role, a->device);
}
int stream_device(hipStream_t s) { return s ? static_cast<Stream *>(s)->device : t_device; }
void check_stream(hipStream_t s, const char *what) {
    if (stream_device(s) != t_device) die(what, "the stream", stream_device(s));
}
void register_alloc(void *p, size_t bytes, bool host) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    g_allocs[(uintptr_t)p] = Alloc{bytes ? bytes : 1, t_device, host};
}
void unregister_alloc(void *p, bool host, const char *what) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_allocs.find((uintptr_t)p);
    if (it == g_allocs.end() || it->second.host != host) {
        fprintf(stderr, "emu: %s(%p): not a live allocation of this kind (double free?)\n", what, p);
        abort();
    }
    if (!host && it->second.device != t_device) die(what, "the allocation", it->second.device);
    g_allocs.erase(it);
}
}  // namespace

int device_count() {
    static const int n = [] {
        const char *e = getenv("TFHE_EMU_DEVICES");
        const int v = e ? atoi(e) : 1;
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}
int current_device() { return t_device; }

void check_launch(const void *fn, size_t smem_bytes, hipStream_t stream, const char *kernel) {
    check_stream(stream, kernel);
    if (fn && smem_bytes > 64 * 1024) {  // above 64 KiB the limit is raised per kernel AND per device
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto it = g_lds_attr.find(std::make_pair(fn, t_device));
        if (it == g_lds_attr.end() || (size_t)it->second < smem_bytes) {
            fprintf(stderr, "emu: %s: %zu bytes of dynamic LDS without hipFuncSetAttribute(MaxDynamicSharedMemorySize) on device %d\n",
                    kernel, smem_bytes, t_device);
            abort();
        }
    }
}
// kernel arguments are plain structs and pointers: any aligned 8-byte word of their object representation that points
// into a live allocation of another device is a device pointer used on the wrong device (heap addresses cannot be
// mistaken for the small integers, strides and gadget masks that share these structs)
void check_words(const char *kernel, const void *obj, size_t bytes) {
    if (bytes < 8 || device_count() == 1) return;
    const unsigned char *b = static_cast<const unsigned char *>(obj);
    std::lock_guard<std::mutex> lk(g_dev_mu);
    for (size_t off = 0; off + 8 <= bytes; off += 8) {
        uintptr_t w;
        memcpy(&w, b + off, 8);
        const Alloc *a = find_alloc((const void *)w);
        if (a && !a->host && a->device != t_device) die(kernel, "a pointer among the kernel arguments", a->device);
    }
}
}  // namespace emu

hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= emu::device_count()) return hipErrorInvalidDevice;
    emu::t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = emu::t_device;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int *d) {
    *d = emu::device_count();
    return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int device) {
    if (device < 0 || device >= emu::device_count() || !buf || len < 13) return hipErrorInvalidDevice;
    snprintf(buf, (size_t)len, "0000:%02x:00.0", 0xe0 + device);
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int device) {
    if (device < 0 || device >= emu::device_count()) return hipErrorInvalidDevice;
    *v = 3;
    return hipSuccess;
}

static void emu_copy_checks(void *d, const void *s, hipMemcpyKind kind, const char *what) {
    emu::check_ptr(d, kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice, what, "the destination");
    emu::check_ptr(s, kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice, what, "the source");
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind kind) {
    emu_copy_checks(d, s, kind, "hipMemcpy");
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind kind, hipStream_t stream) {
    emu::check_stream(stream, "hipMemcpyAsync");
    emu_copy_checks(d, s, kind, "hipMemcpyAsync");
    if (emu::t_capture) {
        emu::t_capture->nodes.push_back([d, s, n]() { memcpy(d, s, n); });
        return hipSuccess;
    }
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) {
    emu::check_ptr(d, true, "hipMemset", "the destination");
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t stream) {
    emu::check_stream(stream, "hipMemsetAsync");
    emu::check_ptr(d, true, "hipMemsetAsync", "the destination");
    if (emu::t_capture) {
        emu::t_capture->nodes.push_back([d, v, n]() { memset(d, v, n); });
        return hipSuccess;
    }
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = new emu::Stream{emu::t_device};
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    emu::check_stream(s, "hipStreamDestroy");
    delete static_cast<emu::Stream *>(s);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new EmuEvent{0, emu::t_device};
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t stream) {
    emu::check_stream(stream, "hipEventRecord");
    if (e->device != emu::t_device) emu::die("hipEventRecord", "the event", e->device);
    e->t = emu_now_ms();
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t e, unsigned) {
    emu::check_stream(stream, "hipStreamWaitEvent");
    if (e->device != emu::t_device) emu::die("hipStreamWaitEvent", "the event", e->device);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    if (a->device != b->device) emu::die("hipEventElapsedTime", "one of the two events", a->device);
    *ms = (float)(b->t - a->t);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    if (e->device != emu::t_device) emu::die("hipEventDestroy", "the event", e->device);
    delete e;
    return hipSuccess;
}
hipError_t emu_func_set_attribute(const void *fn, int bytes) {
    if (emu::t_capture) {  // the host library's own rule (tfhe_amd_bootstrap_streamed): every kernel of a schedule is configured before its capture
        fprintf(stderr, "emu: hipFuncSetAttribute inside a stream capture\n");
        abort();
    }
    std::lock_guard<std::mutex> lk(emu::g_dev_mu);
    emu::g_lds_attr[std::make_pair(fn, emu::t_device)] = bytes;
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode) {
    emu::check_stream(stream, "hipStreamBeginCapture");
    if (emu::t_capture) return hipErrorInvalidValue;
    emu::t_capture = new EmuGraph();
    emu::t_capture->device = emu::t_device;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *graph) {
    *graph = emu::t_capture;
    emu::t_capture = nullptr;
    return *graph ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *exec, hipGraph_t graph, void *, void *, unsigned long long) {
    *exec = new EmuGraph(*graph);
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t graph) {
    delete graph;
    return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) {
    delete exec;
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t stream) {
    emu::check_stream(stream, "hipGraphLaunch");
    if (exec->device != emu::t_device) emu::die("hipGraphLaunch", "the captured graph", exec->device);
    for (auto &node : exec->nodes) node();
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int device) {
    if (device < 0 || device >= emu::device_count()) return hipErrorInvalidDevice;
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "CPU emulation of the kernels (tests/emu), not a device; emulated ordinal %d of %d", device,
             emu::device_count());
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
    p->multiProcessorCount = 3;
    p->warpSize = 64;
    p->sharedMemPerBlock = p->sharedMemPerBlockOptin = 160 * 1024;
    p->pciBusID = 0xe0 + device;
    return hipSuccess;
}

unsigned long long emu_ref_ticks() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull;
}
double emu_now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

hipError_t hipMalloc(void **p, size_t bytes) {
    // exact size: ASan then flags any out-of-bounds device-pointer access made by a kernel
    *p = malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorInvalidValue;
    emu::register_alloc(*p, bytes, false);
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    emu::unregister_alloc(p, false, "hipFree");
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
    *p = malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorInvalidValue;
    emu::register_alloc(*p, bytes, true);
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    if (!p) return hipSuccess;
    emu::unregister_alloc(p, true, "hipHostFree");
    free(p);
    return hipSuccess;
}
