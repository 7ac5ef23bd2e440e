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
#include <mutex>
#include <thread>
#include <vector>

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
    ucontext_t ctx;
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
    ucontext_t sched;
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

void yield() { swapcontext(&t_blk->fibers[t_blk->current].ctx, &t_blk->sched); }

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
    swapcontext(&f.ctx, &b->sched);
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
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
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
            swapcontext(&b.sched, &f.ctx);
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
            for (int j = 0; j < 16; j++) sum += (int)ap[j] * (int)bp[j];
        }
        c[e] += sum;
    }
    return c;
}

}  // namespace emu
struct EmuGraph {
    std::vector<std::function<void()>> nodes;
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

hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    if (emu::t_capture) {
        emu::t_capture->nodes.push_back([d, s, n]() { memcpy(d, s, n); });
        return hipSuccess;
    }
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    if (emu::t_capture) {
        emu::t_capture->nodes.push_back([d, v, n]() { memset(d, v, n); });
        return hipSuccess;
    }
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (emu::t_capture) return hipErrorInvalidValue;
    emu::t_capture = new EmuGraph();
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
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t) {
    for (auto &node : exec->nodes) node();
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "CPU emulation of the kernels (tests/emu), not a device");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
    p->multiProcessorCount = 3;
    p->warpSize = 64;
    p->sharedMemPerBlock = p->sharedMemPerBlockOptin = 160 * 1024;
    return hipSuccess;
}

double emu_now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

hipError_t hipMalloc(void **p, size_t bytes) {
    // exact size: ASan then flags any out-of-bounds device-pointer access made by a kernel
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
