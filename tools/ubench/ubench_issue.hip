// ubench_issue.hip -- measured issue cost of the instructions the blind-rotation kernel is made of
// (diagnostic tool, not part of the product).  Each test runs ITER x 128 copies of one instruction
// (or a short pattern) on independent registers in every wave of the launch and reports shader cycles per
// instruction per SIMD (s_memtime of the median wave, and wall time x measured clock) and the shader
// clock (s_memtime / s_memrealtime).  Occupancy is pinned by LDS: blocks of 256 threads (one wave per
// SIMD each), k blocks per CU, each asking for 160 KB / k of dynamic LDS, grid = 256 k.
//   hipcc --offload-arch=gfx950 -O3 ubench_issue.hip -o ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long cyc, real; };

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define REP128(X) REP64(X) REP64(X)

enum Op { FMA64, ADD64, MUL64, FMA64NEG, FMA32, ADDU32, SUBU32, XOR32, AND32, LSHL32, ASHR32, BFEI32, BFEU32, CNDMASK, LSHLADD,
          CVT64I32, TRUNC64, CVTI32F64, DPPMOV, PERM32SWAP, PERM16SWAP, MIX_F64_INT, MIX_F64_2INT, DSW64, DSW2_64, DSR64, DSR2_64, DSR128,
          DSW64_FMA2, DSR64_FMA2, NOPS, DEP1, DEP2, DEP4, BFLY, BFLY_LDS, MADU64, MULHI32, MULLO32, ADDC64 };

template <int OP>
__global__ void __launch_bounds__(256) k_issue(Stamp *out, int iters, double seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    double a0 = seed + t, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 0.9999999;
    int i0 = t, i1 = t + 1, i2 = t + 2, i3 = t + 3, i4 = t + 4, i5 = t + 5, i6 = t + 6, i7 = t + 7;
    int j0 = t, j1 = t + 1, j2 = t + 2, j3 = t + 3, j4 = t + 4, j5 = t + 5, j6 = t + 6, j7 = t + 7;
    unsigned long long q0 = t, q1 = t + 1, q2 = t + 2, q3 = t + 3, q4 = t + 4, q5 = t + 5, q6 = t + 6, q7 = t + 7;
    const unsigned lds = (unsigned)(t * 8);       // conflict-free 8-byte slots, 2 KB per wave-instruction group
    const unsigned lds16 = (unsigned)(t * 16);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define A(k) a##k
#define I(k) i##k
#define Q(k) q##k
#define J(k) j##k
    for (int it = 0; it < iters; it++) {
        if (OP == FMA64) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(k)) : "v"(b), "v"(c));
            REP128(X)
#undef X
        } else if (OP == FMA64NEG) {
#define X(k) asm volatile("v_fma_f64 %0, -%0, %1, %2" : "+v"(A(k)) : "v"(b), "v"(c));
            REP128(X)
#undef X
        } else if (OP == ADD64) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(A(k)) : "v"(c));
            REP128(X)
#undef X
        } else if (OP == MUL64) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A(k)) : "v"(b));
            REP128(X)
#undef X
        } else if (OP == FMA32) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == ADDU32) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == SUBU32) {
#define X(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == XOR32) {
#define X(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == AND32) {
#define X(k) asm volatile("v_and_b32 %0, 0x3ff, %0" : "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == LSHL32) {
#define X(k) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == ASHR32) {
#define X(k) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == BFEI32) {
#define X(k) asm volatile("v_bfe_i32 %0, %0, 3, 10" : "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == BFEU32) {
#define X(k) asm volatile("v_bfe_u32 %0, %0, 3, 10" : "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == CNDMASK) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(I(k)) : "v"(t) : "vcc");
            REP128(X)
#undef X
        } else if (OP == LSHLADD) {
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == CVT64I32) {
#define X(k) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(A(k)) : "v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == CVTI32F64) {
#define X(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(I(k)) : "v"(A(k)));
            REP128(X)
#undef X
        } else if (OP == TRUNC64) {
#define X(k) asm volatile("v_trunc_f64 %0, %0" : "+v"(A(k)));
            REP128(X)
#undef X
        } else if (OP == DPPMOV) {
#define X(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == PERM32SWAP) {
#define X(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(I(k)), "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == PERM16SWAP) {
#define X(k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(I(k)), "+v"(I(k)));
            REP128(X)
#undef X
        } else if (OP == MADU64) {        // the 32 x 32 + 64 -> 64 multiply-add of the Real96 (128-bit fixed point) products
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(Q(k)) : "v"(I(k)), "v"(t) : "vcc");
            REP128(X)
#undef X
        } else if (OP == MULHI32) {
#define X(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == MULLO32) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(I(k)) : "v"(t));
            REP128(X)
#undef X
        } else if (OP == ADDC64) {        // 64 x (add_co ; addc_co): one 64-bit add through the carry
#define X(k) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(I(k)), "+v"(J(k)) : "v"(t) : "vcc");
            REP64(X)
#undef X
        } else if (OP == MIX_F64_INT) {   // 64 x (fma_f64 ; add_u32)
#define X(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_add_u32 %1, %1, %4" : "+v"(A(k)), "+v"(I(k)) : "v"(b), "v"(c), "v"(t));
            REP64(X)
#undef X
        } else if (OP == MIX_F64_2INT) {  // 64 x (add_f64 ; xor ; add_u32)
#define X(k) asm volatile("v_add_f64 %0, %0, %2\n\tv_xor_b32 %1, %1, %3\n\tv_add_u32 %1, %1, %3" : "+v"(A(k)), "+v"(I(k)) : "v"(c), "v"(t));
            REP64(X)
#undef X
        } else if (OP == DSW64) {
#define X(k) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(lds), "v"(A(k)), "i"(k * 2048) : "memory");
            REP128(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == DSW2_64) {
#define X(k) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(lds16), "v"(A(k)), "v"(a0), "i"(k * 2), "i"(k * 2 + 1) : "memory");
            REP128(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == DSR64) {
#define X(k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(A(k)) : "v"(lds), "i"(k * 2048) : "memory");
            REP128(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == DSR2_64) {
            double2 q0, q1, q2, q3, q4, q5, q6, q7;
#define X(k) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(q##k) : "v"(lds), "i"(k * 8), "i"(k * 8 + 64) : "memory");
            REP128(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a0 += q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x;
        } else if (OP == DSR128) {
            double2 q0, q1, q2, q3, q4, q5, q6, q7;
#define X(k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q##k) : "v"(lds16), "i"(k * 4096) : "memory");
            REP128(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a0 += q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x;
        } else if (OP == DSW64_FMA2) {  // 64 x (ds_write_b64 ; 2 fma_f64): do LDS stores hide under fp64?
#define X(k) asm volatile("ds_write_b64 %0, %1 offset:%2\n\tv_fma_f64 %1, %1, %3, %4\n\tv_fma_f64 %1, %1, %3, %4" :: "v"(lds), "v"(A(k)), "i"(k * 2048), "v"(b), "v"(c) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == DSR64_FMA2) {  // 64 x (ds_read_b64 ; 2 fma_f64)
            double q0, q1, q2, q3, q4, q5, q6, q7;
#define X(k) asm volatile("ds_read_b64 %0, %2 offset:%3\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %1, %1, %4, %5" : "=v"(q##k), "+v"(A(k)) : "v"(lds), "i"(k * 2048), "v"(b), "v"(c) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a0 += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
        } else if (OP == DEP1) {   // every fma depends on the previous one
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
            REP128(X)
#undef X
        } else if (OP == DEP2) {   // two interleaved chains
#define X(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
            REP64(X)
#undef X
        } else if (OP == DEP4) {   // four interleaved chains
#define X(k) asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == BFLY || OP == BFLY_LDS) {
            // compiler-scheduled radix-2 DIF butterflies on 8 complex values (3 stages = 12 butterflies
            // = 96 fp64 instructions), as the kernel's register passes; BFLY_LDS adds 8 LDS twiddle reads
            double xr[8] = {a0, a1, a2, a3, a4, a5, a6, a7}, xi[8] = {a7, a6, a5, a4, a3, a2, a1, a0};
            const double2 *twl = reinterpret_cast<const double2 *>(smem);
#pragma unroll
            for (int rep = 0; rep < 4; rep++) {
#pragma unroll
                for (int s_ = 4; s_ >= 1; s_ >>= 1)
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        if (m & s_) continue;
                        double cw = b, sw = c;
                        if (OP == BFLY_LDS) { const double2 w2 = twl[(t & 63) + 64 * ((m + s_ + rep) & 7)]; cw = w2.x; sw = w2.y; }
                        const double sr = xr[m] + xr[m + s_], si = xi[m] + xi[m + s_];
                        const double dr = xr[m] - xr[m + s_], di = xi[m] - xi[m + s_];
                        xr[m] = sr; xi[m] = si;
                        xr[m + s_] = __builtin_fma(-di, sw, dr * cw);
                        xi[m + s_] = __builtin_fma(di, cw, dr * sw);
                    }
            }
            a0 = xr[0]; a1 = xr[1]; a2 = xr[2]; a3 = xr[3]; a4 = xr[4]; a5 = xr[5]; a6 = xr[6]; a7 = xr[7];
            i0 += (int)(xi[0] + xi[1] + xi[2] + xi[3] + xi[4] + xi[5] + xi[6] + xi[7]);
        } else if (OP == NOPS) {
#define X(k) asm volatile("s_nop 0");
            REP128(X)
#undef X
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + (double)(j0 + j1 + j2 + j3 + j4 + j5 + j6 + j7) +
               (double)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
    if (s == 1.2345e-300) smem[0] = 1;  // keep everything alive
    if ((t & 63) == 0) {
        Stamp st{c1 - c0, r1 - r0};
        out[blockIdx.x * 4 + t / 64] = st;
    }
}

template <int OP>
static void run(const char *name, int per_iter, int k, Stamp *d_out, int iters) {
    const int blocks = 256 * k;
    const int lds = (160 * 1024 / k) & ~1023;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_issue<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    k_issue<OP><<<blocks, 256, lds>>>(d_out, iters / 10, 1.0);  // warm
    CHECK(hipEventRecord(e0));
    k_issue<OP><<<blocks, 256, lds>>>(d_out, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (auto &s : h) { cyc.push_back((double)s.cyc); clk.push_back((double)s.cyc / ((double)s.real / 100e6)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double med = cyc[cyc.size() / 2], n = (double)iters * per_iter, ck = clk[clk.size() / 2];
    printf("%-26s waves/SIMD %d  cyc/instr/SIMD %6.2f (wall x clock: %6.2f)  clock %.3f GHz  wall %.3f ms\n", name, k,
           med / n / k, ms * 1e-3 * ck / (n * k), ck / 1e9, ms);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    if (argc > 2 && std::string(argv[2]) == "int") {  // only the integer-multiply cases (round 5: the Real96 transforms' bound)
        Stamp *d;
        CHECK(hipMalloc(&d, sizeof(Stamp) * 256 * 8 * 4));
        for (int k : {1, 2, 4}) {
#define R(OP, n) run<OP>(#OP, n, k, d, iters);
            R(MADU64, 128) R(MULHI32, 128) R(MULLO32, 128) R(ADDC64, 128) R(ADDU32, 128) R(FMA64, 128)
#undef R
            printf("\n");
        }
        return 0;
    }
    Stamp *d_out;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * 256 * 8 * 4));
    for (int k : {1, 2}) {
#define R(OP, n) run<OP>(#OP, n, k, d_out, iters);
        R(FMA64, 128) R(FMA64NEG, 128) R(ADD64, 128) R(MUL64, 128) R(CVT64I32, 128) R(CVTI32F64, 128) R(TRUNC64, 128)
        R(FMA32, 128) R(ADDU32, 128) R(SUBU32, 128) R(XOR32, 128) R(AND32, 128) R(LSHL32, 128) R(ASHR32, 128) R(BFEI32, 128) R(BFEU32, 128)
        R(CNDMASK, 128) R(LSHLADD, 128) R(DPPMOV, 128) R(PERM32SWAP, 128) R(PERM16SWAP, 128) R(NOPS, 128)
        R(MIX_F64_INT, 64) R(MIX_F64_2INT, 64)
        R(DEP1, 128) R(DEP2, 128) R(DEP4, 128) R(BFLY, 384) R(BFLY_LDS, 384)
        R(DSW64, 128) R(DSW2_64, 128) R(DSR64, 128) R(DSR2_64, 128) R(DSR128, 128) R(DSW64_FMA2, 64) R(DSR64_FMA2, 64)
        printf("\n");
    }
    return 0;
}
