// tfhe_amd.hip -- host side of the C ABI (include/tfhe_amd.h): context, twiddle tables,
// key upload, kernel dispatch.  Built by hipcc for gfx950 (build.sh).  There is no CPU
// path in this library: without a device, context creation fails.
#include "tfhe_kernels.h"
#include "tfhe_kernels_generic.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/tfhe_amd.h"

using namespace tfhe;

// ------------------------------------------------------------------ context
struct tfhe_amd_gsw {
    tfhe_amd_ctx *ctx;
    double2 *data_d;  // [count][2l][2][PPL][64], scaled by 2/N
    int count;
    size_t sample_complex;  // complex values per sample = 2l*2*N/2
};

struct tfhe_amd_ctx {
    tfhe_amd_params p;
    int device;
    int logn;
    bool generic;  // N is neither 1024 nor 2048: the team-per-polynomial kernels of tfhe_kernels_generic.h
    hipStream_t stream;  // the stream work is issued on: `own`, or the caller's (tfhe_amd_set_stream)
    hipStream_t own;     // created with the context, alive until it is destroyed (switching streams costs nothing)
    std::string err;
    std::vector<double> fft_trig, ifft_trig;  // reference layout
    double2 *tw_d;                            // [2*NC]
    Gadget gd;
    const tfhe_amd_gsw *bk;
    int32_t *ks_d;   // reference layout [N][t][base][n_out+1]
    int8_t *ksm_d;   // matrix-core layout [hblocks][N*KPI][4][64][16] int8 (null: shape not covered)
    // growable scratch
    void *ws_lwe;
    size_t ws_lwe_bytes;
    void *ws_acc;
    size_t ws_acc_bytes;
    void *ws_gen;  // generic-N kernels: transform / accumulator areas that do not fit the LDS
    size_t ws_gen_bytes;
    int32_t *vp_rot_d;  // rotation constants of tfhe_amd_lut_eval
    void *hp_tw_d;      // Real96 twiddles: powomega [2N] then powombar [2N] (HpCplx), lazily built
    hipStream_t probe_stream;  // tfhe_amd_clock_probe: a second stream, so that the probe runs BESIDE the work queued on `stream`
    void *probe_d;             // its stamps
    int br_split_max;      // TFHE_AMD_OPT_BR_SPLIT: largest batch served by k_blind_rotate_split (< 0: BR_SPLIT_AUTO_MAX, 0: never)
    int ks_force_gather;   // TFHE_AMD_OPT_KS_GATHER: != 0 per-sample gather kernel even where the matrix-core kernel applies
    // TFHE_AMD_OPT_STREAMED_GRAPH: the n+3 launches of tfhe_amd_bootstrap_streamed captured once into a
    // hipGraph and replayed while the call's arguments stay the same
    bool streamed_graph;
    unsigned streamed_warm;  // bit 9 * ks_gather + 3 * class of the 0-step launches + class of the 1-step launches: that schedule variant has run uncaptured once
    struct {
        void *exec;  // hipGraphExec_t
        const void *x, *out;
        int32_t mu;
        int batch;
        int ks_gather;
    } sg;
    std::vector<const void *> lds_configured;  // kernels whose dynamic-LDS limit is raised on this device
    std::vector<std::pair<const void *, long long>> resident_blocks;  // persistent-wave kernels: workgroups the device keeps resident
};

namespace {

// Which blind-rotation form serves a Torus32 / N = 1024 batch (all three give the same bits; measured on one box,
// profiles/r03_latency.jsonl, blind rotation + extraction of n = 630):
//   up to BR_SPLIT_AUTO_MAX   k_blind_rotate_split: one ciphertext per 4-wave workgroup, two workgroups per CU
//                             (2.5 ms up to 256 ciphertexts, 3.6 ms up to 512; l = 2 only)
//   up to BR_LONE_WAVE_MAX    k_blind_rotate in 4-wave workgroups: one wave per SIMD on every CU (5.0 ms up to 1024)
//   above                     k_blind_rotate in 8-wave workgroups: two waves per SIMD (7.5 ms up to 2048, 16.6 ms for 4096)
constexpr int BR_SPLIT_AUTO_MAX = 512;
constexpr int BR_LONE_WAVE_MAX = 1024;

int fail(tfhe_amd_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}
#define HIPCHECK(c, expr)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), TFHE_AMD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
// every entry point that allocates or launches starts here: a process may hold contexts on several
// GPUs, and HIP allocates on / launches from the calling thread's current device
#define ENTER(c) HIPCHECK((c), hipSetDevice((c)->device))
#define REQUIRE(c, cond, msg) \
    do {                      \
        if (!(cond)) return fail((c), TFHE_AMD_ERR_PARAM, msg); \
    } while (0)

// cos/sin(2 pi i / n) with the angle folded into the first quadrant so that symmetric
// entries are bit-identical (the reference's accurate_cos/accurate_sin,
// CB/spqlios/spqlios-fft-impl.cpp:99-113).  Host libm; tables are pinned by SHA-256 in tests.
double quad_cos(long i, long n) {
    i = ((i % n) + n) % n;
    const double w = 2. * M_PI;
    if (i >= 3 * n / 4) return cos(w * (double)(n - i) / (double)n);
    if (i >= n / 2) return -cos(w * (double)(i - n / 2) / (double)n);
    if (i >= n / 4) return -cos(w * (double)(n / 2 - i) / (double)n);
    return cos(w * (double)i / (double)n);
}
double quad_sin(long i, long n) {
    i = ((i % n) + n) % n;
    const double w = 2. * M_PI;
    if (i >= 3 * n / 4) return -sin(w * (double)(n - i) / (double)n);
    if (i >= n / 2) return -sin(w * (double)(i - n / 2) / (double)n);
    if (i >= n / 4) return sin(w * (double)(n / 2 - i) / (double)n);
    return sin(w * (double)i / (double)n);
}
// append `count` (cos,sin) entries for angles mult*i in the reference's [4 cos | 4 sin] packing
void ref_block(std::vector<double> &v, int count, long mult, long n) {
    for (int i = 0; i < count; i += 4) {
        for (int k = 0; k < 4; k++) v.push_back(quad_cos(mult * (i + k), n));
        for (int k = 0; k < 4; k++) v.push_back(quad_sin(mult * (i + k), n));
    }
}

// Builds (a) the two tables exactly as new_ifft_table/new_fft_table lay them out and
// (b) the kernels' complex table; verifies fft table == conj(ifft table) entry by entry
// (with the quarter-turn exception of tfhe_kernels.h flip_sign_if), which the kernels rely on.
bool build_tables(int N, std::vector<double> &fft_trig, std::vector<double> &ifft_trig, std::vector<double2> &tw) {
    const long n = 2L * N;
    const int NC = N / 2;
    ifft_trig.clear();
    fft_trig.clear();
    ref_block(ifft_trig, NC, 1, n);
    for (int nn = NC; nn >= 8; nn /= 2) ref_block(ifft_trig, nn / 2, n / nn, n);
    for (int h = 4; h < NC; h *= 2) ref_block(fft_trig, h, -(n / (2 * h)), n);
    ref_block(fft_trig, NC, -1, n);
    tw.assign((size_t)2 * NC, make_double2(0., 0.));
    auto rc = [](const std::vector<double> &v, size_t base, int e) { return v[base + 8 * (e >> 2) + (e & 3)]; };
    auto rs = [](const std::vector<double> &v, size_t base, int e) { return v[base + 8 * (e >> 2) + 4 + (e & 3)]; };
    bool conj_ok = true;
    // twist: ifft block 0, fft last block
    const size_t fft_twist = fft_trig.size() - (size_t)2 * NC;
    for (int j = 0; j < NC; j++) {
        tw[j] = make_double2(rc(ifft_trig, 0, j), rs(ifft_trig, 0, j));
        conj_ok &= (rc(fft_trig, fft_twist, j) == tw[j].x) && (rs(fft_trig, fft_twist, j) == -tw[j].y);
    }
    size_t ib = (size_t)2 * NC;  // ifft stage blocks: h = NC/2 .. 4
    for (int h = NC / 2; h >= 4; h /= 2) {
        // fft block for half-size h starts after blocks 4, 8, .., h/2: 2*(4+8+..+h/2) = 2*(h-4)
        const size_t fb = (size_t)2 * (h - 4);
        const int base = 2 * NC - 2 * h;
        for (int o = 0; o < h; o++) {
            tw[base + o] = make_double2(rc(ifft_trig, ib, o), rs(ifft_trig, ib, o));
            // conjugate, except cos at the quarter turn, which has the opposite sign (see flip_sign_if)
            const double want_c = (o == h / 2) ? -tw[base + o].x : tw[base + o].x;
            conj_ok &= (rc(fft_trig, fb, o) == want_c) && (rs(fft_trig, fb, o) == -tw[base + o].y);
        }
        ib += (size_t)2 * h;
    }
    return conj_ok;
}

int ilog2(int v) {
    int r = 0;
    while ((1 << r) < v) r++;
    return r;
}

// dynamic LDS above 64 KiB must be allowed per function AND per device: remembered per context
// (a process may hold contexts on several GPUs)
template <typename KernelT>
int set_lds(tfhe_amd_ctx *c, KernelT kernel, size_t bytes) {
    const void *key = reinterpret_cast<const void *>(kernel);
    for (const void *k : c->lds_configured)
        if (k == key) return TFHE_AMD_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    c->lds_configured.push_back(key);
    return TFHE_AMD_OK;
}

// grid of a persistent-wave kernel: as many workgroups as the device keeps resident (occupancy of this kernel
// with its dynamic LDS x number of CUs), never more than the work needs
template <typename KernelT>
int persistent_grid(tfhe_amd_ctx *c, KernelT kernel, int block, size_t lds, int needed, int *grid) {
    // constant per (kernel, its LDS size, device): queried once per context, not on every small-batch plugin call
    const void *key = reinterpret_cast<const void *>(kernel);
    long long resident = 0;
    for (const auto &e : c->resident_blocks)
        if (e.first == key) resident = e.second;
    if (!resident) {
        int per_cu = 0, cus = 0;
        HIPCHECK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, key, block, lds));
        HIPCHECK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        resident = (long long)(per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
        c->resident_blocks.emplace_back(key, resident);
    }
    *grid = (int)(needed < resident ? needed : resident);
    return TFHE_AMD_OK;
}

// one instantiation per (torus, N): waves per workgroup chosen so LDS fits 160 KiB
template <typename T, int LOGN, int WAVES, int PAIR, int LC = 0, int BGC = 0>
int launch_br_t(tfhe_amd_ctx *c, const BlindRotateArgs<T> &a) {
    using Lds = BlindRotateLds<T, LOGN, WAVES>;
    auto kernel = k_blind_rotate<T, LOGN, WAVES, PAIR, LC, BGC>;
    if (int rc = set_lds(c, kernel, Lds::total)) return rc;
    const int blocks = (a.batch + WAVES - 1) / WAVES;
    TFHE_LAUNCH((k_blind_rotate<T, LOGN, WAVES, PAIR, LC, BGC>), dim3(blocks), dim3(WAVES * 64), Lds::total, c->stream, a);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
// one CMux step in place in global memory, two workgroups per CU (tfhe_kernels.h k_cmux_stream)
template <int LC, int BGC>
int launch_cmux_stream(tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) {
    static_assert(StreamLds::total <= 80 * 1024, "two workgroups per CU");
    auto kernel = k_cmux_stream<LC, BGC>;
    if (int rc = set_lds(c, kernel, StreamLds::total)) return rc;
    int grid = 0;
    if (int rc = persistent_grid(c, kernel, 256, StreamLds::total, (a.batch + 3) / 4, &grid)) return rc;
    TFHE_LAUNCH((k_cmux_stream<LC, BGC>), dim3(grid), dim3(256), StreamLds::total, c->stream, a);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
bool cmux_stream_applies(const tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) {
    return c->logn == 10 && c->p.l == 2 && a.n_steps == 1 && (a.flags & ~(uint32_t)BR_MODSWITCH) == 0 && !a.gsw_sel && a.sel_div <= 0 &&
           a.acc_io && a.rot;
}
// latency-shaped kernel (one ciphertext per 4-wave workgroup, tfhe_kernels.h k_blind_rotate_split): gate gadget
// length, N = 1024, plain blind rotations only
template <int BGC>
int launch_br_split(tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) {
    auto kernel = k_blind_rotate_split<BGC>;
    if (int rc = set_lds(c, kernel, SplitLds::total)) return rc;
    TFHE_LAUNCH((k_blind_rotate_split<BGC>), dim3(a.batch), dim3(256), SplitLds::total, c->stream, a);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
bool br_split_applies(const tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) {
    if (c->logn != 10 || c->p.l != 2 || a.n_steps < 1) return false;
    if ((a.flags & (BR_NO_ROTATE | BR_CMUX_DATA)) || a.gsw_sel || a.sel_div > 0) return false;
    const int limit = c->br_split_max < 0 ? BR_SPLIT_AUTO_MAX : c->br_split_max;
    return a.batch <= limit;
}
// which of the three forms above serves this call: 0 = k_blind_rotate_split, 1 = 4-wave workgroups, 2 = 8-wave workgroups
int br32_class(const tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) {
    if (br_split_applies(c, a)) return 0;
    return (c->logn == 10 && a.batch <= BR_LONE_WAVE_MAX) ? 1 : 2;
}
template <typename T>
int launch_br_gen(tfhe_amd_ctx *c, const BlindRotateArgs<T> &a);
int launch_br32(tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) {
    if (c->generic) return launch_br_gen<int32_t>(c, a);
    if (br_split_applies(c, a)) {
        if (c->p.Bgbit == 10) return launch_br_split<10>(c, a);
        if (c->p.Bgbit == 8) return launch_br_split<8>(c, a);
        return launch_br_split<0>(c, a);
    }
    // N=1024/Torus32: digits in pairs.  PPL=16 shapes transform one digit at a time (a pair would need 128 more
    // registers than the file has).
    if (c->logn == 11) return launch_br_t<int32_t, 11, 4, 1>(c, a);
    // Workgroup width: 8 waves (2 per SIMD, one workgroup per CU) is the throughput form; up to BR_LONE_WAVE_MAX
    // ciphertexts 4-wave workgroups put ONE wave on every SIMD of every CU instead of two on half of them -- a lone wave
    // walks a CMux in 19 k clocks instead of 28 k: 5.0 ms for any batch up to 1024 against 7.4 ms (profiles/r03_latency.jsonl).
    // Gadget length (and, for the gate set, Bgbit) fixed at compile time: the two transform groups
    // of a CMux are unrolled and each digit is one bit-field extract (tfhe_kernels.h, cmux_step)
    if (a.batch <= BR_LONE_WAVE_MAX) {
        if (c->p.l == 2 && c->p.Bgbit == 10) return launch_br_t<int32_t, 10, 4, 2, 2, 10>(c, a);  // gate set
        if (c->p.l == 2 && c->p.Bgbit == 8) return launch_br_t<int32_t, 10, 4, 2, 2, 8>(c, a);    // circuit bootstrap's output gadget
        if (c->p.l == 2) return launch_br_t<int32_t, 10, 4, 2, 2>(c, a);
        return launch_br_t<int32_t, 10, 4, 2>(c, a);
    }
#ifndef TFHE_NO_CMUX_STREAM
    // the one-launch-per-CMux schedule (tfhe_amd_bootstrap_streamed, tfhe_amd_mux_rotate): accumulators in place in global memory
    if (cmux_stream_applies(c, a)) {
        if (c->p.Bgbit == 10) return launch_cmux_stream<2, 10>(c, a);
        if (c->p.Bgbit == 8) return launch_cmux_stream<2, 8>(c, a);
        return launch_cmux_stream<2, 0>(c, a);
    }
#endif
    if (c->p.l == 2 && c->p.Bgbit == 10) return launch_br_t<int32_t, 10, 8, 2, 2, 10>(c, a);
    if (c->p.l == 2 && c->p.Bgbit == 8) return launch_br_t<int32_t, 10, 8, 2, 2, 8>(c, a);
    if (c->p.l == 2) return launch_br_t<int32_t, 10, 8, 2, 2>(c, a);
    return launch_br_t<int32_t, 10, 8, 2>(c, a);
}
int launch_br64(tfhe_amd_ctx *c, const BlindRotateArgs<int64_t> &a) {
    if (c->generic) return launch_br_gen<int64_t>(c, a);
    // N=2048: accumulator in registers, 4 waves per workgroup (one per SIMD); tfhe_kernels.h, BlindRotateLds::ACCREG.
    // (Two waves per ciphertext -- 128 lanes x 8 points, everything in 256 registers, 8 waves per CU -- was built and
    // measured in round 3: bit-identical, 1.35 instead of 1.57 vector instructions per fp64 instruction, but 10-20 % SLOWER:
    // its third transpose and 64 barriers per CMux put the LDS (ds_write_b128: 79 B/clk per CU) on the critical path.
    // profiles/r03_cb_team_experiment.txt; the kernel is in the history, commit d3c637a.)
    return c->logn == 10 ? launch_br_t<int64_t, 10, 4, 2>(c, a) : launch_br_t<int64_t, 11, 4, 1>(c, a);
}

// bytes one launch reads + writes, against the 256 MB Infinity Cache: beyond it the lane-contiguous accesses are
// nontemporal (tfhe_kernels.h stream_load / stream_store)
constexpr size_t STREAMING_WORKING_SET = (size_t)256 << 20;
template <typename TIN, int LOGN, bool PACK, int WAVES, bool NT>
int launch_ifft_nt(tfhe_amd_ctx *c, double *out_d, const TIN *in_d, int batch) {
    using Lds = FftLds<LOGN, WAVES>;
    auto kernel = k_ifft_batch<TIN, LOGN, WAVES, PACK, NT>;
    if (int rc = set_lds(c, kernel, Lds::total)) return rc;
    int grid = 0;
    if (int rc = persistent_grid(c, kernel, WAVES * 64, Lds::total, (batch + WAVES - 1) / WAVES, &grid)) return rc;
    TFHE_LAUNCH((k_ifft_batch<TIN, LOGN, WAVES, PACK, NT>), dim3(grid), dim3(WAVES * 64), Lds::total,
                c->stream, out_d, in_d, (const double2 *)c->tw_d, batch);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
template <typename TIN, int LOGN, bool PACK, int WAVES>
int launch_ifft_w(tfhe_amd_ctx *c, double *out_d, const TIN *in_d, int batch) {
    const size_t bytes = (size_t)batch * (1u << LOGN) * (sizeof(TIN) + sizeof(double));
    // (the key-conversion form writes a table that the next kernel reads: never streamed past the cache)
    if (!PACK && bytes > STREAMING_WORKING_SET) return launch_ifft_nt<TIN, LOGN, PACK, WAVES, true>(c, out_d, in_d, batch);
    return launch_ifft_nt<TIN, LOGN, PACK, WAVES, false>(c, out_d, in_d, batch);
}
// 4 waves (= polynomials) per workgroup: measured against 8 and 12 on MI355X, in the non-persistent form in round 2
// (profiles/r02_config4_fft.jsonl) and in the persistent form in round 3 (profiles/r03_config4_ab.txt: 8 equal within
// noise at N=2048, 12 slower by up to 40 % on the Lagrange -> coefficient side)
constexpr int FFT_WAVES = 4;
template <typename TIN, int LOGN, bool PACK = false>
int launch_ifft_t(tfhe_amd_ctx *c, double *out_d, const TIN *in_d, int batch) {
    return launch_ifft_w<TIN, LOGN, PACK, FFT_WAVES>(c, out_d, in_d, batch);
}
template <typename TOUT, int LOGN, int WAVES, bool NT>
int launch_fft_nt(tfhe_amd_ctx *c, TOUT *out_d, const double *in_d, int batch) {
    using Lds = FftLds<LOGN, WAVES>;
    auto kernel = k_fft_batch<TOUT, LOGN, WAVES, NT>;
    if (int rc = set_lds(c, kernel, Lds::total)) return rc;
    int grid = 0;
    if (int rc = persistent_grid(c, kernel, WAVES * 64, Lds::total, (batch + WAVES - 1) / WAVES, &grid)) return rc;
    TFHE_LAUNCH((k_fft_batch<TOUT, LOGN, WAVES, NT>), dim3(grid), dim3(WAVES * 64), Lds::total,
                c->stream, out_d, in_d, (const double2 *)c->tw_d, batch);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
template <typename TOUT, int LOGN, int WAVES>
int launch_fft_w(tfhe_amd_ctx *c, TOUT *out_d, const double *in_d, int batch) {
    const size_t bytes = (size_t)batch * (1u << LOGN) * (sizeof(TOUT) + sizeof(double));
    // nontemporal stores: +17..20 % at N = 1024, +4 % at N = 2048 / Torus64, -6 % at N = 2048 / Torus32 (r03_config4_ab.txt)
    if (bytes > STREAMING_WORKING_SET && !(LOGN == 11 && sizeof(TOUT) == 4))
        return launch_fft_nt<TOUT, LOGN, WAVES, true>(c, out_d, in_d, batch);
    return launch_fft_nt<TOUT, LOGN, WAVES, false>(c, out_d, in_d, batch);
}
template <typename TOUT, int LOGN>
int launch_fft_t(tfhe_amd_ctx *c, TOUT *out_d, const double *in_d, int batch) {
    return launch_fft_w<TOUT, LOGN, FFT_WAVES>(c, out_d, in_d, batch);
}

// ---- ring degrees other than 1024 / 2048 (tfhe_kernels_generic.h) -------------------------------------------
// dynamic LDS the generic kernels may ask for; their per-function limit is always raised to this (a process may hold
// contexts of several ring degrees, and the attribute is per function, not per context)
constexpr size_t GEN_LDS_MAX = (size_t)160 * 1024;
constexpr size_t GEN_WORK_MAX = (size_t)256 << 20;  // global scratch of one launch when the LDS is too small
// largest ring degree served one wavefront per ciphertext (kg_blind_rotate<T, true, true>).  Measured, MI355X, batch 4096 x 64 CMux
// (profiles/r06_generic_n.txt), against a workgroup per ciphertext: N = 256: 1.72 against 1.93 ms; N = 512: 2.87 against 3.17 ms with
// two digits at a time (80 KB per four waves: two workgroups per CU; all four digits, 112 KB: 4.14 ms); N = 64: equal
constexpr size_t GEN_WAVE_MAX_N = 512;
int grow(tfhe_amd_ctx *c, void **buf, size_t *have, size_t need);
// transforms: grid and work area of a launch over `batch` polynomials
struct GenFftPlan {
    int grid;
    size_t lds;
    double *work;
};
bool gen_fft_in_lds(const tfhe_amd_ctx *c) {
    const int block = gen_block(c->p.N);
    return (size_t)(block / gen_team_size(c->p.N / 2, block)) * c->p.N * sizeof(double) <= GEN_LDS_MAX;
}
template <typename KernelT>
int gen_fft_plan(tfhe_amd_ctx *c, KernelT kernel, int batch, GenFftPlan *pl) {
    const int N = c->p.N, block = gen_block(N), teams = block / gen_team_size(N / 2, block);
    const size_t bytes = (size_t)teams * N * sizeof(double);
    const int needed = (batch + teams - 1) / teams;
    if (int rc = set_lds(c, kernel, GEN_LDS_MAX)) return rc;
    if (gen_fft_in_lds(c)) {
        pl->lds = bytes;
        pl->work = nullptr;
        return persistent_grid(c, kernel, block, bytes, needed, &pl->grid);
    }
    long long g = (long long)(GEN_WORK_MAX / bytes);
    if (g < 1) g = 1;
    pl->grid = (int)(needed < g ? needed : g);
    pl->lds = 0;
    if (int rc = grow(c, &c->ws_gen, &c->ws_gen_bytes, (size_t)pl->grid * bytes)) return rc;
    pl->work = (double *)c->ws_gen;
    return TFHE_AMD_OK;
}
template <typename TIN, bool PACK, bool LDS>
int launch_gen_ifft_p(tfhe_amd_ctx *c, double *out_d, const TIN *in_d, int batch) {
    GenFftPlan pl;
    if (int rc = gen_fft_plan(c, kg_ifft_batch<TIN, PACK, LDS>, batch, &pl)) return rc;
    TFHE_LAUNCH((kg_ifft_batch<TIN, PACK, LDS>), dim3(pl.grid), dim3(gen_block(c->p.N)), pl.lds, c->stream, out_d, in_d,
                (const double2 *)c->tw_d, batch, c->logn, pl.work);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
template <typename TIN, bool PACK>
int launch_gen_ifft(tfhe_amd_ctx *c, double *out_d, const TIN *in_d, int batch) {
    return gen_fft_in_lds(c) ? launch_gen_ifft_p<TIN, PACK, true>(c, out_d, in_d, batch) : launch_gen_ifft_p<TIN, PACK, false>(c, out_d, in_d, batch);
}
template <typename TOUT, bool LDS>
int launch_gen_fft_p(tfhe_amd_ctx *c, TOUT *out_d, const double *in_d, int batch) {
    GenFftPlan pl;
    if (int rc = gen_fft_plan(c, kg_fft_batch<TOUT, LDS>, batch, &pl)) return rc;
    TFHE_LAUNCH((kg_fft_batch<TOUT, LDS>), dim3(pl.grid), dim3(gen_block(c->p.N)), pl.lds, c->stream, out_d, in_d,
                (const double2 *)c->tw_d, batch, c->logn, pl.work);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
template <typename TOUT>
int launch_gen_fft(tfhe_amd_ctx *c, TOUT *out_d, const double *in_d, int batch) {
    return gen_fft_in_lds(c) ? launch_gen_fft_p<TOUT, true>(c, out_d, in_d, batch) : launch_gen_fft_p<TOUT, false>(c, out_d, in_d, batch);
}
// blind rotation: accumulator, digit and Fourier-accumulator areas in LDS while they fit (all three; else the two
// transform areas; else none), the rest in a per-workgroup slice of global scratch
template <typename T>
int launch_br_gen(tfhe_amd_ctx *c, const BlindRotateArgs<T> &a) {
    const size_t N = (size_t)c->p.N;
    const size_t acc_b = sizeof(T) * 2 * N, dig1_b = sizeof(double) * N, fac_b = sizeof(double) * 2 * N;
    GenBrPlace g;
    memset(&g, 0, sizeof(g));
    g.logn = c->logn;
    // digits transformed together (one barrier sequence for all of them): all 2l rows -- then the Fourier accumulator lives in the
    // first two digit buffers (GenBrPlace::fac_in_dig) --, else one polynomial's l, else one
    const int nd_try[3] = {2 * c->p.l, c->p.l, 1};
    auto areas = [&](int nd) { return acc_b + nd * dig1_b + (nd == 2 * c->p.l ? 0 : fac_b); };
    size_t lds = 0, glob = 0;
    g.nd = 0;
    for (int k = 0; k < 3 && !g.nd; k++)
        if (areas(nd_try[k]) <= GEN_LDS_MAX) g.nd = nd_try[k];
    if (g.nd) {
        g.acc_lds = 0;
        g.dig_lds = (long long)acc_b;
        g.fac_lds = (long long)(acc_b + g.nd * dig1_b);
        lds = areas(g.nd);
    } else if (dig1_b + fac_b <= GEN_LDS_MAX) {
        g.nd = 1;
        g.acc_lds = -1;
        g.dig_lds = 0;
        g.fac_lds = (long long)dig1_b;
        lds = dig1_b + fac_b;
        glob = acc_b;
    } else {
        g.nd = 2 * c->p.l;
        g.acc_lds = g.dig_lds = g.fac_lds = -1;
        glob = areas(g.nd);
    }
    g.fac_in_dig = g.nd == 2 * c->p.l;
    int block = (int)(N / 4);  // NC/2 butterflies per layer
    block = block < 64 ? 64 : (block > GEN_BR_BLOCK ? GEN_BR_BLOCK : block);
    const bool all_lds = glob == 0;
#ifndef TFHE_GEN_NO_WAVE
    // One WAVEFRONT per ciphertext up to N = 512: four independent ciphertexts per 256-thread workgroup, no workgroup barrier at
    // all (kg_blind_rotate<T, true, true>), and as many digits at a time as leave room for TWO such workgroups on a CU
    if (all_lds && N <= GEN_WAVE_MAX_N) {
        int nd_w = 0;
        for (int k = 0; k < 3 && !nd_w; k++)
            if (4 * areas(nd_try[k]) <= GEN_LDS_MAX / 2) nd_w = nd_try[k];
        if (nd_w) {
            g.nd = nd_w;
            g.fac_in_dig = nd_w == 2 * c->p.l;
            g.wave_bytes = (long long)areas(nd_w);
            const size_t wg_lds = 4 * (size_t)g.wave_bytes;
            if (int rc = set_lds(c, kg_blind_rotate<T, true, true>, GEN_LDS_MAX)) return rc;
            int grid = 0;
            if (int rc = persistent_grid(c, kg_blind_rotate<T, true, true>, GEN_WAVE_BLOCK, wg_lds, (a.batch + 3) / 4, &grid)) return rc;
            TFHE_LAUNCH((kg_blind_rotate<T, true, true>), dim3(grid), dim3(GEN_WAVE_BLOCK), wg_lds, c->stream, a, g);
            HIPCHECK(c, hipGetLastError());
            return TFHE_AMD_OK;
        }
    }
#endif
    if (int rc = all_lds ? set_lds(c, kg_blind_rotate<T, true>, GEN_LDS_MAX) : set_lds(c, kg_blind_rotate<T, false>, GEN_LDS_MAX)) return rc;
    int grid = 0;
    if (int rc = all_lds ? persistent_grid(c, kg_blind_rotate<T, true>, block, lds, a.batch, &grid)
                         : persistent_grid(c, kg_blind_rotate<T, false>, block, lds, a.batch, &grid))
        return rc;
    if (glob) {
        const long long cap = (long long)(GEN_WORK_MAX / glob);
        if (grid > cap) grid = cap < 1 ? 1 : (int)cap;
        if (int rc = grow(c, &c->ws_gen, &c->ws_gen_bytes, (size_t)grid * glob)) return rc;
        g.work = (unsigned char *)c->ws_gen;
        g.work_stride = (long long)glob;
    }
    if (all_lds)
        TFHE_LAUNCH((kg_blind_rotate<T, true>), dim3(grid), dim3(block), lds, c->stream, a, g);
    else
        TFHE_LAUNCH((kg_blind_rotate<T, false>), dim3(grid), dim3(block), lds, c->stream, a, g);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}

void drop_streamed_graph(tfhe_amd_ctx *c) {
    if (c->sg.exec) (void)hipGraphExecDestroy((hipGraphExec_t)c->sg.exec);
    c->sg.exec = nullptr;
}

// workspaces only grow; a captured streamed-schedule graph holds their addresses and dies with them
int grow(tfhe_amd_ctx *c, void **buf, size_t *have, size_t need) {
    if (*have >= need) return TFHE_AMD_OK;
    drop_streamed_graph(c);
    if (*buf) {
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        HIPCHECK(c, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    HIPCHECK(c, hipMalloc(buf, need));
    *have = need;
    return TFHE_AMD_OK;
}

size_t torus_bytes(const tfhe_amd_ctx *c) { return (size_t)c->p.torus_bits / 8; }

int pack_rows(tfhe_amd_ctx *c, double2 *dst_d, const double *src_d, long long rows) {
    const long long total = rows * (c->p.N / 2);
    const int blocks = (int)((total + 255) / 256);
    if (c->generic)
        TFHE_LAUNCH_FLAT(kg_pack_gsw, dim3(blocks), dim3(256), c->stream, dst_d, src_d, rows, c->logn);
    else if (c->logn == 10)
        TFHE_LAUNCH_FLAT((k_pack_gsw<10>), dim3(blocks), dim3(256), c->stream, dst_d, src_d, rows);
    else
        TFHE_LAUNCH_FLAT((k_pack_gsw<11>), dim3(blocks), dim3(256), c->stream, dst_d, src_d, rows);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}

// ---- key switch on the matrix cores (k_ks_mfma): any row length, basebit 1..3, t * basebit <= 32,
// t * base <= 96
// Small batches: how many slices to cut K into so that the launch has a few hundred workgroups (a lone 256-sample
// tile is 8 * ceil(hblocks / 8) of them, each streaming its columns of the WHOLE key).  1 = no split (plain stores).
int ks_mfma_ksplit(int count, int row_ints, int n_in, int t, int bb) {
    const int kpi = ks_mfma_kpi(t, bb);
    const int chunks = ks_mfma_steps(n_in, kpi) / ks_mfma_chunk(kpi);
    const int base = 8 * ((count + 255) / 256) * (((row_ints + 31) / 32 + 7) / 8);
    int ks = 384 / (base > 0 ? base : 1);
    if (ks > chunks / 4) ks = chunks / 4;  // at least four chunks per slice
    if (ks > 16) ks = 16;
    return ks < 1 ? 1 : ks;
}
bool ks_mfma_supported(int t, int bb) { return bb >= 1 && bb <= 3 && t * bb <= 32 && ks_mfma_kpi(t, bb) <= 3; }
size_t ks_mfma_bytes(int n_in, int t, int bb, int row_ints) {
    return (size_t)((row_ints + 31) / 32) * ks_mfma_steps(n_in, ks_mfma_kpi(t, bb)) * 4096;
}
// tab_d: reference layout [n_in][t][base][row_ints] on the device
int ks_mfma_pack(hipStream_t stream, int8_t *dst_d, const int32_t *tab_d, int n_in, int t, int bb, int row_ints) {
    const int hblocks = (row_ints + 31) / 32;
    const long long frags = (long long)hblocks * ks_mfma_steps(n_in, ks_mfma_kpi(t, bb)) * 256;
    TFHE_LAUNCH_FLAT(k_ks_mfma_pack, dim3((unsigned)((frags + 255) / 256)), dim3(256), stream, dst_d, tab_d, n_in, t, bb,
                     row_ints, hblocks, frags);
    return hipGetLastError() == hipSuccess ? TFHE_AMD_OK : TFHE_AMD_ERR_DEVICE;
}
template <typename XT, int BB>
int launch_ks_mfma_b(hipStream_t stream, const KsMfmaArgs &a, int kpi, unsigned grid) {
    switch (kpi) {
        case 1: TFHE_LAUNCH((k_ks_mfma<XT, BB, 1>), dim3(grid), dim3(256), 2 * 8 * 4096, stream, a); break;
        case 2: TFHE_LAUNCH((k_ks_mfma<XT, BB, 2>), dim3(grid), dim3(256), 2 * 8 * 4096, stream, a); break;
        default: TFHE_LAUNCH((k_ks_mfma<XT, BB, 3>), dim3(grid), dim3(256), 2 * 6 * 4096, stream, a); break;
    }
    return hipGetLastError() == hipSuccess ? TFHE_AMD_OK : TFHE_AMD_ERR_DEVICE;
}
template <typename XT>
int launch_ks_mfma(hipStream_t stream, KsMfmaArgs a, int bb) {
    a.hblocks = (a.row_ints + 31) / 32;
    const int mtiles = (a.count + 255) / 256;
    const int kpi = ks_mfma_kpi(a.t, bb);
    if (a.ksplit < 1) a.ksplit = 1;
    const unsigned grid = 8u * (unsigned)mtiles * (unsigned)((a.hblocks + 7) / 8) * (unsigned)a.ksplit;
    switch (bb) {
        case 1: return launch_ks_mfma_b<XT, 1>(stream, a, kpi, grid);
        case 2: return launch_ks_mfma_b<XT, 2>(stream, a, kpi, grid);
        default: return launch_ks_mfma_b<XT, 3>(stream, a, kpi, grid);
    }
}

// common part of every blind-rotation-shaped call
template <typename T>
void fill_common(const tfhe_amd_ctx *c, BlindRotateArgs<T> &a, const tfhe_amd_gsw *g, int index, int steps, int batch) {
    memset(&a, 0, sizeof(a));
    a.bk = g->data_d + (size_t)index * g->sample_complex;
    a.bk_step_stride = (long long)g->sample_complex;
    a.tw = c->tw_d;
    a.gd = c->gd;
    a.n_steps = steps;
    a.batch = batch;
}
template <typename T, int LOGN>
int launch_exact_t(tfhe_amd_ctx *c, void *acc_d, const void *gsw_torus_d, int batch) {
    const size_t lds = ExactLds<T, LOGN>::total(c->p.l);
    if (lds > 160 * 1024) return fail(c, TFHE_AMD_ERR_PARAM, "exact external product: the digits of this gadget do not fit the LDS");
    auto kernel = k_extprod_exact<T, LOGN>;
    if (int rc = set_lds(c, kernel, lds)) return rc;
    TFHE_LAUNCH((k_extprod_exact<T, LOGN>), dim3(batch), dim3(256), lds, c->stream, (T *)acc_d, (const T *)gsw_torus_d,
                c->gd, batch);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
template <typename T>
int launch_exact_gen(tfhe_amd_ctx *c, void *acc_d, const void *gsw_torus_d, int batch) {
    const size_t N = (size_t)c->p.N;
    const size_t lds = sizeof(int32_t) * 2 * c->p.l * N + sizeof(T) * 4 * N;  // digits, [g | -g], results
    if (lds > GEN_LDS_MAX) return fail(c, TFHE_AMD_ERR_PARAM, "exact external product: the digits of this gadget do not fit the LDS");
    auto kernel = kg_extprod_exact<T>;
    if (int rc = set_lds(c, kernel, GEN_LDS_MAX)) return rc;
    TFHE_LAUNCH((kg_extprod_exact<T>), dim3(batch), dim3(256), lds, c->stream, (T *)acc_d, (const T *)gsw_torus_d, c->gd, batch,
                c->logn);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
int launch_br(tfhe_amd_ctx *c, const BlindRotateArgs<int32_t> &a) { return launch_br32(c, a); }
int launch_br(tfhe_amd_ctx *c, const BlindRotateArgs<int64_t> &a) { return launch_br64(c, a); }

template <typename T>
int cmux_t(tfhe_amd_ctx *c, void *out_d, const tfhe_amd_gsw *g, const int32_t *sel_d, const void *d0_d, const void *d1_d,
           int batch) {
    BlindRotateArgs<T> a;
    fill_common(c, a, g, 0, 1, batch);
    a.acc_io = (T *)out_d;
    a.cmux_d0 = (const T *)d0_d;
    a.cmux_d1 = (const T *)d1_d;
    a.gsw_sel = sel_d;
    a.gsw_sample_stride = (long long)g->sample_complex;
    a.cmux_stride = 2ll * c->p.N;
    a.flags = BR_NO_ROTATE | BR_CMUX_DATA;
    return launch_br(c, a);
}

// LUT evaluation by vertical packing (see tfhe_amd.h).  items = batch, TGSW sample b*d + i = bit i of item b.
//   tree:   level j (bit logN + j) halves the table: new[p] = CMux(bit, cur[2p], cur[2p+1]);
//           level 0 reads the shared plaintext table, later levels the previous level's TLWE samples
//   rotate: acc <- CMux(bit i, acc, X^{-2^i} acc), i < min(d, logN)  == a blind rotation whose "key" is
//           the item's own TGSW samples and whose rotations are the constants 2N - 2^i; extraction fused
constexpr int MAX_LOGN = 20;  // ring degrees 16 .. 2^20
constexpr int VP_ROT_STRIDE = 32, VP_ROT_ROWS = MAX_LOGN + 1;
template <typename T>
int lut_eval_t(tfhe_amd_ctx *c, void *lwe_out_d, const tfhe_amd_gsw *bits, int d, const void *lut_d, int batch) {
    const int N = c->p.N, logn = c->logn;
    const int low = d < logn ? d : logn, levels = d - low;
    if (!c->vp_rot_d) {
        // row r = the rotations of an r-bit selection, (2N - 2^i)_{i<r}, then 0 in the slot the
        // test-vector initialisation reads as "barb"
        int32_t rot[VP_ROT_ROWS * VP_ROT_STRIDE] = {0};
        for (int r = 0; r <= logn; r++)
            for (int i = 0; i < r; i++) rot[r * VP_ROT_STRIDE + i] = 2 * N - (1 << i);
        HIPCHECK(c, hipMalloc((void **)&c->vp_rot_d, sizeof(rot)));
        HIPCHECK(c, hipMemcpyAsync(c->vp_rot_d, rot, sizeof(rot), hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));  // `rot` is a stack array
    }
    BlindRotateArgs<T> a;
    const T *cur = nullptr;
    if (levels > 0) {
        // two ping-pong buffers: level 0 writes batch * 2^(levels-1) samples, level 1 half of that, ...
        const size_t sample = (size_t)2 * N * sizeof(T), first = ((size_t)batch << (levels - 1)) * sample;
        if (int rc = grow(c, &c->ws_acc, &c->ws_acc_bytes, first + first / 2)) return rc;
        T *buf[2] = {(T *)c->ws_acc, (T *)((char *)c->ws_acc + first)};
        for (int j = 0; j < levels; j++) {
            const int pairs = 1 << (levels - 1 - j);
            if ((long long)batch * pairs > 0x7fffffffLL) return fail(c, TFHE_AMD_ERR_PARAM, "LUT tree level too large");
            fill_common(c, a, bits, 0, 1, batch * pairs);
            a.acc_io = buf[j & 1];
            a.gsw_sample_stride = (long long)bits->sample_complex;
            a.sel_div = pairs;
            a.sel_mul = d;
            a.sel_add = low + j;
            a.flags = BR_NO_ROTATE | BR_CMUX_DATA;
            if (j == 0) {  // shared plaintext table: item (b, p) reads polynomials 2p and 2p + 1
                a.cmux_d0 = (const T *)lut_d;
                a.cmux_d1 = (const T *)lut_d + N;
                a.cmux_stride = 2ll * N;
                a.cmux_period = pairs;
                a.flags |= BR_CMUX_TRIVIAL;
            } else {       // previous level: item (b, p) reads samples (b, 2p) and (b, 2p + 1)
                a.cmux_d0 = cur;
                a.cmux_d1 = cur + 2 * N;
                a.cmux_stride = 4ll * N;
            }
            if (int rc = launch_br(c, a)) return rc;
            cur = buf[j & 1];
        }
    }
    fill_common(c, a, bits, 0, low, batch);
    a.gsw_sample_stride = (long long)bits->sample_complex;
    a.sel_div = 1;
    a.sel_mul = d;
    a.rot = c->vp_rot_d + low * VP_ROT_STRIDE;
    a.rot_stride = 0;
    a.lwe_out = (T *)lwe_out_d;
    a.flags = BR_EXTRACT;
    if (levels > 0) {
        a.acc_io = const_cast<T *>(cur);
    } else {  // d <= logN: the table is one polynomial, the accumulator starts as its trivial sample
        a.tv = (const T *)lut_d;
        a.tv_stride = 0;
        a.flags |= BR_INIT_TESTVEC;
    }
    return launch_br(c, a);
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

int tfhe_amd_device_info(int device, char *buf, size_t len) {
    if (!buf || len == 0) return TFHE_AMD_ERR_PARAM;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return TFHE_AMD_ERR_DEVICE;
    int rt = 0, drv = 0;
    (void)hipRuntimeGetVersion(&rt);
    (void)hipDriverGetVersion(&drv);
    char pci[32] = "?";
    (void)hipDeviceGetPCIBusId(pci, (int)sizeof(pci), device);
    snprintf(buf, len, "%s (%s), PCI %s, %d CUs, %d MHz, %.1f GiB, LDS/workgroup %zu KiB (opt-in %zu KiB), L2 %d MiB, wave %d, "
                       "HIP runtime %d driver %d",
             p.name, p.gcnArchName, pci, p.multiProcessorCount, p.clockRate / 1000, (double)p.totalGlobalMem / (1 << 30),
             p.sharedMemPerBlock >> 10, (size_t)p.sharedMemPerBlockOptin >> 10, p.l2CacheSize >> 20, p.warpSize, rt, drv);
    return TFHE_AMD_OK;
}

int tfhe_amd_device_count(int *count) {
    if (!count) return TFHE_AMD_ERR_PARAM;
    *count = 0;
    return hipGetDeviceCount(count) == hipSuccess ? TFHE_AMD_OK : TFHE_AMD_ERR_DEVICE;
}

// "domain:bus:device.function" of the physical GPU behind an ordinal: the ordinal is per process (a launcher may show
// every rank one device, all of them "device 0"), the bus id is what tells two GPUs apart
int tfhe_amd_device_pci_bus_id(int device, char *buf, size_t len) {
    if (!buf || len < 13) return TFHE_AMD_ERR_PARAM;
    return hipDeviceGetPCIBusId(buf, (int)len, device) == hipSuccess ? TFHE_AMD_OK : TFHE_AMD_ERR_DEVICE;
}

const char *tfhe_amd_version(void) { return "experimental-tfhe_amd 0.1 (gfx950)"; }

int tfhe_amd_ctx_create(const tfhe_amd_params *p, int device, tfhe_amd_ctx **out) {
    if (!p || !out) return TFHE_AMD_ERR_PARAM;
    *out = nullptr;
    if (p->k != 1) return TFHE_AMD_ERR_PARAM;
    // every ring degree the reference's plugin accepts (new_fft_table: a power of two >= 16, spqlios-fft-impl.cpp:157-160)
    if (p->N < 16 || p->N > (1 << MAX_LOGN) || (p->N & (p->N - 1))) return TFHE_AMD_ERR_PARAM;
    if (p->torus_bits != 32 && p->torus_bits != 64) return TFHE_AMD_ERR_PARAM;
    if (p->l < 1 || p->l > 8 || p->Bgbit < 1 || p->l * p->Bgbit > p->torus_bits - 1) return TFHE_AMD_ERR_PARAM;
    if (p->n < 1) return TFHE_AMD_ERR_PARAM;
    if (p->ks_t < 0 || (p->ks_t > 0 && (p->ks_basebit < 1 || p->ks_t * p->ks_basebit > 31 || p->ks_n_out < 1)))
        return TFHE_AMD_ERR_PARAM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return TFHE_AMD_ERR_DEVICE;  // no GPU: fail loudly, there is no CPU fallback
    tfhe_amd_ctx *c = new tfhe_amd_ctx();
    c->p = *p;
    c->device = device;
    c->logn = ilog2(p->N);
    c->generic = p->N != 1024 && p->N != 2048;
    c->ws_gen = nullptr;
    c->ws_gen_bytes = 0;
    c->stream = nullptr;
    c->own = nullptr;
    c->tw_d = nullptr;
    c->bk = nullptr;
    c->ks_d = nullptr;
    c->ksm_d = nullptr;
    c->ks_force_gather = 0;
    c->br_split_max = -1;
    c->streamed_graph = false;
    c->streamed_warm = 0;
    memset(&c->sg, 0, sizeof(c->sg));
    c->ws_lwe = c->ws_acc = nullptr;
    c->vp_rot_d = nullptr;
    c->hp_tw_d = nullptr;
    c->probe_stream = nullptr;
    c->probe_d = nullptr;
    c->ws_lwe_bytes = c->ws_acc_bytes = 0;
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return TFHE_AMD_ERR_DEVICE;
    }
    if (hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return TFHE_AMD_ERR_DEVICE;
    }
    c->stream = c->own;
    // from here on the context owns device objects: failures go through tfhe_amd_ctx_destroy
    std::vector<double2> tw;
    if (!build_tables(p->N, c->fft_trig, c->ifft_trig, tw)) {
        tfhe_amd_ctx_destroy(c);
        return TFHE_AMD_ERR_PARAM;  // libm broke the conjugate symmetry the kernels rely on
    }
    if (hipMalloc((void **)&c->tw_d, tw.size() * sizeof(double2)) != hipSuccess ||
        hipMemcpy(c->tw_d, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess) {
        tfhe_amd_ctx_destroy(c);
        return TFHE_AMD_ERR_DEVICE;
    }
    // gadget offset: Torus32 per TGswParams ctor (tgsw_functions.cpp:29-35), Torus64 per poc:349-350
    c->gd.Bgbit = p->Bgbit;
    c->gd.l = p->l;
    if (p->torus_bits == 32) {
        uint32_t s = 0;
        for (int i = 0; i < p->l; i++) s += 1u << (32 - (i + 1) * p->Bgbit);
        c->gd.offset = (uint32_t)(s * (1u << (p->Bgbit - 1)));
        c->gd.flip = (uint32_t)(s * (1u << (p->Bgbit - 1)));  // Bg/2 << (32-(i+1)Bgbit), every i: same bits
    } else {
        uint64_t s = 0;
        for (int i = 0; i <= p->l; i++) s |= 1ull << (63 - i * p->Bgbit);
        c->gd.offset = s;
        uint64_t f = 0;  // Bg/2 at every digit position: bit (Bgbit - 1) of the field at 64 - (i+1) Bgbit
        for (int i = 0; i < p->l; i++) f |= 1ull << (63 - i * p->Bgbit);
        c->gd.flip = f;
    }
    *out = c;
    return TFHE_AMD_OK;
}

void tfhe_amd_ctx_destroy(tfhe_amd_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->own && c->own != c->stream) (void)hipStreamSynchronize(c->own);
    if (c->tw_d) (void)hipFree(c->tw_d);
    if (c->ks_d) (void)hipFree(c->ks_d);
    if (c->ksm_d) (void)hipFree(c->ksm_d);
    if (c->ws_lwe) (void)hipFree(c->ws_lwe);
    if (c->ws_acc) (void)hipFree(c->ws_acc);
    if (c->ws_gen) (void)hipFree(c->ws_gen);
    if (c->vp_rot_d) (void)hipFree(c->vp_rot_d);
    if (c->hp_tw_d) (void)hipFree(c->hp_tw_d);
    if (c->probe_d) (void)hipFree(c->probe_d);
    if (c->probe_stream) (void)hipStreamDestroy(c->probe_stream);
    drop_streamed_graph(c);
    if (c->own) (void)hipStreamDestroy(c->own);
    delete c;
}

const char *tfhe_amd_last_error(const tfhe_amd_ctx *c) { return c ? c->err.c_str() : "null context"; }

int tfhe_amd_set_stream(tfhe_amd_ctx *c, void *s) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    hipStream_t next = s ? (hipStream_t)s : c->own;
    if (next != c->stream) {
        drop_streamed_graph(c);  // a captured schedule replays on the stream it was captured on
        c->stream = next;        // (no wait: work already queued on the previous stream stays ordered there)
    }
    return TFHE_AMD_OK;
}

int tfhe_amd_set_option(tfhe_amd_ctx *c, int option, int value) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    switch (option) {
        case TFHE_AMD_OPT_KS_GATHER:
            c->ks_force_gather = value != 0;  // normalised: it also indexes the warm-up bits of the streamed schedule
            return TFHE_AMD_OK;
        case TFHE_AMD_OPT_STREAMED_GRAPH:
            c->streamed_graph = value != 0;
            return TFHE_AMD_OK;
        case TFHE_AMD_OPT_BR_SPLIT:
            c->br_split_max = value;
            drop_streamed_graph(c);  // a captured schedule holds the kernel choice
            return TFHE_AMD_OK;
        default:
            return fail(c, TFHE_AMD_ERR_PARAM, "unknown option");
    }
}

int tfhe_amd_sync(tfhe_amd_ctx *c) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}

int tfhe_amd_event_create(tfhe_amd_ctx *c, void **event) {
    if (!c || !event) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    hipEvent_t e;
    HIPCHECK(c, hipEventCreate(&e));
    *event = (void *)e;
    return TFHE_AMD_OK;
}
int tfhe_amd_event_record(tfhe_amd_ctx *c, void *event) {
    if (!c || !event) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipEventRecord((hipEvent_t)event, c->stream));
    return TFHE_AMD_OK;
}
// host waits for a recorded event; the context's CURRENT stream waits for one (work queued after this call starts only once
// the event has happened): the two halves of a copy / compute pipeline over several streams of one context
int tfhe_amd_event_sync(tfhe_amd_ctx *c, void *event) {
    if (!c || !event) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipEventSynchronize((hipEvent_t)event));
    return TFHE_AMD_OK;
}
int tfhe_amd_stream_wait_event(tfhe_amd_ctx *c, void *event) {
    if (!c || !event) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipStreamWaitEvent(c->stream, (hipEvent_t)event, 0));
    return TFHE_AMD_OK;
}
int tfhe_amd_event_elapsed_ms(tfhe_amd_ctx *c, void *start, void *stop, float *ms) {
    if (!c || !start || !stop || !ms) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipEventSynchronize((hipEvent_t)stop));
    HIPCHECK(c, hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return TFHE_AMD_OK;
}
int tfhe_amd_event_destroy(tfhe_amd_ctx *c, void *event) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (event) HIPCHECK(c, hipEventDestroy((hipEvent_t)event));
    return TFHE_AMD_OK;
}

int tfhe_amd_build_tables(int N, double *fft_trig, double *ifft_trig) {  // host only: no context, no device
    if (N < 16 || N > (1 << MAX_LOGN) || (N & (N - 1))) return TFHE_AMD_ERR_PARAM;
    std::vector<double> f, r;
    std::vector<double2> tw;
    if (!build_tables(N, f, r, tw)) return TFHE_AMD_ERR_PARAM;
    if (fft_trig) memcpy(fft_trig, f.data(), f.size() * 8);
    if (ifft_trig) memcpy(ifft_trig, r.data(), r.size() * 8);
    return TFHE_AMD_OK;
}

int tfhe_amd_get_tables(const tfhe_amd_ctx *c, double *fft_trig, double *ifft_trig) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    if (fft_trig) memcpy(fft_trig, c->fft_trig.data(), c->fft_trig.size() * 8);
    if (ifft_trig) memcpy(ifft_trig, c->ifft_trig.data(), c->ifft_trig.size() * 8);
    return TFHE_AMD_OK;
}

int tfhe_amd_malloc(tfhe_amd_ctx *c, void **dptr, size_t bytes) {
    if (!c || !dptr) return TFHE_AMD_ERR_PARAM;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipMalloc(dptr, bytes ? bytes : 1));
    return TFHE_AMD_OK;
}
int tfhe_amd_free(tfhe_amd_ctx *c, void *dptr) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (dptr) {
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        HIPCHECK(c, hipFree(dptr));
    }
    return TFHE_AMD_OK;
}
int tfhe_amd_host_alloc(tfhe_amd_ctx *c, void **hptr, size_t bytes) {
    if (!c || !hptr) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return TFHE_AMD_OK;
}
int tfhe_amd_host_free(tfhe_amd_ctx *c, void *hptr) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (hptr) {
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        HIPCHECK(c, hipHostFree(hptr));
    }
    return TFHE_AMD_OK;
}
int tfhe_amd_memcpy_h2d(tfhe_amd_ctx *c, void *dst_d, const void *src, size_t bytes) {
    if (!c || (!dst_d && bytes) || (!src && bytes)) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}
// the same copies WITHOUT the wait: ordered on the context's current stream like a launch; the host buffer must stay valid
// (and should be page-locked: tfhe_amd_host_alloc) until that stream is synchronised
int tfhe_amd_memcpy_h2d_async(tfhe_amd_ctx *c, void *dst_d, const void *src, size_t bytes) {
    if (!c || (!dst_d && bytes) || (!src && bytes)) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, c->stream));
    return TFHE_AMD_OK;
}
int tfhe_amd_memcpy_d2h_async(tfhe_amd_ctx *c, void *dst, const void *src_d, size_t bytes) {
    if (!c || (!dst && bytes) || (!src_d && bytes)) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, c->stream));
    return TFHE_AMD_OK;
}
// more streams on the context's device (for tfhe_amd_set_stream): work of independent batches issued alternately on two
// streams overlaps one batch's copies with the other's kernels
int tfhe_amd_stream_create(tfhe_amd_ctx *c, void **stream) {
    if (!c || !stream) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    hipStream_t s = nullptr;
    HIPCHECK(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return TFHE_AMD_OK;
}
int tfhe_amd_stream_sync(tfhe_amd_ctx *c, void *stream) {
    if (!c || !stream) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipStreamSynchronize((hipStream_t)stream));
    return TFHE_AMD_OK;
}
int tfhe_amd_stream_destroy(tfhe_amd_ctx *c, void *stream) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (stream) {
        REQUIRE(c, (hipStream_t)stream != c->stream, "the stream is the context's current one: tfhe_amd_set_stream(ctx, NULL) first");
        HIPCHECK(c, hipStreamSynchronize((hipStream_t)stream));
        HIPCHECK(c, hipStreamDestroy((hipStream_t)stream));
    }
    return TFHE_AMD_OK;
}
int tfhe_amd_memcpy_d2h(tfhe_amd_ctx *c, void *dst, const void *src_d, size_t bytes) {
    if (!c || (!dst && bytes) || (!src_d && bytes)) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    HIPCHECK(c, hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}

// ---- keys
static int gsw_alloc(tfhe_amd_ctx *c, int count, tfhe_amd_gsw **out) {
    tfhe_amd_gsw *g = new tfhe_amd_gsw();
    g->ctx = c;
    g->count = count;
    g->sample_complex = (size_t)2 * c->p.l * 2 * (c->p.N / 2);
    g->data_d = nullptr;
    if (hipMalloc((void **)&g->data_d, g->sample_complex * count * sizeof(double2)) != hipSuccess) {
        delete g;
        return fail(c, TFHE_AMD_ERR_ALLOC, "hipMalloc(gsw)");
    }
    *out = g;
    return TFHE_AMD_OK;
}

int tfhe_amd_gsw_from_fft(tfhe_amd_ctx *c, const double *gsw_fft, int count, tfhe_amd_gsw **out) {
    if (!c || !gsw_fft || !out || count < 1) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    const long long rows = (long long)count * 2 * c->p.l * 2;
    const size_t bytes = (size_t)rows * c->p.N * sizeof(double);
    double *tmp = nullptr;
    HIPCHECK(c, hipMalloc((void **)&tmp, bytes));
    int rc = TFHE_AMD_OK;
    tfhe_amd_gsw *g = nullptr;
    if (hipMemcpyAsync(tmp, gsw_fft, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = fail(c, TFHE_AMD_ERR_DEVICE, "upload gsw");
    if (!rc) rc = gsw_alloc(c, count, &g);
    if (!rc) rc = pack_rows(c, g->data_d, tmp, rows);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (rc) {
        if (g) tfhe_amd_gsw_free(g);
        return rc;
    }
    *out = g;
    return TFHE_AMD_OK;
}

// tGswToFFTConvert on device-resident coefficients: every polynomial through
// execute_reverse_torus32/64, written directly in the key layout (k_ifft_batch<PACK>)
int tfhe_amd_gsw_from_torus_d(tfhe_amd_ctx *c, const void *gsw_torus_d, int count, tfhe_amd_gsw **out) {
    if (!c || !gsw_torus_d || !out || count < 1) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    const long long rows = (long long)count * 2 * c->p.l * 2;
    REQUIRE(c, rows <= 0x7fffffffLL, "too many TGSW samples for one conversion");
    tfhe_amd_gsw *g = nullptr;
    int rc = gsw_alloc(c, count, &g);
    if (rc) return rc;
    double *dst = reinterpret_cast<double *>(g->data_d);
    if (c->generic)
        rc = c->p.torus_bits == 32 ? launch_gen_ifft<int32_t, true>(c, dst, (const int32_t *)gsw_torus_d, (int)rows)
                                   : launch_gen_ifft<int64_t, true>(c, dst, (const int64_t *)gsw_torus_d, (int)rows);
    else if (c->p.torus_bits == 32)
        rc = c->logn == 10 ? launch_ifft_t<int32_t, 10, true>(c, dst, (const int32_t *)gsw_torus_d, (int)rows)
                           : launch_ifft_t<int32_t, 11, true>(c, dst, (const int32_t *)gsw_torus_d, (int)rows);
    else
        rc = c->logn == 10 ? launch_ifft_t<int64_t, 10, true>(c, dst, (const int64_t *)gsw_torus_d, (int)rows)
                           : launch_ifft_t<int64_t, 11, true>(c, dst, (const int64_t *)gsw_torus_d, (int)rows);
    if (rc) {
        tfhe_amd_gsw_free(g);
        return rc;
    }
    *out = g;
    return TFHE_AMD_OK;
}

int tfhe_amd_gsw_from_torus(tfhe_amd_ctx *c, const void *gsw_torus, int count, tfhe_amd_gsw **out) {
    if (!c || !gsw_torus || !out || count < 1) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    const size_t tbytes = (size_t)count * 2 * c->p.l * 2 * c->p.N * torus_bytes(c);
    void *tor = nullptr;
    HIPCHECK(c, hipMalloc(&tor, tbytes));
    int rc = TFHE_AMD_OK;
    if (hipMemcpyAsync(tor, gsw_torus, tbytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = fail(c, TFHE_AMD_ERR_DEVICE, "upload gsw");
    if (!rc) rc = tfhe_amd_gsw_from_torus_d(c, tor, count, out);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tor);
    return rc;
}

void tfhe_amd_gsw_free(tfhe_amd_gsw *g) {
    if (!g) return;
    if (g->ctx && g->ctx->bk == g) {
        g->ctx->bk = nullptr;
        drop_streamed_graph(g->ctx);
    }
    if (g->data_d) {
        if (g->ctx) (void)hipSetDevice(g->ctx->device);
        (void)hipStreamSynchronize(g->ctx->stream);
        (void)hipFree(g->data_d);
    }
    delete g;
}

int tfhe_amd_gsw_export_fft(tfhe_amd_ctx *c, const tfhe_amd_gsw *g, int index, double *out) {
    if (!c || !g || !out || index < 0 || index >= g->count) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    const int N = c->p.N, NC = N / 2, PPL = NC >= 64 ? NC / 64 : 1;
    std::vector<double2> h(g->sample_complex);
    HIPCHECK(c, hipMemcpyAsync(h.data(), g->data_d + (size_t)index * g->sample_complex,
                               g->sample_complex * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const double unscale = (double)N / 2.0;  // exact: power of two
    const int rows = 2 * c->p.l * 2;
    if (c->generic) {  // [row][NC] complex in the reference's order
        for (int r = 0; r < rows; r++)
            for (int j = 0; j < NC; j++) {
                const double2 v = h[(size_t)r * NC + j];
                out[(size_t)r * N + j] = v.x * unscale;
                out[(size_t)r * N + NC + j] = v.y * unscale;
            }
        return TFHE_AMD_OK;
    }
    for (int r = 0; r < rows; r++)
        for (int m = 0; m < PPL; m++)
            for (int t = 0; t < 64; t++) {
                const double2 v = h[((size_t)r * PPL + m) * 64 + t];
                out[(size_t)r * N + PPL * t + m] = v.x * unscale;
                out[(size_t)r * N + NC + PPL * t + m] = v.y * unscale;
            }
    return TFHE_AMD_OK;
}

int tfhe_amd_set_bootstrap_key(tfhe_amd_ctx *c, const tfhe_amd_gsw *bk) {
    if (!c) return TFHE_AMD_ERR_PARAM;
    REQUIRE(c, bk && bk->ctx == c && bk->count == c->p.n, "bootstrap key must hold exactly n TGSW samples of this context");
    c->bk = bk;
    drop_streamed_graph(c);
    return TFHE_AMD_OK;
}

int tfhe_amd_load_keyswitch_key(tfhe_amd_ctx *c, const int32_t *ks) {
    if (!c || !ks) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, c->p.ks_t > 0, "context has no key-switch parameters");
    drop_streamed_graph(c);
    const size_t bytes = (size_t)c->p.N * c->p.ks_t * ((size_t)1 << c->p.ks_basebit) * (c->p.ks_n_out + 1) * 4;
    if (!c->ks_d) HIPCHECK(c, hipMalloc((void **)&c->ks_d, bytes));
    HIPCHECK(c, hipMemcpyAsync(c->ks_d, ks, bytes, hipMemcpyHostToDevice, c->stream));
    if (ks_mfma_supported(c->p.ks_t, c->p.ks_basebit)) {
        if (!c->ksm_d) HIPCHECK(c, hipMalloc((void **)&c->ksm_d, ks_mfma_bytes(c->p.N, c->p.ks_t, c->p.ks_basebit, c->p.ks_n_out + 1)));
        if (int rc = ks_mfma_pack(c->stream, c->ksm_d, c->ks_d, c->p.N, c->p.ks_t, c->p.ks_basebit, c->p.ks_n_out + 1))
            return fail(c, rc, "k_ks_mfma_pack launch");
    }
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}

// the key-switch key from DEVICE memory (or any pointer hipMemcpyDefault resolves): what a rank does with the bytes a
// broadcast delivered.  Same layout as tfhe_amd_load_keyswitch_key; the matrix-core layout is rebuilt locally.
int tfhe_amd_load_keyswitch_key_d(tfhe_amd_ctx *c, const int32_t *ks_any) {
    if (!c || !ks_any) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, c->p.ks_t > 0, "context has no key-switch parameters");
    drop_streamed_graph(c);
    const size_t bytes = (size_t)c->p.N * c->p.ks_t * ((size_t)1 << c->p.ks_basebit) * (c->p.ks_n_out + 1) * 4;
    if (!c->ks_d) HIPCHECK(c, hipMalloc((void **)&c->ks_d, bytes));
    HIPCHECK(c, hipMemcpyAsync(c->ks_d, ks_any, bytes, hipMemcpyDefault, c->stream));
    if (ks_mfma_supported(c->p.ks_t, c->p.ks_basebit)) {
        if (!c->ksm_d) HIPCHECK(c, hipMalloc((void **)&c->ksm_d, ks_mfma_bytes(c->p.N, c->p.ks_t, c->p.ks_basebit, c->p.ks_n_out + 1)));
        if (int rc = ks_mfma_pack(c->stream, c->ksm_d, c->ks_d, c->p.N, c->p.ks_t, c->p.ks_basebit, c->p.ks_n_out + 1))
            return fail(c, rc, "k_ks_mfma_pack launch");
    }
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}
int tfhe_amd_keyswitch_key_bytes(const tfhe_amd_ctx *c, size_t *bytes) {
    if (!c || !bytes) return TFHE_AMD_ERR_PARAM;
    *bytes = c->p.ks_t > 0 ? (size_t)c->p.N * c->p.ks_t * ((size_t)1 << c->p.ks_basebit) * (c->p.ks_n_out + 1) * 4 : 0;
    return TFHE_AMD_OK;
}
int tfhe_amd_keyswitch_key_export(tfhe_amd_ctx *c, void *dst_any) {
    if (!c || !dst_any) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (!c->ks_d) return fail(c, TFHE_AMD_ERR_STATE, "no key-switch key");
    size_t bytes = 0;
    (void)tfhe_amd_keyswitch_key_bytes(c, &bytes);
    HIPCHECK(c, hipMemcpyAsync(dst_any, c->ks_d, bytes, hipMemcpyDefault, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}

// TGSW samples in the KERNEL layout, as bytes: what one device hands to another (or to a broadcast) so that the
// receiver neither regenerates nor re-converts the key.  [count][2l][2][N/128][64] double2, scaled by 2/N.
int tfhe_amd_gsw_packed_bytes(const tfhe_amd_ctx *c, int count, size_t *bytes) {
    if (!c || !bytes || count < 0) return TFHE_AMD_ERR_PARAM;
    *bytes = (size_t)count * 2 * c->p.l * 2 * (c->p.N / 2) * sizeof(double2);
    return TFHE_AMD_OK;
}
int tfhe_amd_gsw_export_packed(tfhe_amd_ctx *c, const tfhe_amd_gsw *g, void *dst_any) {
    if (!c || !g || !dst_any) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, g->ctx == c, "TGSW handle belongs to another context");
    HIPCHECK(c, hipMemcpyAsync(dst_any, g->data_d, g->sample_complex * g->count * sizeof(double2), hipMemcpyDefault, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}
int tfhe_amd_gsw_from_packed(tfhe_amd_ctx *c, const void *src_any, int count, tfhe_amd_gsw **out) {
    if (!c || !src_any || !out || count < 1) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    tfhe_amd_gsw *g = nullptr;
    if (int rc = gsw_alloc(c, count, &g)) return rc;
    if (hipMemcpyAsync(g->data_d, src_any, g->sample_complex * count * sizeof(double2), hipMemcpyDefault, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        tfhe_amd_gsw_free(g);
        return fail(c, TFHE_AMD_ERR_DEVICE, "copy of the packed TGSW samples");
    }
    *out = g;
    return TFHE_AMD_OK;
}

// ---- diagnostics: the shader clock the chip holds WHILE the context's queued work runs
// A few one-wave workgroups on a second stream stamp s_memtime (shader cycles) and s_memrealtime (100 MHz) around a
// sleep loop of `duration_us` (MI355X_MICROARCH.md, DVFS give-back item 6: clock = d memtime / d memrealtime x 100 MHz).
// They need a wave slot and no LDS, so they run beside kernels that fill the CUs' LDS.  Call right after queueing work
// on the context's stream; blocks for about duration_us.
int tfhe_amd_clock_probe(tfhe_amd_ctx *c, int duration_us, double *ghz_median, double *ghz_min, double *ghz_max) {
    if (!c || duration_us < 1 || !ghz_median) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    constexpr int PROBES = 32;
    if (!c->probe_stream) HIPCHECK(c, hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
    if (!c->probe_d) HIPCHECK(c, hipMalloc(&c->probe_d, PROBES * sizeof(ClockStamp)));
    TFHE_LAUNCH_FLAT(k_clock_probe, dim3(PROBES), dim3(64), c->probe_stream, (ClockStamp *)c->probe_d,
                     (unsigned long long)duration_us * 100ull);
    HIPCHECK(c, hipGetLastError());
    ClockStamp h[PROBES];
    HIPCHECK(c, hipMemcpyAsync(h, c->probe_d, sizeof(h), hipMemcpyDeviceToHost, c->probe_stream));
    HIPCHECK(c, hipStreamSynchronize(c->probe_stream));
    double g[PROBES];
    int m = 0;
    for (int i = 0; i < PROBES; i++)
        if (h[i].r1 > h[i].r0) g[m++] = (double)(h[i].c1 - h[i].c0) / (double)(h[i].r1 - h[i].r0) * 0.1;
    if (m == 0) return fail(c, TFHE_AMD_ERR_DEVICE, "clock probe: no stamps");
    for (int i = 1; i < m; i++)  // insertion sort: 32 values
        for (int j = i; j > 0 && g[j] < g[j - 1]; j--) std::swap(g[j], g[j - 1]);
    *ghz_median = g[m / 2];
    if (ghz_min) *ghz_min = g[0];
    if (ghz_max) *ghz_max = g[m - 1];
    return TFHE_AMD_OK;
}

// ---- L1
int tfhe_amd_ifft_int32(tfhe_amd_ctx *c, double *out_d, const int32_t *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic) return launch_gen_ifft<int32_t, false>(c, out_d, in_d, batch);
    return c->logn == 10 ? launch_ifft_t<int32_t, 10>(c, out_d, in_d, batch) : launch_ifft_t<int32_t, 11>(c, out_d, in_d, batch);
}
int tfhe_amd_ifft_torus64(tfhe_amd_ctx *c, double *out_d, const int64_t *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic) return launch_gen_ifft<int64_t, false>(c, out_d, in_d, batch);
    return c->logn == 10 ? launch_ifft_t<int64_t, 10>(c, out_d, in_d, batch) : launch_ifft_t<int64_t, 11>(c, out_d, in_d, batch);
}
// k_fft_batch reads its input with 16-byte loads (tfhe_kernels.h): Lagrange-domain inputs must be 16-byte aligned
#define REQUIRE_ALIGNED16(c, p) REQUIRE(c, ((uintptr_t)(p) & 15) == 0, "Lagrange-domain input must be 16-byte aligned")
int tfhe_amd_fft_torus32(tfhe_amd_ctx *c, int32_t *out_d, const double *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE_ALIGNED16(c, in_d);
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic) return launch_gen_fft<int32_t>(c, out_d, in_d, batch);
    return c->logn == 10 ? launch_fft_t<int32_t, 10>(c, out_d, in_d, batch) : launch_fft_t<int32_t, 11>(c, out_d, in_d, batch);
}
int tfhe_amd_fft_torus64(tfhe_amd_ctx *c, int64_t *out_d, const double *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE_ALIGNED16(c, in_d);
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic) return launch_gen_fft<int64_t>(c, out_d, in_d, batch);
    return c->logn == 10 ? launch_fft_t<int64_t, 10>(c, out_d, in_d, batch) : launch_fft_t<int64_t, 11>(c, out_d, in_d, batch);
}
// the bare core transforms of spqlios-fft.h:52-53 (`ifft`, `fft`): N doubles -> N doubles, no conversion, no scale.
// Same element type on both sides makes aliasing easy: the persistent kernels prefetch the next polynomial while they
// store the current one, so ANY overlap of the two ranges (out = in, or out = in + N inside a batch) is refused.
static bool ranges_overlap(const void *a, const void *b, size_t bytes) {
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    if (bytes == 0) return x == y;
    return x < y + bytes && y < x + bytes;
}
int tfhe_amd_ifft_f64(tfhe_amd_ctx *c, double *out_d, const double *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, !ranges_overlap(out_d, in_d, (size_t)batch * c->p.N * sizeof(double)), "the transforms are out of place on the device: the two ranges overlap");
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic) return launch_gen_ifft<double, false>(c, out_d, in_d, batch);
    return c->logn == 10 ? launch_ifft_t<double, 10>(c, out_d, in_d, batch) : launch_ifft_t<double, 11>(c, out_d, in_d, batch);
}
int tfhe_amd_fft_f64(tfhe_amd_ctx *c, double *out_d, const double *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE_ALIGNED16(c, in_d);
    REQUIRE(c, !ranges_overlap(out_d, in_d, (size_t)batch * c->p.N * sizeof(double)), "the transforms are out of place on the device: the two ranges overlap");
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic) return launch_gen_fft<double>(c, out_d, in_d, batch);
    return c->logn == 10 ? launch_fft_t<double, 10>(c, out_d, in_d, batch) : launch_fft_t<double, 11>(c, out_d, in_d, batch);
}
int tfhe_amd_lagrange_addmul(tfhe_amd_ctx *c, double *res_d, const double *a_d, const double *b_d, int batch, int b_shared) {
    if (!c || !res_d || !a_d || !b_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    const int Ns2 = c->p.N / 2;
    const long long total = (long long)batch * Ns2;
    TFHE_LAUNCH_FLAT(k_lagrange_addmul, dim3((unsigned)((total + 255) / 256)), dim3(256), c->stream, res_d, a_d, b_d, Ns2,
                (long long)(b_shared ? 0 : c->p.N), total);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}

// ---- L2
static int run_steps(tfhe_amd_ctx *c, void *acc_d, const tfhe_amd_gsw *g, int index, int steps, const int32_t *rot_d,
                     int rot_stride, int batch, uint32_t flags) {
    if (c->p.torus_bits == 32) {
        BlindRotateArgs<int32_t> a;
        fill_common(c, a, g, index, steps, batch);
        a.acc_io = (int32_t *)acc_d;
        a.rot = rot_d;
        a.rot_stride = rot_stride;
        a.flags = flags;
        return launch_br32(c, a);
    }
    BlindRotateArgs<int64_t> a;
    fill_common(c, a, g, index, steps, batch);
    a.acc_io = (int64_t *)acc_d;
    a.rot = rot_d;
    a.rot_stride = rot_stride;
    a.flags = flags;
    return launch_br64(c, a);
}

int tfhe_amd_extern_mul(tfhe_amd_ctx *c, void *acc_d, const tfhe_amd_gsw *g, int index, int batch) {
    if (!c || !acc_d || !g || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, g->ctx == c && index >= 0 && index < g->count, "bad TGSW handle/index");
    if (batch == 0) return TFHE_AMD_OK;
    return run_steps(c, acc_d, g, index, 1, nullptr, 0, batch, BR_NO_ROTATE);
}
int tfhe_amd_mux_rotate(tfhe_amd_ctx *c, void *acc_d, const tfhe_amd_gsw *g, int index, const int32_t *barai_d, int batch) {
    if (!c || !acc_d || !g || !barai_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, g->ctx == c && index >= 0 && index < g->count, "bad TGSW handle/index");
    if (batch == 0) return TFHE_AMD_OK;
    return run_steps(c, acc_d, g, index, 1, barai_d, 1, batch, 0);
}

// CMux on data: out = gsw[sel] (x) (d1 - d0) + d0 -- one launch, subtraction and addition fused
// into the kernel's load/store
int tfhe_amd_cmux(tfhe_amd_ctx *c, void *out_d, const tfhe_amd_gsw *g, const int32_t *sel_d, const void *d0_d,
                  const void *d1_d, int batch) {
    if (!c || !out_d || !g || !d0_d || !d1_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, g->ctx == c, "TGSW handle belongs to another context");
    if (batch == 0) return TFHE_AMD_OK;
    return c->p.torus_bits == 32 ? cmux_t<int32_t>(c, out_d, g, sel_d, d0_d, d1_d, batch)
                                 : cmux_t<int64_t>(c, out_d, g, sel_d, d0_d, d1_d, batch);
}

int tfhe_amd_lut_eval(tfhe_amd_ctx *c, void *lwe_out_d, const tfhe_amd_gsw *bits, int d, const void *lut_d, int batch) {
    if (!c || !lwe_out_d || !bits || !lut_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, bits->ctx == c, "TGSW handle belongs to another context");
    REQUIRE(c, c->p.torus_bits == 32, "LUT evaluation consumes TGSW32 samples (circuit-bootstrap outputs)");
    REQUIRE(c, d >= 1 && d <= c->logn + 20, "LUT evaluation: 1 <= d <= log2(N) + 20");
    REQUIRE(c, (long long)batch * d <= bits->count, "LUT evaluation needs batch * d TGSW samples (bit i of item b at b*d + i)");
    if (batch == 0) return TFHE_AMD_OK;
    return lut_eval_t<int32_t>(c, lwe_out_d, bits, d, lut_d, batch);
}

// ---- L3
int tfhe_amd_blind_rotate(tfhe_amd_ctx *c, void *acc_d, const int32_t *bara_d, int batch) {
    if (!c || !acc_d || !bara_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (!c->bk) return fail(c, TFHE_AMD_ERR_STATE, "no bootstrapping key");
    if (batch == 0) return TFHE_AMD_OK;
    return run_steps(c, acc_d, c->bk, 0, c->p.n, bara_d, c->p.n, batch, 0);
}

int tfhe_amd_blind_rotate_extract(tfhe_amd_ctx *c, void *lwe_out_d, const void *v_d, int v_per_sample,
                                  const int32_t *rot_d, int batch) {
    if (!c || !lwe_out_d || !v_d || !rot_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (!c->bk) return fail(c, TFHE_AMD_ERR_STATE, "no bootstrapping key");
    if (batch == 0) return TFHE_AMD_OK;
    const uint32_t flags = BR_INIT_TESTVEC | BR_EXTRACT;
    if (c->p.torus_bits == 32) {
        BlindRotateArgs<int32_t> a;
        fill_common(c, a, c->bk, 0, c->p.n, batch);
        a.rot = rot_d;
        a.rot_stride = c->p.n + 1;
        a.tv = (const int32_t *)v_d;
        a.tv_stride = v_per_sample ? c->p.N : 0;
        a.lwe_out = (int32_t *)lwe_out_d;
        a.flags = flags;
        return launch_br32(c, a);
    }
    BlindRotateArgs<int64_t> a;
    fill_common(c, a, c->bk, 0, c->p.n, batch);
    a.rot = rot_d;
    a.rot_stride = c->p.n + 1;
    a.tv = (const int64_t *)v_d;
    a.tv_stride = v_per_sample ? c->p.N : 0;
    a.lwe_out = (int64_t *)lwe_out_d;
    a.flags = flags;
    return launch_br64(c, a);
}

int tfhe_amd_bootstrap_woks(tfhe_amd_ctx *c, int32_t *lwe_out_d, int32_t mu, const int32_t *x_d, int batch) {
    if (!c || !lwe_out_d || !x_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, c->p.torus_bits == 32, "tfhe_bootstrap_woKS_FFT is a Torus32 operation");
    if (!c->bk) return fail(c, TFHE_AMD_ERR_STATE, "no bootstrapping key");
    if (batch == 0) return TFHE_AMD_OK;
    BlindRotateArgs<int32_t> a;
    fill_common(c, a, c->bk, 0, c->p.n, batch);
    a.rot = x_d;
    a.rot_stride = c->p.n + 1;
    a.tv_const = mu;
    a.lwe_out = lwe_out_d;
    a.flags = BR_INIT_TESTVEC | BR_EXTRACT | BR_MODSWITCH | BR_TV_CONST;
    return launch_br32(c, a);
}

int tfhe_amd_keyswitch(tfhe_amd_ctx *c, int32_t *out_d, const int32_t *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (!c->ks_d) return fail(c, TFHE_AMD_ERR_STATE, "no key-switch key");
    if (batch == 0) return TFHE_AMD_OK;
    if (c->ksm_d && c->ks_force_gather == 0) {
        KsMfmaArgs a;
        memset(&a, 0, sizeof(a));
        a.out = out_d;
        a.stride_in_group = c->p.ks_n_out + 1;
        a.group = batch;
        a.x = in_d;
        a.bm = c->ksm_d;
        a.x_stride = c->p.N + 1;
        a.n_in = c->p.N;
        a.t = c->p.ks_t;
        a.row_ints = c->p.ks_n_out + 1;
        a.count = batch;
        a.b_index = c->p.N;
        a.b_col = c->p.ks_n_out;
        a.ksplit = ks_mfma_ksplit(batch, a.row_ints, a.n_in, a.t, c->p.ks_basebit);
        if (a.ksplit > 1)  // the slices add into the output
            HIPCHECK(c, hipMemsetAsync(out_d, 0, (size_t)batch * (c->p.ks_n_out + 1) * 4, c->stream));
        if (int rc = launch_ks_mfma<int32_t>(c->stream, a, c->p.ks_basebit)) return fail(c, rc, "k_ks_mfma launch");
        return TFHE_AMD_OK;
    }
    TFHE_LAUNCH_FLAT(k_keyswitch32, dim3(batch), dim3(256), c->stream, out_d, in_d, (const int32_t *)c->ks_d, c->p.N,
                c->p.ks_n_out, c->p.ks_t, c->p.ks_basebit, batch);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}

int tfhe_amd_bootstrap(tfhe_amd_ctx *c, int32_t *out_d, int32_t mu, const int32_t *x_d, int batch) {
    if (!c || !out_d || !x_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    int rc = grow(c, &c->ws_lwe, &c->ws_lwe_bytes, (size_t)batch * (c->p.N + 1) * 4);
    if (rc) return rc;
    rc = tfhe_amd_bootstrap_woks(c, (int32_t *)c->ws_lwe, mu, x_d, batch);
    if (rc) return rc;
    return tfhe_amd_keyswitch(c, out_d, (const int32_t *)c->ws_lwe, batch);
}

static int streamed_plain(tfhe_amd_ctx *c, int32_t *out_d, int32_t mu, const int32_t *x_d, int batch) {
    const int n = c->p.n, N = c->p.N;
    int rc = grow(c, &c->ws_lwe, &c->ws_lwe_bytes, (size_t)batch * (N + 1) * 4);
    if (rc) return rc;
    rc = grow(c, &c->ws_acc, &c->ws_acc_bytes, (size_t)batch * 2 * N * 4);
    if (rc) return rc;
    BlindRotateArgs<int32_t> a;
    // 1. acc = (0, X^{-barb} * mu)           (0 CMux steps, store the accumulator)
    fill_common(c, a, c->bk, 0, 0, batch);
    a.rot = x_d + n;  // entry [n_steps = 0] of each row must be b
    a.rot_stride = n + 1;
    a.tv_const = mu;
    a.acc_io = (int32_t *)c->ws_acc;
    a.flags = BR_INIT_TESTVEC | BR_MODSWITCH | BR_TV_CONST;
    rc = launch_br32(c, a);
    // 2. one launch per CMux step, accumulators round-trip through HBM
    for (int i = 0; i < n && !rc; i++) {
        fill_common(c, a, c->bk, i, 1, batch);
        a.rot = x_d + i;
        a.rot_stride = n + 1;
        a.acc_io = (int32_t *)c->ws_acc;
        a.flags = BR_MODSWITCH;
        rc = launch_br32(c, a);
    }
    if (rc) return rc;
    // 3. sample extraction (0 steps)
    fill_common(c, a, c->bk, 0, 0, batch);
    a.rot = x_d;
    a.rot_stride = n + 1;
    a.acc_io = (int32_t *)c->ws_acc;
    a.lwe_out = (int32_t *)c->ws_lwe;
    a.flags = BR_EXTRACT;
    rc = launch_br32(c, a);
    if (rc) return rc;
    return tfhe_amd_keyswitch(c, out_d, (const int32_t *)c->ws_lwe, batch);
}

int tfhe_amd_bootstrap_streamed(tfhe_amd_ctx *c, int32_t *out_d, int32_t mu, const int32_t *x_d, int batch) {
    if (!c || !out_d || !x_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, c->p.torus_bits == 32, "Torus32 operation");
    if (!c->bk) return fail(c, TFHE_AMD_ERR_STATE, "no bootstrapping key");
    if (!c->ks_d) return fail(c, TFHE_AMD_ERR_STATE, "no key-switch key");
    if (batch == 0) return TFHE_AMD_OK;
    if (c->streamed_graph) {
        // workspace growth and per-kernel LDS attributes must not happen inside a capture: grow now,
        // and let the first call of each schedule variant run as plain launches
        if (int rc = grow(c, &c->ws_lwe, &c->ws_lwe_bytes, (size_t)batch * (c->p.N + 1) * 4)) return rc;
        if (int rc = grow(c, &c->ws_acc, &c->ws_acc_bytes, (size_t)batch * 2 * c->p.N * 4)) return rc;
        // one warm-up per (key-switch variant, blind-rotation kernel class): the class follows the batch size, and the
        // first launch of a kernel sets its LDS attribute (hipFuncSetAttribute), which must not happen inside a capture
        // (the schedule's two 0-step launches -- initialisation and extraction -- never take the split kernel: their class
        // follows the batch alone, so it is part of the key too)
        BlindRotateArgs<int32_t> probe, probe0;
        fill_common(c, probe, c->bk, 0, 1, batch);
        fill_common(c, probe0, c->bk, 0, 0, batch);
        const unsigned vbit = 1u << ((c->ks_force_gather ? 9 : 0) + 3 * br32_class(c, probe0) + br32_class(c, probe));
        if (!(c->streamed_warm & vbit)) {
            c->streamed_warm |= vbit;
            return streamed_plain(c, out_d, mu, x_d, batch);
        }
        const bool hit = c->sg.exec && c->sg.x == x_d && c->sg.out == out_d && c->sg.mu == mu && c->sg.batch == batch &&
                         c->sg.ks_gather == c->ks_force_gather;
        if (!hit) {
            drop_streamed_graph(c);
            hipGraph_t graph = nullptr;
            HIPCHECK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int rc = streamed_plain(c, out_d, mu, x_d, batch);
            const hipError_t e = hipStreamEndCapture(c->stream, &graph);  // always end the capture
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            HIPCHECK(c, e);
            hipGraphExec_t exec = nullptr;
            const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HIPCHECK(c, ei);
            c->sg.exec = exec;
            c->sg.x = x_d;
            c->sg.out = out_d;
            c->sg.mu = mu;
            c->sg.batch = batch;
            c->sg.ks_gather = c->ks_force_gather;
        }
        HIPCHECK(c, hipGraphLaunch((hipGraphExec_t)c->sg.exec, c->stream));
        return TFHE_AMD_OK;
    }
    return streamed_plain(c, out_d, mu, x_d, batch);
}

int tfhe_amd_bootstrap_host(tfhe_amd_ctx *c, int32_t *out, int32_t mu, const int32_t *x, int batch) {
    if (!c || !out || !x || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    const size_t bytes = (size_t)batch * (c->p.n + 1) * 4;
    void *in_d = nullptr, *out_d = nullptr;
    HIPCHECK(c, hipMalloc(&in_d, bytes));
    if (hipMalloc(&out_d, bytes) != hipSuccess) {
        (void)hipFree(in_d);
        return fail(c, TFHE_AMD_ERR_ALLOC, "hipMalloc");
    }
    int rc = tfhe_amd_memcpy_h2d(c, in_d, x, bytes);
    if (!rc) rc = tfhe_amd_bootstrap(c, (int32_t *)out_d, mu, (const int32_t *)in_d, batch);
    if (!rc) rc = tfhe_amd_memcpy_d2h(c, out, out_d, bytes);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(in_d);
    (void)hipFree(out_d);
    return rc;
}

int tfhe_amd_cb_bootstrap_woks(tfhe_amd_ctx *c, int64_t *lwe_out_d, int64_t mu, const int32_t *abar_d, int batch) {
    if (!c || !lwe_out_d || !abar_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    REQUIRE(c, c->p.torus_bits == 64, "circuitBootstrapWoKS works on Torus64");
    if (!c->bk) return fail(c, TFHE_AMD_ERR_STATE, "no bootstrapping key");
    if (batch == 0) return TFHE_AMD_OK;
    BlindRotateArgs<int64_t> a;
    fill_common(c, a, c->bk, 0, c->p.n, batch);
    a.rot = abar_d;
    a.rot_stride = c->p.n + 1;
    a.tv_const = mu / 2;
    a.out_b_add = mu / 2;
    a.lwe_out = lwe_out_d;
    a.flags = BR_INIT_TESTVEC | BR_EXTRACT | BR_TV_HALF;
    return launch_br64(c, a);
}

int tfhe_amd_modswitch(tfhe_amd_ctx *c, int32_t *out_d, const int32_t *x_d, int batch) {
    if (!c || !out_d || !x_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    const long long total = (long long)batch * (c->p.n + 1);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (c->generic)
        TFHE_LAUNCH_FLAT(kg_modswitch, dim3(blocks), dim3(256), c->stream, out_d, x_d, total, c->logn);
    else if (c->logn == 10)
        TFHE_LAUNCH_FLAT((k_modswitch<10>), dim3(blocks), dim3(256), c->stream, out_d, x_d, total);
    else
        TFHE_LAUNCH_FLAT((k_modswitch<11>), dim3(blocks), dim3(256), c->stream, out_d, x_d, total);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}

// exact external product: the reference's FFT-free backend (poc:285-316, CB/poc_karatsuba.cpp)
int tfhe_amd_extern_mul_exact(tfhe_amd_ctx *c, void *acc_d, const void *gsw_torus_d, int batch) {
    if (!c || !acc_d || !gsw_torus_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    if (c->generic)
        return c->p.torus_bits == 32 ? launch_exact_gen<int32_t>(c, acc_d, gsw_torus_d, batch)
                                     : launch_exact_gen<int64_t>(c, acc_d, gsw_torus_d, batch);
    if (c->p.torus_bits == 32)
        return c->logn == 10 ? launch_exact_t<int32_t, 10>(c, acc_d, gsw_torus_d, batch)
                             : launch_exact_t<int32_t, 11>(c, acc_d, gsw_torus_d, batch);
    return c->logn == 10 ? launch_exact_t<int64_t, 10>(c, acc_d, gsw_torus_d, batch)
                         : launch_exact_t<int64_t, 11>(c, acc_d, gsw_torus_d, batch);
}

// ---- Real96 high-precision anticyclic transforms (high-precision-anticyclic-fft/src/code.cpp)
static int hp_prepare(tfhe_amd_ctx *c) {
    if (c->generic) return fail(c, TFHE_AMD_ERR_PARAM, "Real96 transforms: N = 1024 or 2048 (the reference's code.cpp is N = 2048 only)");
    if (c->hp_tw_d) return TFHE_AMD_OK;
    const int n = 2 * c->p.N;
    std::vector<uint64_t> tw((size_t)2 * n * 4);
    if (tfhe_amd_hp_twiddles(n, tw.data(), tw.data() + (size_t)n * 4) != TFHE_AMD_OK)
        return fail(c, TFHE_AMD_ERR_PARAM, "Real96 twiddle tables");
    HIPCHECK(c, hipMalloc(&c->hp_tw_d, tw.size() * 8));
    HIPCHECK(c, hipMemcpyAsync(c->hp_tw_d, tw.data(), tw.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return TFHE_AMD_OK;
}
int tfhe_amd_hp_ifft(tfhe_amd_ctx *c, uint64_t *out_d, const int64_t *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    if (int rc = hp_prepare(c)) return rc;
    const HpCplx *pw = (const HpCplx *)c->hp_tw_d;
    if (c->logn == 10)
        TFHE_LAUNCH((k_hp_ifft<10>), dim3(batch), dim3(256), HpGeom<10>::lds_bytes, c->stream, (HpCplx *)out_d, in_d, pw, batch);
    else
        TFHE_LAUNCH((k_hp_ifft<11>), dim3(batch), dim3(256), HpGeom<11>::lds_bytes, c->stream, (HpCplx *)out_d, in_d, pw, batch);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}
int tfhe_amd_hp_fft(tfhe_amd_ctx *c, int64_t *out_d, const uint64_t *in_d, int batch) {
    if (!c || !out_d || !in_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    ENTER(c);
    if (batch == 0) return TFHE_AMD_OK;
    if (int rc = hp_prepare(c)) return rc;
    const HpCplx *pwbar = (const HpCplx *)c->hp_tw_d + 2 * c->p.N;
    if (c->logn == 10)
        TFHE_LAUNCH((k_hp_fft<10>), dim3(batch), dim3(256), HpGeom<10>::lds_bytes, c->stream, out_d, (const HpCplx *)in_d, pwbar, batch);
    else
        TFHE_LAUNCH((k_hp_fft<11>), dim3(batch), dim3(256), HpGeom<11>::lds_bytes, c->stream, out_d, (const HpCplx *)in_d, pwbar, batch);
    HIPCHECK(c, hipGetLastError());
    return TFHE_AMD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- circuit bootstrap
struct tfhe_amd_cb {
    tfhe_amd_cb_params p;
    tfhe_amd_ctx *c10;  // Torus32, ring N1: preKeySwitch (n_in = N1 -> n_out = n0)
    tfhe_amd_ctx *c2;   // Torus64, ring N2, n = n0: modswitch to 2*N2, blind rotation
    tfhe_amd_gsw *bk;
    int32_t *privks_d[2];  // reference layout (kept only where the matrix-core kernel does not cover the shape)
    int8_t *privksm_d[2];  // matrix-core layout (k_ks_mfma)
    void *ws_pre, *ws_abar, *ws_boot;
    size_t ws_pre_bytes, ws_abar_bytes, ws_boot_bytes;
    std::string err;
};

namespace {
int cb_fail(tfhe_amd_cb *cb, int code, const std::string &m) {
    if (cb) cb->err = m;
    return code;
}
int cb_pass(tfhe_amd_cb *cb, tfhe_amd_ctx *c, int rc) {
    if (rc && cb) cb->err = c->err;
    return rc;
}
// `count` samples in groups of `group`: sample s goes to out + (s % group)*stride_in_group + (s / group)*stride_of_group
template <int BB>
int launch_privks_t(tfhe_amd_cb *cb, int32_t *out_d, long long stride_in_group, long long stride_of_group, int group,
                    const int64_t *x_d, const int32_t *tab, int count) {
    // samples per tile / ints per thread and pass: see k_privks
    constexpr int TB = (BB == 3) ? 32 : 16, EPT = (BB == 3) ? 2 : 8;
    tfhe_amd_ctx *c = cb->c2;
    const int n2 = cb->p.N2, row = 2 * cb->p.N1;
    const int tiles = (count + TB - 1) / TB;
    // enough i-slices to fill the chip: ~4 workgroups per CU
    int slices = (1024 + tiles - 1) / tiles;
    if (slices < 1) slices = 1;
    int i_per_block = ((n2 + 1 + slices - 1) / slices + 63) / 64 * 64;
    slices = (n2 + 1 + i_per_block - 1) / i_per_block;
    // (fallback: only reached when t21 * basebit > 32, i.e. the digits span both words of each input;
    // the other shapes go through k_ks_mfma)
    TFHE_LAUNCH((k_privks<int64_t, TB, BB, EPT, 256>), dim3(tiles, slices), dim3(256), 0, c->stream, out_d,
                stride_in_group, stride_of_group, group, x_d, n2 + 1, tab, n2, cb->p.t21, row, count, i_per_block);
    if (hipGetLastError() != hipSuccess) return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "k_privks launch");
    return TFHE_AMD_OK;
}
int launch_privks(tfhe_amd_cb *cb, int32_t *out_d, long long stride_in_group, long long stride_of_group, int group, int u,
                  const int64_t *x_d, int count) {
    if (cb->privksm_d[u]) {  // one dense int8 contraction on the matrix cores (tfhe_kernels.h, k_ks_mfma)
        KsMfmaArgs a;
        memset(&a, 0, sizeof(a));
        a.out = out_d;
        a.stride_in_group = stride_in_group;
        a.stride_of_group = stride_of_group;
        a.group = group;
        a.x = x_d;
        a.bm = cb->privksm_d[u];
        a.x_stride = cb->p.N2 + 1;
        a.n_in = cb->p.N2 + 1;  // the b coefficient is input n2 of the digit loop (poc:676-680)
        a.t = cb->p.t21;
        a.row_ints = 2 * cb->p.N1;
        a.count = count;
        a.b_index = 0;
        a.b_col = -1;
        if (launch_ks_mfma<int64_t>(cb->c2->stream, a, cb->p.bb21)) return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "k_ks_mfma launch");
        return TFHE_AMD_OK;
    }
    const int32_t *tab = cb->privks_d[u];
    switch (cb->p.bb21) {
        case 1: return launch_privks_t<1>(cb, out_d, stride_in_group, stride_of_group, group, x_d, tab, count);
        case 2: return launch_privks_t<2>(cb, out_d, stride_in_group, stride_of_group, group, x_d, tab, count);
        default: return launch_privks_t<3>(cb, out_d, stride_in_group, stride_of_group, group, x_d, tab, count);
    }
}
}  // namespace

extern "C" {

int tfhe_amd_cb_create(const tfhe_amd_cb_params *p, int device, tfhe_amd_cb **out) {
    if (!p || !out) return TFHE_AMD_ERR_PARAM;
    *out = nullptr;
    if (p->bb21 < 1 || p->bb21 > 3 || p->t21 < 1 || p->t21 * p->bb21 > 63 || p->l1 < 1 || p->l1 * p->Bgbit1 > 63)
        return TFHE_AMD_ERR_PARAM;
    if (p->N1 % 4 != 0) return TFHE_AMD_ERR_PARAM;
    tfhe_amd_params p10 = {32, p->n0, p->N1, 1, p->l1, p->Bgbit1, p->t10, p->bb10, p->n0};
    tfhe_amd_params p2 = {64, p->n0, p->N2, 1, p->l2, p->Bgbit2, 0, 0, 0};
    tfhe_amd_cb *cb = new tfhe_amd_cb();
    cb->p = *p;
    cb->bk = nullptr;
    cb->privks_d[0] = cb->privks_d[1] = nullptr;
    cb->privksm_d[0] = cb->privksm_d[1] = nullptr;
    cb->ws_pre = cb->ws_abar = cb->ws_boot = nullptr;
    cb->ws_pre_bytes = cb->ws_abar_bytes = cb->ws_boot_bytes = 0;
    cb->c10 = cb->c2 = nullptr;
    int rc = tfhe_amd_ctx_create(&p10, device, &cb->c10);
    if (!rc) rc = tfhe_amd_ctx_create(&p2, device, &cb->c2);
    if (!rc) rc = tfhe_amd_set_stream(cb->c10, cb->c2->stream);  // one stream for the whole pipeline
    if (rc) {
        tfhe_amd_cb_destroy(cb);
        return rc;
    }
    *out = cb;
    return TFHE_AMD_OK;
}

void tfhe_amd_cb_destroy(tfhe_amd_cb *cb) {
    if (!cb) return;
    if (cb->c2) (void)hipSetDevice(cb->c2->device);  // the frees below belong to that device
    if (cb->c2) (void)hipStreamSynchronize(cb->c2->stream);
    if (cb->bk) tfhe_amd_gsw_free(cb->bk);
    for (int u = 0; u < 2; u++) {
        if (cb->privks_d[u]) (void)hipFree(cb->privks_d[u]);
        if (cb->privksm_d[u]) (void)hipFree(cb->privksm_d[u]);
    }
    if (cb->ws_pre) (void)hipFree(cb->ws_pre);
    if (cb->ws_abar) (void)hipFree(cb->ws_abar);
    if (cb->ws_boot) (void)hipFree(cb->ws_boot);
    if (cb->c10) tfhe_amd_ctx_destroy(cb->c10);  // borrowed c2's stream: destroyed first, never owns it
    if (cb->c2) tfhe_amd_ctx_destroy(cb->c2);
    delete cb;
}

const char *tfhe_amd_cb_last_error(const tfhe_amd_cb *cb) { return cb ? cb->err.c_str() : "null handle"; }
tfhe_amd_ctx *tfhe_amd_cb_ctx_lvl10(tfhe_amd_cb *cb) { return cb ? cb->c10 : nullptr; }
tfhe_amd_ctx *tfhe_amd_cb_ctx_lvl2(tfhe_amd_cb *cb) { return cb ? cb->c2 : nullptr; }

int tfhe_amd_cb_set_stream(tfhe_amd_cb *cb, void *s) {
    if (!cb) return TFHE_AMD_ERR_PARAM;
    int rc = cb_pass(cb, cb->c2, tfhe_amd_set_stream(cb->c2, s));
    if (!rc) rc = cb_pass(cb, cb->c10, tfhe_amd_set_stream(cb->c10, cb->c2->stream));
    return rc;
}
int tfhe_amd_cb_sync(tfhe_amd_cb *cb) {
    if (!cb) return TFHE_AMD_ERR_PARAM;
    return cb_pass(cb, cb->c2, tfhe_amd_sync(cb->c2));
}

int tfhe_amd_cb_load_preks(tfhe_amd_cb *cb, const int32_t *preks) {
    if (!cb || !preks) return TFHE_AMD_ERR_PARAM;
    return cb_pass(cb, cb->c10, tfhe_amd_load_keyswitch_key(cb->c10, preks));
}
static int cb_set_bk(tfhe_amd_cb *cb, tfhe_amd_gsw *g) {
    if (cb->bk) tfhe_amd_gsw_free(cb->bk);
    cb->bk = g;
    return cb_pass(cb, cb->c2, tfhe_amd_set_bootstrap_key(cb->c2, g));
}
int tfhe_amd_cb_load_bk_torus(tfhe_amd_cb *cb, const int64_t *bk) {
    if (!cb || !bk) return TFHE_AMD_ERR_PARAM;
    tfhe_amd_gsw *g = nullptr;
    int rc = cb_pass(cb, cb->c2, tfhe_amd_gsw_from_torus(cb->c2, bk, cb->p.n0, &g));
    return rc ? rc : cb_set_bk(cb, g);
}
int tfhe_amd_cb_load_bk_fft(tfhe_amd_cb *cb, const double *bkfft) {
    if (!cb || !bkfft) return TFHE_AMD_ERR_PARAM;
    tfhe_amd_gsw *g = nullptr;
    int rc = cb_pass(cb, cb->c2, tfhe_amd_gsw_from_fft(cb->c2, bkfft, cb->p.n0, &g));
    return rc ? rc : cb_set_bk(cb, g);
}
int tfhe_amd_cb_load_privks_plane(tfhe_amd_cb *cb, int u, const int32_t *plane) {
    if (!cb || !plane || u < 0 || u > 1) return TFHE_AMD_ERR_PARAM;
    if (hipSetDevice(cb->c2->device) != hipSuccess) return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "hipSetDevice");
    const size_t bytes = (size_t)(cb->p.N2 + 1) * cb->p.t21 * ((size_t)1 << cb->p.bb21) * 2 * cb->p.N1 * 4;
    tfhe_amd_ctx *c = cb->c2;
    // The plane is built in NEW buffers and swapped in only when complete: a failed (re)load leaves the handle
    // without this plane (calls then answer ERR_STATE), never with a half-written one.
    (void)hipStreamSynchronize(c->stream);  // no kernel may still read the buffers released below
    if (cb->privks_d[u]) (void)hipFree(cb->privks_d[u]);
    if (cb->privksm_d[u]) (void)hipFree(cb->privksm_d[u]);
    cb->privks_d[u] = nullptr;
    cb->privksm_d[u] = nullptr;
    int32_t *ref_d = nullptr;
    int8_t *mat_d = nullptr;
    auto bail = [&](int code, const char *what) {
        (void)hipStreamSynchronize(c->stream);  // the asynchronous upload / pack must not outlive their buffers
        if (ref_d) (void)hipFree(ref_d);
        if (mat_d) (void)hipFree(mat_d);
        return cb_fail(cb, code, what);
    };
    if (hipMalloc((void **)&ref_d, bytes) != hipSuccess) return bail(TFHE_AMD_ERR_ALLOC, "hipMalloc(privKS plane)");
    if (hipMemcpyAsync(ref_d, plane, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return bail(TFHE_AMD_ERR_DEVICE, "upload privKS plane");
    if (ks_mfma_supported(cb->p.t21, cb->p.bb21)) {
        // re-layout for the matrix-core kernel, then drop the reference-layout copy
        const int n_in = cb->p.N2 + 1, row = 2 * cb->p.N1;
        if (hipMalloc((void **)&mat_d, ks_mfma_bytes(n_in, cb->p.t21, cb->p.bb21, row)) != hipSuccess)
            return bail(TFHE_AMD_ERR_ALLOC, "hipMalloc(privKS plane, matrix-core layout)");
        if (ks_mfma_pack(c->stream, mat_d, ref_d, n_in, cb->p.t21, cb->p.bb21, row))
            return bail(TFHE_AMD_ERR_DEVICE, "k_ks_mfma_pack launch");
        if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(TFHE_AMD_ERR_DEVICE, "pack privKS plane");
        (void)hipFree(ref_d);
        cb->privksm_d[u] = mat_d;
        return TFHE_AMD_OK;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(TFHE_AMD_ERR_DEVICE, "upload privKS plane");
    cb->privks_d[u] = ref_d;
    return TFHE_AMD_OK;
}

int tfhe_amd_privks(tfhe_amd_cb *cb, int32_t *out_d, int u, const int64_t *x_d, int batch) {
    if (!cb || !out_d || !x_d || u < 0 || u > 1 || batch < 0) return TFHE_AMD_ERR_PARAM;
    if (hipSetDevice(cb->c2->device) != hipSuccess) return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "hipSetDevice");
    if (!cb->privks_d[u] && !cb->privksm_d[u]) return cb_fail(cb, TFHE_AMD_ERR_STATE, "privKS plane not loaded");
    if (batch == 0) return TFHE_AMD_OK;
    const long long row = 2LL * cb->p.N1;
    if (hipMemsetAsync(out_d, 0, (size_t)batch * row * 4, cb->c2->stream) != hipSuccess)
        return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "memset");
    return launch_privks(cb, out_d, row, 0, batch, u, x_d, batch);
}

int tfhe_amd_circuit_bootstrap(tfhe_amd_cb *cb, int32_t *out_d, const int32_t *x_d, int batch) {
    if (!cb || !out_d || !x_d || batch < 0) return TFHE_AMD_ERR_PARAM;
    if (hipSetDevice(cb->c2->device) != hipSuccess) return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "hipSetDevice");
    if (!cb->bk || !(cb->privks_d[0] || cb->privksm_d[0]) || !(cb->privks_d[1] || cb->privksm_d[1]) || !cb->c10->ks_d)
        return cb_fail(cb, TFHE_AMD_ERR_STATE, "preKS, bk and both privKS planes must be loaded");
    if (batch == 0) return TFHE_AMD_OK;
    const tfhe_amd_cb_params &p = cb->p;
    tfhe_amd_ctx *c2 = cb->c2;
    int rc = grow(c2, &cb->ws_pre, &cb->ws_pre_bytes, (size_t)batch * (p.n0 + 1) * 4);
    if (!rc) rc = grow(c2, &cb->ws_abar, &cb->ws_abar_bytes, (size_t)batch * (p.n0 + 1) * 4);
    if (!rc) rc = grow(c2, &cb->ws_boot, &cb->ws_boot_bytes, (size_t)p.l1 * batch * (p.N2 + 1) * 8);
    if (rc) return cb_pass(cb, c2, rc);
    // preKeySwitch lvl1 -> lvl0, then preModSwitch to [0, 2*N2)
    rc = cb_pass(cb, cb->c10, tfhe_amd_keyswitch(cb->c10, (int32_t *)cb->ws_pre, x_d, batch));
    if (!rc) rc = cb_pass(cb, c2, tfhe_amd_modswitch(c2, (int32_t *)cb->ws_abar, (const int32_t *)cb->ws_pre, batch));
    if (rc) return rc;
    const long long tlwe = 2LL * p.N1, out_stride = 2LL * p.l1 * tlwe;
    if (hipMemsetAsync(out_d, 0, (size_t)batch * out_stride * 4, c2->stream) != hipSuccess)
        return cb_fail(cb, TFHE_AMD_ERR_DEVICE, "memset");
    // the l1 blind rotations first (one per gadget level w, outputs kept side by side) ...
    int64_t *boot = (int64_t *)cb->ws_boot;
    const size_t boot_level = (size_t)batch * (p.N2 + 1);
    for (int w = 0; w < p.l1 && !rc; w++) {
        const int64_t mu1 = (int64_t)(1ull << (64 - (w + 1) * p.Bgbit1));  // poc:846
        rc = cb_pass(cb, c2, tfhe_amd_cb_bootstrap_woks(c2, boot + w * boot_level, mu1, (const int32_t *)cb->ws_abar, batch));
    }
    // ... then ONE private key switch per plane u over all l1*batch samples: the 1.3 GB plane is the
    // cost, and it is streamed once per tile of samples whatever level they belong to.
    // result->samples[u][w] of input b  <-  sample w*batch + b
    for (int u = 0; u <= 1 && !rc; u++)
        rc = launch_privks(cb, out_d + (size_t)u * p.l1 * tlwe, out_stride, tlwe, batch, u, boot, p.l1 * batch);
    return rc;
}

}  // extern "C"
