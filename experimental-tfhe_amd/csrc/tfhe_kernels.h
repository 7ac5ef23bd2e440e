// tfhe_kernels.h -- CDNA4 (gfx950) kernels of the TFHE bootstrapping hot path.
//
// One 64-lane wavefront owns one ciphertext for the whole blind rotation:
//   * the TLWE accumulator (k+1 = 2 polynomials) lives in the wave's slice of LDS,
//   * every polynomial transform is a wave-level anticyclic FFT over Z[X]/(X^N+1):
//     N/2 complex points, N/128 per lane, radix-2 butterflies grouped in three
//     register passes with two LDS transposes in between (no workgroup barrier:
//     a wave's DS instructions execute in order),
//   * the Fourier-domain accumulator of the external product never leaves registers,
//   * bootstrapping-key rows are read straight from HBM/L2 with one coalesced 16-byte
//     load per lane per complex value, in a layout fixed at key-upload time,
//   * the two waves that share a SIMD keep each other's pace through progress counters in LDS
//     and s_setprio (WaveLds::balance): the hardware otherwise favours the older wave and the
//     younger one finishes alone.
// (The Torus64 / N = 2048 instantiation of the circuit bootstrap keeps the accumulator in
// registers instead and runs one wave per SIMD: BlindRotateLds::ACCREG.)
//
// Arithmetic follows the reference's FMA assembly operation by operation (SURVEY.md
// App. A; CB/spqlios/spqlios-{i,}fft-fma.s, lagrangehalfc_impl_fma.s), so Torus32 /
// Torus64 results are bit-identical to the CPU path.  Compile with -ffp-contract=off:
// every fused multiply-add below is explicit.
//
// Reference functions covered (CB/ = circuit-bootstrapping/src/):
//   ifft_wave / fft_wave            spqlios-ifft-fma.s:9-275 / spqlios-fft-fma.s:9-285
//   load/convert, round/store       fft_processor_spqlios.cpp:27-170 (execute_*)
//   mac_row                         lagrangehalfc_impl_fma.s:78-135
//   decomposition                   tgsw_functions.cpp:224-337 ; poc_CircuitBootstrapping.cpp:492-515
//   (X^a-1) rotation                numeric_functions.cpp:304-323
//   cmux_step                       lwe_functions.cpp:328-333 + tgsw_functions.cpp:424-449
//   k_blind_rotate                  lwe_functions.cpp:337-430 (+ poc:530-659 test vector form)
#pragma once
#include "devport.h"

#include <type_traits>

// How many steps ahead of its use a butterfly's twiddle is requested from LDS (WaveFFT::ifft / fft, TWD), by kernel form:
// PAIR = two waves per SIMD (the 8-wave gate kernel), LONE = one wave per SIMD (Torus64 kernel, 4-wave gate form),
// SPLIT = the latency kernel's four lone waves.  Values decided by A/B on MI355X (profiles/r04_twiddle_distance.txt:
// distances 1 / 2 / 3 / 4: gate kernel 16.35 / 16.24 / 16.15 / 16.13 ms per 4096, Torus64 17.44 / 17.40 / 17.36 / 17.28 ms
// (6: 17.35, 8: 19.3 -- spills), one bootstrap 2.55 / 2.52 / 2.50 / 2.46 ms).
#ifndef TFHE_TWD_PAIR
#define TFHE_TWD_PAIR 4
#endif
#ifndef TFHE_TWD_LONE
#define TFHE_TWD_LONE 4
#endif
#ifndef TFHE_TWD_SPLIT
#define TFHE_TWD_SPLIT 4
#endif

// Workgroup barrier every K CMux steps of the blind rotation, by workgroup width (0 = none).  The 8 (or 4) waves of a workgroup
// stream the SAME key row per step (64 KB at N = 1024, l = 2) from L2, each for itself -- 9 TB/s of L2 -> L1 traffic per launch, and
// the chip, power-limited under this fp64 load, pays for it in clock.  Re-aligned by a barrier the waves run close enough for the
// CU's 32 KB L1 to serve the followers; they drift apart again within a few steps, and a barrier costs the wait for the slowest
// wave.  Round 5, MI355X, batch 4096 (profiles/r05_br_sync.txt): K = 16 -> L2 requests -32 %, cycles per CMux +2.0 %, shader clock
// 2.22 -> 2.28 GHz, kernel time -1.1 % on a box that holds 2.2 GHz and -2.2 % on one that holds 2.05; K = 1 / 2: slower (the
// barrier waits outweigh a -67 % in L2 requests); K = 4 .. 32: the same within noise.  Bit-identical (the barrier orders nothing
// the results depend on).
#ifndef TFHE_BR_SYNC8
#define TFHE_BR_SYNC8 16
#endif
#ifndef TFHE_BR_SYNC4
#define TFHE_BR_SYNC4 0
#endif

namespace tfhe {

template <int WAVES>
struct BR_SYNC_EVERY {
    static constexpr int value = (WAVES == 8) ? TFHE_BR_SYNC8 : (WAVES == 4 ? TFHE_BR_SYNC4 : 0);
};

// ------------------------------------------------------------------ geometry
template <int LOGN>
struct Geom {
    static constexpr int N = 1 << LOGN;       // ring degree
    static constexpr int NC = N / 2;          // complex points
    static constexpr int LB = LOGN - 1;       // log2(NC)
    static constexpr int PPL = NC / 64;       // complex points per lane: 8 (N=1024), 16 (N=2048)
    static constexpr int R = (PPL == 8) ? 3 : 4;
    static constexpr int CB = LB - 2 * R;     // index bits left for the last pass: 3 / 2
    static constexpr int XCH = NC + 64;       // doubles in the (padded) transpose buffer
    static constexpr int TW = 2 * NC;         // complex twiddles (2*NC-4 used)
    static_assert(PPL == 8 || PPL == 16, "N must be 1024 or 2048");
    static_assert(LB - R == 6, "one wave = 64 lanes");
    // twiddle table: [0,NC) = omega^j (twist); stage with half-size h at tw_base(h)+off,
    // value e^{2 pi i off / (2h)}  (= the reference's ifft table; its fft table is the conjugate)
    TFHE_HOST_DEVICE static constexpr int tw_base(int h) { return 2 * NC - 2 * h; }
    // lane/register -> point index j for the three register passes
    TFHE_HOST_DEVICE static int jA(int t, int m) { return t + 64 * m; }
    TFHE_HOST_DEVICE static int jB(int t, int m) {
        return ((t >> CB) << 6) + (m << CB) + (t & ((1 << CB) - 1));
    }
    TFHE_HOST_DEVICE static int jC(int t, int m) { return PPL * t + m; }
    // padded LDS index of point j for transpose 1 (A<->B) and transpose 2 (B<->C);
    // both directions of both transposes are bank-conflict free (LDS)
    TFHE_HOST_DEVICE static int idx1(int j) { return j + ((j >> 6) << CB); }
    TFHE_HOST_DEVICE static int idx2(int j) { return j + (j >> R); }
    // padded index of register m relative to register 0 of the same lane, for the four
    // (map, padding) pairs the transposes read with; independent of the lane (compile-time)
    enum ReadMap { RD_A1 = 0, RD_B1 = 1, RD_B2 = 2, RD_C2 = 3, RD_A2 = 4 };
    TFHE_HOST_DEVICE static constexpr int roff(int map, int m) {
        return map == RD_A1 ? m * (64 + (1 << CB))                 // idx1(jA): t + 64m + (m << CB)
             : map == RD_B1 ? (m << CB)                            // idx1(jB)
             : map == RD_B2 ? (m << CB) + ((m << CB) >> R)         // idx2(jB): 9m (N=1024), 4m + (m>>2) (N=2048)
             : map == RD_A2 ? m * (64 + (64 >> R))                 // idx2(jA): t + 64m + ((t + 64m) >> R)
                            : m;                                   // idx2(jC)
    }
};

// -------------------------------------------------------------- butterflies
// ifft stage (spqlios-ifft-fma.s:113-157): (a,b) -> (a+b, (a-b)*w), w=(c,s)
TFHE_DEVICE void dif_bfly(double &ar, double &ai, double &br, double &bi, double c, double s) {
    const double sr = ar + br, si = ai + bi;
    const double dr = ar - br, di = ai - bi;
    ar = sr;
    ai = si;
    br = __builtin_fma(-di, s, dr * c);
    bi = __builtin_fma(di, c, dr * s);
}
// fft stage (spqlios-fft-fma.s:189-234) with the fft table (c,-s) written through (c,s):
// tr = fma(-i1,-s, r1*c) = fma(i1,s,r1*c); ti = fma(i1,c, r1*(-s)) = fma(i1,c,-(r1*s)).
TFHE_DEVICE void dit_bfly(double &ar, double &ai, double &br, double &bi, double c, double s) {
    const double tr = __builtin_fma(bi, s, br * c);
    const double ti = __builtin_fma(bi, c, -(br * s));
    const double r0 = ar, i0 = ai;
    br = r0 - tr;
    bi = i0 - ti;
    ar = r0 + tr;
    ai = i0 + ti;
}

// The reference's fft table is the conjugate of its ifft table EXCEPT at the quarter turn
// (off == h/2): accurate_cos folds +n/4 to -cos(pi/2) = -6.1e-17 and -n/4 to +cos(pi/2)
// (spqlios-fft-impl.cpp:99-105), so there cos_fft = -cos_ifft.  One table serves both
// directions; the fft butterflies flip the sign of c on exactly those entries.
TFHE_DEVICE double flip_sign_if(double c, bool cond) { return cond ? -c : c; }

// ---- where the butterflies get their twiddles: the workgroup's LDS copy of the table (one
// ds_read_b128 per use).  (A variant holding the lane's twiddles in registers at one wave per SIMD
// was measured slower on MI355X: profiles/r02_variants.txt, "br1".)
template <int LOGN>
struct TwLds {
    using G = Geom<LOGN>;
    const double2 *tw;
    int t;
#define TFHE_TW_SRC tw
    TFHE_DEVICE double2 twist(int m) const { return TFHE_TW_SRC[G::jA(t, m)]; }
    TFHE_DEVICE double2 passA(int s, int m) const { return TFHE_TW_SRC[G::tw_base(64 * s) + t + 64 * (m & (s - 1))]; }
    TFHE_DEVICE double2 passB(int s, int m) const {
        return TFHE_TW_SRC[G::tw_base(s << G::CB) + ((m & (s - 1)) << G::CB) + (t & ((1 << G::CB) - 1))];
    }
    // the four twiddles of the stride-4 stage (N = 1024) are the same in every lane: held in scalar registers,
    // fetched once per kernel from the global table (uniform address -> s_load); 24 LDS reads less per CMux, +1 %
    double2 c4[4];
    TFHE_DEVICE void load_uniform(const double2 *global_table) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            c4[m].x = tfhe_uniform_load_f64(reinterpret_cast<const double *>(global_table + G::tw_base(4) + m), 0);
            c4[m].y = tfhe_uniform_load_f64(reinterpret_cast<const double *>(global_table + G::tw_base(4) + m), 1);
        }
    }
    TFHE_DEVICE double2 passC(int m) const { return c4[m & 3]; }
#undef TFHE_TW_SRC
};
template <int LOGN>
struct WaveFFT {
    using G = Geom<LOGN>;
    static constexpr int PPL = G::PPL;

    // The wave's transpose buffer with the LANE part of the padded index of every (map, padding) pair:
    // point index of register m under map M = lane[M] + roff(M, m), roff a compile-time constant, so a
    // transpose is 2 address registers + immediate offsets (and the compiler pairs the accesses).
    struct Xch {
        double *buf;
        int lane[5];
    };
    TFHE_DEVICE static Xch make_xch(double *buf, int t) {
        Xch x;
        x.buf = buf;
        const int hi = t >> G::CB, lo = t & ((1 << G::CB) - 1);
        x.lane[G::RD_A1] = t;                                           // idx1(jA)
        x.lane[G::RD_B1] = (hi << 6) + lo + (hi << G::CB);              // idx1(jB)
        x.lane[G::RD_B2] = (hi << 6) + lo + (hi << (6 - G::R));         // idx2(jB)
        x.lane[G::RD_C2] = (PPL + 1) * t;                               // idx2(jC)
        x.lane[G::RD_A2] = t + (t >> G::R);                             // idx2(jA)
#pragma unroll
        for (int k = 0; k < 5; k++) TFHE_OPAQUE(x.lane[k]);  // one register each, never re-derived from t
        return x;
    }
    // one wave-wide transpose through LDS: write with map WMAP, read with map RMAP
    template <int WMAP, int RMAP>
    TFHE_DEVICE static void transpose(double (&x)[PPL], const Xch &X) {
        double *w = X.buf + X.lane[WMAP], *r = X.buf + X.lane[RMAP];
#pragma unroll
        for (int m = 0; m < PPL; m++) w[G::roff(WMAP, m)] = x[m];
        TFHE_WAVE_FENCE();
#pragma unroll
        for (int m = 0; m < PPL; m++) x[m] = r[G::roff(RMAP, m)];
        TFHE_WAVE_FENCE();
    }

    // The same transpose on COMPLEX points (16-byte elements: re and im of a point side by side), for a
    // buffer of XCH double2: one ds_write_b128 / ds_read_b128 per point instead of two 8-byte accesses per
    // plane -- the reads cost half the LDS cycles of the paired 8-byte form (MI355X_MICROARCH.md, LDS
    // table).  The padded maps are conflict-free in 16-byte units too (tools/lds_conflicts.py).
    template <int WMAP, int RMAP>
    TFHE_DEVICE static void transpose_cplx(double (&xr)[PPL], double (&xi)[PPL], const Xch &X) {
        double2 *w = reinterpret_cast<double2 *>(X.buf) + X.lane[WMAP], *r = reinterpret_cast<double2 *>(X.buf) + X.lane[RMAP];
        // (one ds_write_b128 per point: two ds_write_b64 -- 6 LDS cycles each against 13 -- measured equal, 17.15 vs 17.14 ms
        // per 4096 blind rotations on one box, round 3)
#pragma unroll
        for (int m = 0; m < PPL; m++) w[G::roff(WMAP, m)] = make_double2(xr[m], xi[m]);
        TFHE_WAVE_FENCE();
#pragma unroll
        for (int m = 0; m < PPL; m++) {
            const double2 v = r[G::roff(RMAP, m)];
            xr[m] = v.x;
            xi[m] = v.y;
        }
        TFHE_WAVE_FENCE();
    }
    // CPLX: the caller's buffer holds XCH double2 (the blind-rotation kernel at N = 1024, where LDS has room)
    template <int WMAP, int RMAP, bool CPLX>
    TFHE_DEVICE static void transpose_point(double (&xr)[PPL], double (&xi)[PPL], const Xch &X) {
        if (CPLX) {
            transpose_cplx<WMAP, RMAP>(xr, xi, X);
        } else {
            transpose<WMAP, RMAP>(xr, X);
            transpose<WMAP, RMAP>(xi, X);
        }
    }

    // ---- twiddle schedule of a register pass: entry k = (stage k / (PPL/2), butterfly group k % (PPL/2)); DESC: strides
    // PPL/2 .. 1 (ifft), else 1 .. PPL/2 (fft).  PASS_LEN entries per pass.
    static constexpr int PASS_LEN = (PPL / 2) * G::R;
    template <bool DESC>
    TFHE_HOST_DEVICE static constexpr int pass_s(int k) {
        return DESC ? ((PPL / 2) >> (k / (PPL / 2))) : (1 << (k / (PPL / 2)));
    }
    TFHE_HOST_DEVICE static constexpr int pass_m(int s, int k) { return ((k % (PPL / 2)) / s) * 2 * s + ((k % (PPL / 2)) % s); }

    // coefficient -> Lagrange for NP polynomials at once (every twiddle is read from LDS once
    // and used by all NP).  In: lane t register m = point jA(t,m) (re = coef j, im = coef j+N/2).
    // Out: register m = position jC(t,m) of the reference's output order.
    // TWD: how many steps ahead of its use a twiddle is requested from LDS.  1 is enough where a second wave shares the
    // SIMD (its instructions fill the rest of the latency); a LONE wave (one per SIMD: the Torus64 kernel, the latency
    // kernel, the 4-wave form) has only its own butterflies to put in front of a read -- 8 fp64 instructions per polynomial,
    // ~40 cycles, against an LDS latency of ~100+ -- and waited at every butterfly: see DESIGN.md 2 (round 4).
    // The first twiddles of a pass are requested BEFORE the transpose in front of it: behind it (the fences keep LDS reads
    // in program order) they would be the LAST of the wave's outstanding LDS operations, and the first butterfly would wait
    // with lgkmcnt(0) for every polynomial's transpose instead of for the first polynomial's reads.
    template <int NP, class TW, bool CPLX = false, int TWD = 1>
    TFHE_DEVICE static void ifft(double (&xr)[NP][PPL], double (&xi)[NP][PPL], const TW &tw, const Xch &X, int t) {
        static_assert(TWD >= 1 && TWD <= 8, "twiddle prefetch distance");
        // twist by omega^j (spqlios-ifft-fma.s:63-78), then pass A: strides 64*s, s = PPL/2 .. 1 -- one schedule
        constexpr int LA = PPL + PASS_LEN;
        auto tw_a = [&](int k) {
            return k < PPL ? tw.twist(k) : tw.passA(pass_s<true>(k - PPL), pass_m(pass_s<true>(k - PPL), k - PPL));
        };
        double2 ra[TWD];
#pragma unroll
        for (int k = 0; k < TWD; k++) ra[k] = tw_a(k);
#pragma unroll
        for (int k = 0; k < LA; k++) {
            const double2 w = ra[k % TWD];
            if (k + TWD < LA) ra[k % TWD] = tw_a(k + TWD);
            if (k < PPL) {
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const double r = xr[p][k], i = xi[p][k];
                    xr[p][k] = __builtin_fma(-i, w.y, r * w.x);
                    xi[p][k] = __builtin_fma(i, w.x, r * w.y);
                }
            } else {
                const int s = pass_s<true>(k - PPL), m = pass_m(s, k - PPL);
#pragma unroll
                for (int p = 0; p < NP; p++) dif_bfly(xr[p][m], xi[p][m], xr[p][m + s], xi[p][m + s], w.x, w.y);
            }
        }
        // pass B: strides s<<CB
        auto tw_b = [&](int k) { return tw.passB(pass_s<true>(k), pass_m(pass_s<true>(k), k)); };
        double2 rb[TWD];
#pragma unroll
        for (int k = 0; k < TWD; k++) rb[k] = tw_b(k);
#pragma unroll
        for (int p = 0; p < NP; p++) transpose_point<G::RD_A1, G::RD_B1, CPLX>(xr[p], xi[p], X);
#pragma unroll
        for (int k = 0; k < PASS_LEN; k++) {
            const double2 w = rb[k % TWD];
            if (k + TWD < PASS_LEN) rb[k % TWD] = tw_b(k + TWD);
            const int s = pass_s<true>(k), m = pass_m(s, k);
#pragma unroll
            for (int p = 0; p < NP; p++) dif_bfly(xr[p][m], xi[p][m], xr[p][m + s], xi[p][m + s], w.x, w.y);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) transpose_point<G::RD_B2, G::RD_C2, CPLX>(xr[p], xi[p], X);
        // pass C: (N=1024 only) stride 4 general stage, then size-4 and size-2 steps
        if (G::CB == 3) {
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                if (m & 4) continue;
                const double2 w = tw.passC(m);
#pragma unroll
                for (int p = 0; p < NP; p++) dif_bfly(xr[p][m], xi[p][m], xr[p][m + 4], xi[p][m + 4], w.x, w.y);
            }
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int b = 0; b < PPL; b += 4) {  // spqlios-ifft-fma.s:194-213
                const double r0 = xr[p][b], r1 = xr[p][b + 1], r2 = xr[p][b + 2], r3 = xr[p][b + 3];
                const double i0 = xi[p][b], i1 = xi[p][b + 1], i2 = xi[p][b + 2], i3 = xi[p][b + 3];
                xr[p][b] = r0 + r2;
                xr[p][b + 1] = r1 + r3;
                xr[p][b + 2] = r0 - r2;
                xr[p][b + 3] = i3 - i1;
                xi[p][b] = i0 + i2;
                xi[p][b + 1] = i1 + i3;
                xi[p][b + 2] = i0 - i2;
                xi[p][b + 3] = r1 - r3;
            }
#pragma unroll
            for (int b = 0; b < PPL; b += 2) {  // :247-263
                const double r0 = xr[p][b], r1 = xr[p][b + 1], i0 = xi[p][b], i1 = xi[p][b + 1];
                xr[p][b] = r0 + r1;
                xr[p][b + 1] = r0 - r1;
                xi[p][b] = i0 + i1;
                xi[p][b + 1] = i0 - i1;
            }
        }
    }

    // Lagrange -> coefficient for NP polynomials (caller has applied the 2/N scale).
    // In: register m = position jC(t,m).  Out: register m = point jA(t,m).  TWD: as in ifft.
    template <int NP, class TW, bool CPLX = false, int TWD = 1>
    TFHE_DEVICE static void fft(double (&xr)[NP][PPL], double (&xi)[NP][PPL], const TW &tw, const Xch &X, int t) {
        static_assert(TWD >= 1 && TWD <= 8, "twiddle prefetch distance");
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int b = 0; b < PPL; b += 2) {  // spqlios-fft-fma.s:79-95
                const double r0 = xr[p][b], r1 = xr[p][b + 1], i0 = xi[p][b], i1 = xi[p][b + 1];
                xr[p][b] = r0 + r1;
                xr[p][b + 1] = r0 - r1;
                xi[p][b] = i0 + i1;
                xi[p][b + 1] = i0 - i1;
            }
#pragma unroll
            for (int b = 0; b < PPL; b += 4) {  // :134-152
                const double r0 = xr[p][b], r1 = xr[p][b + 1], r2 = xr[p][b + 2], r3 = xr[p][b + 3];
                const double i0 = xi[p][b], i1 = xi[p][b + 1], i2 = xi[p][b + 2], i3 = xi[p][b + 3];
                xr[p][b] = r0 + r2;
                xr[p][b + 1] = r1 + i3;
                xr[p][b + 2] = r0 - r2;
                xr[p][b + 3] = r1 - i3;
                xi[p][b] = i0 + i2;
                xi[p][b + 1] = i1 - r3;
                xi[p][b + 2] = i0 - i2;
                xi[p][b + 3] = i1 + r3;
            }
        }
        if (G::CB == 3) {
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                if (m & 4) continue;
                const double2 w = tw.passC(m);
                const double wc = (m & 3) == 2 ? -w.x : w.x;  // quarter turn
#pragma unroll
                for (int p = 0; p < NP; p++) dit_bfly(xr[p][m], xi[p][m], xr[p][m + 4], xi[p][m + 4], wc, w.y);
            }
        }
        // pass B: strides (s << CB), s = 1 .. PPL/2; its first twiddles are requested before the transpose (see ifft)
        auto tw_b = [&](int k) { return tw.passB(pass_s<false>(k), pass_m(pass_s<false>(k), k)); };
        double2 rb[TWD];
#pragma unroll
        for (int k = 0; k < TWD; k++) rb[k] = tw_b(k);
#pragma unroll
        for (int p = 0; p < NP; p++) transpose_point<G::RD_C2, G::RD_B2, CPLX>(xr[p], xi[p], X);
        const int c = t & ((1 << G::CB) - 1);
#pragma unroll
        for (int k = 0; k < PASS_LEN; k++) {
            const double2 w = rb[k % TWD];
            if (k + TWD < PASS_LEN) rb[k % TWD] = tw_b(k + TWD);
            const int s = pass_s<false>(k), m = pass_m(s, k);
            // quarter turn: off == (s<<CB)/2
            const bool mq = (s == 1) ? true : ((m & (s - 1)) == s / 2);
            const bool lq = (s == 1) ? (c == (1 << (G::CB - 1))) : (c == 0);
            const double wc = mq ? flip_sign_if(w.x, lq) : w.x;
#pragma unroll
            for (int p = 0; p < NP; p++) dit_bfly(xr[p][m], xi[p][m], xr[p][m + s], xi[p][m + s], wc, w.y);
        }
        // pass A (strides 64 s, s = 1 .. PPL/2), then the final twist by conj(omega^j), four rounded products
        // (spqlios-fft-fma.s:255-274): re' = re*c - im*(-s) = re*c + im*s ; im' = re*(-s) + im*c = im*c - re*s
        constexpr int LA = PASS_LEN + PPL;
        auto tw_a = [&](int k) {
            return k < PASS_LEN ? tw.passA(pass_s<false>(k), pass_m(pass_s<false>(k), k)) : tw.twist(k - PASS_LEN);
        };
        double2 ra[TWD];
#pragma unroll
        for (int k = 0; k < TWD; k++) ra[k] = tw_a(k);
#pragma unroll
        for (int p = 0; p < NP; p++) transpose_point<G::RD_B1, G::RD_A1, CPLX>(xr[p], xi[p], X);
#pragma unroll
        for (int k = 0; k < LA; k++) {
            const double2 w = ra[k % TWD];
            if (k + TWD < LA) ra[k % TWD] = tw_a(k + TWD);
            if (k < PASS_LEN) {
                const int s = pass_s<false>(k), m = pass_m(s, k);
                // quarter turn: off == 32*s
                const bool mq = (s == 1) ? true : ((m & (s - 1)) == s / 2);
                const bool lq = (s == 1) ? (t == 32) : (t == 0);
                const double wc = mq ? flip_sign_if(w.x, lq) : w.x;
#pragma unroll
                for (int p = 0; p < NP; p++) dit_bfly(xr[p][m], xi[p][m], xr[p][m + s], xi[p][m + s], wc, w.y);
            } else {
                const int m = k - PASS_LEN;
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const double r = xr[p][m], i = xi[p][m];
                    const double rc = r * w.x, rs = r * w.y, ic = i * w.x, is = i * w.y;
                    xr[p][m] = rc + is;
                    xi[p][m] = ic - rs;
                }
            }
        }
    }
};

// ------------------------------------------------------------- torus helpers
template <typename T>
struct Torus;
template <>
struct Torus<int32_t> {
    using U = uint32_t;
    static constexpr int BITS = 32;
    static constexpr bool HAS_FAST = true;
    // int32_t(int64_t(x)), fft_processor_spqlios.cpp:102 (truncate toward zero, wrap).  Beyond the int64 range
    // the C expression is undefined; the reference as compiled for x86 (cvttsd2si: 0x8000000000000000) yields 0,
    // which is what this returns too (such values do not occur: the external product stays below 2^52)
    TFHE_DEVICE static int32_t from_double(double x) {
        return (__builtin_fabs(x) < 0x1p63) ? (int32_t)(int64_t)x : 0;
    }
    // The same value for |x| < 2^51 in 2 fp64 operations instead of 5: t = trunc(x) is an integer
    // below 2^51, so t + 1.5*2^52 is exact (ulp 1) and carries t mod 2^32 in its low word.
    // `guard` ORs the high words of t; guard_ok() then bounds every exponent seen with one compare
    // (an OR can only over-estimate), and callers fall back to from_double when it fails.
    TFHE_DEVICE static int32_t from_double_fast(double x, uint32_t &guard) {
        const double tr = __builtin_trunc(x);
        guard |= (uint32_t)((uint64_t)__builtin_bit_cast(int64_t, tr) >> 32);
        const double y = tr + 0x1.8p52;
        return (int32_t)(uint32_t)(uint64_t)__builtin_bit_cast(int64_t, y);
    }
    TFHE_DEVICE static bool guard_ok(uint32_t guard) { return (guard & 0x7FF00000u) < 0x43200000u; }
    TFHE_DEVICE static double to_double(int32_t v) { return (double)v; }
};
template <>
struct Torus<int64_t> {
    using U = uint64_t;
    static constexpr int BITS = 64;
    static constexpr bool HAS_FAST = true;
    // trunc(x) mod 2^64 for |x| < 2^83 in 5 fp64 + 3 integer operations instead of ~20 integer ones (the
    // bit-field form below).  t = trunc(x);  A = rne(t / 2^32) read from the low word of t * 2^-32 + 1.5 * 2^52
    // (exact while |t / 2^32| < 2^51);  B = t - A * 2^32 is an integer with |B| <= 2^31, exact in one fma, and
    // B + 1.5 * 2^52 carries B mod 2^32 in its low word.  t mod 2^64 = ((A mod 2^32) - (B < 0)) * 2^32 + (B mod 2^32).
    // `guard` ORs the high words of t (bounds every exponent seen, as in the Torus32 form).
    TFHE_DEVICE static int64_t from_double_fast(double x, uint32_t &guard) {
        const double tr = __builtin_trunc(x);
        guard |= (uint32_t)((uint64_t)__builtin_bit_cast(int64_t, tr) >> 32);
        const double y1 = __builtin_fma(tr, 0x1p-32, 0x1.8p52);
        const double a = y1 - 0x1.8p52;
        const double b = __builtin_fma(a, -0x1p32, tr);
        const double y2 = b + 0x1.8p52;
        const uint32_t lo = (uint32_t)(uint64_t)__builtin_bit_cast(int64_t, y2);
        // B < 0 <=> the mantissa of y2 (= 2^51 + B) has bit 51 clear (taken from y2, not from the sign of b: b may be -0)
        const uint32_t nonneg = (uint32_t)((uint64_t)__builtin_bit_cast(int64_t, y2) >> 51) & 1u;
        const uint32_t hi = (uint32_t)(uint64_t)__builtin_bit_cast(int64_t, y1) + nonneg - 1u;
        return (int64_t)(((uint64_t)hi << 32) | lo);
    }
    TFHE_DEVICE static bool guard_ok(uint32_t guard) { return (guard & 0x7FF00000u) < 0x45200000u; }  // |t| < 2^83
    // fft_processor_spqlios.cpp:131-142: mantissa shifted by (exponent-1075), truncation,
    // modulo 2^64; shifts of 64 or more (|x| < 2^-11: undefined in the reference) give 0.
    TFHE_DEVICE static int64_t from_double(double x) {
        // Branch-free: a right shift of 53..63 already gives 0 (mant < 2^53), so the count is clamped
        // instead of tested; the sign is applied as (v ^ s) - s with s = 0 or ~0.
        const uint64_t bits = (uint64_t)__builtin_bit_cast(int64_t, x);
        const uint64_t mant = (bits & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
        const int trans = (int)((bits >> 52) & 0x7FF) - 1075;
        const int rsh = -trans < 63 ? -trans : 63;
        const uint64_t left = trans > 63 ? 0 : (mant << (trans & 63));
        const uint64_t right = mant >> (rsh & 63);
        const uint64_t v = trans > 0 ? left : right;
        const uint64_t s = (uint64_t)(__builtin_bit_cast(int64_t, x) >> 63);
        return (int64_t)((v ^ s) - s);
    }
    TFHE_DEVICE static double to_double(int64_t v) { return (double)v; }  // round to nearest even
};

// gadget decomposition parameters, computed on the host (tgsw_functions.cpp:24-36 for
// Torus32: no rounding bit; poc:349-350 for Torus64: with rounding bit)
struct Gadget {
    uint64_t offset;
    uint64_t flip;  // Bg/2 at every digit position (see ifft_mac_digits)
    int32_t Bgbit;
    int32_t l;
};

// one coefficient of X^a * p, a in [0, 2N)   (numeric_functions.cpp:327-347)
template <typename T, int LOGN>
TFHE_DEVICE T rot_only(const T *p, int i, int a) {
    using U = typename Torus<T>::U;
    constexpr int N = 1 << LOGN;
    const int idx = (i - a) & (2 * N - 1);
    const U src = (U)p[idx & (N - 1)];
    return (T)((idx & N) ? (U)(0 - src) : src);
}

// --------------------------------------------------------------- CMux step
// Shared (LDS) state of one wave.  Polynomial q of the accumulator starts at LDS byte offset
// acc_lds + q * N * sizeof(T), a multiple of its own size (BlindRotateLds puts the accumulators
// first; the kernel checks the segment base), which lets the rotated read form an address with one
// AND-OR.
template <typename T, int LOGN>
struct WaveLds {
    unsigned char *smem;  // workgroup's dynamic LDS block
    uint32_t acc_lds;     // LDS byte offset of acc (a multiple of the polynomial size)
    T *acc;               // [2][N] accumulator
    typename WaveFFT<LOGN>::Xch xch;  // [Geom::XCH] transpose buffer + lane maps
    TwLds<LOGN> tw;       // twiddle source
    // Issue balance between the two waves of a SIMD (8-wave workgroups).  Left alone, the SIMD favours one of
    // its two waves (the older): measured on MI355X, waves 0-3 of a workgroup finished their 630 CMux steps in
    // 5.7 ms and waves 4-7 in 8.8 ms, the second wave of every SIMD running its last third alone at half the
    // SIMD's throughput.  Each wave therefore publishes its progress in LDS and takes the high issue priority
    // (s_setprio) only while it is not ahead of its partner: both finish together (7.9 ms), -7 % kernel time.
    uint32_t progress_lds;   // LDS byte offset of int[WAVES]: steps done by each wave of the workgroup
    int self, partner;       // partner == self: alone on its SIMD, nothing to do
    TFHE_DEVICE void balance(int step, int lane) const {
        if (partner == self) return;
        if (lane == 0) tfhe_lds_poke32(progress_lds + 4u * (uint32_t)self, (uint32_t)step);
        const int other = TFHE_UNIFORM((int)tfhe_lds_peek32(progress_lds + 4u * (uint32_t)partner));
        if (other < step)
            TFHE_SETPRIO(0);
        else
            TFHE_SETPRIO(3);
    }
};

// Fourier-domain multiply-accumulate of one decomposed limb with one key row
// (lagrangehalfc_impl_fma.s:96-107), bk already in registers.  FIRST: the accumulator is still the
// +0 of tLweFFTClear (tgsw_functions.cpp:438) -- fma(a, b, -(+0)) and fma(a, b, +0) are a*b up to the
// sign of a zero result (the one tolerated difference), so the row needs no zeroed input.
template <int PPL, bool FIRST>
TFHE_DEVICE void mac_row(double (&fr)[2][PPL], double (&fi)[2][PPL], const double (&xr)[PPL],
                         const double (&xi)[PPL], const double2 (&bk)[2][PPL]) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
#pragma unroll
        for (int m = 0; m < PPL; m++) {
            const double ar = xr[m], ai = xi[m], br = bk[q][m].x, bi = bk[q][m].y;
            const double tneg = FIRST ? ai * bi : __builtin_fma(ai, bi, -fr[q][m]);
            fr[q][m] = __builtin_fma(ar, br, -tneg);
            const double u = FIRST ? ar * bi : __builtin_fma(ar, bi, fi[q][m]);
            fi[q][m] = __builtin_fma(ai, br, u);
        }
    }
}

// ND consecutive gadget digits (d .. d+ND-1) of one accumulator polynomial: extract, transform
// together, multiply-accumulate with their key rows in digit order (the MAC chain is sequential in
// the row index: lagrangehalfc AddMul accumulates in place, tgsw_functions.cpp:441-443).
// BGC: Bgbit when it is known at compile time (0: read gd.Bgbit) -- one v_bfe_i32 per digit.
// one output polynomial q of the same multiply-accumulate (key half-row in registers)
template <int PPL, bool FIRST = false>
TFHE_DEVICE void mac_half_row(double (&fr)[PPL], double (&fi)[PPL], const double (&xr)[PPL], const double (&xi)[PPL],
                              const double2 (&bk)[PPL]) {
#pragma unroll
    for (int m = 0; m < PPL; m++) {
        const double ar = xr[m], ai = xi[m], br = bk[m].x, bi = bk[m].y;
        const double tneg = FIRST ? ai * bi : __builtin_fma(ai, bi, -fr[m]);
        fr[m] = __builtin_fma(ar, br, -tneg);
        const double u = FIRST ? ar * bi : __builtin_fma(ar, bi, fi[m]);
        fi[m] = __builtin_fma(ai, br, u);
    }
}

// HALFROW (register-resident accumulator, PPL = 16): the key row is fetched in two halves of 64 registers
// (the half for output polynomial 0 underneath the transform, the other half while the first is consumed)
// instead of 128 at once.
template <typename T, int LOGN, int ND, int BGC, bool FIRST, bool CPLX, bool HALFROW = false, int TWD = 1>
TFHE_DEVICE void ifft_mac_digits(const WaveLds<T, LOGN> &w, const double2 *__restrict__ bkrow, int row0, int d0,
                                 const typename Torus<T>::U (&lo)[Geom<LOGN>::PPL],
                                 const typename Torus<T>::U (&hi)[Geom<LOGN>::PPL], const Gadget &gd,
                                 double (&fr)[2][Geom<LOGN>::PPL], double (&fi)[2][Geom<LOGN>::PPL], int t) {
    using U = typename Torus<T>::U;
    constexpr int PPL = Geom<LOGN>::PPL;
    const int Bgbit = BGC ? BGC : gd.Bgbit;
    double2 bk[HALFROW ? 1 : 2][PPL];  // key row of the first digit, fetched underneath the transform
    // wave-uniform row pointer + 32-bit lane offset: one address register for the whole row
    const unsigned char *kb = reinterpret_cast<const unsigned char *>(bkrow);
    const uint32_t lane16 = (uint32_t)t * 16u;
    // buffer addressing: resource descriptor of the row (scalar registers) + ONE 32-bit lane offset + a scalar
    // window offset + immediate.  With flat 64-bit addresses hipcc spent two VALU adds and a register pair per
    // 8 KB window of the row (-30 VALU instructions per CMux, +0.9 %).
    const TFHE_BUFFER_RSRC rsrc = TFHE_MAKE_BUFFER_RSRC(kb);
#define TFHE_BK(row, qq, m) tfhe_buffer_load_d2(rsrc, lane16, (uint32_t)((((row) * 2 + (qq)) * PPL + (m)) * 64) * 16u)
#pragma unroll
    for (int qq = 0; qq < (HALFROW ? 1 : 2); qq++)
#pragma unroll
        for (int m = 0; m < PPL; m++) bk[qq][m] = TFHE_BK(row0, qq, m);
    double xr[ND][PPL], xi[ND][PPL];
#pragma unroll
    for (int e = 0; e < ND; e++) {
        const int decal = Torus<T>::BITS - (d0 + e + 1) * Bgbit;
        // lo/hi arrive with the top bit of every digit field flipped (Gadget::flip), so the field read as a
        // signed Bgbit-bit number IS (field - Bg/2): one v_bfe_i32 per digit
        if constexpr (Torus<T>::BITS == 32) {
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                xr[e][m] = (double)(((int32_t)((uint32_t)lo[m] << (32 - decal - Bgbit))) >> (32 - Bgbit));
                xi[e][m] = (double)(((int32_t)((uint32_t)hi[m] << (32 - decal - Bgbit))) >> (32 - Bgbit));
            }
        } else {
            // the same on 64 bits, from the field's 32-bit word -- or, where it straddles the words, from a 32-bit
            // window on both.  Which of the three it is depends on the (wave-uniform) digit index only: one
            // scalar branch around the whole polynomial.
            auto extract = [&](auto mode) {
                constexpr int MODE = decltype(mode)::value;
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    const uint64_t a = (uint64_t)lo[m], b = (uint64_t)hi[m];
                    if (MODE == 0) {
                        xr[e][m] = (double)TFHE_SBFE((uint32_t)(a >> 32), decal - 32, Bgbit);
                        xi[e][m] = (double)TFHE_SBFE((uint32_t)(b >> 32), decal - 32, Bgbit);
                    } else if (MODE == 1) {
                        xr[e][m] = (double)TFHE_SBFE(TFHE_ALIGNBIT((uint32_t)(a >> 32), (uint32_t)a, decal), 0, Bgbit);
                        xi[e][m] = (double)TFHE_SBFE(TFHE_ALIGNBIT((uint32_t)(b >> 32), (uint32_t)b, decal), 0, Bgbit);
                    } else {
                        xr[e][m] = (double)TFHE_SBFE((uint32_t)a, decal, Bgbit);
                        xi[e][m] = (double)TFHE_SBFE((uint32_t)b, decal, Bgbit);
                    }
                }
            };
            if (decal >= 32) {
                TFHE_KEEP_BRANCH();
                extract(std::integral_constant<int, 0>{});
            } else if (decal + Bgbit > 32) {
                TFHE_KEEP_BRANCH();
                extract(std::integral_constant<int, 1>{});
            } else {
                TFHE_KEEP_BRANCH();
                extract(std::integral_constant<int, 2>{});
            }
        }
    }
    WaveFFT<LOGN>::template ifft<ND, TwLds<LOGN>, CPLX, TWD>(xr, xi, w.tw, w.xch, t);
    if constexpr (HALFROW) {
        static_assert(!HALFROW || ND == 1, "half-row form: one digit at a time");
        mac_half_row<PPL, FIRST>(fr[0], fi[0], xr[0], xi[0], bk[0]);
        // (requesting this half-row earlier -- in front of the first MAC, or with the first half-row under the transform --
        // was measured in round 4: 19.07 / 21.82 ms against 18.27 per 1024 x 500 CMux.  Its 64 registers do not exist:
        // accumulator 128 + Fourier accumulator 128 + rotated coefficients 64 + transform 64 + one half-row 64 already
        // fill 448 of the wave's 512, and the kernel then spills to scratch.  profiles/r04_cb_experiments.txt)
#pragma unroll
        for (int m = 0; m < PPL; m++) bk[0][m] = TFHE_BK(row0, 1, m);
        mac_half_row<PPL, FIRST>(fr[1], fi[1], xr[0], xi[0], bk[0]);
    } else {
#pragma unroll
        for (int e = 0; e < ND; e++) {
            if (e > 0) {
#pragma unroll
                for (int qq = 0; qq < 2; qq++)
#pragma unroll
                    for (int m = 0; m < PPL; m++) bk[qq][m] = TFHE_BK(row0 + e, qq, m);
            }
            if (FIRST && e == 0)
                mac_row<PPL, true>(fr, fi, xr[e], xi[e], bk);
            else
                mac_row<PPL, false>(fr, fi, xr[e], xi[e], bk);
        }
    }
#undef TFHE_BK
}

// The 2*PPL coefficients (j = t + 64m and j + N/2) of polynomial q of (X^a - 1) * acc, gadget offset
// added and digit-field tops flipped  (torusPolynomialMulByXaiMinusOne, numeric_functions.cpp:304-323).
// Source index (j - a) mod 2N: its low LOGN bits address the coefficient, bit LOGN negates it.  The
// lane's byte offset of the K = 0 source, B = ((t - a) mod 2N) * sizeof(T), is formed once; per
// coefficient (Torus32): one add (x = B + 4K), one AND-OR (wrap inside the polynomial | its LDS
// offset), one signed bit-field extract (s = the sign as 0 / -1), then ((src ^ s) + (offset - acc[j]))
// as one XOR-ADD, - s, ^ flip: 7 VALU operations, where the plain expression compiles to 10.
template <typename T, int LOGN, int GRP32 = 8>
TFHE_DEVICE void rotated_minus_one(const WaveLds<T, LOGN> &w, int q, int a, typename Torus<T>::U offset,
                                   typename Torus<T>::U flip, typename Torus<T>::U (&lo)[Geom<LOGN>::PPL],
                                   typename Torus<T>::U (&hi)[Geom<LOGN>::PPL], int t) {
    using G = Geom<LOGN>;
    using U = typename Torus<T>::U;
    constexpr int PPL = G::PPL, N = G::N, NC = G::NC;
    const T *p = w.acc + q * N;
    if (sizeof(T) == 4) {
        constexpr uint32_t PB = N * 4;  // polynomial size in bytes; bit LOGN + 2 of a byte offset = the sign
        const uint32_t poly_lds = w.acc_lds + (uint32_t)q * PB;
        const uint32_t B = ((uint32_t)(t - a) & (2 * N - 1)) * 4;
        // the rotated reads are issued GRP at a time and then consumed (left to itself hipcc waits for each
        // read right behind its issue: one LDS round trip per coefficient); 8 measured best of 4 / 8 / 16 with two
        // waves per SIMD; the lone waves of k_blind_rotate_split: 16 measured equal to 8 (2.48 vs 2.47 ms, round 3)
        constexpr int GRP = GRP32;
#pragma unroll
        for (int g = 0; g < 2 * PPL; g += GRP) {
            uint32_t src[GRP];
#pragma unroll
            for (int e = 0; e < GRP; e++) {
                const int K = 64 * ((g + e) >> 1) + ((g + e) & 1) * NC;
                src[e] = tfhe_lds_load32(w.smem, tfhe_and_or(B + (uint32_t)K * 4, PB - 4, poly_lds));
            }
            TFHE_SCHED_BARRIER();
#pragma unroll
            for (int e = 0; e < GRP; e++) {
                const int m = (g + e) >> 1, h = (g + e) & 1;
                const int K = 64 * m + h * NC;
                const uint32_t x = B + (uint32_t)K * 4;
                const uint32_t s = tfhe_sign_mask<LOGN + 2>(x);
                const uint32_t v = (tfhe_xad(src[e], s, (uint32_t)offset - (uint32_t)p[t + K]) - s) ^ (uint32_t)flip;
                if (h == 0)
                    lo[m] = (U)v;
                else
                    hi[m] = (U)v;
            }
        }
    } else {
        int base = (t - a) & (2 * N - 1);
        TFHE_OPAQUE(base);  // one add per coefficient; expanded, the index terms outlive the transforms and spill
#pragma unroll
        for (int m = 0; m < PPL; m++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int K = 64 * m + h * NC;
                const int idx = base + K;
                const U src = (U)p[idx & (N - 1)];
                const U rot = (idx & N) ? (U)(0 - src) : src;
                const U v = ((rot - (U)p[t + K]) + offset) ^ flip;
                if (h == 0)
                    lo[m] = v;
                else
                    hi[m] = v;
            }
        }
    }
}

// The same straight from GLOBAL memory (Torus32; k_cmux_stream: the accumulator is never staged in LDS).  For a fixed
// register index the 64 lanes read 64 consecutive coefficients (rotated: consecutive modulo N), so every load instruction
// is one or two contiguous runs -- as coalesced as the copy into LDS it replaces.
template <int LOGN>
TFHE_DEVICE void rotated_minus_one_g(const int32_t *p, int a, uint32_t offset, uint32_t flip, uint32_t (&lo)[Geom<LOGN>::PPL],
                                     uint32_t (&hi)[Geom<LOGN>::PPL], int t) {
    using G = Geom<LOGN>;
    constexpr int PPL = G::PPL, N = G::N, NC = G::NC;
    const int base = (t - a) & (2 * N - 1);
    uint32_t src[2 * PPL], own[2 * PPL];
#pragma unroll
    for (int e = 0; e < 2 * PPL; e++) {  // every load is issued before the first use
        const int K = 64 * (e >> 1) + (e & 1) * NC;
        src[e] = tfhe_global_load32(p, (base + K) & (N - 1));
        own[e] = tfhe_global_load32(p, t + K);
    }
#pragma unroll
    for (int e = 0; e < 2 * PPL; e++) {
        const int K = 64 * (e >> 1) + (e & 1) * NC;
        const uint32_t s = ((base + K) & N) ? 0xFFFFFFFFu : 0u;
        const uint32_t v = ((((src[e] ^ s) - s) - own[e]) + offset) ^ flip;
        if (e & 1)
            hi[e >> 1] = v;
        else
            lo[e >> 1] = v;
    }
}

// The same for an accumulator held in registers (BlindRotateLds::ACCREG): own[h][m] = coefficient
// t + 64m + h*N/2 of the polynomial.  The polynomial is written to the wave's LDS scratch (which the
// transposes reuse afterwards: the wave's DS operations execute in order), the rotated source is read back.
template <typename T, int LOGN>
TFHE_DEVICE void rotated_minus_one_reg(const WaveLds<T, LOGN> &w, const typename Torus<T>::U (&own)[2][Geom<LOGN>::PPL], int a,
                                       typename Torus<T>::U offset, typename Torus<T>::U flip,
                                       typename Torus<T>::U (&lo)[Geom<LOGN>::PPL], typename Torus<T>::U (&hi)[Geom<LOGN>::PPL],
                                       int t) {
    using G = Geom<LOGN>;
    using U = typename Torus<T>::U;
    constexpr int PPL = G::PPL, N = G::N, NC = G::NC;
    U *scratch = reinterpret_cast<U *>(w.acc);
    TFHE_WAVE_FENCE();  // earlier reads of this region (transposes) are ordered before these writes
#pragma unroll
    for (int m = 0; m < PPL; m++) {
        scratch[t + 64 * m] = own[0][m];
        scratch[t + 64 * m + NC] = own[1][m];
    }
    TFHE_WAVE_FENCE();
    int base = (t - a) & (2 * N - 1);
    TFHE_OPAQUE(base);
#pragma unroll
    for (int m = 0; m < PPL; m++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int idx = base + 64 * m + h * NC;
            const U src = scratch[idx & (N - 1)];
            const U rot = (idx & N) ? (U)(0 - src) : src;
            const U v = ((rot - own[h][m]) + offset) ^ flip;
            if (h == 0)
                lo[m] = v;
            else
                hi[m] = v;
        }
    }
    TFHE_WAVE_FENCE();
}

// One CMux step on the wave's accumulator:
//   rotate:  acc <- bk_row (x) ((X^a - 1) * acc) + acc   (tfhe_MuxRotate_FFT, a != 0)
//   !rotate: acc <- bk_row (x) acc                       (tGswFFTExternMulToTLwe)
// `rotate` is wave-uniform.  bkrow: device layout [2l][2][PPL][64] complex, pre-scaled by 2/N.
// The inverse transforms run PAIR digits at a time (their key row is prefetched into registers
// under the transform); the two forward transforms run together, in place on the Fourier
// accumulator, sharing every twiddle read.
// LC: the gadget length when it is known at compile time (0: read gd.l); BGC: likewise Bgbit.  With
// LC == PAIR (the gate set: l = 2 in one pair) each polynomial is one transform group and the two
// groups are unrolled: no loop-carried Fourier accumulator (its zero-initialisation disappears into the
// first multiply), key-row addresses become immediates.
// ACCREG: the accumulator is accr[q][h][m] (registers) instead of w.acc (LDS); see BlindRotateLds.
// ACCGLOBAL: w.acc points into GLOBAL memory (the caller's accumulator array, updated in place): see k_cmux_stream.
template <typename T, int LOGN, int PAIR, int LC = 0, int BGC = 0, bool CPLX = false, bool KEEP_ROT = false, bool ACCREG = false,
          bool ACCGLOBAL = false>
TFHE_DEVICE void cmux_step(const WaveLds<T, LOGN> &w, const double2 *__restrict__ bkrow, int a, bool rotate,
                           const Gadget &gd, int t, typename Torus<T>::U (&accr)[2][2][ACCREG ? Geom<LOGN>::PPL : 1]) {
    using G = Geom<LOGN>;
    using U = typename Torus<T>::U;
    constexpr int PPL = G::PPL, N = G::N, NC = G::NC;
    const U offset = (U)gd.offset, flip = (U)gd.flip;

    double fr[2][PPL], fi[2][PPL];  // Fourier accumulator (tLweFFTClear)
    // twiddle prefetch distance of the transforms (WaveFFT::ifft): KEEP_ROT marks the forms with ONE wave per SIMD
    // (the instantiation that reads the gadget length at run time has no registers to spare at two waves per SIMD: 1)
    constexpr int TWD = KEEP_ROT ? TFHE_TWD_LONE : (LC > 0 ? TFHE_TWD_PAIR : 1);
    constexpr bool UNROLLED = (LC > 0 && LC == PAIR);
    const int l = LC ? LC : gd.l;
    const int groups = (l + PAIR - 1) / PAIR;

    // one group = PAIR digits of one accumulator polynomial; for l == PAIR (the gate set) a polynomial is
    // one group and nothing is read twice.
    int qrt = 0;  // run-time polynomial index of the rolled loop below
    // the (rotated) coefficients j and j+N/2 of polynomial q, offset added, digit tops flipped
    auto read_poly = [&](auto qc, U (&lo)[PPL], U (&hi)[PPL]) {
        constexpr int q = decltype(qc)::value < 0 ? 0 : decltype(qc)::value;
        const int qq = decltype(qc)::value < 0 ? qrt : q;  // run-time polynomial index (LDS accumulator only)
        if constexpr (ACCREG) {
            if (rotate) {
                rotated_minus_one_reg<T, LOGN>(w, accr[q], a, offset, flip, lo, hi, t);
            } else {
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    lo[m] = (accr[q][0][m] + offset) ^ flip;
                    hi[m] = (accr[q][1][m] + offset) ^ flip;
                }
            }
        } else if (ACCGLOBAL && rotate) {
            if constexpr (ACCGLOBAL) rotated_minus_one_g<LOGN>(w.acc + qq * N, a, (uint32_t)offset, (uint32_t)flip, lo, hi, t);
        } else if (rotate) {
            rotated_minus_one<T, LOGN>(w, qq, a, offset, flip, lo, hi, t);
        } else {
            const T *p = w.acc + qq * N;
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                const int j = G::jA(t, m);
                lo[m] = ((U)p[j] + offset) ^ flip;
                hi[m] = ((U)p[j + NC] + offset) ^ flip;
            }
        }
    };
    // digits d .. of polynomial q: p = bloc*l + i  (tgsw_functions.cpp:435-443); a trailing odd digit goes alone
    auto digits = [&](int q, int d, const U (&lo)[PPL], const U (&hi)[PPL], auto first) {
        constexpr bool FIRST = decltype(first)::value;
        if (PAIR == 2 && ((LC && LC % 2 == 0) || d + 1 < l)) {
            ifft_mac_digits<T, LOGN, 2, BGC, FIRST, CPLX, false, TWD>(w, bkrow, q * l + d, d, lo, hi, gd, fr, fi, t);
        } else {
            ifft_mac_digits<T, LOGN, 1, BGC, FIRST, CPLX, ACCREG, TWD>(w, bkrow, q * l + d, d, lo, hi, gd, fr, fi, t);
        }
    };
    if (UNROLLED) {
        {
            U lo[PPL], hi[PPL];
            read_poly(std::integral_constant<int, 0>{}, lo, hi);
            digits(0, 0, lo, hi, std::true_type{});
        }
        {
            U lo[PPL], hi[PPL];
            read_poly(std::integral_constant<int, 1>{}, lo, hi);
            digits(1, 0, lo, hi, std::false_type{});
        }
    } else if (ACCREG) {
        // the accumulator registers are indexed by q: the two polynomials are unrolled (the digit loop is not)
        {
            U lo[PPL], hi[PPL];
            read_poly(std::integral_constant<int, 0>{}, lo, hi);
            digits(0, 0, lo, hi, std::true_type{});  // first row as multiplies: no zero-initialised accumulator (-0.9 %)
#pragma unroll 1
            for (int gi = 1; gi < groups; gi++) digits(0, gi * PAIR, lo, hi, std::false_type{});
        }
        {
            U lo[PPL], hi[PPL];
            read_poly(std::integral_constant<int, 1>{}, lo, hi);
#pragma unroll 1
            for (int gi = 0; gi < groups; gi++) digits(1, gi * PAIR, lo, hi, std::false_type{});
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int m = 0; m < PPL; m++) fr[q][m] = fi[q][m] = 0.0;
        // KEEP_ROT (kernels with one wave per SIMD, 512 registers): the rotated coefficients are formed
        // ONCE per polynomial and kept across its digit groups -- at Torus64 the 64-bit rotation arithmetic
        // of a group costs as many VALU instructions as its transform.  Otherwise (256 registers) they are
        // re-read for every group: they then die at the digit extraction, which is what lets the transform
        // pair + key row + Fourier accumulator fit the register file.
#pragma unroll 1
        for (int q = 0; q < 2; q++) {
            U lo[PPL], hi[PPL];
            qrt = q;
            if (KEEP_ROT) read_poly(std::integral_constant<int, -1>{}, lo, hi);
#pragma unroll 1
            for (int gi = 0; gi < groups; gi++) {
                if (!KEEP_ROT) read_poly(std::integral_constant<int, -1>{}, lo, hi);
                digits(q, gi * PAIR, lo, hi, std::false_type{});
            }
        }
    }
    // back to coefficients (both polynomials together), round, accumulate into acc
    // (tLweFromFFTConvert + tLweAddTo)
    WaveFFT<LOGN>::template fft<2, TwLds<LOGN>, CPLX, TWD>(fr, fi, w.tw, w.xch, t);
    U r0[2][PPL], r1[2][PPL];
    bool exact_path = true;
    if (Torus<T>::HAS_FAST) {  // Torus32: short rounding sequence, valid while every |x| < 2^51
        uint32_t guard = 0;
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                r0[q][m] = (U)Torus<T>::from_double_fast(fr[q][m], guard);
                r1[q][m] = (U)Torus<T>::from_double_fast(fi[q][m], guard);
            }
        exact_path = TFHE_WAVE_ANY(!Torus<T>::guard_ok(guard));  // wave-uniform
    }
    if (exact_path) {
        TFHE_KEEP_BRANCH();
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                TFHE_OPAQUE(fr[q][m]);
                TFHE_OPAQUE(fi[q][m]);
                r0[q][m] = (U)Torus<T>::from_double(fr[q][m]);
                r1[q][m] = (U)Torus<T>::from_double(fi[q][m]);
            }
    }
    if constexpr (ACCREG) {  // acc (+)= result, in registers
        // the wave-uniform `rotate` as ONE scalar branch around the whole update (inside the loops hipcc emits a pair of
        // v_cndmask per coefficient: 128 per CMux)
        if (rotate) {
            TFHE_KEEP_BRANCH();
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    accr[q][0][m] += r0[q][m];
                    accr[q][1][m] += r1[q][m];
                }
        } else {
            TFHE_KEEP_BRANCH();
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    accr[q][0][m] = r0[q][m];
                    accr[q][1][m] = r1[q][m];
                }
        }
        return;
    }
    if constexpr (ACCGLOBAL) {  // acc (+)= result in place in global memory: every lane owns its coefficients
        uint32_t old0[2][PPL], old1[2][PPL];
        if (rotate) {
            TFHE_KEEP_BRANCH();
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    old0[q][m] = tfhe_global_load32(w.acc + q * N, G::jA(t, m));
                    old1[q][m] = tfhe_global_load32(w.acc + q * N, G::jA(t, m) + NC);
                }
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    r0[q][m] += old0[q][m];
                    r1[q][m] += old1[q][m];
                }
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                tfhe_global_store32(w.acc + q * N, G::jA(t, m), (uint32_t)r0[q][m]);
                tfhe_global_store32(w.acc + q * N, G::jA(t, m) + NC, (uint32_t)r1[q][m]);
            }
        return;
    }
    // (the wave-uniform `rotate` test sits outside the unrolled loops: inside them hipcc keeps one
    // scalar branch pair per store)
    if (rotate) {
        // acc += result, in place in LDS: one DS add per coefficient (no read, no VALU add); every lane
        // owns its coefficients, and the wave's DS operations execute in order
#pragma unroll
        for (int q = 0; q < 2; q++) {
            U *p = reinterpret_cast<U *>(w.acc + q * N);
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                const int j = G::jA(t, m);
                TFHE_LDS_ADD(&p[j], r0[q][m]);
                TFHE_LDS_ADD(&p[j + NC], r1[q][m]);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            T *p = w.acc + q * N;
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                const int j = G::jA(t, m);
                p[j] = (T)r0[q][m];
                p[j + NC] = (T)r1[q][m];
            }
        }
    }
    TFHE_WAVE_FENCE();
}

// ----------------------------------------------------- blind-rotation kernel
enum : uint32_t {
    BR_INIT_TESTVEC = 1u << 0,  // acc = (0, X^{2N-barb} * v)    else: load acc from acc_io
    BR_EXTRACT = 1u << 1,       // write sample-extracted LWE (index 0)   else: store acc to acc_io
    BR_MODSWITCH = 1u << 2,     // `rot` holds Torus32 LWE samples; rotations = modSwitchFromTorus32(.,2N)
    BR_TV_CONST = 1u << 3,      // test vector = tv_const everywhere (tfhe_bootstrap_woKS_FFT)
    BR_TV_HALF = 1u << 4,       // test vector = -tv_const for j<N/2, +tv_const above (poc:551-553)
    BR_NO_ROTATE = 1u << 5,     // plain external product steps (tGswFFTExternMulToTLwe): no X^a, no +acc
    BR_CMUX_DATA = 1u << 6,     // CMux on data: acc = d1 - d0 on load, + d0 on store (d0 = cmux_d0, d1 = cmux_d1)
    BR_CMUX_TRIVIAL = 1u << 7,  // with BR_CMUX_DATA: d0/d1 are plaintext polynomials [N] = noiseless trivial TLWE (0, d)
};

template <typename T>
struct BlindRotateArgs {
    const double2 *bk;      // [n_steps][2l][2][PPL][64], pre-scaled
    const double2 *tw;      // [2*NC] twiddles
    const int32_t *rot;     // [batch][rot_stride]: rotations (or LWE a_i); entry n_steps = barb / b
    T *acc_io;              // [batch][2][N] (in and/or out)
    const T *tv;            // test vector(s) [N] (tv_stride 0) or [batch][N]
    const T *cmux_d0;       // BR_CMUX_DATA: [batch][2][N]
    const T *cmux_d1;       // BR_CMUX_DATA: [batch][2][N]
    const int32_t *gsw_sel; // per-sample TGSW index into bk (null: computed, see sel_div)
    long long gsw_sample_stride;  // complex elements per TGSW sample (for gsw_sel / sel_div)
    long long cmux_stride;  // BR_CMUX_DATA: elements between the d0 (and d1) of consecutive items
    int32_t cmux_period;    // BR_CMUX_DATA: item index taken modulo this (0: no wrap) -- shared plaintext table
    int32_t sel_div, sel_mul, sel_add;  // gsw_sel == null, sel_div > 0: TGSW sample (ct / sel_div) * sel_mul + sel_add
    T *lwe_out;             // [batch][N+1]
    long long tv_stride;
    long long bk_step_stride;  // complex elements between consecutive steps (0: same row every step)
    T tv_const;
    T out_b_add;            // added to b of the extracted sample (poc:648: + mu/2)
    Gadget gd;
    int32_t n_steps;
    int32_t rot_stride;
    int32_t batch;
    uint32_t flags;
};

// modSwitchFromTorus32(phase, 2N), numeric_functions.cpp:54-60, for Msize = 2N a power of two
template <int LOGN>
TFHE_DEVICE int modswitch_2N(int32_t phase) {
    const int sh = 63 - LOGN;  // interv = 2^(63-LOGN)
    const uint64_t half = 1ull << (sh - 1);
    return (int)((((uint64_t)(uint32_t)phase << 32) + half) >> sh);
}

template <typename T, int LOGN, int WAVES>
struct BlindRotateLds {
    using G = Geom<LOGN>;
    static constexpr size_t tw_bytes = sizeof(double2) * G::TW;
    // ACCREG (Torus64, N = 2048): the 32 KB accumulator lives in the wave's REGISTERS (128 per lane; the
    // kernel runs one wave per SIMD anyway) and LDS holds only one 16 KB scratch polynomial per wave, which
    // the rotated read goes through and which the transposes reuse afterwards: 4 waves per CU instead of 3
    static constexpr bool ACCREG = (LOGN == 11 && sizeof(T) == 8);
    static constexpr size_t acc_bytes = ACCREG ? 0 : sizeof(T) * 2 * G::N;
    // complex-point transposes (16-byte elements) where the workgroup's LDS has room for the larger buffer:
    // N = 1024 (Torus32: 155,648 B for 8 waves; Torus64: 119,808 B for 4) and the register-accumulator kernel;
    // Torus32 at N = 2048 keeps one 8-byte plane at a time (4 x (16 KB + 17 KB) + 32 KB would not fit)
    static constexpr bool CPLX_XCH = (LOGN == 10) || ACCREG;
    static constexpr size_t xch_min = (CPLX_XCH ? sizeof(double2) : sizeof(double)) * G::XCH;
    static constexpr size_t xch_bytes = (ACCREG && sizeof(T) * G::N > xch_min) ? sizeof(T) * G::N : xch_min;
    static constexpr size_t wave_bytes = acc_bytes + xch_bytes;
    // accumulators first (each polynomial then sits at a multiple of its own size, see WaveLds),
    // then the twiddle table, then the transpose buffers
    static constexpr size_t acc_at(int wave) { return (size_t)wave * acc_bytes; }
    static constexpr size_t tw_at = WAVES * acc_bytes;
    static constexpr size_t xch_at(int wave) { return tw_at + tw_bytes + (size_t)wave * xch_bytes; }
    // after the transpose buffers: per-wave progress counters and SIMD ids (WaveLds::balance)
    static constexpr size_t sync_at = tw_bytes + WAVES * wave_bytes;
    static constexpr size_t total = sync_at + 64;
    static_assert(WAVES <= 8, "sync area holds 8 + 8 ints");
};


template <typename T, int LOGN, int WAVES, int PAIR, int LC = 0, int BGC = 0>
TFHE_GLOBAL void __launch_bounds__(WAVES * 64) k_blind_rotate(BlindRotateArgs<T> A) {
    using G = Geom<LOGN>;
    using U = typename Torus<T>::U;
    using Lds = BlindRotateLds<T, LOGN, WAVES>;
    constexpr int N = G::N, PPL = G::PPL;
    TFHE_DYN_LDS(smem);
    TFHE_PROBE_KERNEL_BEGIN();
    const uint32_t sync_lds = tfhe_lds_offset(smem + Lds::sync_at);  // int [wave]: progress; [8 + wave]: SIMD of the wave
    const int my_simd = TFHE_SIMD_ID();
    if ((threadIdx.x & 63) == 0) {
        tfhe_lds_poke32(sync_lds + 4u * (threadIdx.x >> 6), (uint32_t)-1);
        tfhe_lds_poke32(sync_lds + 4u * (8 + (threadIdx.x >> 6)), (uint32_t)my_simd);
    }
    {
        double2 *tw = reinterpret_cast<double2 *>(smem + Lds::tw_at);
        for (int i = threadIdx.x; i < G::TW; i += WAVES * 64) tw[i] = A.tw[i];
        __syncthreads();
    }

    const int wave = TFHE_UNIFORM((int)(threadIdx.x >> 6));
    const int t = threadIdx.x & 63;
    int partner = wave;  // the other wave of this workgroup on the same SIMD (8-wave workgroups: exactly one)
    for (int k = 0; k < WAVES; k++)
        if (k != wave && (int)tfhe_lds_peek32(sync_lds + 4u * (uint32_t)(8 + k)) == my_simd) partner = k;
    const int ct = TFHE_UNIFORM((int)(blockIdx.x * WAVES) + wave);
    if (ct >= A.batch) return;

    WaveLds<T, LOGN> w;
    w.smem = smem;
    w.acc = reinterpret_cast<T *>(smem + (Lds::ACCREG ? Lds::xch_at(wave) : Lds::acc_at(wave)));  // ACCREG: the scratch polynomial
    w.acc_lds = tfhe_lds_offset(w.acc);
    if (!Lds::ACCREG && (w.acc_lds & (uint32_t)(sizeof(T) * N - 1))) TFHE_TRAP();  // rotated_minus_one relies on it: fail loudly
    w.xch = WaveFFT<LOGN>::make_xch(reinterpret_cast<double *>(smem + Lds::xch_at(wave)), t);
    w.tw.tw = reinterpret_cast<const double2 *>(smem + Lds::tw_at);
    w.tw.t = t;
    w.tw.load_uniform(A.tw);
    w.progress_lds = sync_lds;
    w.self = wave;
    w.partner = TFHE_UNIFORM(partner);

    const int32_t *rot = A.rot + (size_t)ct * A.rot_stride;
    // The accumulator: [2][N] in the wave's LDS slice, or (ACCREG) accr[q][h][m] = coefficient
    // t + 64m + h*N/2 of polynomial q in registers.  ACC_AT(m) = flat coefficient t + 64m, m < 4*PPL.
    constexpr bool ACCREG = Lds::ACCREG;
    U accr[2][2][ACCREG ? PPL : 1];
#define ACC_REG(m) accr[(m) / (2 * PPL)][((m) / PPL) & 1][ACCREG ? (m) % PPL : 0]
#define ACC_SET(m, v)                  \
    do {                               \
        if constexpr (ACCREG)          \
            ACC_REG(m) = (U)(v);       \
        else                           \
            w.acc[t + 64 * (m)] = (T)(v); \
    } while (0)
#define ACC_GET(m) (ACCREG ? ACC_REG(m) : (U)w.acc[ACCREG ? 0 : t + 64 * (m)])
    // ---- accumulator initialisation
    if (A.flags & BR_INIT_TESTVEC) {
        int barb = rot[A.n_steps];
        if (A.flags & BR_MODSWITCH) barb = modswitch_2N<LOGN>(barb);
        const int a0 = (2 * N - barb) & (2 * N - 1);  // lwe_functions.cpp:385-386
        const T *tv = A.tv + (size_t)ct * A.tv_stride;
#pragma unroll
        for (int m = 0; m < 2 * PPL; m++) {
            const int j = t + 64 * m;
            // coefficient j of X^{a0} * v
            const int idx = (j - a0) & (2 * N - 1);
            const int src = idx & (N - 1);
            U v;
            if (A.flags & BR_TV_CONST)
                v = (U)A.tv_const;
            else if (A.flags & BR_TV_HALF)
                v = (src < N / 2) ? (U)(0 - (U)A.tv_const) : (U)A.tv_const;
            else
                v = (U)tv[src];
            ACC_SET(m, (U)0);
            ACC_SET(2 * PPL + m, (idx & N) ? (U)(0 - v) : v);
        }
    } else if (A.flags & BR_CMUX_DATA) {
        const size_t item = A.cmux_period ? (size_t)(ct % A.cmux_period) : (size_t)ct;
        const T *d0 = A.cmux_d0 + item * A.cmux_stride, *d1 = A.cmux_d1 + item * A.cmux_stride;
        if (A.flags & BR_CMUX_TRIVIAL) {
#pragma unroll
            for (int m = 0; m < 2 * PPL; m++) {
                ACC_SET(m, (U)0);
                ACC_SET(2 * PPL + m, (U)d1[t + 64 * m] - (U)d0[t + 64 * m]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4 * PPL; m++) ACC_SET(m, (U)d1[t + 64 * m] - (U)d0[t + 64 * m]);
        }
    } else {
        const T *src = A.acc_io + (size_t)ct * 2 * N;
#pragma unroll
        for (int m = 0; m < 4 * PPL; m++) ACC_SET(m, (U)src[t + 64 * m]);
    }
    TFHE_WAVE_FENCE();
    const double2 *bk0 = A.bk;
    if (A.gsw_sel)
        bk0 += (size_t)TFHE_UNIFORM(A.gsw_sel[ct]) * A.gsw_sample_stride;
    else if (A.sel_div > 0)
        bk0 += (size_t)((ct / A.sel_div) * A.sel_mul + A.sel_add) * A.gsw_sample_stride;

    // ---- CMux loop (lwe_functions.cpp:337-361)
    TFHE_PROBE_LOOP_BEGIN();
    // the rotation amount is requested one step ahead, as a scalar load
    int a_next = ((A.flags & BR_NO_ROTATE) || A.n_steps <= 0) ? 0 : tfhe_uniform_load32(rot, 0);
#pragma unroll 1
    for (int i = 0; i < A.n_steps; i++) {
        // every wave of a workgroup reads the same bootstrapping-key row in the same order: re-aligned every few steps, the waves
        // behind find the row's lines in the CU's L1 (see TFHE_BR_SYNC8 at the top of this file)
        if constexpr (WAVES > 1 && BR_SYNC_EVERY<WAVES>::value > 0) {
            if ((i % BR_SYNC_EVERY<WAVES>::value) == 0) __syncthreads();
        }
        const double2 *bkrow = bk0 + (size_t)i * A.bk_step_stride;
        int a = 0;
        const bool rotate = !(A.flags & BR_NO_ROTATE);
        if (rotate) {
            a = a_next;
            a_next = (i + 1 < A.n_steps) ? tfhe_uniform_load32(rot, i + 1) : 0;
            if (A.flags & BR_MODSWITCH) a = modswitch_2N<LOGN>(a);
            a = TFHE_UNIFORM(a);
            if (a == 0) continue;  // :348-350
        }
        w.balance(i, t);
        cmux_step<T, LOGN, PAIR, LC, BGC, Lds::CPLX_XCH, (WAVES <= 4), ACCREG>(w, bkrow, a, rotate, A.gd, t, accr);
    }

    TFHE_PROBE_LOOP_END(wave, t);
    // ---- output
    if (A.flags & BR_EXTRACT) {  // tLweExtractLweSampleIndex, index 0 (tlwe_functions.cpp:351-363)
        T *out = A.lwe_out + (size_t)ct * (N + 1);
        if constexpr (ACCREG) {  // the mask polynomial through the LDS scratch: lane t needs coefficient N - j
            TFHE_WAVE_FENCE();
#pragma unroll
            for (int m = 0; m < 2 * PPL; m++) w.acc[t + 64 * m] = (T)ACC_REG(m);
            TFHE_WAVE_FENCE();
        }
#pragma unroll
        for (int m = 0; m < 2 * PPL; m++) {
            const int j = t + 64 * m;
            out[j] = (j == 0) ? w.acc[0] : (T)(0 - (U)w.acc[N - j]);
        }
        if (t == 0) out[N] = (T)(ACC_GET(2 * PPL) + (U)A.out_b_add);  // lane 0: coefficient 0 of the body polynomial
    } else if (A.flags & BR_CMUX_DATA) {
        const size_t item = A.cmux_period ? (size_t)(ct % A.cmux_period) : (size_t)ct;
        const T *d0 = A.cmux_d0 + item * A.cmux_stride;
        T *dst = A.acc_io + (size_t)ct * 2 * N;
        // every d0 value is loaded before the first store: dst may alias d0/d1, and with loads and
        // stores interleaved the compiler has to wait for each load (one L2 round trip per element)
        if (A.flags & BR_CMUX_TRIVIAL) {
            U v[2 * PPL];
#pragma unroll
            for (int m = 0; m < 2 * PPL; m++) v[m] = (U)d0[t + 64 * m];
#pragma unroll
            for (int m = 0; m < 2 * PPL; m++) {
                dst[t + 64 * m] = (T)ACC_GET(m);
                dst[N + t + 64 * m] = (T)(ACC_GET(2 * PPL + m) + v[m]);
            }
        } else {
            U v[4 * PPL];
#pragma unroll
            for (int m = 0; m < 4 * PPL; m++) v[m] = (U)d0[t + 64 * m];
#pragma unroll
            for (int m = 0; m < 4 * PPL; m++) dst[t + 64 * m] = (T)(ACC_GET(m) + v[m]);
        }
    } else {
        T *dst = A.acc_io + (size_t)ct * 2 * N;
#pragma unroll
        for (int m = 0; m < 4 * PPL; m++) dst[t + 64 * m] = (T)ACC_GET(m);
    }
#undef ACC_REG
#undef ACC_SET
#undef ACC_GET
    TFHE_PROBE_KERNEL_END(t);
}

// ------------------------------------------- one CMux step per launch, accumulators in place in global memory
// BASELINE config 2 as worded ("one external-product kernel per CMux"; tfhe_amd_bootstrap_streamed): acc_io[ct] <- CMux step.
// k_blind_rotate copies the 8 KB accumulator into LDS, computes, copies it back -- three phases that the single 8-wave
// workgroup of a CU walks in lockstep, so the chip alternates between an HBM-bound and a compute-bound state (36.6 against
// 25.7 us per CMux inside the persistent kernel, round 5).  Here the rotated coefficients are read straight from global memory
// (coalesced: rotated_minus_one_g) and the result is added in place, so a wave needs no accumulator in LDS: 16 KB of twiddles +
// 4 x 9 KB of transposes = 53 KB per 4-wave workgroup, two workgroups per CU, their memory phases under each other's arithmetic.
// Torus32, N = 1024, gate gadget length; flags: BR_MODSWITCH only.  Round 6, MI355X, batch 4096 (profiles/r06_streamed_ab.txt):
// 36.8 -> 34.6 us per launch, 0.228 -> 0.242 of 8 TB/s on the contract bytes, bit-identical.  (Tried beside it and dropped:
// k_blind_rotate in 4-wave workgroups of 67 KB with 8-byte transposes, two per CU: 39.6 us; touching the wave's next
// accumulator into L2 a CMux ahead: 35.0 us; one ciphertext per wave without the persistent loop: the same 34.6.)
struct StreamLds {
    using G = Geom<10>;
    static constexpr int WAVES = 4;
    static constexpr size_t tw_bytes = sizeof(double2) * G::TW;
    static constexpr size_t xch_bytes = sizeof(double2) * G::XCH;
    static constexpr size_t total = tw_bytes + WAVES * xch_bytes;
};
template <int LC, int BGC>
TFHE_GLOBAL void __launch_bounds__(256, 2) k_cmux_stream(BlindRotateArgs<int32_t> A) {
    using G = Geom<10>;
    using L = StreamLds;
    TFHE_DYN_LDS(smem);
    {
        double2 *tw = reinterpret_cast<double2 *>(smem);
        for (int i = threadIdx.x; i < G::TW; i += 256) tw[i] = A.tw[i];
        __syncthreads();
    }
    const int wave = TFHE_UNIFORM((int)(threadIdx.x >> 6));
    const int t = threadIdx.x & 63;
    WaveLds<int32_t, 10> w;
    w.smem = smem;
    w.acc_lds = 0;
    w.xch = WaveFFT<10>::make_xch(reinterpret_cast<double *>(smem + L::tw_bytes + (size_t)wave * L::xch_bytes), t);
    w.tw.tw = reinterpret_cast<const double2 *>(smem);
    w.tw.t = t;
    w.tw.load_uniform(A.tw);
    w.progress_lds = 0;
    w.self = w.partner = wave;
    uint32_t none[2][2][1];
    // persistent waves: the grid is what the chip holds (two workgroups per CU); a wave walks its ciphertexts
    const int stride = TFHE_UNIFORM((int)(gridDim.x * L::WAVES));
#pragma unroll 1
    for (int ct = TFHE_UNIFORM((int)(blockIdx.x * L::WAVES) + wave); ct < A.batch; ct += stride) {
        int a = tfhe_uniform_load32(A.rot + (size_t)ct * A.rot_stride, 0);
        if (A.flags & BR_MODSWITCH) a = modswitch_2N<10>(a);
        a = TFHE_UNIFORM(a);
        if (a == 0) continue;  // lwe_functions.cpp:348-350
        w.acc = A.acc_io + (size_t)ct * 2 * G::N;  // GLOBAL memory
        cmux_step<int32_t, 10, 2, LC, BGC, true, false, false, true>(w, A.bk, a, true, A.gd, t, none);
    }
}

// ------------------------------------------- latency-shaped blind rotation (small batches)
// The reference bootstraps ONE sample per call (lwe_functions.cpp:434-446).  With one wave per ciphertext
// (k_blind_rotate) a lone bootstrap is a single wave walking 630 CMux steps of ~2,560 vector instructions
// each; this kernel gives the ciphertext a whole 4-wave workgroup -- one wave on each SIMD of a CU --
// and splits every CMux over them (gate gadget: l = 2, so the external product has 2l = 4 rows):
//   phase 1, all four waves:  wave p = (q, d) forms the rotated polynomial q of (X^a - 1) * acc, extracts
//            gadget digit d and transforms it (one inverse FFT per SIMD); the Lagrange-domain digit goes to
//            the workgroup's hand-over buffer in LDS (lane-contiguous 16-byte points, conflict-free);
//   phase 2, waves 0 and 1:   wave q' owns OUTPUT polynomial q' -- it multiply-accumulates the four digits
//            with its half of the key row in the reference's row order p = 0..3 (the fma chain of
//            lagrangehalfc_impl_fma.s:96-107 is sequential in p, so one wave must own one output polynomial),
//            runs the forward transform, rounds and adds into the accumulator in LDS.
// Two workgroup barriers per CMux (digits complete / accumulator updated).  The key half-rows of step i+1 are
// requested right after the MAC of step i (128 registers per wave, in flight for a whole step), so a lone
// ciphertext never waits for L2 / HBM.  Arithmetic per element is exactly that of k_blind_rotate (same
// helper functions, same operation order): outputs are bit-identical; tests run both kernels on the same inputs.
struct SplitLds {
    using G = Geom<10>;
    static constexpr int WAVES = 4;
    static constexpr size_t acc_bytes = sizeof(int32_t) * 2 * G::N;   // at offset 0: rotated_minus_one's AND-OR addressing
    static constexpr size_t tw_at = acc_bytes;
    static constexpr size_t tw_bytes = sizeof(double2) * G::TW;
    static constexpr size_t xch_bytes = sizeof(double2) * G::XCH;      // complex-point transposes
    static constexpr size_t xch_at(int wave) { return tw_at + tw_bytes + (size_t)wave * xch_bytes; }
    // hand-over buffer: the Lagrange-domain digits of waves 0 and 1 get 2 x 8 KB of their own (those waves need their
    // transpose buffers again in phase 2); waves 2 and 3 are idle in phase 2 and leave their digit in their own
    // transpose buffer
    static constexpr size_t hand_at = tw_at + tw_bytes + WAVES * xch_bytes;
    static constexpr size_t hand_bytes = sizeof(double2) * 2 * G::NC;
    static constexpr size_t hand_row(int p) { return p < 2 ? hand_at + (size_t)p * sizeof(double2) * G::NC : xch_at(p); }
    static constexpr size_t total = hand_at + hand_bytes;              // 77,824 B: two workgroups per CU
    static_assert(xch_bytes >= sizeof(double2) * G::NC, "a digit fits a transpose buffer");
};

// BGC: Bgbit when known at compile time (0: read A.gd.Bgbit).  Gadget length 2, Torus32, N = 1024.
// Flags honoured: BR_INIT_TESTVEC (+ BR_MODSWITCH, BR_TV_CONST, BR_TV_HALF), BR_EXTRACT; without them the
// accumulator is loaded from / stored to acc_io.  (BR_NO_ROTATE / BR_CMUX_DATA / per-sample keys stay on
// k_blind_rotate: the host never routes them here.)
template <int BGC>
TFHE_GLOBAL void __launch_bounds__(256, 2) k_blind_rotate_split(BlindRotateArgs<int32_t> A) {
    using T = int32_t;
    using U = uint32_t;
    using G = Geom<10>;
    using L = SplitLds;
    constexpr int N = G::N, NC = G::NC, PPL = G::PPL;
    TFHE_DYN_LDS(smem);
    const int tid = (int)threadIdx.x;
    const int wave = TFHE_UNIFORM(tid >> 6);
    const int t = tid & 63;
    const int ct = (int)blockIdx.x;
    if (ct >= A.batch) return;  // whole workgroup
    T *acc = reinterpret_cast<T *>(smem);
    {
        double2 *tw = reinterpret_cast<double2 *>(smem + L::tw_at);
        for (int i = tid; i < G::TW; i += 256) tw[i] = A.tw[i];
    }
    const int32_t *rot = A.rot + (size_t)ct * A.rot_stride;
    // ---- accumulator initialisation, 8 coefficients per thread (flat index e over [2][N])
    if (A.flags & BR_INIT_TESTVEC) {
        int barb = rot[A.n_steps];
        if (A.flags & BR_MODSWITCH) barb = modswitch_2N<10>(barb);
        const int a0 = (2 * N - barb) & (2 * N - 1);  // lwe_functions.cpp:385-386
        const T *tv = A.tv + (size_t)ct * A.tv_stride;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int j = tid + 256 * m;
            const int idx = (j - a0) & (2 * N - 1);
            const int src = idx & (N - 1);
            U v;
            if (A.flags & BR_TV_CONST)
                v = (U)A.tv_const;
            else if (A.flags & BR_TV_HALF)
                v = (src < N / 2) ? (U)(0 - (U)A.tv_const) : (U)A.tv_const;
            else
                v = (U)tv[src];
            acc[j] = 0;
            acc[N + j] = (T)((idx & N) ? (U)(0 - v) : v);
        }
    } else {
        const T *src = A.acc_io + (size_t)ct * 2 * N;
#pragma unroll
        for (int m = 0; m < 8; m++) acc[tid + 256 * m] = src[tid + 256 * m];
    }
    __syncthreads();

    WaveLds<T, 10> w;
    w.smem = smem;
    w.acc = acc;
    w.acc_lds = tfhe_lds_offset(acc);
    if (w.acc_lds & (uint32_t)(sizeof(T) * N - 1)) TFHE_TRAP();  // rotated_minus_one relies on it: fail loudly
    w.xch = WaveFFT<10>::make_xch(reinterpret_cast<double *>(smem + L::xch_at(wave)), t);
    w.tw.tw = reinterpret_cast<const double2 *>(smem + L::tw_at);
    w.tw.t = t;
    w.tw.load_uniform(A.tw);
    w.progress_lds = 0;
    w.self = w.partner = wave;

    const U offset = (U)A.gd.offset, flip = (U)A.gd.flip;
    const int Bgbit = BGC ? BGC : A.gd.Bgbit;
    const int q = wave >> 1, d = wave & 1;          // phase 1: row p = q * l + d = wave
    const int decal = 32 - (d + 1) * Bgbit;         // wave-uniform
    const bool fwd = wave < 2;                      // phase 2: this wave owns output polynomial `wave`

    // rotation of step i (0 = skipped, lwe_functions.cpp:348-350); wave-uniform scalar loads
    auto rotation = [&](int i) {
        int a = tfhe_uniform_load32(rot, i);
        if (A.flags & BR_MODSWITCH) a = modswitch_2N<10>(a);
        return TFHE_UNIFORM(a);
    };
    auto next_step = [&](int i) {  // first step >= i with a non-zero rotation
        while (i < A.n_steps && rotation(i) == 0) i++;
        return i;
    };
    // This wave's half (output polynomial `wave`) of the key row of a step: rows p = 0..3, [PPL][64] complex each.
    // Rows 0 and 1 are requested a whole step ahead (right after the previous MAC: 64 registers carried through
    // phase 1); rows 2 and 3 at the end of phase 1, when its registers are free -- they arrive under the MAC of rows
    // 0 and 1 from L2, where the idle waves 2 and 3 pulled them during the previous phase 2 (`touch`).  Carrying all
    // four rows (128 registers) through phase 1 made hipcc park them in AGPRs: ~420 v_accvgpr moves per CMux.
    double2 bkA[2][PPL], bkB[2][PPL];
    const uint32_t lane16 = (uint32_t)t * 16u + (uint32_t)(wave & 1) * (uint32_t)(PPL * 64 * 16);
    auto key_rsrc = [&](int i) {
        return TFHE_MAKE_BUFFER_RSRC(reinterpret_cast<const unsigned char *>(A.bk + (size_t)i * A.bk_step_stride));
    };
    auto request_rows = [&](double2 (&dst)[2][PPL], int i, int p0) {
        const TFHE_BUFFER_RSRC rsrc = key_rsrc(i);
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int m = 0; m < PPL; m++)
                dst[p][m] = tfhe_buffer_load_d2(rsrc, lane16, (uint32_t)((((p0 + p) * 2) * PPL + m) * 64) * 16u);
    };
    // waves 2, 3: one dword of every 128-byte line of rows 2 and 3 (both output halves: 32 KB = 256 lines, 2 per lane)
    // of step i's key row -> this XCD's L2.  The values are folded into `sink`, which is what keeps the loads alive.
    uint32_t sink = 0, touched[2] = {0, 0};
    auto touch = [&](int i) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(A.bk + (size_t)i * A.bk_step_stride) + (size_t)2 * 2 * PPL * 64 * 4;
        const int line = ((wave & 1) * 64 + t) * 2;
#pragma unroll
        for (int k = 0; k < 2; k++) touched[k] = row[(size_t)(line + k) * 32];
    };

    int i = next_step(0);
    if (fwd && i < A.n_steps) {
        TFHE_KEEP_BRANCH();
        request_rows(bkA, i, 0);
    }
#pragma unroll 1
    while (i < A.n_steps) {
        const int a = rotation(i);
        const int inext = next_step(i + 1);
        // ---- phase 1: digit d of polynomial q of (X^a - 1) * acc, to the Lagrange domain
        {
            U lo[PPL], hi[PPL];
            rotated_minus_one<T, 10>(w, q, a, offset, flip, lo, hi, t);
            double xr[1][PPL], xi[1][PPL];
#pragma unroll
            for (int m = 0; m < PPL; m++) {  // field tops are flipped: the signed field IS digit - Bg/2 (Gadget::flip)
                xr[0][m] = (double)TFHE_SBFE(lo[m], decal, Bgbit);
                xi[0][m] = (double)TFHE_SBFE(hi[m], decal, Bgbit);
            }
            WaveFFT<10>::template ifft<1, TwLds<10>, true, TFHE_TWD_SPLIT>(xr, xi, w.tw, w.xch, t);
            double2 *h = reinterpret_cast<double2 *>(smem + L::hand_row(wave)) + t;
#pragma unroll
            for (int m = 0; m < PPL; m++) h[64 * m] = make_double2(xr[0][m], xi[0][m]);
        }
        TFHE_ORDER();
        if (fwd) {
            TFHE_KEEP_BRANCH();
            request_rows(bkB, i, 2);
        }
        __syncthreads();  // the four digits are in the hand-over buffer
        // ---- phase 2 (waves 0, 1): MAC in row order, forward transform, round, acc += result
        if (fwd) {
            TFHE_KEEP_BRANCH();
            double fr[1][PPL], fi[1][PPL];
            // digits read one row ahead of their MAC (two rows = 64 registers in flight; left to itself hipcc issues
            // all four rows' reads up front: 128 registers next to the 128 of the key rows, and spills)
            double ar[2][PPL], ai[2][PPL];
            auto read_digit = [&](int p) {
                const double2 *h = reinterpret_cast<const double2 *>(smem + L::hand_row(p)) + t;
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    const double2 v = h[64 * m];
                    ar[p & 1][m] = v.x;
                    ai[p & 1][m] = v.y;
                }
            };
            read_digit(0);
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (p < 3) read_digit(p + 1);
                TFHE_ORDER();
                if (p == 0)
                    mac_half_row<PPL, true>(fr[0], fi[0], ar[0], ai[0], bkA[0]);
                else
                    mac_half_row<PPL, false>(fr[0], fi[0], ar[p & 1], ai[p & 1], p == 1 ? bkA[1] : bkB[p - 2]);
#pragma unroll
                for (int m = 0; m < PPL; m++) {  // pins the row's arithmetic here (the optimiser otherwise sinks all four
                    TFHE_OPAQUE(fr[0][m]);       // rows below the key request, and every digit and key register stays live)
                    TFHE_OPAQUE(fi[0][m]);
                }
                TFHE_ORDER();
            }
            TFHE_ORDER();  // not above the MAC: the old and the new rows would both be live
            if (inext < A.n_steps) {
                TFHE_KEEP_BRANCH();
                request_rows(bkA, inext, 0);  // in flight during the transform below and the whole of the next phase 1
            }
            TFHE_ORDER();
            WaveFFT<10>::template fft<1, TwLds<10>, true, TFHE_TWD_SPLIT>(fr, fi, w.tw, w.xch, t);
            U r0[PPL], r1[PPL];
            uint32_t guard = 0;
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                r0[m] = (U)Torus<T>::from_double_fast(fr[0][m], guard);
                r1[m] = (U)Torus<T>::from_double_fast(fi[0][m], guard);
            }
            if (TFHE_WAVE_ANY(!Torus<T>::guard_ok(guard))) {  // |x| >= 2^51 somewhere: the reference's own form
                TFHE_KEEP_BRANCH();
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    TFHE_OPAQUE(fr[0][m]);
                    TFHE_OPAQUE(fi[0][m]);
                    r0[m] = (U)Torus<T>::from_double(fr[0][m]);
                    r1[m] = (U)Torus<T>::from_double(fi[0][m]);
                }
            }
            U *pacc = reinterpret_cast<U *>(acc + wave * N);
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                TFHE_LDS_ADD(&pacc[G::jA(t, m)], r0[m]);
                TFHE_LDS_ADD(&pacc[G::jA(t, m) + NC], r1[m]);
            }
        } else {
            TFHE_KEEP_BRANCH();
            sink ^= touched[0] ^ touched[1];  // the previous step's touches have long landed
            if (inext < A.n_steps) {
                TFHE_KEEP_BRANCH();
                touch(inext);
            }
        }
        __syncthreads();  // accumulator updated; hand-over buffer free
        i = inext;
    }
    if (sink == 0x9E3779B9u && A.batch < 0) A.lwe_out[0] = (T)sink;  // never true: keeps `touch` observable
    // ---- output
    if (A.flags & BR_EXTRACT) {  // tLweExtractLweSampleIndex, index 0 (tlwe_functions.cpp:351-363)
        T *out = A.lwe_out + (size_t)ct * (N + 1);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int j = tid + 256 * m;
            out[j] = (j == 0) ? acc[0] : (T)(0 - (U)acc[N - j]);
        }
        if (tid == 0) out[N] = (T)((U)acc[N] + (U)A.out_b_add);
    } else {
        T *dst = A.acc_io + (size_t)ct * 2 * N;
#pragma unroll
        for (int m = 0; m < 8; m++) dst[tid + 256 * m] = acc[tid + 256 * m];
    }
}

// ------------------------------------------- standalone batched transforms
// NT: lane-contiguous accesses of data this launch touches once, when the launch's working set is larger than the
// 256 MB Infinity Cache (the host decides: launch_ifft_w / launch_fft_w) -- nontemporal loads and stores.  Measured
// (profiles/r03_config4_ab.txt): +5..11 % on the coefficient -> Lagrange kernels at 4 x the cache size, -9 % when the
// working set fits the cache (repeated calls then hit it), and never on the strided 16-byte loads of k_fft_batch
// (the eight loads of a line must meet in the cache: 0.34 instead of 0.64 of 8 TB/s with nt).
template <bool NT, typename V>
TFHE_DEVICE V stream_load(const V *p) {
    return NT ? tfhe_nontemporal_load(p) : *p;
}
template <bool NT, typename V>
TFHE_DEVICE void stream_store(V v, V *p) {
    if (NT)
        tfhe_nontemporal_store(v, p);
    else
        *p = v;
}
// FFT plugin boundary (CB/spqlios/lagrangehalfc_impl.h:8-31), one wave per polynomial.
template <int LOGN, int WAVES>
struct FftLds {
    using G = Geom<LOGN>;
    static constexpr size_t tw_bytes = sizeof(double2) * G::TW;
    static constexpr size_t wave_bytes = sizeof(double) * G::XCH;
    static constexpr size_t total = tw_bytes + WAVES * wave_bytes;
};

// execute_reverse_int / _torus32 / _torus64: torus or int coefficients -> LagrangeHalfC.
// PACK: write the key layout of the blind-rotation kernel instead ([row][PPL][64] complex, scaled by
// 2/N -- what k_pack_gsw makes of the reference layout), i.e. tGswToFFTConvert straight into a
// device-resident key without the intermediate LagrangeHalfC array.
//
// PERSISTENT waves: the grid is sized to the chip (launch_ifft_w / launch_fft_w), every wave walks the batch
// with stride gridDim.x * WAVES.  The twiddle table is staged once per workgroup instead of once per WAVES
// polynomials (32 KB from L2 + a barrier in front of every 4 transforms at N = 2048), and a wave requests its
// NEXT polynomial before it transforms the current one, so its HBM latency runs under the butterflies.
template <typename TIN, int LOGN, int WAVES, bool PACK = false, bool NT = false>
TFHE_GLOBAL void __launch_bounds__(WAVES * 64)
    k_ifft_batch(double *__restrict__ out, const TIN *__restrict__ in, const double2 *__restrict__ twg, int batch) {
    using G = Geom<LOGN>;
    constexpr int N = G::N, NC = G::NC, PPL = G::PPL;
    TFHE_DYN_LDS(smem);
    double2 *tw = reinterpret_cast<double2 *>(smem);
    const int wave = TFHE_UNIFORM((int)(threadIdx.x >> 6));
    const int t = threadIdx.x & 63;
    const int stride = TFHE_UNIFORM((int)(gridDim.x * WAVES));
    int b = TFHE_UNIFORM((int)(blockIdx.x * WAVES) + wave);
    // the first polynomial's loads are issued BEFORE the twiddle table is staged: their HBM latency runs
    // under the staging and its barrier
    TIN raw_r[PPL], raw_i[PPL];
    auto request = [&](int poly) {
        const TIN *p = in + (size_t)poly * N;
#pragma unroll
        for (int m = 0; m < PPL; m++) {
            raw_r[m] = stream_load<NT>(&p[G::jA(t, m)]);
            raw_i[m] = stream_load<NT>(&p[G::jA(t, m) + NC]);
        }
    };
    if (b < batch) request(b);
    for (int i = threadIdx.x; i < G::TW; i += WAVES * 64) tw[i] = twg[i];
    __syncthreads();
    const typename WaveFFT<LOGN>::Xch xch = WaveFFT<LOGN>::make_xch(
        reinterpret_cast<double *>(smem + FftLds<LOGN, WAVES>::tw_bytes) + (size_t)wave * G::XCH, t);
    TwLds<LOGN> twp;
    twp.tw = tw;
    twp.t = t;
    twp.load_uniform(twg);
#pragma unroll 1
    for (; b < batch; b += stride) {
        double xr[1][PPL], xi[1][PPL];
#pragma unroll
        for (int m = 0; m < PPL; m++) {
            xr[0][m] = (double)raw_r[m];
            xi[0][m] = (double)raw_i[m];
        }
        if (b + stride < batch) request(b + stride);
        WaveFFT<LOGN>::template ifft<1, TwLds<LOGN>>(xr, xi, twp, xch, t);
        if (PACK) {
            double2 *o = reinterpret_cast<double2 *>(out) + (size_t)b * NC;
            const double scale = 2.0 / (double)N;  // exact: a power of two
#pragma unroll
            for (int m = 0; m < PPL; m++) o[64 * m + t] = make_double2(xr[0][m] * scale, xi[0][m] * scale);
            continue;
        }
        // The reference's order has lane t holding PPL CONSECUTIVE outputs (jC): stored from there, every
        // store instruction would touch 64 different 128-byte lines.  One more pass through the wave's LDS
        // buffer turns it into the lane-contiguous order jA, 512 contiguous bytes per store instruction.
        // (Measured: 16-byte stores straight from the jC order reach 0.44-0.53 of 8 TB/s against 0.62-0.67 with
        // the LDS pass -- partial-line writes; profiles/r03_config4_ab.txt.  LOADS in that order are fine: k_fft_batch.)
        WaveFFT<LOGN>::template transpose<G::RD_C2, G::RD_A2>(xr[0], xch);
        WaveFFT<LOGN>::template transpose<G::RD_C2, G::RD_A2>(xi[0], xch);
        double *o = out + (size_t)b * N;
#pragma unroll
        for (int m = 0; m < PPL; m++) {
            stream_store<NT>(xr[0][m], &o[G::jA(t, m)]);
            stream_store<NT>(xi[0][m], &o[G::jA(t, m) + NC]);
        }
    }
}

// execute_direct_torus32 / _torus64: LagrangeHalfC -> torus coefficients (scale 2/N first); persistent waves
// with the next polynomial requested ahead, as k_ifft_batch.  TOUT = double: the bare core transform `fft` of
// spqlios-fft.h:52 (no 2/N scale, no rounding: N doubles in, N doubles out) -- what the reference's C core exposes.
template <typename TOUT, int LOGN, int WAVES, bool NT = false>
TFHE_GLOBAL void __launch_bounds__(WAVES * 64)
    k_fft_batch(TOUT *__restrict__ out, const double *__restrict__ in, const double2 *__restrict__ twg, int batch) {
    using G = Geom<LOGN>;
    constexpr int N = G::N, NC = G::NC, PPL = G::PPL;
    TFHE_DYN_LDS(smem);
    double2 *tw = reinterpret_cast<double2 *>(smem);
    const int wave = TFHE_UNIFORM((int)(threadIdx.x >> 6));
    const int t = threadIdx.x & 63;
    const int stride = TFHE_UNIFORM((int)(gridDim.x * WAVES));
    int b = TFHE_UNIFORM((int)(blockIdx.x * WAVES) + wave);
    constexpr bool RAW = std::is_same<TOUT, double>::value;
    const double scale = 2.0 / (double)N;  // fft_processor_spqlios.cpp:78
    double raw_r[PPL], raw_i[PPL];
    // Loads straight in the transform's input order (lane t holds the PPL consecutive points jC = PPL t + m of each
    // half): 16-byte loads, 64 lines per instruction but every line read whole by the PPL / 2 instructions of the
    // half -- the L1/L2 merge them.  Measured against lane-contiguous loads + one more pass through LDS: +1..6 %
    // (profiles/r03_config4_ab.txt); the same access pattern on the STORE side of k_ifft_batch loses 25 %.
    auto request = [&](int poly) {
        const double2 *p2 = reinterpret_cast<const double2 *>(in + (size_t)poly * N);
#pragma unroll
        for (int m = 0; m < PPL; m += 2) {
            const double2 a = p2[(PPL * t + m) / 2], c = p2[(NC + PPL * t + m) / 2];
            raw_r[m] = a.x;
            raw_r[m + 1] = a.y;
            raw_i[m] = c.x;
            raw_i[m + 1] = c.y;
        }
    };
    if (b < batch) request(b);
    for (int i = threadIdx.x; i < G::TW; i += WAVES * 64) tw[i] = twg[i];
    __syncthreads();
    const typename WaveFFT<LOGN>::Xch xch = WaveFFT<LOGN>::make_xch(
        reinterpret_cast<double *>(smem + FftLds<LOGN, WAVES>::tw_bytes) + (size_t)wave * G::XCH, t);
    TwLds<LOGN> twp;
    twp.tw = tw;
    twp.t = t;
    twp.load_uniform(twg);
#pragma unroll 1
    for (; b < batch; b += stride) {
        double xr[1][PPL], xi[1][PPL];
#pragma unroll
        for (int m = 0; m < PPL; m++) {
            xr[0][m] = RAW ? raw_r[m] : raw_r[m] * scale;
            xi[0][m] = RAW ? raw_i[m] : raw_i[m] * scale;
        }
        if (b + stride < batch) request(b + stride);
        WaveFFT<LOGN>::template fft<1, TwLds<LOGN>>(xr, xi, twp, xch, t);
        if constexpr (RAW) {
            double *o = out + (size_t)b * N;
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                stream_store<NT>(xr[0][m], &o[G::jA(t, m)]);
                stream_store<NT>(xi[0][m], &o[G::jA(t, m) + NC]);
            }
        } else {
            // rounding: the short exact sequences of the blind-rotation kernels (Torus<T>::from_double_fast), the
            // reference's own form where the wave's guard trips (|x| >= 2^51 resp. 2^83)
            TOUT *o = out + (size_t)b * N;
            TOUT r0[PPL], r1[PPL];
            uint32_t guard = 0;
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                r0[m] = Torus<TOUT>::from_double_fast(xr[0][m], guard);
                r1[m] = Torus<TOUT>::from_double_fast(xi[0][m], guard);
            }
            if (TFHE_WAVE_ANY(!Torus<TOUT>::guard_ok(guard))) {
                TFHE_KEEP_BRANCH();
#pragma unroll
                for (int m = 0; m < PPL; m++) {
                    TFHE_OPAQUE(xr[0][m]);
                    TFHE_OPAQUE(xi[0][m]);
                    r0[m] = Torus<TOUT>::from_double(xr[0][m]);
                    r1[m] = Torus<TOUT>::from_double(xi[0][m]);
                }
            }
#pragma unroll
            for (int m = 0; m < PPL; m++) {
                stream_store<NT>(r0[m], &o[G::jA(t, m)]);
                stream_store<NT>(r1[m], &o[G::jA(t, m) + NC]);
            }
        }
    }
}

// LagrangeHalfCPolynomialAddMulASM over a batch: res[b] += a[b] * bb[b or 0]
TFHE_GLOBAL void k_lagrange_addmul(double *__restrict__ res, const double *__restrict__ a,
                                   const double *__restrict__ bb, int Ns2, long long b_stride, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const long long poly = gid / Ns2;
    const int i = (int)(gid - poly * Ns2);
    const double *pa = a + poly * 2 * Ns2, *pb = bb + poly * b_stride;
    double *pr = res + poly * 2 * Ns2;
    const double ar = pa[i], ai = pa[Ns2 + i], br = pb[i], bi = pb[Ns2 + i];
    const double tneg = __builtin_fma(ai, bi, -pr[i]);
    pr[i] = __builtin_fma(ar, br, -tneg);
    const double u = __builtin_fma(ar, bi, pr[Ns2 + i]);
    pr[Ns2 + i] = __builtin_fma(ai, br, u);
}

// Key upload: LagrangeHalfC polynomials (reference order) -> kernel layout, scaled by 2/N.
// src: [rows][N] doubles (re[0..NC) | im[0..NC));  dst: [rows][PPL][64] complex with
// dst[row][m][t] = (re, im)[jC(t,m)] * 2/N.
template <int LOGN>
TFHE_GLOBAL void k_pack_gsw(double2 *__restrict__ dst, const double *__restrict__ src, long long rows) {
    using G = Geom<LOGN>;
    constexpr int N = G::N, NC = G::NC, PPL = G::PPL;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * NC) return;
    const long long row = gid / NC;
    const int e = (int)(gid - row * NC);
    const int m = e >> 6, t = e & 63;
    const double scale = 2.0 / (double)N;
    const double *p = src + row * N;
    const int pos = PPL * t + m;
    dst[gid] = make_double2(p[pos] * scale, p[pos + NC] * scale);
}

// preModSwitch (poc:472-484) / modSwitchFromTorus32(., 2N) over flat arrays
template <int LOGN>
TFHE_GLOBAL void k_modswitch(int32_t *__restrict__ out, const int32_t *__restrict__ in, long long total) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) out[gid] = modswitch_2N<LOGN>(in[gid]);
}

// ------------------------------------------- diagnostics: shader clock under load (tfhe_amd_clock_probe)
// One lane per workgroup stamps the shader-cycle counter and the 100 MHz reference counter, sleeps (s_sleep: no issue
// slots taken from the waves it runs beside) until `ticks` reference ticks have passed, and stamps again.
struct ClockStamp {
    unsigned long long c0, r0, c1, r1;
};
TFHE_GLOBAL void k_clock_probe(ClockStamp *out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    ClockStamp s;
    s.c0 = TFHE_SHADER_CYCLES();
    s.r0 = TFHE_REF_TICKS();
    do {
        TFHE_SLEEP();
        s.r1 = TFHE_REF_TICKS();
    } while (s.r1 - s.r0 < ticks);
    s.c1 = TFHE_SHADER_CYCLES();
    s.r1 = TFHE_REF_TICKS();
    out[blockIdx.x] = s;
}

// ------------------------------------------- exact (FFT-free) external product
// The reference's `#ifndef USE_FFT` backend (poc:285-316): the same external product with every
// polynomial product computed exactly in Z_{2^W}[X]/(X^N+1) (torus{32,64}PolynomialMultAddKaratsuba,
// CB/poc_karatsuba.cpp; here the plain negacyclic convolution -- same ring element, no rounding).
// A verification backend: what the fp64 path approximates.  One workgroup per sample; the digits of
// the sample stay in LDS, each key polynomial is staged in LDS as its negacyclic extension
// [g | -g] so that coefficient (i - j) mod 2N needs no sign logic.
template <typename T, int LOGN>
struct ExactLds {
    static constexpr int N = 1 << LOGN;
    static constexpr size_t dig_bytes(int l) { return sizeof(int32_t) * 2 * (size_t)l * N; }
    static constexpr size_t total(int l) { return dig_bytes(l) + sizeof(T) * 2 * N; }
};
template <typename T, int LOGN>
TFHE_GLOBAL void __launch_bounds__(256)
    k_extprod_exact(T *__restrict__ acc_io, const T *__restrict__ gsw, Gadget gd, int batch) {
    using U = typename Torus<T>::U;
    constexpr int N = 1 << LOGN, R = N / 256, BITS = Torus<T>::BITS;
    TFHE_DYN_LDS(smem);
    int32_t *dig = reinterpret_cast<int32_t *>(smem);                                    // [2l][N]
    U *gext = reinterpret_cast<U *>(smem + ExactLds<T, LOGN>::dig_bytes(gd.l));            // [2N]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= batch) return;
    T *acc = acc_io + (size_t)b * 2 * N;
    // gadget decomposition (tgsw_functions.cpp:224-337 / poc:492-515; offset from the host)
    const U mask = ((U)1 << gd.Bgbit) - 1;
    const int32_t halfBg = 1 << (gd.Bgbit - 1);
    for (int e = tid; e < 2 * N; e += 256) {
        const U v = (U)acc[e] + (U)gd.offset;
        const int poly = e >> LOGN, c = e & (N - 1);
        for (int p = 0; p < gd.l; p++)
            dig[(poly * gd.l + p) * N + c] = (int32_t)((v >> (BITS - (p + 1) * gd.Bgbit)) & mask) - halfBg;
    }
    U res[2][R];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int r = 0; r < R; r++) res[q][r] = 0;
    for (int p = 0; p < 2 * gd.l; p++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            __syncthreads();  // digits written / previous polynomial consumed
            const T *g = gsw + ((size_t)p * 2 + q) * N;
            for (int e = tid; e < N; e += 256) {
                gext[e] = (U)g[e];
                gext[N + e] = (U)0 - (U)g[e];
            }
            __syncthreads();
            const int32_t *dp = dig + p * N;
            for (int j = 0; j < N; j++) {
                const U d = (U)(T)dp[j];  // sign-extended digit, wave-uniform (LDS broadcast)
#pragma unroll
                for (int r = 0; r < R; r++) res[q][r] += d * gext[(tid + 256 * r - j) & (2 * N - 1)];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[q * N + tid + 256 * r] = (T)res[q][r];
}

// ------------------------------------- Real96 high-precision anticyclic transforms
// high-precision-anticyclic-fft/src/code.cpp ("HP"): the same radix-2 structure as spqlios, on
// 128-bit fixed point (a Real96 is a two's-complement integer v standing for v / 2^64, HP:17-40),
// n = 2N, N/2 complex points.  One 256-thread workgroup per polynomial, points in LDS (32 B each:
// 32 KB at N = 2048), one workgroup barrier per stage, twiddles read from the [n] table in global
// memory (128 KB at n = 4096: L2-resident).  Integer arithmetic: results are exact functions of
// the inputs, so parity with the CPU restatement is equality.
typedef unsigned __int128 u128;
struct HpCplx {
    u128 re, im;
};
// Real96 product (what HP:79-95 and :148-169 compute): (a * b) >> 64, exact modulo 2^128, for a twiddle b in
// [-1, 1) stored sign-extended.  Derivation of the three correction terms:
TFHE_DEVICE u128 hp_intmul(u128 a, u128 b) {
    /* A = a as a signed 128-bit integer = ahi_s * 2^64 + alo (ahi_s signed, alo unsigned);
     * B = the twiddle, |B| < 2^64, sign-extended: B = blo - 2^64 * bneg with blo its low word.
     *   floor(A * blo / 2^64) = ahi_s * blo + floor(alo * blo / 2^64)
     *                         = ahi_u * blo - 2^64 * [ahi_s < 0] * blo + hi64(alo * blo)      (ahi_s = ahi_u - 2^64 [ahi_s < 0])
     * and modulo 2^128 the middle term is ((0 - blo) mod 2^64) << 64.  The bneg part is exact:
     * (2^64 * bneg * A) / 2^64 = bneg * A.  Hence, modulo 2^128:                                          */
    const uint64_t alo = (uint64_t)a, ahi_u = (uint64_t)(a >> 64), blo = (uint64_t)b;
    const int a_negative = (int)(ahi_u >> 63), b_negative = (int)((uint64_t)(b >> 64) >> 63);
    u128 r = (u128)blo * ahi_u;                  /* ahi_u * blo                  */
    r += ((u128)blo * alo) >> 64;                /* + hi64(alo * blo)            */
    if (a_negative) r += (u128)(0 - blo) << 64;  /* - 2^64 * blo  (mod 2^128)    */
    if (b_negative) r -= a;                      /* - bneg * A                   */
    return r;
}
// std::complex<Real96> product: data on the left, twiddle on the right
TFHE_DEVICE HpCplx hp_cmul(const HpCplx &a, const HpCplx &w) {
    HpCplx r;
    r.re = hp_intmul(a.re, w.re) - hp_intmul(a.im, w.im);
    r.im = hp_intmul(a.re, w.im) + hp_intmul(a.im, w.re);
    return r;
}
template <int LOGN>
struct HpGeom {
    static constexpr int N = 1 << LOGN, NS4 = N / 2, n = 2 * N;
    static constexpr size_t lds_bytes = sizeof(HpCplx) * NS4;
};
// iFFT (HP:391-444): Torus64 coefficients -> N/2 complex Real96 values
template <int LOGN>
TFHE_GLOBAL void __launch_bounds__(256)
    k_hp_ifft(HpCplx *__restrict__ out, const int64_t *__restrict__ in, const HpCplx *__restrict__ pw, int batch) {
    using G = HpGeom<LOGN>;
    constexpr int NS4 = G::NS4;
    TFHE_DYN_LDS(smem);
    HpCplx *buf = reinterpret_cast<HpCplx *>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= batch) return;
    const int64_t *p = in + (size_t)b * G::N;
    for (int j = tid; j < NS4; j += 256) {
        HpCplx v;
        v.re = (u128)(__int128)p[j];  // t64tor96, HP:184-189
        v.im = (u128)(__int128)p[j + NS4];
        buf[j] = hp_cmul(v, pw[j]);
    }
    __syncthreads();
    for (int nn = NS4; nn >= 2; nn >>= 1) {
        const int halfnn = nn >> 1, step = 2 * (NS4 / halfnn);
        for (int bf = tid; bf < NS4 / 2; bf += 256) {
            const int off = bf & (halfnn - 1), i1 = ((bf - off) << 1) + off, i2 = i1 + halfnn;
            const HpCplx t1 = buf[i1], t2 = buf[i2];
            HpCplx sum, dif;
            sum.re = t1.re + t2.re;
            sum.im = t1.im + t2.im;
            dif.re = t1.re - t2.re;
            dif.im = t1.im - t2.im;
            buf[i1] = sum;
            buf[i2] = hp_cmul(dif, pw[step * off]);
        }
        __syncthreads();
    }
    HpCplx *o = out + (size_t)b * NS4;
    for (int j = tid; j < NS4; j += 256) o[j] = buf[j];
}
// FFT (HP:446-512): N/2 complex Real96 values -> Torus64 coefficients, divided by N/2 (">> 10", HP:499-500)
template <int LOGN>
TFHE_GLOBAL void __launch_bounds__(256)
    k_hp_fft(int64_t *__restrict__ out, const HpCplx *__restrict__ in, const HpCplx *__restrict__ pwbar, int batch) {
    using G = HpGeom<LOGN>;
    constexpr int NS4 = G::NS4;
    TFHE_DYN_LDS(smem);
    HpCplx *buf = reinterpret_cast<HpCplx *>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= batch) return;
    const HpCplx *p = in + (size_t)b * NS4;
    for (int j = tid; j < NS4; j += 256) buf[j] = p[j];
    __syncthreads();
    for (int nn = 2; nn <= NS4; nn <<= 1) {
        const int halfnn = nn >> 1, step = 2 * (NS4 / halfnn);
        for (int bf = tid; bf < NS4 / 2; bf += 256) {
            const int off = bf & (halfnn - 1), i1 = ((bf - off) << 1) + off, i2 = i1 + halfnn;
            const HpCplx t1 = buf[i1], t2 = hp_cmul(buf[i2], pwbar[step * off]);
            HpCplx sum, dif;
            sum.re = t1.re + t2.re;
            sum.im = t1.im + t2.im;
            dif.re = t1.re - t2.re;
            dif.im = t1.im - t2.im;
            buf[i1] = sum;
            buf[i2] = dif;
        }
        __syncthreads();
    }
    int64_t *o = out + (size_t)b * G::N;
    for (int j = tid; j < NS4; j += 256) {
        const HpCplx v = hp_cmul(buf[j], pwbar[j]);
        o[j] = (int64_t)(uint64_t)(v.re >> (LOGN - 1));
        o[j + NS4] = (int64_t)(uint64_t)(v.im >> (LOGN - 1));
    }
}

// ----------------------------------------------------------- LWE key switch
// lweKeySwitch (lwe_functions.cpp:136-171) / preKeySwitch (poc:437-465), one workgroup per
// sample, threads over the n_out+1 output coefficients.  ks: [n_in][t][base][n_out+1].
TFHE_GLOBAL void k_keyswitch32(int32_t *__restrict__ out, const int32_t *__restrict__ in,
                               const int32_t *__restrict__ ks, int n_in, int n_out, int t, int basebit,
                               int batch) {
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int base = 1 << basebit;
    const uint32_t mask = (uint32_t)base - 1;
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    const int row = n_out + 1;
    const int32_t *x = in + (size_t)b * (n_in + 1);
    for (int h = threadIdx.x; h < row; h += blockDim.x) {
        uint32_t acc = (h == n_out) ? (uint32_t)x[n_in] : 0u;
        for (int i = 0; i < n_in; i++) {
            const uint32_t aibar = (uint32_t)x[i] + prec_offset;
            for (int j = 0; j < t; j++) {
                const uint32_t aij = (aibar >> (32 - (j + 1) * basebit)) & mask;
                if (aij != 0) acc -= (uint32_t)ks[(((size_t)i * t + j) * base + aij) * row + h];
            }
        }
        out[(size_t)b * row + h] = (int32_t)acc;
    }
}

// ------------------------------------------------ private key switch (circuit bootstrap)
// circuitPrivKS (poc:667-698): LWE64 sample of dimension n2 -> TLWE32 sample, through the key
// privKS[u][i][j][d] (TLWE32 rows of 2*N1 ints, poc:405-419).  Same digit loop as the LWE key
// switch but on 64-bit coefficients (b included as index n2) and with 8 KB rows, so the key
// (1.3 GB per u at the PoC parameters) streams from HBM and the kernel is bound by how often it
// has to be streamed: a workgroup owns a tile of TB samples and a slice of the i range and reads
// each (i,j) block of base-1 candidate rows ONCE for the whole tile.  Reading all base-1 rows for
// a tile costs as much as gathering one row for each of base-1 samples, so the tile only pays
// off beyond that: TB = 32 at base 8 (each thread then keeps EPT = 2 ints of the row per pass
// and a workgroup makes 2*N1/(256*EPT) passes), TB = 16 at base 2 and 4 (EPT = 8, one pass).
// Partial sums are added into the pre-zeroed output with integer atomics (exact, order-free).
//   tab: plane u, reference layout [n2+1][t][base][2*N1]; x: [count][x_stride], inputs 0..n2 of each row used;
//   out: sample s at out + (s % group) * stride_in_group + (s / group) * stride_of_group, 2*N1 ints
//        (the circuit bootstrap runs its l1 gadget levels as l1 groups of one launch)
// FALLBACK for shapes the matrix-core kernel (k_ks_mfma below) does not cover (t * basebit > 32); the
// PoC's 10 x 3 goes through k_ks_mfma.
template <typename XT, int TB, int BB, int EPT, int THREADS>
TFHE_GLOBAL void __launch_bounds__(THREADS)
    k_privks(int32_t *__restrict__ out, long long stride_in_group, long long stride_of_group, int group,
             const XT *__restrict__ x, int x_stride, const int32_t *__restrict__ tab, int n2, int t, int row_ints,
             int count, int i_per_block) {
    using UX = typename std::make_unsigned<XT>::type;
    constexpr int NR = (1 << BB) - 1, BASE = 1 << BB, SEG = THREADS * EPT, W = 8 * (int)sizeof(XT);
    constexpr int ROWS_PER_BLOCK = BASE, FIRST_ROW = 1;
    constexpr UX mask = (UX)BASE - 1;
    const int lane = threadIdx.x & 63;
    const int tile0 = blockIdx.x * TB;
    const int i_begin = blockIdx.y * i_per_block;
    const int i_end = (i_begin + i_per_block < n2 + 1) ? i_begin + i_per_block : n2 + 1;
    const UX prec_offset = (UX)1 << (W - (1 + BB * t));
    // row segments of SEG ints: thread owns ints [seg*SEG + EPT*tid, +EPT)
    for (int seg = 0; seg * SEG < row_ints; seg++) {
        const int e0 = seg * SEG + EPT * (int)threadIdx.x;
        if (e0 >= row_ints) continue;  // (row_ints is a multiple of 8)
        uint32_t acc[TB][EPT];
#pragma unroll
        for (int b = 0; b < TB; b++)
#pragma unroll
            for (int e = 0; e < EPT; e++) acc[b][e] = 0u;
#pragma unroll 1
        for (int i0 = i_begin; i0 < i_end; i0 += 64) {
            // lane L: x[b][i0+L] + prec_offset (0 => all digits 0); the words that carry digits
            constexpr bool TWO_WORDS = sizeof(XT) == 8;
            using AB = typename std::conditional<TWO_WORDS, uint64_t, uint32_t>::type;
            constexpr int WA = TWO_WORDS ? 64 : 32;
            int a0[TB], a1[TWO_WORDS ? TB : 1];
            UX raw[TB];
#pragma unroll
            for (int b = 0; b < TB; b++) {  // unconditional loads from clamped addresses, all issued ...
                const int bs = (tile0 + b < count) ? tile0 + b : count - 1, li = (i0 + lane < i_end) ? i0 + lane : i_end - 1;
                raw[b] = (UX)x[(size_t)bs * x_stride + li];
            }
#pragma unroll
            for (int b = 0; b < TB; b++) TFHE_OPAQUE(raw[b]);  // ... before the first one is waited for
#pragma unroll
            for (int b = 0; b < TB; b++) {
                const bool ok = (tile0 + b < count) && (i0 + lane < i_end);
                const UX v = ok ? (UX)(raw[b] + prec_offset) : (UX)0;
                a0[b] = (int)(uint32_t)v;
                if (TWO_WORDS) a1[TWO_WORDS ? b : 0] = (int)(uint32_t)((uint64_t)v >> 32);
            }
            const int cnt = (i_end - i0 < 64) ? (i_end - i0) : 64;
#pragma unroll 1
            for (int ii = 0; ii < cnt; ii++) {
                AB ab[TB];
#pragma unroll
                for (int b = 0; b < TB; b++) {
                    uint64_t v = (uint32_t)TFHE_READLANE(a0[b], ii);
                    if (TWO_WORDS) v |= (uint64_t)(uint32_t)TFHE_READLANE(a1[TWO_WORDS ? b : 0], ii) << 32;
                    ab[b] = (AB)v;
                }
#pragma unroll 1
                for (int j = 0; j < t; j++) {
                    const int sh = WA - (j + 1) * BB;
                    const int32_t *rows = tab + (((size_t)(i0 + ii) * t + j) * ROWS_PER_BLOCK + FIRST_ROW) * row_ints + e0;
                    uint32_t r[NR][EPT];
#pragma unroll
                    for (int d = 0; d < NR; d++)
#pragma unroll
                        for (int e = 0; e < EPT; e++) r[d][e] = (uint32_t)rows[(size_t)d * row_ints + e];
#pragma unroll
                    for (int b = 0; b < TB; b++) {
                        const uint32_t dig = (uint32_t)((ab[b] >> sh) & (AB)mask);  // wave-uniform
                        if (dig == 0) continue;
#pragma unroll
                        for (int d = 0; d < NR; d++) {
                            if (dig == (uint32_t)(d + 1)) {
                                TFHE_KEEP_BRANCH();
#pragma unroll
                                for (int e = 0; e < EPT; e++) acc[b][e] -= r[d][e];
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < TB; b++) {
            const int s = tile0 + b;
            if (s >= count) continue;
            uint32_t *o = reinterpret_cast<uint32_t *>(out) + (size_t)(s % group) * stride_in_group +
                          (size_t)(s / group) * stride_of_group + e0;
#pragma unroll
            for (int e = 0; e < EPT; e++)
                if (acc[b][e]) atomicAdd(&o[e], acc[b][e]);
        }
    }
}
// ------------------------------------------------ key switch on the matrix cores
// lweKeySwitch (lwe_functions.cpp:136-171), preKeySwitch (poc:437-465) and circuitPrivKS (poc:667-698)
// are one dense contraction over exact integers:
//     out[s][h] = seed[s][h] - sum_{i,j} tab[i][j][dig(s,i,j)][h]        (digit 0 contributes nothing)
//               = seed[s][h] - sum_k A[s][k] * tab[k][h],   k = (i, j, d),  A[s][(i,j,d)] = [dig(s,i,j) == d]
// A is a one-hot matrix generated in registers from the input words; tab is split into its four byte
// limbs, each stored as (limb - 128) so that it is a signed int8 -- the digit-0 rows (value 0) as -128
// too, so that EVERY (i,j) position contributes exactly one row and the bias is the constant
// 128 * n_in * t per limb.  v_mfma_i32_32x32x32_i8 accumulates the limb sums exactly (|sum| <= 2^7 *
// n_in * t < 2^31), the epilogue recombines them modulo 2^32: bit-identical to the reference's loop.
// The key is read once per 256-sample tile (the VALU kernels of round 1 read it once per 16 or 32 samples:
// 1.31 -> 0.77 ms for 4096 gate key switches, 7.1 -> 1.4 ms per privKS plane of 768 samples on MI355X).
//
// Key layout (built once at upload by k_ks_mfma_pack): Bm[hblock][kstep][limb][lane][16] int8 --
// a 1 KB block is exactly the B operand of one MFMA (lane l: column h = 32*hblock + (l & 31), rows
// k = 16*(l >> 5) + 0..15), so a workgroup streams its key slice with fully coalesced 16-byte loads.
// K is laid out per input coefficient i in KPI = ceil(t * base / 32) steps of 32 (positions j >= t
// are zero rows: the one-hot entries generated there multiply zeros).
typedef int v4i __attribute__((vector_size(16)));
typedef int v16i __attribute__((vector_size(64)));

TFHE_HOST_DEVICE int ks_mfma_kpi(int t, int bb) { return (t * (1 << bb) + 31) / 32; }
// K-steps per LDS buffer (a multiple of KPI) and the padded number of K-steps of a key slice (whole chunks;
// the padding steps are zero blocks)
TFHE_HOST_DEVICE int ks_mfma_chunk(int kpi) { return kpi == 3 ? 6 : 8; }
TFHE_HOST_DEVICE int ks_mfma_steps(int n_in, int kpi) {
    const int ch = ks_mfma_chunk(kpi);
    return (n_in * kpi + ch - 1) / ch * ch;
}

TFHE_GLOBAL void k_ks_mfma_pack(int8_t *__restrict__ dst, const int32_t *__restrict__ tab, int n_in, int t, int bb,
                                int row_ints, int hblocks, long long total_frags) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte fragment each
    if (gid >= total_frags) return;
    const int base = 1 << bb, kpi = ks_mfma_kpi(t, bb);
    const int l = (int)(gid & 63), limb = (int)((gid >> 6) & 3);
    const long long ksg = gid >> 8;  // hblock * steps + kstep
    const long long steps = ks_mfma_steps(n_in, kpi);
    const int hb = (int)(ksg / steps), ks = (int)(ksg - (long long)hb * steps);
    const int i = ks / kpi, sub = ks - i * kpi;
    const int h = hb * 32 + (l & 31);
    int8_t bytes[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int kk = 32 * sub + 16 * (l >> 5) + j;
        const int jj = kk >> bb, d = kk & (base - 1);
        int v = 0;
        if (i < n_in && jj < t && h < row_ints) {
            const uint32_t w = d ? (uint32_t)tab[(((size_t)i * t + jj) * base + d) * row_ints + h] : 0u;
            v = (int)((w >> (8 * limb)) & 255u) - 128;
        }
        bytes[j] = (int8_t)v;
    }
    int8_t *o = dst + gid * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) o[j] = bytes[j];
}

struct KsMfmaArgs {
    int32_t *out;             // sample s -> out + (s % group) * stride_in_group + (s / group) * stride_of_group + h
    long long stride_in_group, stride_of_group;
    const void *x;            // [count][x_stride] input words (int32 or int64)
    const int8_t *bm;         // packed key
    int32_t group, x_stride, n_in, t, row_ints, hblocks, count;
    int32_t b_index, b_col;   // seed: out[s][b_col] starts from x[s][b_index] (LWE key switch); b_col < 0: all zero (privKS)
    int32_t ksplit;           // > 1: the K dimension is cut into this many slices, one workgroup each, results added
                              // atomically into a ZEROED out (small batches: a 256-sample tile alone is 24 workgroups
                              // streaming the whole key -- 0.18 ms for one gate key switch; integer sums: any order, same bits)
};

// KPI = ks_mfma_kpi(t, BB) as a template parameter: a chunk of CH K-steps then covers CH / KPI whole input
// coefficients, and which word and which digits a K-step needs is known at compile time.
template <typename XT, int BB, int KPI>
TFHE_GLOBAL void __launch_bounds__(256, 2) k_ks_mfma(KsMfmaArgs A) {
    using UX = typename std::make_unsigned<XT>::type;
    constexpr int W = 8 * (int)sizeof(XT);
    constexpr int CH = (KPI == 3) ? 6 : 8;  // K-steps per LDS buffer, a multiple of KPI (= ks_mfma_chunk)
    constexpr int WPC = CH / KPI;           // input coefficients (words) per chunk
    constexpr int TILE = 256;               // samples per workgroup: 4 waves x 2 row blocks x 32
    TFHE_DYN_LDS(smem);                                               // 2 x CH x 4 KB
    const int wave = TFHE_UNIFORM((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 31, hl = lane >> 5;
    // workgroups that share a key slice (same hblock) get consecutive ids of one XCD (ids b, b+8, ...
    // run on the same XCD): the slice is then served by that XCD's L2
    const int mtiles = (A.count + TILE - 1) / TILE;
    const int per_slice = 8 * mtiles * ((A.hblocks + 7) / 8);
    const int kz = TFHE_UNIFORM((int)blockIdx.x / per_slice);
    const int bid = (int)blockIdx.x - kz * per_slice, y = bid >> 3;
    const int hb = (bid & 7) + 8 * (y / mtiles), mt = y % mtiles;
    if (hb >= A.hblocks) return;
    const int steps = ks_mfma_steps(A.n_in, KPI), chunks_all = steps / CH;
    const int ksplit = A.ksplit > 1 ? A.ksplit : 1;
    const int c_begin = (int)((long long)kz * chunks_all / ksplit), c_end = (int)((long long)(kz + 1) * chunks_all / ksplit);
    const int chunks = c_end - c_begin;  // this workgroup's slice of K: chunks c_begin .. c_end - 1
    if (chunks <= 0) return;
    const int s0 = mt * TILE + wave * 64;
    const bool live = s0 < A.count;  // wave-uniform: a wave without samples only helps staging the key
    const UX prec = (UX)1 << (W - (1 + BB * A.t));
    const XT *xrow[2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int s = s0 + rb * 32 + r;
        xrow[rb] = reinterpret_cast<const XT *>(A.x) + (size_t)(s < A.count ? s : A.count - 1) * A.x_stride;
    }
    // input words of one chunk: the WPC coefficients i its K-steps belong to.  Loaded raw (nothing may
    // touch them until the chunk they are for: arithmetic here would wait for the loads, and with them for
    // the key loads issued before); finish_words adds the rounding offset and keeps the 32 bits that carry
    // the digits (t * BB <= 32)
    // A lane reads ITS sample's WPC consecutive words as one or two 16-byte loads (rows are only word-aligned:
    // the packed type tells the compiler so): per wave-instruction the 32 rows are 32 cache lines whatever the
    // width, and one dword per instruction made these loads as expensive as the chunk's MFMAs.  The last
    // chunk may reach past n_in (zero key rows there): it is read word by word, clamped.
    struct __attribute__((packed, aligned(4))) Words {
        UX v[WPC];
    };
    auto load_words = [&](int c, UX (&w)[2][WPC]) {
        const int i0 = c * WPC;
        if (i0 + WPC <= A.n_in) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const Words tmp = *reinterpret_cast<const Words *>(xrow[rb] + i0);
#pragma unroll
                for (int e = 0; e < WPC; e++) w[rb][e] = tmp.v[e];
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < WPC; e++) {
            const int i = i0 + e;
#pragma unroll
            for (int rb = 0; rb < 2; rb++) w[rb][e] = (UX)xrow[rb][i < A.n_in ? i : A.n_in - 1];
        }
    };
    auto finish_words = [&](uint32_t (&w)[2][WPC], const UX (&raw)[2][WPC]) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int e = 0; e < WPC; e++) w[rb][e] = (uint32_t)((UX)(raw[rb][e] + prec) >> (W - 32));
    };
    const v4i *gsrc = reinterpret_cast<const v4i *>(A.bm) + (size_t)hb * steps * 256;
    auto load_key = [&](int c, v4i (&stage)[CH]) {
#pragma unroll
        for (int e = 0; e < CH; e++) stage[e] = gsrc[(size_t)(c * CH + e) * 256 + threadIdx.x];
    };
    auto store_key = [&](int buf, const v4i (&stage)[CH]) {
        v4i *dst = reinterpret_cast<v4i *>(smem) + buf * (CH * 256);
#pragma unroll
        for (int e = 0; e < CH; e++) dst[e * 256 + threadIdx.x] = stage[e];
    };
    v16i acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int l = 0; l < 4; l++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[rb][l][e] = 0;

    uint32_t wcur[2][WPC];
    UX wraw[2][WPC];
    const int lane_shift = hl * ((16 >> BB) * BB);  // digit bits covered by the lower lane half's 16 k's
    v4i stage[CH];
    load_key(c_begin, stage);
    load_words(c_begin, wraw);
    finish_words(wcur, wraw);
    store_key(0, stage);
    __syncthreads();
    // Two copies of the K loop, chosen once per wave: a wave without samples only stages the key.  (One
    // loop with the MFMAs under a wave-uniform `if` makes hipcc move all 128 accumulators between the VGPR
    // and AGPR halves of the register file on every chunk.)
    auto k_loop = [&](auto compute_tag) {
      constexpr bool COMPUTE = decltype(compute_tag)::value;
#pragma unroll 1
      for (int c = 0; c < chunks; c++) {
        const bool more = c + 1 < chunks;
        if (more) {  // next chunk's key slice and input words: in flight underneath this chunk's MFMAs
            load_key(c_begin + c + 1, stage);
            load_words(c_begin + c + 1, wraw);
        }
        const v4i *kb = reinterpret_cast<const v4i *>(smem) + (c & 1) * (CH * 256) + lane;
        if (COMPUTE) {
            // the B fragments (one ds_read_b128 per limb) of K-step e+1 are requested before the MFMAs of
            // step e: their LDS latency runs under 8 MFMAs
            v4i bf[2][4];
#pragma unroll
            for (int l = 0; l < 4; l++) bf[0][l] = kb[l * 64];
#pragma unroll
            for (int e = 0; e < CH; e++) {
                if (e + 1 < CH) {
#pragma unroll
                    for (int l = 0; l < 4; l++) bf[(e + 1) & 1][l] = kb[((e + 1) * 4 + l) * 64];
                }
                const int sub = e % KPI;  // K-step inside its coefficient: first digit position ((32 * sub) >> BB) + (16 >> BB) * hl
                v4i a[2];
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    // first needed digit moved to the top of the word; digits at positions >= t select zero rows
                    const uint32_t ws = wcur[rb][e / KPI] << ((((32 * sub) >> BB) * BB + lane_shift) & 31);
                    if (BB == 2) {
#pragma unroll
                        for (int u = 0; u < 4; u++) a[rb][u] = (int)(1u << (((ws >> (27 - 2 * u)) & 0x18u)));
                    } else if (BB == 3) {
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const uint64_t one = 1ull << (((ws >> (29 - 3 * u)) & 7u) * 8u);
                            a[rb][2 * u] = (int)(uint32_t)one;
                            a[rb][2 * u + 1] = (int)(uint32_t)(one >> 32);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t d0 = (ws >> (31 - 2 * u)) & 1u, d1 = (ws >> (30 - 2 * u)) & 1u;
                            a[rb][u] = (int)((1u << (8 * d0)) | (1u << (16 + 8 * d1)));
                        }
                    }
                }
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    acc[0][l] = TFHE_MFMA_I8(a[0], bf[e & 1][l], acc[0][l]);
                    acc[1][l] = TFHE_MFMA_I8(a[1], bf[e & 1][l], acc[1][l]);
                }
            }
            // issue order of the chunk: every MFMA is followed by a few of the next K-step's one-hot VALU
            // instructions and every other one by a B-fragment read, so the VALU work and the LDS latency run
            // while the matrix pipe is busy (left alone hipcc issues a K-step's 8 MFMAs back to back and its
            // 26 VALU instructions after them, with the matrix pipe idle)
#pragma unroll
            for (int g = 0; g < CH * 8; g++) {
                TFHE_SCHED_GROUP(0x008, 1);  // 1 MFMA
                TFHE_SCHED_GROUP(0x002, 3);  // 3 VALU (2 and 4 measured within 2 %)
                if ((g & 1) == 0) TFHE_SCHED_GROUP(0x100, 1);  // 1 DS read
            }
        }
        if (more) {
            store_key((c + 1) & 1, stage);
            finish_words(wcur, wraw);
        }
        __syncthreads();
      }
    };
    if (live)
        k_loop(std::true_type{});
    else
        k_loop(std::false_type{});
    // epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int h = hb * 32 + r;
    if (h >= A.row_ints) return;
    // every (i, j) position of this workgroup's input coefficients contributed one row stored as (limb - 128)
    const int i_begin = c_begin * WPC, i_end = c_end * WPC < A.n_in ? c_end * WPC : A.n_in;
    const uint32_t bias = 128u * (uint32_t)(i_end > i_begin ? i_end - i_begin : 0) * (uint32_t)A.t;
    if (!live) return;
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int s = s0 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
            if (s >= A.count) continue;
            uint32_t sum = 0;
#pragma unroll
            for (int l = 0; l < 4; l++) sum += ((uint32_t)acc[rb][l][e] + bias) << (8 * l);
            uint32_t seed = 0;
            if (h == A.b_col && kz == 0) seed = (uint32_t)reinterpret_cast<const XT *>(A.x)[(size_t)s * A.x_stride + A.b_index];
            int32_t *o = A.out + (size_t)(s % A.group) * A.stride_in_group + (size_t)(s / A.group) * A.stride_of_group + h;
            if (ksplit > 1)
                atomicAdd(reinterpret_cast<unsigned *>(o), (unsigned)(seed - sum));  // out was zeroed by the host
            else
                *o = (int32_t)(seed - sum);
        }
    }
}

}  // namespace tfhe
