// tfhe_kernels_generic.h -- the same hot path for EVERY ring degree the reference's FFT plugin accepts.
//
// new_fft_table / new_ifft_table / FFT_Processor_Spqlios(N) take any power of two N >= 16
// (CB/spqlios/spqlios-fft-impl.cpp:157-160,400-403, fft_processor_spqlios.cpp:18-25); the reference only
// instantiates 1024 and 2048, and those two keep the wave-per-polynomial kernels of tfhe_kernels.h.  Every other N
// runs here: a TEAM of work-items (up to a 256-thread workgroup) owns one polynomial, the N/2 complex points live in
// a work buffer of 16-byte points (LDS while it fits, a global scratch slice beyond), and a work-item carries its points
// through two radix-2 layers between workgroup barriers (three at the end).
// The per-node arithmetic is the reference's, operation for operation (twist, dif/dit butterflies
// with their FMA placement, the multiplication-free size-4 and size-2 steps, the four-product final twist), through
// the same helper functions as the tuned kernels (dif_bfly, dit_bfly, Torus<T>::from_double, the AddMul chain) --
// so results are bit-identical to the CPU path for any N; which work-item computes a node does not enter.
// Speed is not the point of this file (profiles/r06_generic_n.txt says what it is).
//
// Layout differences from the tuned kernels: Lagrange-domain key rows are kept in the REFERENCE's position order
// ([row][N/2] complex, scaled by 2/N) -- there is no lane geometry to pre-arrange them for.
#pragma once
#include "tfhe_kernels.h"

namespace tfhe {

// modSwitchFromTorus32(phase, 2N), numeric_functions.cpp:54-60, ring degree 2^logn at run time
TFHE_DEVICE int modswitch_rt(int32_t phase, int logn) {
    const int sh = 63 - logn;
    const uint64_t half = 1ull << (sh - 1);
    return (int)((((uint64_t)(uint32_t)phase << 32) + half) >> sh);
}

// ---- work buffers hold one 16-byte complex point per element (re = coefficient / position j, im = j + N/2 of the reference's
// split storage): one ds_read_b128 / ds_write_b128 per point
struct GenC {
    double r, i;
};
TFHE_DEVICE GenC gen_ld(const double2 *buf, int j) {
    const double2 v = buf[j];
    return GenC{v.x, v.y};
}
TFHE_DEVICE void gen_st(double2 *buf, int j, const GenC &v) { buf[j] = make_double2(v.r, v.i); }
// WHERE point j of a polynomial sits in its work buffer: j with its low four bits XOR-ed by bits of j >> 3.  Unswizzled, the
// pass on eight consecutive points per work-item (lane stride 128 bytes) is an 8-way LDS bank conflict in both directions and
// the pass before it a 2-way one -- 56 % of all LDS cycles of the blind rotation at N = 512 were conflict cycles
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r06_generic_n.txt).  With this map every pass of every N is conflict-free
// under the gfx950 banking rules for ds_read_b128 / ds_write_b128 (tools/lds_conflicts.py, generic part).  The map is linear
// over GF(2) and touches bits 0..3 only, so for an index a + b whose parts share no bits, gen_sw(a + b) = gen_sw(a) ^ gen_sw(b):
// a pass swizzles one index per item and XORs pass-uniform constants for the other points.
TFHE_HOST_DEVICE int gen_sw(int j) { return j ^ (((j >> 3) & 7) ^ ((j >> 2) & 8) ^ ((j >> 6) & 1)); }
// the two multiplication-free layers on one group of four consecutive points
// inverse: size-4 (spqlios-ifft-fma.s:194-213) then size-2 (:247-263)
TFHE_DEVICE void gen_ifft_tail4(GenC (&x)[4]) {
    const double r0 = x[0].r, r1 = x[1].r, r2 = x[2].r, r3 = x[3].r;
    const double i0 = x[0].i, i1 = x[1].i, i2 = x[2].i, i3 = x[3].i;
    const double a0 = r0 + r2, a1 = r1 + r3, a2 = r0 - r2, a3 = i3 - i1;
    const double b0 = i0 + i2, b1 = i1 + i3, b2 = i0 - i2, b3 = r1 - r3;
    x[0] = GenC{a0 + a1, b0 + b1};
    x[1] = GenC{a0 - a1, b0 - b1};
    x[2] = GenC{a2 + a3, b2 + b3};
    x[3] = GenC{a2 - a3, b2 - b3};
}
// direct: size-2 (spqlios-fft-fma.s:79-95) then size-4 (:134-152)
TFHE_DEVICE void gen_fft_head4(GenC (&x)[4]) {
    const double r0 = x[0].r + x[1].r, r1 = x[0].r - x[1].r, r2 = x[2].r + x[3].r, r3 = x[2].r - x[3].r;
    const double i0 = x[0].i + x[1].i, i1 = x[0].i - x[1].i, i2 = x[2].i + x[3].i, i3 = x[2].i - x[3].i;
    x[0] = GenC{r0 + r2, i0 + i2};
    x[1] = GenC{r1 + i3, i1 - r3};
    x[2] = GenC{r0 - r2, i0 - i2};
    x[3] = GenC{r1 - i3, i1 + r3};
}

// What separates two phases that exchange points through the work buffer: a workgroup barrier when the team is (part of) a
// workgroup, or -- WAVE: the team is exactly one wavefront -- nothing but a compiler fence, since a wave's LDS instructions execute
// in order (the idiom of the tuned kernels).  WAVE teams of one workgroup never wait for each other.
template <bool WAVE>
TFHE_DEVICE void gen_sync() {
    if constexpr (WAVE)
        TFHE_WAVE_FENCE();
    else
        __syncthreads();
}

// How the `tpp` work-items of a team share a pass of `count` independent items (butterflies, groups of 4 or 8 points) on np
// polynomials: while count >= tpp every work-item walks the items and takes all polynomials of each (one set of twiddles per
// item); a shorter pass is cut into g = tpp / count sub-teams that take the polynomials s, s + g, ... -- instead of leaving
// the work-items beyond `count` idle.  count and tpp are powers of two.
struct GenSplit {
    int tq, g, s, l2;  // work-items per polynomial, sub-teams, this work-item's sub-team and its number in it
};
TFHE_DEVICE GenSplit gen_split(int count, int lt, int tpp) {
    if (count >= tpp) return GenSplit{tpp, 1, 0, lt};
    const int sh = __builtin_ctz((unsigned)count);
    return GenSplit{count, tpp >> sh, lt >> sh, lt & (count - 1)};
}

// One item of a pass on the polynomials of this work-item's sub-team: its NPT points (idx: their swizzled positions, the same in
// every polynomial) are read, f does the butterflies on GenC (&)[NPT], and they are written back.
// LD / ST (GenNone: the buffer): the caller's source of the points (ld(p, k): point k of the item in polynomial p) and sink of the
// results (st(p, k, value)) -- how the blind rotation feeds the first layer from the accumulator's digits and takes the last layer's
// output into the accumulator without a pass through the buffer.
// (Several polynomials in flight per item -- all loads, all butterflies, all stores -- was measured SLOWER at 2 and at 4: N = 512
// 2.33 -> 2.43 / 2.54 ms per 4096 x 64 CMux, registers 147 -> 200; profiles/r06_generic_n.txt.)
struct GenNone {};
template <int NPT, class F, class LD = GenNone, class ST = GenNone>
TFHE_DEVICE void gen_item(double2 *buf, int np, long pstride, const GenSplit &S, const int (&idx)[NPT], F &&f, LD ld = LD(), ST st = ST()) {
    for (int p = S.s; p < np; p += S.g) {
        double2 *x = buf + p * pstride;
        GenC v[NPT];
#pragma unroll
        for (int k = 0; k < NPT; k++) {
            if constexpr (std::is_same<LD, GenNone>::value)
                v[k] = gen_ld(x, idx[k]);
            else
                v[k] = ld(p, k);
        }
        f(v);
#pragma unroll
        for (int k = 0; k < NPT; k++) {
            if constexpr (std::is_same<ST, GenNone>::value)
                gen_st(x, idx[k], v[k]);
            else
                st(p, k, v[k]);
        }
    }
}

// Coefficient -> Lagrange, in place, for `np` polynomials `pstride` points apart; each is NC complex points holding
// a_j + i a_{j+NC} on entry (the fold of spqlios-ifft-fma.s:40-44).  A team of `tpp` work-items (this one is number `lt`)
// shares the work; EVERY work-item of the workgroup must call this (the barriers are workgroup barriers), `active` = false
// for those whose team has no polynomial.  tw: the kernels' table (tfhe_amd.hip build_tables: [0,NC) twist, half-size h at
// 2 NC - 2 h).  The caller has synchronised its writes; on return every result is visible to the whole workgroup.
// Barriers: the layers are the reference's radix-2 layers, node for node, but a work-item carries its points through
// TWO layers (four points) between barriers, and through the last three (h = 4, size 4, size 2: eight consecutive points).
// FIRST (GenNone: the buffer holds them): first(p, j) = point j of polynomial p, asked for exactly once.
template <bool WAVE = false, class FIRST = GenNone>
TFHE_DEVICE void gen_ifft(double2 *buf, int np, long pstride, int NC, const double2 *__restrict__ tw, int lt, int tpp, bool active,
                          FIRST first = FIRST()) {
    {  // layer h = NC/2 with the twist by omega^j (spqlios-ifft-fma.s:63-78) fused: its butterfly owns both points
        const int h = NC >> 1;
        if (active) {
            const double2 *ts = tw + NC;  // 2 NC - 2 h
            const GenSplit S = gen_split(h, lt, tpp);
            const int sh = gen_sw(h);
            for (int bf = S.l2; bf < h; bf += S.tq) {
                const double2 w = ts[bf], w1 = tw[bf], w2 = tw[bf + h];
                const int ia = gen_sw(bf);
                const int idx[2] = {ia, ia ^ sh};
                auto twist_bfly = [&](GenC(&v)[2]) {
                    const GenC ta{__builtin_fma(-v[0].i, w1.y, v[0].r * w1.x), __builtin_fma(v[0].i, w1.x, v[0].r * w1.y)};
                    const GenC tb{__builtin_fma(-v[1].i, w2.y, v[1].r * w2.x), __builtin_fma(v[1].i, w2.x, v[1].r * w2.y)};
                    v[0] = ta;
                    v[1] = tb;
                    dif_bfly(v[0].r, v[0].i, v[1].r, v[1].i, w.x, w.y);
                };
                if constexpr (std::is_same<FIRST, GenNone>::value)
                    gen_item<2>(buf, np, pstride, S, idx, twist_bfly);
                else
                    gen_item<2>(buf, np, pstride, S, idx, twist_bfly, [&](int p, int k) { return first(p, k ? bf + h : bf); });
            }
        }
        gen_sync<WAVE>();
    }
    if (NC > 8) {
        int h = NC >> 2, layers = 0;  // middle layers h = NC/4 .. 8
        for (int t = h; t >= 8; t >>= 1) layers++;
        if (layers & 1) {  // an odd one out: alone
            if (active) {
                const double2 *ts = tw + (2 * NC - 2 * h);
                const GenSplit S = gen_split(NC >> 1, lt, tpp);
                const int sh = gen_sw(h);
                for (int bf = S.l2; bf < (NC >> 1); bf += S.tq) {
                    const int off = bf & (h - 1), i1 = gen_sw(((bf - off) << 1) + off);
                    const double2 w = ts[off];
                    const int idx[2] = {i1, i1 ^ sh};
                    gen_item<2>(buf, np, pstride, S, idx, [&](GenC(&v)[2]) { dif_bfly(v[0].r, v[0].i, v[1].r, v[1].i, w.x, w.y); });
                }
            }
            gen_sync<WAVE>();
            h >>= 1;
        }
        for (; h >= 16; h >>= 2) {  // layers h and h/2 on the four points base + {0, h/2, h, 3h/2}
            if (active) {
                const int hh = h >> 1;
                const double2 *ta = tw + (2 * NC - 2 * h), *tb = tw + (2 * NC - h);
                const GenSplit S = gen_split(NC >> 2, lt, tpp);
                const int s1 = gen_sw(hh), s2 = gen_sw(h), s3 = s1 ^ s2;
                for (int q4 = S.l2; q4 < (NC >> 2); q4 += S.tq) {
                    const int off = q4 & (hh - 1), base = gen_sw(((q4 - off) << 2) + off);
                    const double2 wa0 = ta[off], wa1 = ta[off + hh], wb = tb[off];
                    const int idx[4] = {base, base ^ s1, base ^ s2, base ^ s3};
                    gen_item<4>(buf, np, pstride, S, idx, [&](GenC(&v)[4]) {
                        dif_bfly(v[0].r, v[0].i, v[2].r, v[2].i, wa0.x, wa0.y);
                        dif_bfly(v[1].r, v[1].i, v[3].r, v[3].i, wa1.x, wa1.y);
                        dif_bfly(v[0].r, v[0].i, v[1].r, v[1].i, wb.x, wb.y);
                        dif_bfly(v[2].r, v[2].i, v[3].r, v[3].i, wb.x, wb.y);
                    });
                }
            }
            gen_sync<WAVE>();
        }
        // layer h = 4, then size 4 and size 2, on eight consecutive points
        if (active) {
            const double2 *t4 = tw + (2 * NC - 8);
            const double2 w0 = t4[0], w1 = t4[1], w2 = t4[2], w3 = t4[3];
            const GenSplit S = gen_split(NC >> 3, lt, tpp);
            for (int g = S.l2; g < (NC >> 3); g += S.tq) {
                const int g8 = gen_sw(8 * g);  // point 8 g + k sits at g8 ^ k
                const int idx[8] = {g8, g8 ^ 1, g8 ^ 2, g8 ^ 3, g8 ^ 4, g8 ^ 5, g8 ^ 6, g8 ^ 7};
                gen_item<8>(buf, np, pstride, S, idx, [&](GenC(&v)[8]) {
                    dif_bfly(v[0].r, v[0].i, v[4].r, v[4].i, w0.x, w0.y);
                    dif_bfly(v[1].r, v[1].i, v[5].r, v[5].i, w1.x, w1.y);
                    dif_bfly(v[2].r, v[2].i, v[6].r, v[6].i, w2.x, w2.y);
                    dif_bfly(v[3].r, v[3].i, v[7].r, v[7].i, w3.x, w3.y);
                    GenC lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
                    gen_ifft_tail4(lo);
                    gen_ifft_tail4(hi);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        v[k] = lo[k];
                        v[4 + k] = hi[k];
                    }
                });
            }
        }
        gen_sync<WAVE>();
        return;
    }
    if (active) {  // NC = 8: the fused first layer was h = 4 (gen_sw is the identity below 8)
        const GenSplit S = gen_split(NC >> 2, lt, tpp);
        for (int g = S.l2; g < (NC >> 2); g += S.tq) {
            const int idx[4] = {4 * g, 4 * g + 1, 4 * g + 2, 4 * g + 3};
            gen_item<4>(buf, np, pstride, S, idx, [&](GenC(&v)[4]) { gen_ifft_tail4(v); });
        }
    }
    gen_sync<WAVE>();
}

// Lagrange -> coefficient, in place (the caller has applied the 2/N scale); same calling rules as gen_ifft.
// (the reference's fft table is the conjugate of its ifft table except cos at the quarter turn: flip_sign_if)
// LAST (GenNone: left in the buffer): last(p, j, value) takes point j of polynomial p instead.
template <bool WAVE = false, class LAST = GenNone>
TFHE_DEVICE void gen_fft(double2 *buf, int np, long pstride, int NC, const double2 *__restrict__ tw, int lt, int tpp, bool active,
                         LAST last = LAST()) {
    if (NC > 8) {
        // size 2, size 4, then layer h = 4, on eight consecutive points
        if (active) {
            const double2 *t4 = tw + (2 * NC - 8);
            const double2 w0 = t4[0], w1 = t4[1], w2 = t4[2], w3 = t4[3];
            const GenSplit S = gen_split(NC >> 3, lt, tpp);
            for (int g = S.l2; g < (NC >> 3); g += S.tq) {
                const int g8 = gen_sw(8 * g);
                const int idx[8] = {g8, g8 ^ 1, g8 ^ 2, g8 ^ 3, g8 ^ 4, g8 ^ 5, g8 ^ 6, g8 ^ 7};
                gen_item<8>(buf, np, pstride, S, idx, [&](GenC(&v)[8]) {
                    GenC lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
                    gen_fft_head4(lo);
                    gen_fft_head4(hi);
                    dit_bfly(lo[0].r, lo[0].i, hi[0].r, hi[0].i, w0.x, w0.y);
                    dit_bfly(lo[1].r, lo[1].i, hi[1].r, hi[1].i, w1.x, w1.y);
                    dit_bfly(lo[2].r, lo[2].i, hi[2].r, hi[2].i, -w2.x, w2.y);  // quarter turn: off == h/2
                    dit_bfly(lo[3].r, lo[3].i, hi[3].r, hi[3].i, w3.x, w3.y);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        v[k] = lo[k];
                        v[4 + k] = hi[k];
                    }
                });
            }
        }
        gen_sync<WAVE>();
        int layers = 0;  // middle layers h = 8 .. NC/4
        for (int t = 8; t <= (NC >> 2); t <<= 1) layers++;
        int h = 8;
        for (int pr = 0; pr < (layers >> 1); pr++, h <<= 2) {  // layers h and 2h on the four points base + {0, h, 2h, 3h}
            if (active) {
                const int h2 = h << 1;
                const double2 *ta = tw + (2 * NC - 2 * h), *tb = tw + (2 * NC - 2 * h2);
                const GenSplit S = gen_split(NC >> 2, lt, tpp);
                const int s1 = gen_sw(h), s2 = gen_sw(h2), s3 = s1 ^ s2;
                for (int q4 = S.l2; q4 < (NC >> 2); q4 += S.tq) {
                    const int off = q4 & (h - 1), base = gen_sw(((q4 - off) << 2) + off);
                    const double2 wa = ta[off], wb0 = tb[off], wb1 = tb[off + h];
                    const double wac = flip_sign_if(wa.x, off == (h >> 1));
                    const double wb1c = flip_sign_if(wb1.x, off == 0);  // off + h == h2 / 2
                    const int idx[4] = {base, base ^ s1, base ^ s2, base ^ s3};
                    gen_item<4>(buf, np, pstride, S, idx, [&](GenC(&v)[4]) {
                        dit_bfly(v[0].r, v[0].i, v[1].r, v[1].i, wac, wa.y);
                        dit_bfly(v[2].r, v[2].i, v[3].r, v[3].i, wac, wa.y);
                        dit_bfly(v[0].r, v[0].i, v[2].r, v[2].i, wb0.x, wb0.y);
                        dit_bfly(v[1].r, v[1].i, v[3].r, v[3].i, wb1c, wb1.y);
                    });
                }
            }
            gen_sync<WAVE>();
        }
        if (layers & 1) {  // the odd one out: h == NC/4
            if (active) {
                const double2 *ts = tw + (2 * NC - 2 * h);
                const GenSplit S = gen_split(NC >> 1, lt, tpp);
                const int sh = gen_sw(h);
                for (int bf = S.l2; bf < (NC >> 1); bf += S.tq) {
                    const int off = bf & (h - 1), i1 = gen_sw(((bf - off) << 1) + off);
                    const double2 w = ts[off];
                    const double wc = flip_sign_if(w.x, off == (h >> 1));
                    const int idx[2] = {i1, i1 ^ sh};
                    gen_item<2>(buf, np, pstride, S, idx, [&](GenC(&v)[2]) { dit_bfly(v[0].r, v[0].i, v[1].r, v[1].i, wc, w.y); });
                }
            }
            gen_sync<WAVE>();
        }
    } else {
        if (active) {
            const GenSplit S = gen_split(NC >> 2, lt, tpp);
            for (int g = S.l2; g < (NC >> 2); g += S.tq) {
                const int idx[4] = {4 * g, 4 * g + 1, 4 * g + 2, 4 * g + 3};
                gen_item<4>(buf, np, pstride, S, idx, [&](GenC(&v)[4]) { gen_fft_head4(v); });
            }
        }
        gen_sync<WAVE>();
    }
    {  // layer h = NC/2 with the final twist by conj(omega^j), four rounded products (spqlios-fft-fma.s:255-274)
        const int h = NC >> 1;
        if (active) {
            const double2 *ts = tw + NC;
            const GenSplit S = gen_split(h, lt, tpp);
            const int sh = gen_sw(h);
            for (int bf = S.l2; bf < h; bf += S.tq) {
                const double2 w = ts[bf], w1 = tw[bf], w2 = tw[bf + h];
                const double wc = flip_sign_if(w.x, bf == (h >> 1));
                const int ia = gen_sw(bf);
                const int idx[2] = {ia, ia ^ sh};
                auto bfly_twist = [&](GenC(&v)[2]) {
                    dit_bfly(v[0].r, v[0].i, v[1].r, v[1].i, wc, w.y);
                    const double arc = v[0].r * w1.x, ars = v[0].r * w1.y, aic = v[0].i * w1.x, ais = v[0].i * w1.y;
                    const double brc = v[1].r * w2.x, brs = v[1].r * w2.y, bic = v[1].i * w2.x, bis = v[1].i * w2.y;
                    v[0] = GenC{arc + ais, aic - ars};
                    v[1] = GenC{brc + bis, bic - brs};
                };
                if constexpr (std::is_same<LAST, GenNone>::value)
                    gen_item<2>(buf, np, pstride, S, idx, bfly_twist);
                else
                    gen_item<2>(buf, np, pstride, S, idx, bfly_twist, GenNone(), [&](int p, int k, const GenC &v) { last(p, k ? bf + h : bf, v); });
            }
        }
        gen_sync<WAVE>();
    }
}

// How a workgroup of `block` work-items is cut into teams: one team per polynomial, at most NC/2 butterflies per layer each.
// The host picks the workgroup (gen_block): 256 up to N = 4096, the whole 1024 beyond -- measured on MI355X: 512 instead of 256
// loses 6 % at N = 512, gains 4 % at 4096 and 34 % at 16384 (profiles/r06_generic_n.txt)
constexpr int GEN_BLOCK_MAX = 1024;
constexpr int GEN_WAVE_BLOCK = 256;  // workgroup of the one-wave-per-ciphertext blind rotation: four waves
TFHE_HOST_DEVICE int gen_block(int N) { return N >= 8192 ? GEN_BLOCK_MAX : 256; }
TFHE_HOST_DEVICE int gen_team_size(int NC, int block) { return (NC >> 1) < block ? (NC >> 1) : block; }

// ------------------------------------------------------------ FFT plugin boundary, any N
// execute_reverse_int / _torus32 / _torus64 and the bare `ifft` (TIN = double): coefficients -> LagrangeHalfC.
// PACK: write the key layout of kg_blind_rotate instead ([row][NC] complex in reference order, scaled by 2/N).
// LDS: the transform buffers are dynamic LDS (teams x N doubles; a template parameter so that the compiler addresses them
// with DS instructions instead of flat ones), else slices of the global scratch `work`, one per workgroup.
template <typename TIN, bool PACK, bool LDS>
TFHE_GLOBAL void __launch_bounds__(GEN_BLOCK_MAX)
    kg_ifft_batch(double *__restrict__ out, const TIN *__restrict__ in, const double2 *__restrict__ tw, int batch, int logn,
                  double *__restrict__ work) {
    const int N = 1 << logn, NC = N >> 1;
    const int tpp = gen_team_size(NC, (int)blockDim.x), teams = (int)blockDim.x / tpp;
    const int team = (int)threadIdx.x / tpp, lt = (int)threadIdx.x - team * tpp;
    TFHE_DYN_LDS(smem);
    double2 *buf = reinterpret_cast<double2 *>(LDS ? reinterpret_cast<double *>(smem) : work + (size_t)blockIdx.x * teams * N) + (size_t)team * NC;
    for (int b0 = (int)blockIdx.x * teams; b0 < batch; b0 += (int)gridDim.x * teams) {  // workgroup-uniform
        const int b = b0 + team;
        const bool active = b < batch;
        if (active) {
            const TIN *p = in + (size_t)b * N;
            for (int j = lt; j < NC; j += tpp) buf[gen_sw(j)] = make_double2((double)p[j], (double)p[j + NC]);
        }
        __syncthreads();
        gen_ifft(buf, 1, 0, NC, tw, lt, tpp, active);
        if (active) {
            if (PACK) {
                double2 *o = reinterpret_cast<double2 *>(out) + (size_t)b * NC;
                const double scale = 2.0 / (double)N;  // exact: a power of two
                for (int j = lt; j < NC; j += tpp) {
                    const double2 v = buf[gen_sw(j)];
                    o[j] = make_double2(v.x * scale, v.y * scale);
                }
            } else {
                double *o = out + (size_t)b * N;
                for (int j = lt; j < NC; j += tpp) {
                    const double2 v = buf[gen_sw(j)];
                    o[j] = v.x;
                    o[NC + j] = v.y;
                }
            }
        }
        __syncthreads();  // the buffer is refilled by the next polynomial
    }
}

// execute_direct_torus32 / _torus64 (scale 2/N, transform, round as fft_processor_spqlios.cpp:102,131-142) and the bare
// `fft` (TOUT = double: no scale, no rounding)
template <typename TOUT, bool LDS>
TFHE_GLOBAL void __launch_bounds__(GEN_BLOCK_MAX)
    kg_fft_batch(TOUT *__restrict__ out, const double *__restrict__ in, const double2 *__restrict__ tw, int batch, int logn,
                 double *__restrict__ work) {
    const int N = 1 << logn, NC = N >> 1;
    const int tpp = gen_team_size(NC, (int)blockDim.x), teams = (int)blockDim.x / tpp;
    const int team = (int)threadIdx.x / tpp, lt = (int)threadIdx.x - team * tpp;
    constexpr bool RAW = std::is_same<TOUT, double>::value;
    TFHE_DYN_LDS(smem);
    double2 *buf = reinterpret_cast<double2 *>(LDS ? reinterpret_cast<double *>(smem) : work + (size_t)blockIdx.x * teams * N) + (size_t)team * NC;
    const double scale = 2.0 / (double)N;  // fft_processor_spqlios.cpp:78
    for (int b0 = (int)blockIdx.x * teams; b0 < batch; b0 += (int)gridDim.x * teams) {
        const int b = b0 + team;
        const bool active = b < batch;
        if (active) {
            const double *p = in + (size_t)b * N;
            for (int j = lt; j < NC; j += tpp) buf[gen_sw(j)] = RAW ? make_double2(p[j], p[NC + j]) : make_double2(p[j] * scale, p[NC + j] * scale);
        }
        __syncthreads();
        gen_fft(buf, 1, 0, NC, tw, lt, tpp, active);
        if (active) {
            TOUT *o = out + (size_t)b * N;
            for (int j = lt; j < NC; j += tpp) {
                const double2 v = buf[gen_sw(j)];
                if constexpr (RAW) {
                    o[j] = v.x;
                    o[NC + j] = v.y;
                } else {
                    o[j] = Torus<TOUT>::from_double(v.x);
                    o[NC + j] = Torus<TOUT>::from_double(v.y);
                }
            }
        }
        __syncthreads();
    }
}

// Key upload: LagrangeHalfC polynomials (reference order) -> [rows][NC] complex, scaled by 2/N
TFHE_GLOBAL void kg_pack_gsw(double2 *__restrict__ dst, const double *__restrict__ src, long long rows, int logn) {
    const int N = 1 << logn, NC = N >> 1;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * NC) return;
    const long long row = gid >> (logn - 1);
    const int e = (int)(gid - row * NC);
    const double scale = 2.0 / (double)N;
    const double *p = src + row * N;
    dst[gid] = make_double2(p[e] * scale, p[e + NC] * scale);
}

TFHE_GLOBAL void kg_modswitch(int32_t *__restrict__ out, const int32_t *__restrict__ in, long long total, int logn) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) out[gid] = modswitch_rt(in[gid], logn);
}

// ------------------------------------------------------------ blind rotation, any N
// One workgroup per ciphertext (persistent: the grid walks the batch), every flag of k_blind_rotate honoured with the
// same meaning.  Work areas: acc [2][N] torus, dig [nd][N/2] complex points (nd gadget digits at a time: extract, transform,
// multiply-accumulate, discard), fac [2][N/2] complex points (the Fourier accumulator of tLweFFTClear / tLweFFTAddMulRTo).
// Each of the three is in dynamic LDS when its offset is >= 0, else in the workgroup's slice of `work`.
struct GenBrPlace {
    long long acc_lds, dig_lds, fac_lds;  // byte offsets into dynamic LDS, or -1: global
    long long work_stride;                // bytes of global scratch per workgroup
    unsigned char *work;
    int logn;
    int nd;  // gadget digits transformed together: 2l, l or 1 (dig holds nd polynomials)
    int fac_in_dig;  // nd == 2l: no separate Fourier accumulator -- the multiply-accumulate of point j reads dig[0 .. 2l)[j] and
                     // leaves its two results in dig[0][j] and dig[1][j] (same work-item), which the direct transforms then take
    long long wave_bytes;  // WAVE form: LDS bytes of one wavefront's three areas
};

// ALL_LDS: the three areas are dynamic LDS at the offsets of G (compile-time knowledge: DS instructions, not flat ones)
// WAVE (with ALL_LDS): ONE WAVEFRONT per ciphertext, its three areas in its own slice of the workgroup's LDS (G.wave_bytes apart);
// the waves of a workgroup are independent ciphertexts and no workgroup barrier exists (gen_sync<true>)
constexpr int GEN_BR_BLOCK = 1024;  // widest workgroup of the blind rotation (a ciphertext of N >= 2048: two waves per SIMD between barriers)
template <typename T, bool ALL_LDS, bool WAVE = false>
TFHE_GLOBAL void __launch_bounds__(WAVE ? GEN_WAVE_BLOCK : GEN_BR_BLOCK) kg_blind_rotate(BlindRotateArgs<T> A, GenBrPlace G) {
    static_assert(!WAVE || ALL_LDS, "wave teams keep everything in LDS");
    using U = typename Torus<T>::U;
    constexpr int BITS = Torus<T>::BITS;
    const int logn = G.logn, N = 1 << logn, NC = N >> 1;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nt = WAVE ? 64 : (int)blockDim.x;
    const int team = WAVE ? (int)(threadIdx.x >> 6) : 0, teams = WAVE ? (int)(blockDim.x >> 6) : 1;
    TFHE_DYN_LDS(smem_all);
    unsigned char *smem = smem_all + (WAVE ? (size_t)team * (size_t)G.wave_bytes : 0);
    const int nd = G.nd;
    T *acc;
    double2 *dig, *fac;  // [nd][NC] and [2][NC] complex points
    if constexpr (ALL_LDS) {
        acc = reinterpret_cast<T *>(smem);
        dig = reinterpret_cast<double2 *>(smem + sizeof(T) * 2 * (size_t)N);
        fac = G.fac_in_dig ? dig : dig + (size_t)nd * NC;
    } else {
        unsigned char *wsl = G.work ? G.work + (size_t)blockIdx.x * (size_t)G.work_stride : nullptr;
        size_t woff = 0;
        auto place = [&](long long lds_off, size_t bytes) -> unsigned char * {
            if (lds_off >= 0) return smem + lds_off;
            unsigned char *p = wsl + woff;
            woff += bytes;
            return p;
        };
        acc = reinterpret_cast<T *>(place(G.acc_lds, sizeof(T) * 2 * (size_t)N));
        dig = reinterpret_cast<double2 *>(place(G.dig_lds, sizeof(double) * (size_t)nd * N));
        fac = G.fac_in_dig ? dig : reinterpret_cast<double2 *>(place(G.fac_lds, sizeof(double) * 2 * (size_t)N));
    }
    const U offset = (U)A.gd.offset;
    const int Bgbit = A.gd.Bgbit, l = A.gd.l;
    const U mask = ((U)1 << Bgbit) - 1;
    const int32_t halfBg = 1 << (Bgbit - 1);
    const bool rotate = !(A.flags & BR_NO_ROTATE);

    for (int ct = (int)blockIdx.x * teams + team; ct < A.batch; ct += (int)gridDim.x * teams) {  // team-uniform
        const int32_t *rot = A.rot + (size_t)ct * A.rot_stride;
        // ---- accumulator initialisation (as k_blind_rotate)
        if (A.flags & BR_INIT_TESTVEC) {
            int barb = rot[A.n_steps];
            if (A.flags & BR_MODSWITCH) barb = modswitch_rt(barb, logn);
            const int a0 = (2 * N - barb) & (2 * N - 1);  // lwe_functions.cpp:385-386
            const T *tv = A.tv + (size_t)ct * A.tv_stride;
            for (int j = tid; j < N; j += nt) {
                const int idx = (j - a0) & (2 * N - 1), src = idx & (N - 1);
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
        } else if (A.flags & BR_CMUX_DATA) {
            const size_t item = A.cmux_period ? (size_t)(ct % A.cmux_period) : (size_t)ct;
            const T *d0 = A.cmux_d0 + item * A.cmux_stride, *d1 = A.cmux_d1 + item * A.cmux_stride;
            if (A.flags & BR_CMUX_TRIVIAL) {
                for (int j = tid; j < N; j += nt) {
                    acc[j] = 0;
                    acc[N + j] = (T)((U)d1[j] - (U)d0[j]);
                }
            } else {
                for (int j = tid; j < 2 * N; j += nt) acc[j] = (T)((U)d1[j] - (U)d0[j]);
            }
        } else {
            const T *src = A.acc_io + (size_t)ct * 2 * N;
            for (int j = tid; j < 2 * N; j += nt) acc[j] = src[j];
        }
        gen_sync<WAVE>();
        const double2 *bk0 = A.bk;
        if (A.gsw_sel)
            bk0 += (size_t)A.gsw_sel[ct] * A.gsw_sample_stride;
        else if (A.sel_div > 0)
            bk0 += (size_t)((ct / A.sel_div) * A.sel_mul + A.sel_add) * A.gsw_sample_stride;

        // ---- CMux loop (lwe_functions.cpp:337-361)
        for (int i = 0; i < A.n_steps; i++) {
            int a = 0;
            if (rotate) {
                a = rot[i];
                if (A.flags & BR_MODSWITCH) a = modswitch_rt(a, logn);
                if (a == 0) continue;  // :348-350 (workgroup-uniform)
            }
            const double2 *bkrow = bk0 + (size_t)i * A.bk_step_stride;
            // rows p = q*l + d (tgsw_functions.cpp:435-443) in groups of nd: extract, transform together, multiply-accumulate
            // in row order (the chain of lagrangehalfc_impl_fma.s:96-107 is sequential in p for every point)
            for (int p0 = 0; p0 < 2 * l; p0 += nd) {
                // digit d of polynomial q of (X^a - 1) * acc (numeric_functions.cpp:304-323), or of acc itself; decomposition
                // tgsw_functions.cpp:224-337 / poc:492-515 (offset from the host); point j = (coefficient j, coefficient j + N/2).
                // The first layer of the transform asks for every point once: the digits never sit in the buffer untransformed.
                // (all four reads of a point are issued before the first is used, and `rotate` is a select, not a branch: a branch per
                // coefficient made every read wait for the one before it)
                auto digit_point = [&](int e, int j) {
                    const int q = (p0 + e) >= l ? 1 : 0, d = (p0 + e) - q * l;
                    const T *pa = acc + q * N;
                    const int decal = BITS - (d + 1) * Bgbit;
                    const int i0 = (j - a) & (2 * N - 1), i1 = (i0 + NC) & (2 * N - 1);  // a == 0 without rotation
                    const U c0 = (U)pa[j], c1 = (U)pa[j + NC], s0 = (U)pa[i0 & (N - 1)], s1 = (U)pa[i1 & (N - 1)];
                    U v0 = ((i0 & N) ? (U)(0 - s0) : s0) - c0, v1 = ((i1 & N) ? (U)(0 - s1) : s1) - c1;
                    v0 = rotate ? v0 : c0;
                    v1 = rotate ? v1 : c1;
                    return GenC{(double)((int32_t)(((U)(v0 + offset) >> decal) & mask) - halfBg),
                                (double)((int32_t)(((U)(v1 + offset) >> decal) & mask) - halfBg)};
                };
                gen_ifft<WAVE>(dig, nd, NC, NC, A.tw, tid, nt, true, digit_point);
                // tLweFFTAddMulRTo (tlwe_functions.cpp:318-325), both output polynomials, on an accumulator that starts as
                // the +0 of tLweFFTClear (tgsw_functions.cpp:438)
                for (int j = tid; j < NC; j += nt) {
                    const int sj = gen_sw(j);  // position j of the work buffers
                    double2 f0 = make_double2(0.0, 0.0), f1 = make_double2(0.0, 0.0);
                    if (p0) {
                        f0 = fac[sj];
                        f1 = fac[NC + sj];
                    }
                    // rows in chunks of MC: every load of a chunk (digit from the buffer, two key values from global memory) is issued
                    // before the chain starts.  Wave form, row by row: each row's key values were waited for with nothing else in flight;
                    // two rows per chunk +2 ... 4 % at N = 64 ... 512, four rows cost the third wave per SIMD its registers (N = 256: -8 %);
                    // the workgroup form (four waves per SIMD) measured equal or slower with chunks
                    constexpr int MC = WAVE ? 2 : 1;
                    for (int e0 = 0; e0 < nd; e0 += MC) {
                        double2 x[MC], b0[MC], b1[MC];
#pragma unroll
                        for (int u = 0; u < MC; u++) {
                            if (u == 0 || e0 + u < nd) {
                                x[u] = dig[(size_t)(e0 + u) * NC + sj];
                                const double2 *row = bkrow + (size_t)(p0 + e0 + u) * 2 * NC;
                                b0[u] = row[j];
                                b1[u] = row[NC + j];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < MC; u++) {
                            if (u == 0 || e0 + u < nd) {
                                const double ar = x[u].x, ai = x[u].y;
                                const double t0 = __builtin_fma(ai, b0[u].y, -f0.x);
                                f0.x = __builtin_fma(ar, b0[u].x, -t0);
                                const double u0 = __builtin_fma(ar, b0[u].y, f0.y);
                                f0.y = __builtin_fma(ai, b0[u].x, u0);
                                const double t1 = __builtin_fma(ai, b1[u].y, -f1.x);
                                f1.x = __builtin_fma(ar, b1[u].x, -t1);
                                const double u1 = __builtin_fma(ar, b1[u].y, f1.y);
                                f1.y = __builtin_fma(ai, b1[u].x, u1);
                            }
                        }
                    }
                    fac[sj] = f0;
                    fac[NC + sj] = f1;
                }
                gen_sync<WAVE>();  // the transforms read fac / the next group's first layer overwrites dig
            }
            // tLweFromFFTConvert (key rows carry the 2/N scale) + tLweAddTo: the last layer hands point j of polynomial q
            // (coefficients j and j + N/2) over as it is finished; every coefficient belongs to one work-item
            gen_fft<WAVE>(fac, 2, NC, NC, A.tw, tid, nt, true, [&](int q, int j, const GenC &v) {
                const U r0 = (U)Torus<T>::from_double(v.r), r1 = (U)Torus<T>::from_double(v.i);
                T *pa = acc + q * N;
                pa[j] = (T)(rotate ? (U)pa[j] + r0 : r0);
                pa[j + NC] = (T)(rotate ? (U)pa[j + NC] + r1 : r1);
            });
        }

        // ---- output
        if (A.flags & BR_EXTRACT) {  // tLweExtractLweSampleIndex, index 0 (tlwe_functions.cpp:351-363)
            T *out = A.lwe_out + (size_t)ct * (N + 1);
            for (int j = tid; j < N; j += nt) out[j] = (j == 0) ? acc[0] : (T)(0 - (U)acc[N - j]);
            if (tid == 0) out[N] = (T)((U)acc[N] + (U)A.out_b_add);
        } else if (A.flags & BR_CMUX_DATA) {
            const size_t item = A.cmux_period ? (size_t)(ct % A.cmux_period) : (size_t)ct;
            const T *d0 = A.cmux_d0 + item * A.cmux_stride;
            T *dst = A.acc_io + (size_t)ct * 2 * N;
            // dst may alias d0 / d1 of this very item: every element is read before it is written by the same work-item
            if (A.flags & BR_CMUX_TRIVIAL) {
                for (int j = tid; j < N; j += nt) {
                    const U v = (U)d0[j];
                    dst[j] = acc[j];
                    dst[N + j] = (T)((U)acc[N + j] + v);
                }
            } else {
                for (int j = tid; j < 2 * N; j += nt) dst[j] = (T)((U)acc[j] + (U)d0[j]);
            }
        } else {
            T *dst = A.acc_io + (size_t)ct * 2 * N;
            for (int j = tid; j < 2 * N; j += nt) dst[j] = acc[j];
        }
        gen_sync<WAVE>();  // acc is re-initialised for the workgroup's next ciphertext
    }
}

// ------------------------------------------------------------ exact (FFT-free) external product, any N
// As k_extprod_exact, with the per-thread result registers replaced by a loop over the thread's output coefficients.
template <typename T>
TFHE_GLOBAL void __launch_bounds__(256)
    kg_extprod_exact(T *__restrict__ acc_io, const T *__restrict__ gsw, Gadget gd, int batch, int logn) {
    using U = typename Torus<T>::U;
    constexpr int BITS = Torus<T>::BITS;
    const int N = 1 << logn;
    TFHE_DYN_LDS(smem);
    int32_t *dig = reinterpret_cast<int32_t *>(smem);                                   // [2l][N]
    U *gext = reinterpret_cast<U *>(smem + sizeof(int32_t) * 2 * (size_t)gd.l * N);     // [2N]
    U *res = gext + 2 * N;                                                              // [2][N]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= batch) return;
    T *acc = acc_io + (size_t)b * 2 * N;
    const U mask = ((U)1 << gd.Bgbit) - 1;
    const int32_t halfBg = 1 << (gd.Bgbit - 1);
    for (int e = tid; e < 2 * N; e += 256) {
        const U v = (U)acc[e] + (U)gd.offset;
        const int poly = e >> logn, c = e & (N - 1);
        for (int p = 0; p < gd.l; p++)
            dig[(poly * gd.l + p) * N + c] = (int32_t)((v >> (BITS - (p + 1) * gd.Bgbit)) & mask) - halfBg;
        res[e] = 0;
    }
    for (int p = 0; p < 2 * gd.l; p++) {
        for (int q = 0; q < 2; q++) {
            __syncthreads();
            const T *g = gsw + ((size_t)p * 2 + q) * N;
            for (int e = tid; e < N; e += 256) {
                gext[e] = (U)g[e];
                gext[N + e] = (U)0 - (U)g[e];
            }
            __syncthreads();
            const int32_t *dp = dig + p * N;
            for (int e = tid; e < N; e += 256) {
                U s = res[q * N + e];
                for (int j = 0; j < N; j++) s += (U)(T)dp[j] * gext[(e - j) & (2 * N - 1)];
                res[q * N + e] = s;
            }
        }
    }
    for (int e = tid; e < 2 * N; e += 256) acc[e] = (T)res[e];
}

}  // namespace tfhe
