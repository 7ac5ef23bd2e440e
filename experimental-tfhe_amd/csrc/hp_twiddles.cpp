// hp_twiddles.cpp -- twiddle tables of the Real96 high-precision anticyclic FFT
// (high-precision-anticyclic-fft/src/code.cpp, "HP" below): powomega[i] = (cos, sin)(2 pi i / n) and
// powombar[i] = (cos(i), sin(n - i)) as 2^64-scaled, correctly rounded integers in the reference's
// Real96 encoding (HP:25-40: a 128-bit two's-complement integer v standing for v / 2^64; the
// values 1 are stored as 2^64 - 1, HP:248,265).
//
// The reference computes them with NTL's 150-bit RR (HP:246-278), which this image does not have.
// Here: a 124-fractional-bit fixed-point Taylor evaluation on the first octant plus the exact
// symmetries of the index -- pure integer arithmetic, host only.  (The test oracle builds the same
// table a different way, with libquadmath, and the two must agree entry for entry.)
#include <stdint.h>

#include "../../include/tfhe_amd.h"

namespace {

typedef unsigned __int128 u128;
constexpr int F = 124;  // fractional bits; values stay below 4

// pi * 2^124 (3.243F6A8885A308D313198A2E0370734 4A40...)
const u128 PI_FX = ((u128)0x3243F6A8885A308DULL << 64) | (u128)0x313198A2E0370734ULL;

// (a * b) >> F for 0 <= a, b < 2^126: 256-bit product by 64-bit limbs
u128 mulfx(u128 a, u128 b) {
    const uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    const u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    // 256-bit result r3:r2:r1:r0
    const uint64_t r0 = (uint64_t)p00;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    const uint64_t r1 = (uint64_t)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + p11;  // r3:r2
    (void)r0;
    // shift the 256-bit value right by F = 124 = 64 + 60
    const u128 upper = hi;                     // bits 128..255
    const u128 lower = ((u128)r1 << 64) | r0;  // bits 0..127
    return (upper << (128 - F)) | (lower >> F);
}

// cos and sin of x in [0, pi/4], x and results in F-bit fixed point
void cos_sin_fx(u128 x, u128 *c, u128 *s) {
    const u128 one = (u128)1 << F;
    u128 term = one;  // x^k / k!
    u128 cpos = one, cneg = 0, spos = 0, sneg = 0;
    for (int k = 1; k <= 40; k++) {
        term = mulfx(term, x) / (unsigned)k;
        if (term == 0) break;
        switch (k & 3) {
            case 1: spos += term; break;
            case 2: cneg += term; break;
            case 3: sneg += term; break;
            default: cpos += term; break;
        }
    }
    *c = cpos - cneg;
    *s = spos - sneg;
}

// round-to-nearest of v / 2^(F-64): the 2^64-scaled magnitude, in [0, 2^64]
u128 scale64(u128 v) { return (v + ((u128)1 << (F - 65))) >> (F - 64); }

// (cos, sin)(2 pi i / n) * 2^64 in Real96 encoding, i in [0, n), n a power of two >= 8
void unit(int i, int n, u128 *re, u128 *im) {
    int logn = 0;
    while ((1 << logn) < n) logn++;
    // fold to the first octant by exact index arithmetic: angle = quad * pi/2 + 2 pi r / n
    const int q = n / 4;
    const int quad = i / q, r = i % q;
    const bool swap = r > q / 2;     // use the complementary angle pi/2 - t
    const int k = swap ? q - r : r;  // t = 2 pi k / n in [0, pi/4]
    const u128 one64 = (u128)1 << 64;
    u128 cm, sm;  // magnitudes of cos t, sin t at scale 2^64
    if (k == 0) {
        cm = one64;  // on an axis: exact
        sm = 0;
    } else {
        // x = pi * 2k / n = (PI_FX * k) >> (logn - 1), PI_FX split so that nothing overflows
        const int sh = logn - 1;
        const u128 hi = (u128)(uint64_t)(PI_FX >> 64) * (unsigned)k, lo = (u128)(uint64_t)PI_FX * (unsigned)k;
        const u128 x = (hi << (64 - sh)) + (lo >> sh);
        u128 c, s;
        cos_sin_fx(x, &c, &s);
        cm = scale64(c);
        sm = scale64(s);
    }
    if (swap) {
        const u128 t = cm;
        cm = sm;
        sm = t;
    }
    // rotate by quad quarter turns: (c, s) -> (-s, c) per turn
    u128 rr, ii;
    switch (quad & 3) {
        case 0: rr = cm; ii = sm; break;
        case 1: rr = (u128)0 - sm; ii = cm; break;
        case 2: rr = (u128)0 - cm; ii = (u128)0 - sm; break;
        default: rr = sm; ii = (u128)0 - cm; break;
    }
    // +1 is not representable with hi = 0: stored as 2^64 - 1 (HP:248 for cos at i = 0, HP:265 for sin
    // at i = n/4); -1 is stored exactly (lo = 0, hi = -1)
    if (rr == one64) rr = one64 - 1;
    if (ii == one64) ii = one64 - 1;
    *re = rr;
    *im = ii;
}

}  // namespace

extern "C" int tfhe_amd_hp_twiddles(int n, uint64_t *powomega, uint64_t *powombar) {
    if (n < 8 || (n & (n - 1)) || n > (1 << 20) || (!powomega && !powombar)) return TFHE_AMD_ERR_PARAM;
    for (int i = 0; i < n; i++) {
        u128 c, s, cb, sb;
        unit(i, n, &c, &s);
        unit((n - i) % n, n, &cb, &sb);  // HP:384-389: powombar[i] = (cos(i), sin(n - i))
        if (powomega) {
            powomega[4 * i + 0] = (uint64_t)c;
            powomega[4 * i + 1] = (uint64_t)(c >> 64);
            powomega[4 * i + 2] = (uint64_t)s;
            powomega[4 * i + 3] = (uint64_t)(s >> 64);
        }
        if (powombar) {
            powombar[4 * i + 0] = (uint64_t)c;
            powombar[4 * i + 1] = (uint64_t)(c >> 64);
            powombar[4 * i + 2] = (uint64_t)sb;
            powombar[4 * i + 3] = (uint64_t)(sb >> 64);
        }
    }
    return TFHE_AMD_OK;
}
