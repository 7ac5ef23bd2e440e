// tfhe_kernels_generic.h -- the same hot path for EVERY ring degree the reference's FFT plugin accepts.
//
// new_fft_table / new_ifft_table / FFT_Processor_Spqlios(N) take any power of two N >= 16
// (CB/spqlios/spqlios-fft-impl.cpp:157-160,400-403, fft_processor_spqlios.cpp:18-25); the reference only
// instantiates 1024 and 2048, and those two keep the wave-per-polynomial kernels of tfhe_kernels.h.  Every other N
// runs here: a TEAM of work-items (up to a 256-thread workgroup) owns one polynomial, the N/2 complex points live in
// a work buffer (LDS while it fits, a global scratch slice beyond), one radix-2 layer per workgroup barrier.
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

// ---- the two multiplication-free layers on one group of four consecutive points
// inverse: size-4 (spqlios-ifft-fma.s:194-213) then size-2 (:247-263)
TFHE_DEVICE void gen_ifft_tail4(double *re, double *im) {
    const double r0 = re[0], r1 = re[1], r2 = re[2], r3 = re[3];
    const double i0 = im[0], i1 = im[1], i2 = im[2], i3 = im[3];
    const double a0 = r0 + r2, a1 = r1 + r3, a2 = r0 - r2, a3 = i3 - i1;
    const double b0 = i0 + i2, b1 = i1 + i3, b2 = i0 - i2, b3 = r1 - r3;
    re[0] = a0 + a1;
    re[1] = a0 - a1;
    re[2] = a2 + a3;
    re[3] = a2 - a3;
    im[0] = b0 + b1;
    im[1] = b0 - b1;
    im[2] = b2 + b3;
    im[3] = b2 - b3;
}
// direct: size-2 (spqlios-fft-fma.s:79-95) then size-4 (:134-152)
TFHE_DEVICE void gen_fft_head4(double *re, double *im) {
    const double r0 = re[0] + re[1], r1 = re[0] - re[1], r2 = re[2] + re[3], r3 = re[2] - re[3];
    const double i0 = im[0] + im[1], i1 = im[0] - im[1], i2 = im[2] + im[3], i3 = im[2] - im[3];
    re[0] = r0 + r2;
    re[1] = r1 + i3;
    re[2] = r0 - r2;
    re[3] = r1 - i3;
    im[0] = i0 + i2;
    im[1] = i1 - r3;
    im[2] = i0 - i2;
    im[3] = i1 + r3;
}

// Coefficient -> Lagrange, in place, for `np` polynomials `pstride` doubles apart; each is re[0..NC) | im[0..NC) and holds
// a_j + i a_{j+NC} on entry (the fold of spqlios-ifft-fma.s:40-44 is the storage order itself).  A team of `tpp`
// work-items (this one is number `lt`) shares the work; EVERY work-item of the workgroup must call this (the barriers
// are workgroup barriers), `active` = false for those whose team has no polynomial.  tw: the kernels' table
// (tfhe_amd.hip build_tables: [0,NC) twist, half-size h at 2 NC - 2 h).  The caller has synchronised its writes; on
// return every result is visible to the whole workgroup.
TFHE_DEVICE void gen_ifft(double *buf, int np, long pstride, int NC, const double2 *__restrict__ tw, int lt, int tpp, bool active) {
    // twist by omega^j (spqlios-ifft-fma.s:63-78) fused into the first layer (h = NC/2: its butterfly owns both points)
    bool first = true;
    for (int h = NC >> 1; h >= 4; h >>= 1) {
        if (active) {
            const double2 *ts = tw + (2 * NC - 2 * h);
            for (int bf = lt; bf < (NC >> 1); bf += tpp) {
                const int off = bf & (h - 1), i1 = ((bf - off) << 1) + off, i2 = i1 + h;
                const double2 w = ts[off];
                double2 w1 = w, w2 = w;
                if (first) {
                    w1 = tw[i1];
                    w2 = tw[i2];
                }
                for (int p = 0; p < np; p++) {
                    double *re = buf + p * pstride, *im = re + NC;
                    double ar = re[i1], ai = im[i1], br = re[i2], bi = im[i2];
                    if (first) {
                        const double tr = __builtin_fma(-ai, w1.y, ar * w1.x), ti = __builtin_fma(ai, w1.x, ar * w1.y);
                        const double ur = __builtin_fma(-bi, w2.y, br * w2.x), ui = __builtin_fma(bi, w2.x, br * w2.y);
                        ar = tr;
                        ai = ti;
                        br = ur;
                        bi = ui;
                    }
                    dif_bfly(ar, ai, br, bi, w.x, w.y);
                    re[i1] = ar;
                    im[i1] = ai;
                    re[i2] = br;
                    im[i2] = bi;
                }
            }
        }
        first = false;
        __syncthreads();
    }
    if (active) {
        for (int g = lt; g < (NC >> 2); g += tpp)
            for (int p = 0; p < np; p++) gen_ifft_tail4(buf + p * pstride + 4 * g, buf + p * pstride + NC + 4 * g);
    }
    __syncthreads();
}

// Lagrange -> coefficient, in place (the caller has applied the 2/N scale); same calling rules as gen_ifft.
TFHE_DEVICE void gen_fft(double *buf, int np, long pstride, int NC, const double2 *__restrict__ tw, int lt, int tpp, bool active) {
    if (active) {
        for (int g = lt; g < (NC >> 2); g += tpp)
            for (int p = 0; p < np; p++) gen_fft_head4(buf + p * pstride + 4 * g, buf + p * pstride + NC + 4 * g);
    }
    __syncthreads();
    for (int h = 4; h <= (NC >> 1); h <<= 1) {
        const bool last = h == (NC >> 1);
        if (active) {
            const double2 *ts = tw + (2 * NC - 2 * h);
            for (int bf = lt; bf < (NC >> 1); bf += tpp) {
                const int off = bf & (h - 1), i1 = ((bf - off) << 1) + off, i2 = i1 + h;
                const double2 w = ts[off];
                // the reference's fft table is the conjugate of its ifft table except cos at the quarter turn (flip_sign_if)
                const double wc = flip_sign_if(w.x, off == (h >> 1));
                double2 w1 = w, w2 = w;
                if (last) {
                    w1 = tw[i1];
                    w2 = tw[i2];
                }
                for (int p = 0; p < np; p++) {
                    double *re = buf + p * pstride, *im = re + NC;
                    double ar = re[i1], ai = im[i1], br = re[i2], bi = im[i2];
                    dit_bfly(ar, ai, br, bi, wc, w.y);
                    if (last) {
                        // final twist by conj(omega^j), four rounded products (spqlios-fft-fma.s:255-274)
                        const double arc = ar * w1.x, ars = ar * w1.y, aic = ai * w1.x, ais = ai * w1.y;
                        const double brc = br * w2.x, brs = br * w2.y, bic = bi * w2.x, bis = bi * w2.y;
                        ar = arc + ais;
                        ai = aic - ars;
                        br = brc + bis;
                        bi = bic - brs;
                    }
                    re[i1] = ar;
                    im[i1] = ai;
                    re[i2] = br;
                    im[i2] = bi;
                }
            }
        }
        __syncthreads();
    }
}

// how a 256-thread workgroup is cut into teams: one team per polynomial, NC/2 butterflies per layer
TFHE_HOST_DEVICE int gen_team_size(int NC) { return (NC >> 1) < 256 ? (NC >> 1) : 256; }
constexpr int GEN_BLOCK = 256;

// ------------------------------------------------------------ FFT plugin boundary, any N
// execute_reverse_int / _torus32 / _torus64 and the bare `ifft` (TIN = double): coefficients -> LagrangeHalfC.
// PACK: write the key layout of kg_blind_rotate instead ([row][NC] complex in reference order, scaled by 2/N).
// work: the transform buffers -- null: dynamic LDS (teams x N doubles), else global scratch, one slice per workgroup.
template <typename TIN, bool PACK>
TFHE_GLOBAL void __launch_bounds__(GEN_BLOCK)
    kg_ifft_batch(double *__restrict__ out, const TIN *__restrict__ in, const double2 *__restrict__ tw, int batch, int logn,
                  double *__restrict__ work) {
    const int N = 1 << logn, NC = N >> 1;
    const int tpp = gen_team_size(NC), teams = GEN_BLOCK / tpp;
    const int team = (int)threadIdx.x / tpp, lt = (int)threadIdx.x - team * tpp;
    TFHE_DYN_LDS(smem);
    double *buf = (work ? work + (size_t)blockIdx.x * teams * N : reinterpret_cast<double *>(smem)) + (size_t)team * N;
    for (int b0 = (int)blockIdx.x * teams; b0 < batch; b0 += (int)gridDim.x * teams) {  // workgroup-uniform
        const int b = b0 + team;
        const bool active = b < batch;
        if (active) {
            const TIN *p = in + (size_t)b * N;
            for (int j = lt; j < NC; j += tpp) {
                buf[j] = (double)p[j];
                buf[NC + j] = (double)p[j + NC];
            }
        }
        __syncthreads();
        gen_ifft(buf, 1, 0, NC, tw, lt, tpp, active);
        if (active) {
            if (PACK) {
                double2 *o = reinterpret_cast<double2 *>(out) + (size_t)b * NC;
                const double scale = 2.0 / (double)N;  // exact: a power of two
                for (int j = lt; j < NC; j += tpp) o[j] = make_double2(buf[j] * scale, buf[NC + j] * scale);
            } else {
                double *o = out + (size_t)b * N;
                for (int j = lt; j < N; j += tpp) o[j] = buf[j];
            }
        }
        __syncthreads();  // the buffer is refilled by the next polynomial
    }
}

// execute_direct_torus32 / _torus64 (scale 2/N, transform, round as fft_processor_spqlios.cpp:102,131-142) and the bare
// `fft` (TOUT = double: no scale, no rounding)
template <typename TOUT>
TFHE_GLOBAL void __launch_bounds__(GEN_BLOCK)
    kg_fft_batch(TOUT *__restrict__ out, const double *__restrict__ in, const double2 *__restrict__ tw, int batch, int logn,
                 double *__restrict__ work) {
    const int N = 1 << logn, NC = N >> 1;
    const int tpp = gen_team_size(NC), teams = GEN_BLOCK / tpp;
    const int team = (int)threadIdx.x / tpp, lt = (int)threadIdx.x - team * tpp;
    constexpr bool RAW = std::is_same<TOUT, double>::value;
    TFHE_DYN_LDS(smem);
    double *buf = (work ? work + (size_t)blockIdx.x * teams * N : reinterpret_cast<double *>(smem)) + (size_t)team * N;
    const double scale = 2.0 / (double)N;  // fft_processor_spqlios.cpp:78
    for (int b0 = (int)blockIdx.x * teams; b0 < batch; b0 += (int)gridDim.x * teams) {
        const int b = b0 + team;
        const bool active = b < batch;
        if (active) {
            const double *p = in + (size_t)b * N;
            for (int j = lt; j < N; j += tpp) buf[j] = RAW ? p[j] : p[j] * scale;
        }
        __syncthreads();
        gen_fft(buf, 1, 0, NC, tw, lt, tpp, active);
        if (active) {
            TOUT *o = out + (size_t)b * N;
            if constexpr (RAW) {
                for (int j = lt; j < N; j += tpp) o[j] = buf[j];
            } else {
                for (int j = lt; j < N; j += tpp) o[j] = Torus<TOUT>::from_double(buf[j]);
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
// same meaning.  Work areas: acc [2][N] torus, dig [N] doubles (one gadget digit at a time: extract, transform,
// multiply-accumulate, discard), fac [2][N] doubles (the Fourier accumulator of tLweFFTClear / tLweFFTAddMulRTo).
// Each of the three is in dynamic LDS when its offset is >= 0, else in the workgroup's slice of `work`.
struct GenBrPlace {
    long long acc_lds, dig_lds, fac_lds;  // byte offsets into dynamic LDS, or -1: global
    long long work_stride;                // bytes of global scratch per workgroup
    unsigned char *work;
    int logn;
};

template <typename T>
TFHE_GLOBAL void __launch_bounds__(GEN_BLOCK) kg_blind_rotate(BlindRotateArgs<T> A, GenBrPlace G) {
    using U = typename Torus<T>::U;
    constexpr int BITS = Torus<T>::BITS;
    const int logn = G.logn, N = 1 << logn, NC = N >> 1;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    TFHE_DYN_LDS(smem);
    unsigned char *wsl = G.work ? G.work + (size_t)blockIdx.x * (size_t)G.work_stride : nullptr;
    size_t woff = 0;
    auto place = [&](long long lds_off, size_t bytes) -> unsigned char * {
        if (lds_off >= 0) return smem + lds_off;
        unsigned char *p = wsl + woff;
        woff += bytes;
        return p;
    };
    T *acc = reinterpret_cast<T *>(place(G.acc_lds, sizeof(T) * 2 * (size_t)N));
    double *dig = reinterpret_cast<double *>(place(G.dig_lds, sizeof(double) * (size_t)N));
    double *fac = reinterpret_cast<double *>(place(G.fac_lds, sizeof(double) * 2 * (size_t)N));
    const U offset = (U)A.gd.offset;
    const int Bgbit = A.gd.Bgbit, l = A.gd.l;
    const U mask = ((U)1 << Bgbit) - 1;
    const int32_t halfBg = 1 << (Bgbit - 1);
    const bool rotate = !(A.flags & BR_NO_ROTATE);

    for (int ct = (int)blockIdx.x; ct < A.batch; ct += (int)gridDim.x) {  // workgroup-uniform
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
        __syncthreads();
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
            for (int j = tid; j < 2 * N; j += nt) fac[j] = 0.0;  // tLweFFTClear (tgsw_functions.cpp:438)
            for (int q = 0; q < 2; q++) {
                const T *pa = acc + q * N;
                for (int d = 0; d < l; d++) {
                    const int decal = BITS - (d + 1) * Bgbit;
                    // digit d of polynomial q of (X^a - 1) * acc (numeric_functions.cpp:304-323), or of acc itself;
                    // decomposition tgsw_functions.cpp:224-337 / poc:492-515 (offset from the host)
                    for (int j = tid; j < N; j += nt) {
                        U v = (U)pa[j];
                        if (rotate) {
                            const int idx = (j - a) & (2 * N - 1);
                            const U src = (U)pa[idx & (N - 1)];
                            v = ((idx & N) ? (U)(0 - src) : src) - v;
                        }
                        dig[j] = (double)((int32_t)(((U)(v + offset) >> decal) & mask) - halfBg);
                    }
                    __syncthreads();
                    gen_ifft(dig, 1, 0, NC, A.tw, tid, nt, true);
                    // tLweFFTAddMulRTo (tlwe_functions.cpp:318-325): row p = q*l + d, both output polynomials;
                    // the chain of lagrangehalfc_impl_fma.s:96-107 on an accumulator that started as +0
                    const double2 *row = bkrow + (size_t)(q * l + d) * 2 * NC;
                    for (int j = tid; j < NC; j += nt) {
                        const double ar = dig[j], ai = dig[NC + j];
                        for (int q2 = 0; q2 < 2; q2++) {
                            const double2 b = row[(size_t)q2 * NC + j];
                            double *fr = fac + q2 * N + j, *fi = fr + NC;
                            const double tneg = __builtin_fma(ai, b.y, -*fr);
                            *fr = __builtin_fma(ar, b.x, -tneg);
                            const double u = __builtin_fma(ar, b.y, *fi);
                            *fi = __builtin_fma(ai, b.x, u);
                        }
                    }
                    __syncthreads();  // the next digit's fill overwrites dig (not always from the work-item that read it: N < 2 x block)
                }
            }
            __syncthreads();
            // tLweFromFFTConvert (key rows carry the 2/N scale) + tLweAddTo
            gen_fft(fac, 2, N, NC, A.tw, tid, nt, true);
            for (int j = tid; j < 2 * N; j += nt) {
                const U r = (U)Torus<T>::from_double(fac[j]);
                acc[j] = (T)(rotate ? (U)acc[j] + r : r);
            }
            __syncthreads();
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
        __syncthreads();  // acc is re-initialised for the workgroup's next ciphertext
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
