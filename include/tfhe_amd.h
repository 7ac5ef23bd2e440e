/*
 * tfhe_amd.h -- C ABI of the MI355X-native TFHE bootstrapping engine.
 *
 * This is the drop-in boundary for the hot path of tfhe/experimental-tfhe: every entry
 * point below names the reference interface it replaces (CB/ = the reference's
 * circuit-bootstrapping/src/).  The reference processes ONE sample per call on the host;
 * here every operation takes a BATCH of independent samples resident in GPU memory and
 * runs asynchronously on the context's HIP stream.  Batch-1 shims with the reference's
 * exact names and struct types are in include/tfhe_amd_compat.hpp.
 *
 * Conventions
 *   - plain C: pointers, sizes, int status (0 = TFHE_AMD_OK); no HIP or torch types.
 *   - `*_d` pointers are device pointers (hipMalloc, tfhe_amd_malloc or any framework's
 *     allocator on the context's device); everything else is host memory.
 *   - Torus32 = int32_t, Torus64 = int64_t (CB/poc_types.h:13-14), wrap-around arithmetic.
 *   - LWE sample of dimension n: n+1 torus values, a[0..n-1] then b (CB/poc_types.h:137-158).
 *   - TLWE sample (k=1): 2N torus values, polynomial a then polynomial b (poc_types.h:164-197).
 *   - LagrangeHalfC polynomial: N doubles, re[0..N/2) then im[0..N/2) in the order
 *     produced by the reference's ifft (CB/spqlios/lagrangehalfc_impl_fma.s:91-93).
 *   - TGSW sample in Lagrange form: [(k+1)l][k+1][N] doubles = `allsamples` rows
 *     p = bloc*l + i, each with polynomials a[0..k] (poc_types.h:239-251).
 *   - a context is bound to one device and one stream and is not thread-safe; use one
 *     context per host thread / per GPU.  Distinct contexts are independent.
 *   - the library never falls back to the CPU: without a usable gfx950 device
 *     tfhe_amd_ctx_create fails with TFHE_AMD_ERR_DEVICE.
 */
#ifndef TFHE_AMD_H
#define TFHE_AMD_H

#include <stddef.h>
#include <stdint.h>

/* The reference's PoC header defines `k` as a macro (`#define k 1`, CB/poc_types.h:10); this header names
 * a struct field k.  Keep both usable in one translation unit, whichever is included first. */
#pragma push_macro("k")
#undef k

#ifdef __cplusplus
extern "C" {
#endif

enum {
    TFHE_AMD_OK = 0,
    TFHE_AMD_ERR_PARAM = 1,   /* unsupported or inconsistent parameters / null pointer */
    TFHE_AMD_ERR_DEVICE = 2,  /* HIP error (see tfhe_amd_last_error)                  */
    TFHE_AMD_ERR_STATE = 3,   /* a required key has not been loaded                   */
    TFHE_AMD_ERR_ALLOC = 4
};

/* Parameters of one bootstrapping level.  Reference: LweParams/TLweParams/TGswParams
 * (CB/lwe_functions.cpp:17, CB/tgsw_functions.cpp:15-38) and the PoC's Globals
 * (CB/poc_types.h:267-283).  Supported: N = every power of two from 16 to 2^20 -- what the reference's plugin accepts
 * (new_fft_table / FFT_Processor_Spqlios(N): a power of two >= 16, CB/spqlios/spqlios-fft-impl.cpp:157-160,
 * fft_processor_spqlios.cpp:18-25); 1024 and 2048, the two the reference instantiates, run the wave-per-polynomial
 * kernels, every other degree the team-per-polynomial kernels of csrc/tfhe_kernels_generic.h (same bits, a little under
 * half the tuned kernels' rate: profiles/r06_generic_n.txt).  k = 1, l in [1,8], l*Bgbit <= torus_bits - 1,
 * torus_bits in {32, 64}.  (Real96 transforms, tfhe_amd_hp_*: N in {1024, 2048} only.) */
typedef struct tfhe_amd_params {
    int32_t torus_bits; /* 32: Torus32 accumulator (gate bootstrap); 64: Torus64 (circuit bootstrap lvl2) */
    int32_t n;          /* LWE dimension = number of CMux steps of a blind rotation */
    int32_t N;          /* ring degree */
    int32_t k;          /* must be 1 (the reference hard-codes it: CB/poc_types.h:10) */
    int32_t l;          /* gadget length */
    int32_t Bgbit;      /* log2 of the gadget base */
    int32_t ks_t;       /* key-switch length t (0: no key switch on this context) */
    int32_t ks_basebit; /* key-switch base bits */
    int32_t ks_n_out;   /* output dimension of the key switch (gate bootstrap: = n) */
} tfhe_amd_params;

typedef struct tfhe_amd_ctx tfhe_amd_ctx;
typedef struct tfhe_amd_gsw tfhe_amd_gsw; /* device-resident TGSW samples, kernel layout */

/* ---- context ------------------------------------------------------------------------ */
int tfhe_amd_ctx_create(const tfhe_amd_params *params, int device, tfhe_amd_ctx **out);
void tfhe_amd_ctx_destroy(tfhe_amd_ctx *ctx);
const char *tfhe_amd_last_error(const tfhe_amd_ctx *ctx);
const char *tfhe_amd_version(void);
/* one-line description of a device (name, gfx target, PCI bus id, CUs, clock, memory, LDS limits, runtime version) for logs */
int tfhe_amd_device_info(int device, char *buf, size_t len);
/* number of HIP devices this process sees */
int tfhe_amd_device_count(int *count);
/* "domain:bus:device.function" of the GPU behind an ordinal (hipDeviceGetPCIBusId): ordinals are per process -- a launcher
 * may show every rank ONE device, all of them "device 0" -- the bus id is what tells two GPUs apart; len >= 13 */
int tfhe_amd_device_pci_bus_id(int device, char *buf, size_t len);
/* shader clock (GHz) read by 32 one-wave probes on a second stream: s_memtime / s_memrealtime around a sleep of duration_us;
 * median, and optionally min / max over the probes.  Blocks for about duration_us.  Diagnostic only.  The probes run BESIDE
 * the work queued on the context's stream only where a CU has a wave slot and registers to spare: the transforms and key
 * switches leave them, the blind-rotation kernels do not (two waves per SIMD hold all 512 registers) -- the probes then run
 * after them and read the idle clock (2.4 GHz).  The clock UNDER a blind rotation comes from the probe build of the library
 * (libtfhe_amd_probe.so, tools/wave_probe.py), which stamps inside the kernel. */
int tfhe_amd_clock_probe(tfhe_amd_ctx *ctx, int duration_us, double *ghz_median, double *ghz_min, double *ghz_max);
/* use an existing hipStream_t (passed as void*); NULL = the context's own stream.  The switch does NOT wait and inserts no
 * dependency: work already queued on the previous stream stays ordered there, work queued after the call is ordered on the new
 * stream only.  A caller that moves a context between streams and needs later work to follow earlier work -- the context's
 * scratch (used by tfhe_amd_bootstrap, _streamed, _lut_eval, the circuit bootstrap's workspaces) is shared by both -- orders
 * them itself: tfhe_amd_event_record before the switch + tfhe_amd_stream_wait_event after it (device side, no host wait), or
 * tfhe_amd_sync before it.  (The pool's copy / compute pipeline relies on the switch being free: it ties its three streams
 * with events of its own.) */
int tfhe_amd_set_stream(tfhe_amd_ctx *ctx, void *hip_stream);
int tfhe_amd_sync(tfhe_amd_ctx *ctx);
/* Scheduling options.  They select among kernels that compute the SAME results bit for bit.  (The
 * blind-rotation schedule variants, the streamed key switch and the wider transform workgroups of round 1
 * were measured on MI355X and removed: profiles/r02_variants.txt, profiles/r02_config4_fft.jsonl.)
 *   TFHE_AMD_OPT_KS_GATHER   != 0: per-sample gather key switch (the kernel for shapes the matrix-core one
 *        does not cover: basebit > 3 or t * basebit > 32) even where the matrix-core kernel applies
 *   TFHE_AMD_OPT_STREAMED_GRAPH  != 0: tfhe_amd_bootstrap_streamed captures its n+3 launches into a hipGraph
 *        (after one plain call per schedule) and replays it while (x_d, out_d, mu, batch) repeat
 *   TFHE_AMD_OPT_BR_SPLIT   which blind-rotation kernel serves a Torus32 / N = 1024 / l = 2 batch: value = the
 *        largest batch that runs on the latency-shaped kernel (one ciphertext per 4-wave workgroup, the 2l inverse
 *        transforms of a CMux on the four SIMDs of a CU at once; what a one-sample call of tfhe_bootstrap_FFT,
 *        lwe_functions.cpp:434-446, gets), larger batches run one wave per ciphertext (in 4-wave workgroups -- one
 *        wave per SIMD -- up to 1024 samples, 8-wave workgroups above).  < 0 (default): the measured crossover
 *        (512); 0: never */
enum { TFHE_AMD_OPT_KS_GATHER = 2, TFHE_AMD_OPT_STREAMED_GRAPH = 4, TFHE_AMD_OPT_BR_SPLIT = 5 };
int tfhe_amd_set_option(tfhe_amd_ctx *ctx, int option, int value);
/* HIP events on the context's stream, for timing kernels without a HIP binding in the host
 * language: create, record (asynchronous), elapsed milliseconds between two recorded events
 * (waits for `stop`), destroy */
int tfhe_amd_event_create(tfhe_amd_ctx *ctx, void **event);
int tfhe_amd_event_record(tfhe_amd_ctx *ctx, void *event);
int tfhe_amd_event_elapsed_ms(tfhe_amd_ctx *ctx, void *start, void *stop, float *ms);
/* host waits for a recorded event / the context's CURRENT stream waits for it (work queued afterwards starts only once the
 * event has happened): with tfhe_amd_set_stream and tfhe_amd_stream_create, a copy / compute pipeline inside one context */
int tfhe_amd_event_sync(tfhe_amd_ctx *ctx, void *event);
int tfhe_amd_stream_wait_event(tfhe_amd_ctx *ctx, void *event);
int tfhe_amd_event_destroy(tfhe_amd_ctx *ctx, void *event);
/* twiddle tables as the reference lays them out (new_fft_table / new_ifft_table,
 * CB/spqlios/spqlios-fft-impl.cpp:158-193,400-437): 2N-8 doubles each; for SHA pinning. */
int tfhe_amd_get_tables(const tfhe_amd_ctx *ctx, double *fft_trig, double *ifft_trig);
/* the same two tables without a context or a device (host libm only), N a power of two in [16, 2^20]; either pointer may be NULL */
int tfhe_amd_build_tables(int N, double *fft_trig, double *ifft_trig);

/* device memory helpers so a host language needs no HIP binding of its own */
int tfhe_amd_malloc(tfhe_amd_ctx *ctx, void **dptr, size_t bytes);
int tfhe_amd_free(tfhe_amd_ctx *ctx, void *dptr);
/* page-locked HOST memory (staging buffers of callers that hand over host data: copies from / to it run at the
 * link's rate, and a gather into it replaces one copy per sample by one copy per batch) */
int tfhe_amd_host_alloc(tfhe_amd_ctx *ctx, void **hptr, size_t bytes);
int tfhe_amd_host_free(tfhe_amd_ctx *ctx, void *hptr);
int tfhe_amd_memcpy_h2d(tfhe_amd_ctx *ctx, void *dst_d, const void *src, size_t bytes);
int tfhe_amd_memcpy_d2h(tfhe_amd_ctx *ctx, void *dst, const void *src_d, size_t bytes);
/* the same copies without the wait: ordered on the context's current stream like a launch; the host buffer must stay valid
 * (and should be page-locked: tfhe_amd_host_alloc) until that stream has been synchronised */
int tfhe_amd_memcpy_h2d_async(tfhe_amd_ctx *ctx, void *dst_d, const void *src, size_t bytes);
int tfhe_amd_memcpy_d2h_async(tfhe_amd_ctx *ctx, void *dst, const void *src_d, size_t bytes);
/* more streams on the context's device, for tfhe_amd_set_stream: independent batches issued alternately on two streams
 * overlap one batch's copies with the other's kernels (what the pool does inside a member).  Entry points that use the
 * context's own scratch (tfhe_amd_bootstrap, _bootstrap_streamed, _lut_eval) must not be in flight on two streams at once;
 * tfhe_amd_bootstrap_woks + tfhe_amd_keyswitch on caller-owned buffers may. */
int tfhe_amd_stream_create(tfhe_amd_ctx *ctx, void **stream);
int tfhe_amd_stream_sync(tfhe_amd_ctx *ctx, void *stream);
int tfhe_amd_stream_destroy(tfhe_amd_ctx *ctx, void *stream);

/* ---- keys --------------------------------------------------------------------------- */
/* TGSW samples already in Lagrange form, host layout [count][(k+1)l][k+1][N] doubles
 * (what init_LweBootstrappingKeyFFT / tGswToFFTConvert produce: CB/lwe_functions.cpp:287-316,
 * CB/tgsw_functions.cpp:389-394; PoC: poc_CircuitBootstrapping.cpp:395-402). */
int tfhe_amd_gsw_from_fft(tfhe_amd_ctx *ctx, const double *gsw_fft, int count, tfhe_amd_gsw **out);
/* TGSW samples in coefficient form, host layout [count][(k+1)l][k+1][N] torus values
 * (int32_t or int64_t per params.torus_bits); converted on the GPU = tGswToFFTConvert. */
int tfhe_amd_gsw_from_torus(tfhe_amd_ctx *ctx, const void *gsw_torus, int count, tfhe_amd_gsw **out);
/* the same from DEVICE memory (asynchronous on the context's stream): turns the TGSW32 outputs of
 * tfhe_amd_circuit_bootstrap ([B][2][l1][2][N1], i.e. B samples in this layout) into selectors for
 * tfhe_amd_cmux / tfhe_amd_lut_eval without leaving the GPU */
int tfhe_amd_gsw_from_torus_d(tfhe_amd_ctx *ctx, const void *gsw_torus_d, int count, tfhe_amd_gsw **out);
void tfhe_amd_gsw_free(tfhe_amd_gsw *gsw);
/* read back sample `index` in the reference's Lagrange layout [(k+1)l][k+1][N] (unscaled) */
int tfhe_amd_gsw_export_fft(tfhe_amd_ctx *ctx, const tfhe_amd_gsw *gsw, int index, double *out);
/* bootstrapping key = TGSW array of exactly n samples (LweBootstrappingKeyFFT::bkFFT,
 * CB/lwe_functions.cpp:272-281); the context keeps a reference, the caller keeps ownership */
int tfhe_amd_set_bootstrap_key(tfhe_amd_ctx *ctx, const tfhe_amd_gsw *bk);
/* key-switch key, host layout [N][ks_t][1<<ks_basebit][ks_n_out+1] int32
 * (LweKeySwitchKey::ks, CB/lwe_functions.cpp:96-110; PoC preKS poc:375) */
int tfhe_amd_load_keyswitch_key(tfhe_amd_ctx *ctx, const int32_t *ks);
/* the same key from DEVICE memory of the context's device (or any pointer hipMemcpyDefault resolves), e.g. the bytes a
 * broadcast delivered; tfhe_amd_keyswitch_key_export writes the resident key in that same layout */
int tfhe_amd_load_keyswitch_key_d(tfhe_amd_ctx *ctx, const int32_t *ks_any);
int tfhe_amd_keyswitch_key_bytes(const tfhe_amd_ctx *ctx, size_t *bytes);
int tfhe_amd_keyswitch_key_export(tfhe_amd_ctx *ctx, void *dst_any);
/* TGSW samples as the bytes of the KERNEL layout ([count][(k+1)l][k+1][N/128][64] complex doubles, scaled by 2/N): what
 * one device hands to another -- SURVEY 8(e): "replicate bkFFT + KS key on every GPU at setup (one hipMemcpy per device or
 * one RCCL broadcast)" -- so that the receiver neither regenerates nor re-converts the key.  src / dst: device memory of
 * the context's device, or host memory. */
int tfhe_amd_gsw_packed_bytes(const tfhe_amd_ctx *ctx, int count, size_t *bytes);
int tfhe_amd_gsw_export_packed(tfhe_amd_ctx *ctx, const tfhe_amd_gsw *gsw, void *dst_any);
int tfhe_amd_gsw_from_packed(tfhe_amd_ctx *ctx, const void *src_any, int count, tfhe_amd_gsw **out);

/* ---- L1: the FFT plugin (class FFT_Processor_Spqlios, CB/spqlios/lagrangehalfc_impl.h:8-36) */
/* execute_reverse_int / execute_reverse_torus32: [batch][N] int32 -> [batch][N] doubles */
int tfhe_amd_ifft_int32(tfhe_amd_ctx *ctx, double *out_d, const int32_t *in_d, int batch);
/* execute_reverse_torus64 */
int tfhe_amd_ifft_torus64(tfhe_amd_ctx *ctx, double *out_d, const int64_t *in_d, int batch);
/* execute_direct_torus32: scale by 2/N, fft, int32_t(int64_t(x)).
 * ALIGNMENT: the Lagrange-domain INPUT of the two direct transforms (and of tfhe_amd_fft_f64) is read with 16-byte
 * loads: in_d must be 16-byte aligned (any hipMalloc / tfhe_amd_malloc pointer, and every polynomial of a batch
 * behind it, is); a misaligned pointer is refused with TFHE_AMD_ERR_PARAM. */
int tfhe_amd_fft_torus32(tfhe_amd_ctx *ctx, int32_t *out_d, const double *in_d, int batch);
/* execute_direct_torus64 */
int tfhe_amd_fft_torus64(tfhe_amd_ctx *ctx, int64_t *out_d, const double *in_d, int batch);
/* the bare core transforms of the reference's C layer, `ifft(tables, data)` / `fft(tables, data)`
 * (CB/spqlios/spqlios-fft.h:52-53; spqlios-ifft-fma.s, spqlios-fft-fma.s): [batch][N] doubles -> [batch][N] doubles,
 * no integer conversion, no 2/N scale; out of place (out_d != in_d) */
int tfhe_amd_ifft_f64(tfhe_amd_ctx *ctx, double *out_d, const double *in_d, int batch);
int tfhe_amd_fft_f64(tfhe_amd_ctx *ctx, double *out_d, const double *in_d, int batch);
/* LagrangeHalfCPolynomialAddMulASM: res[i] += a[i]*b[i] (b_shared != 0: b[0] for every i) */
int tfhe_amd_lagrange_addmul(tfhe_amd_ctx *ctx, double *res_d, const double *a_d, const double *b_d,
                             int batch, int b_shared);

/* ---- Real96 high-precision anticyclic transforms (high-precision-anticyclic-fft/src/code.cpp = HP) --
 * 128-bit fixed point: a Real96 is a two's-complement integer v standing for v / 2^64, stored here as
 * two little-endian uint64 (lo, hi); a complex value is (re, im) = 4 uint64 (HP:17-40, 236).
 * Any context of ring degree N serves (the torus width is not used); the reference runs N = 2048. */
/* twiddle tables, host: powomega[i] = (cos, sin)(2 pi i / n), powombar[i] = (cos(i), sin(n - i)), 2^64-scaled
 * and rounded to nearest, +1 stored as 2^64 - 1 (accurate_cos/accurate_sin HP:246-278, precomp_* HP:378-389).
 * n = 2N; each table [n][4] uint64; either pointer may be NULL */
int tfhe_amd_hp_twiddles(int n, uint64_t *powomega, uint64_t *powombar);
/* iFFT (HP:391-444): in_d [batch][N] Torus64 -> out_d [batch][N/2][4] uint64 */
int tfhe_amd_hp_ifft(tfhe_amd_ctx *ctx, uint64_t *out_d, const int64_t *in_d, int batch);
/* FFT (HP:446-512): in_d [batch][N/2][4] uint64 (not modified) -> out_d [batch][N] Torus64, divided by N/2 */
int tfhe_amd_hp_fft(tfhe_amd_ctx *ctx, int64_t *out_d, const uint64_t *in_d, int batch);

/* ---- L2: ring / TGSW ---------------------------------------------------------------- */
/* tGswFFTExternMulToTLwe (CB/tgsw_functions.cpp:424-449; PoC inline poc:609-620):
 * acc[i] <- gsw[index] (x) acc[i], acc_d: [batch][2][N] torus */
int tfhe_amd_extern_mul(tfhe_amd_ctx *ctx, void *acc_d, const tfhe_amd_gsw *gsw, int index, int batch);
/* tfhe_MuxRotate_FFT (CB/lwe_functions.cpp:328-333), in place:
 * acc[i] <- gsw[index] (x) ((X^barai[i] - 1) acc[i]) + acc[i]; barai in [0,2N), 0 = no-op */
int tfhe_amd_mux_rotate(tfhe_amd_ctx *ctx, void *acc_d, const tfhe_amd_gsw *gsw, int index,
                        const int32_t *barai_d, int batch);

/* The same external product with exact polynomial products in Z_{2^W}[X]/(X^N+1) -- the reference's
 * FFT-free backend (`#ifndef USE_FFT`, poc:285-316, built on torus{32,64}PolynomialMultAddKaratsuba_lvl{1,2},
 * CB/poc_karatsuba.cpp:80-95,188-203).  gsw_torus_d: ONE TGSW sample in coefficient form, device memory,
 * [(k+1)l][k+1][N] torus values.  A verification backend: tfhe_amd_extern_mul approximates this result
 * to within the fp64 rounding of the transforms. */
int tfhe_amd_extern_mul_exact(tfhe_amd_ctx *ctx, void *acc_d, const void *gsw_torus_d, int batch);

/* CMux on data, the consumer of a circuit bootstrap (stub `CMux` at poc:877-879; in library terms
 * tGswFFTExternMulToTLwe applied to d1 - d0, plus d0): out[i] = gsw[sel[i]] (x) (d1[i] - d0[i]) + d0[i].
 * sel_d: [batch] TGSW indices into `gsw` (NULL: index 0 for every sample); d0_d, d1_d, out_d:
 * [batch][2][N] torus; out_d may alias d1_d.  One level of a vertical-packing / LUT tree. */
int tfhe_amd_cmux(tfhe_amd_ctx *ctx, void *out_d, const tfhe_amd_gsw *gsw, const int32_t *sel_d, const void *d0_d,
                  const void *d1_d, int batch);

/* LUT evaluation by vertical packing over TGSW-encrypted bits -- what BASELINE config 3 names after
 * the circuit bootstrap.  The reference stops at the CMux stub (poc:877-879); the algorithm is the
 * published one (Chillotti-Gama-Georgieva-Izabachene, "Faster packed homomorphic operations and
 * efficient circuit bootstrapping for TFHE", vertical packing), built from the two reference
 * operations above: for f: {0,1}^d -> Torus with table polynomials
 *     lut[p] = sum_{i<N} f(p*N + i) X^i,   p < npoly = max(1, 2^(d - log2 N)),
 * and x = sum_i x_i 2^i given as TGSW samples (item b's bit i = sample b*d + i of `bits`):
 *   1. d - log2 N tree levels of tfhe_amd_cmux: level j keeps cur[2p + x_{log2 N + j}] (level 0 reads
 *      the shared plaintext table as noiseless trivial samples, tLweNoiselessTrivial);
 *   2. min(d, log2 N) steps of tfhe_MuxRotate_FFT with the constant rotations 2N - 2^i (X^{-2^i}) and
 *      the item's own bit i as the selector;
 *   3. tLweExtractLweSampleIndex at index 0.
 * lut_d: [npoly][N] Torus32 (plaintext, shared by the batch); lwe_out_d: [batch][N+1].  Torus32 contexts
 * (the circuit bootstrap emits TGSW32), 1 <= d <= log2 N + 20. */
int tfhe_amd_lut_eval(tfhe_amd_ctx *ctx, void *lwe_out_d, const tfhe_amd_gsw *bits, int d, const void *lut_d, int batch);

/* ---- L3: bootstrapping -------------------------------------------------------------- */
/* tfhe_blindRotate_FFT (CB/lwe_functions.cpp:337-361): acc_d [batch][2][N] in place,
 * bara_d [batch][n] rotations in [0,2N) */
int tfhe_amd_blind_rotate(tfhe_amd_ctx *ctx, void *acc_d, const int32_t *bara_d, int batch);
/* tfhe_blindRotateAndExtract_FFT (CB/lwe_functions.cpp:366-395).  v_d: test vector(s), [N]
 * when v_per_sample == 0 else [batch][N]; rot_d: [batch][n+1] = bara[0..n-1] then barb;
 * lwe_out_d: [batch][N+1] */
int tfhe_amd_blind_rotate_extract(tfhe_amd_ctx *ctx, void *lwe_out_d, const void *v_d, int v_per_sample,
                                  const int32_t *rot_d, int batch);
/* tfhe_bootstrap_woKS_FFT (CB/lwe_functions.cpp:399-430), Torus32 contexts: x_d [batch][n+1]
 * LWE samples, lwe_out_d [batch][N+1] */
int tfhe_amd_bootstrap_woks(tfhe_amd_ctx *ctx, int32_t *lwe_out_d, int32_t mu, const int32_t *x_d, int batch);
/* lweKeySwitch (CB/lwe_functions.cpp:163-171) / preKeySwitch (poc:437-465):
 * in_d [batch][N+1] -> out_d [batch][ks_n_out+1] */
int tfhe_amd_keyswitch(tfhe_amd_ctx *ctx, int32_t *out_d, const int32_t *in_d, int batch);
/* tfhe_bootstrap_FFT (CB/lwe_functions.cpp:434-446): x_d [batch][n+1] -> out_d [batch][n+1] */
int tfhe_amd_bootstrap(tfhe_amd_ctx *ctx, int32_t *out_d, int32_t mu, const int32_t *x_d, int batch);
/* the "one external-product kernel per CMux" schedule of the same bootstrap (BASELINE
 * config 2): n launches of tfhe_amd_mux_rotate on an HBM-resident accumulator batch.
 * Identical results; exists so the streamed schedule can be measured. */
int tfhe_amd_bootstrap_streamed(tfhe_amd_ctx *ctx, int32_t *out_d, int32_t mu, const int32_t *x_d, int batch);
/* host-pointer convenience form of tfhe_amd_bootstrap (copies in, runs, copies out, syncs) */
int tfhe_amd_bootstrap_host(tfhe_amd_ctx *ctx, int32_t *out, int32_t mu, const int32_t *x, int batch);

/* circuit-bootstrap blind rotation, circuitBootstrapWoKS (poc:530-659) with the library's
 * rotation semantics (see DESIGN.md "PoC defects"): Torus64 contexts.  abar_d [batch][n+1]
 * (preModSwitch output, poc:472-484), lwe_out_d [batch][N+1] int64. */
int tfhe_amd_cb_bootstrap_woks(tfhe_amd_ctx *ctx, int64_t *lwe_out_d, int64_t mu, const int32_t *abar_d, int batch);
/* preModSwitch (poc:472-484): x_d [batch][n+1] Torus32 -> out_d [batch][n+1] in [0, 2N) */
int tfhe_amd_modswitch(tfhe_amd_ctx *ctx, int32_t *out_d, const int32_t *x_d, int batch);

/* ---- circuit bootstrapping (TLWE -> TRGSW), the PoC of circuit-bootstrapping/src ------------
 * tfhe_CircuitBootstrapFFT (poc:823-873) = preKeySwitch lvl1->lvl0 (poc:437-465), preModSwitch
 * (poc:472-484), then for w < l1: circuitBootstrapWoKS (poc:530-659, Torus64, ring N2) with
 * mu = 2^(64-(w+1)Bgbit1) and k+1 circuitPrivKS (poc:667-698) back to TLWE32 rows of ring N1.
 * Parameters: the PoC's Globals statics (poc:70-85, poc_types.h:267-283). */
typedef struct tfhe_amd_cb_params {
    int32_t n0, N1, N2; /* lvl0 LWE dimension; lvl1 ring degree (= lvl1 LWE dimension); lvl2 ring degree */
    int32_t l1, Bgbit1; /* gadget of the output TGSW (lvl1) */
    int32_t l2, Bgbit2; /* gadget of the bootstrapping key (lvl2, Torus64) */
    int32_t t10, bb10;  /* preKS lvl1 -> lvl0: length, base bits */
    int32_t t21, bb21;  /* privKS lvl2 -> lvl1: length, base bits (1..3) */
} tfhe_amd_cb_params;
typedef struct tfhe_amd_cb tfhe_amd_cb;

int tfhe_amd_cb_create(const tfhe_amd_cb_params *params, int device, tfhe_amd_cb **out);
void tfhe_amd_cb_destroy(tfhe_amd_cb *cb);
const char *tfhe_amd_cb_last_error(const tfhe_amd_cb *cb);
/* both levels' contexts onto one stream; like tfhe_amd_set_stream the switch neither waits nor orders: see there */
int tfhe_amd_cb_set_stream(tfhe_amd_cb *cb, void *hip_stream);
int tfhe_amd_cb_sync(tfhe_amd_cb *cb);
/* the two single-level contexts inside (borrowed, do not destroy): Torus32 lvl1->lvl0 context
 * (preKeySwitch = tfhe_amd_keyswitch) and Torus64 lvl2 context (tfhe_amd_modswitch =
 * preModSwitch, tfhe_amd_cb_bootstrap_woks = circuitBootstrapWoKS, memory helpers) */
tfhe_amd_ctx *tfhe_amd_cb_ctx_lvl10(tfhe_amd_cb *cb);
tfhe_amd_ctx *tfhe_amd_cb_ctx_lvl2(tfhe_amd_cb *cb);
/* Globals::preKS, host layout [N1][t10][1<<bb10][n0+1] int32 (poc:375) */
int tfhe_amd_cb_load_preks(tfhe_amd_cb *cb, const int32_t *preks);
/* Globals::bk in coefficient form [n0][2*l2][2][N2] int64 (poc:388; converted on the GPU like
 * poc:395-402) or already in Lagrange form [n0][2*l2][2][N2] doubles (Globals::bkFFT) */
int tfhe_amd_cb_load_bk_torus(tfhe_amd_cb *cb, const int64_t *bk);
int tfhe_amd_cb_load_bk_fft(tfhe_amd_cb *cb, const double *bkfft);
/* Globals::privKS, host layout [2][N2+1][t21][1<<bb21][2][N1] int32 (poc:408); `u_plane` in
 * {0,1} uploads one [N2+1][t21][base][2][N1] plane (1.3 GB each at the PoC parameters) */
int tfhe_amd_cb_load_privks_plane(tfhe_amd_cb *cb, int u_plane, const int32_t *plane);
/* circuitPrivKS: x_d [batch][N2+1] int64 -> out_d [batch][2][N1] int32 */
int tfhe_amd_privks(tfhe_amd_cb *cb, int32_t *out_d, int u, const int64_t *x_d, int batch);
/* tfhe_CircuitBootstrapFFT: x_d [batch][N1+1] LWE32 -> out_d [batch][2][l1][2][N1] int32
 * (TGswSample32::samples[u][w], each a TLWE32 (a, b)) */
int tfhe_amd_circuit_bootstrap(tfhe_amd_cb *cb, int32_t *out_d, const int32_t *x_d, int batch);

/* ---- several GPUs in one process: a pool of contexts, one per device ----------------------------------------------
 * The reference's only parallel construct is an independent-item loop (parallel/src/test_parallel_multiplications.cpp:62,
 * `#pragma omp parallel for`): samples never interact.  A pool maps that loop onto the GPUs of a node: one context, ONE
 * HOST THREAD and one pinned staging buffer per member, the CALLER'S keys uploaded once to every member
 * (LweBootstrappingKeyFFT ownership: CB/lwe_functions.cpp:287-316 -- the caller keeps the host key, the pool keeps device
 * copies), a batch cut into contiguous slices (member r of m gets floor(count/m) samples, the first count%m members one
 * more), no collective, outputs written straight into the caller's array.  `devices` may name a device more than
 * once (two members then share that GPU, each with its own stream and key copy).  A pool is not re-entrant: one call at
 * a time (calls from several host threads are serialised). */
typedef struct tfhe_amd_pool tfhe_amd_pool;
int tfhe_amd_pool_create(const tfhe_amd_params *params, const int *devices, int n_devices, tfhe_amd_pool **out);
void tfhe_amd_pool_destroy(tfhe_amd_pool *pool);
const char *tfhe_amd_pool_last_error(const tfhe_amd_pool *pool);
int tfhe_amd_pool_size(const tfhe_amd_pool *pool);
int tfhe_amd_pool_device(const tfhe_amd_pool *pool, int member);
/* member's context (borrowed; for options, events, the device-pointer entry points) */
tfhe_amd_ctx *tfhe_amd_pool_ctx(tfhe_amd_pool *pool, int member);
/* one upload per member, in parallel: bkfft = LweBootstrappingKeyFFT::bkFFT flattened, [n][(k+1)l][k+1][N] doubles (as
 * tfhe_amd_gsw_from_fft); ks = LweKeySwitchKey::ks flattened (as tfhe_amd_load_keyswitch_key); either may be NULL.
 * A load that fails on SOME members (the status and tfhe_amd_pool_last_error name the first: "member i (device d): ...")
 * leaves the members with different keys: every operation of the pool then returns TFHE_AMD_ERR_STATE without running
 * until a load of that key has succeeded on all members (tfhe_amd_cb_pool_load_*: the same, per key component). */
int tfhe_amd_pool_load_keys(tfhe_amd_pool *pool, const double *bkfft, const int32_t *ks);
/* the same with the bootstrapping key in coefficient form (converted on every device, tGswToFFTConvert) */
int tfhe_amd_pool_load_keys_torus(tfhe_amd_pool *pool, const void *bk_torus, const int32_t *ks);
/* tfhe_bootstrap_FFT / tfhe_bootstrap_woKS_FFT / lweKeySwitch over `count` samples in HOST memory, sharded over the members:
 * x [count][n+1] -> out [count][n+1];  x [count][n+1] -> out [count][N+1];  x [count][N+1] -> out [count][ks_n_out+1] */
int tfhe_amd_pool_bootstrap_host(tfhe_amd_pool *pool, int32_t *out, int32_t mu, const int32_t *x, int count);
int tfhe_amd_pool_bootstrap_woks_host(tfhe_amd_pool *pool, int32_t *out, int32_t mu, const int32_t *x, int count);
int tfhe_amd_pool_keyswitch_host(tfhe_amd_pool *pool, int32_t *out, const int32_t *x, int count);
/* The same three operations on the CALLER'S OWN representation of the samples (e.g. an array of LweSample pointers): `get`
 * fills a member's pinned staging buffer with rows [first, first + rows) in the flat layout above, `put` takes the result
 * rows; no intermediate flat array, and in the pipelined form the gather of chunk k + 1 and the scatter of chunk k - 1 run while
 * chunk k computes.  Both are called on the members' threads, concurrently for DISJOINT row ranges. */
typedef void (*tfhe_amd_rows_in_fn)(void *user, int first, int rows, int32_t *dst);
typedef void (*tfhe_amd_rows_out_fn)(void *user, int first, int rows, const int32_t *src);
int tfhe_amd_pool_bootstrap_rows(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int32_t mu, int count);
int tfhe_amd_pool_bootstrap_woks_rows(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int32_t mu, int count);
int tfhe_amd_pool_keyswitch_rows(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int count);
/* Inside a member a slice of at least 2 x CHUNK_ROWS samples is PIPELINED: chunks of CHUNK_ROWS whose kernels run back to
 * back on the member's stream while a second stream copies the next chunk in and a third copies the previous one out (three
 * sets of pinned staging buffers; shorter slices go as one piece).  TFHE_AMD_POOL_OPT_CHUNK_ROWS: default 2048 -- the blind
 * rotation needs about that many samples per launch for its full rate; 0 = never pipeline.  Same results bit for bit. */
enum { TFHE_AMD_POOL_OPT_CHUNK_ROWS = 1 };
int tfhe_amd_pool_set_option(tfhe_amd_pool *pool, int option, int value);
/* what the last sharded call did: per member the number of samples and the host seconds its thread spent; arrays of
 * tfhe_amd_pool_size entries, either may be NULL */
int tfhe_amd_pool_last_split(const tfhe_amd_pool *pool, int *counts, double *seconds);

/* the circuit bootstrap the same way (tfhe_CircuitBootstrapFFT over the driver loop poc:1009-1013) */
typedef struct tfhe_amd_cb_pool tfhe_amd_cb_pool;
int tfhe_amd_cb_pool_create(const tfhe_amd_cb_params *params, const int *devices, int n_devices, tfhe_amd_cb_pool **out);
void tfhe_amd_cb_pool_destroy(tfhe_amd_cb_pool *pool);
const char *tfhe_amd_cb_pool_last_error(const tfhe_amd_cb_pool *pool);
int tfhe_amd_cb_pool_size(const tfhe_amd_cb_pool *pool);
tfhe_amd_cb *tfhe_amd_cb_pool_member(tfhe_amd_cb_pool *pool, int member);
int tfhe_amd_cb_pool_load_preks(tfhe_amd_cb_pool *pool, const int32_t *preks);
int tfhe_amd_cb_pool_load_bk_fft(tfhe_amd_cb_pool *pool, const double *bkfft);
int tfhe_amd_cb_pool_load_bk_torus(tfhe_amd_cb_pool *pool, const int64_t *bk);
int tfhe_amd_cb_pool_load_privks_plane(tfhe_amd_cb_pool *pool, int u_plane, const int32_t *plane);
/* x [count][N1+1] LWE32 (host) -> out [count][2][l1][2][N1] int32 (host); _rows: through the caller's callbacks (rows in the
 * flat layouts just named).  A member pipelines a slice of >= 2 x CHUNK_ROWS inputs (default 1024: one Torus64 ciphertext per wave
 * on every SIMD) like the gate pool; tfhe_amd_cb_pool_set_option(pool, TFHE_AMD_POOL_OPT_CHUNK_ROWS, rows) changes it. */
int tfhe_amd_cb_pool_circuit_bootstrap_host(tfhe_amd_cb_pool *pool, int32_t *out, const int32_t *x, int count);
int tfhe_amd_cb_pool_circuit_bootstrap_rows(tfhe_amd_cb_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int count);
int tfhe_amd_cb_pool_set_option(tfhe_amd_cb_pool *pool, int option, int value);

/* ---- harness: synthetic keys and samples (the reference's keygen/encrypt/phase,
 *      poc:88-134,191-227,342-423; PRNG: splitmix64, one stream per key element, csrc/keygen.cpp) -- host side ------------------ */
int tfhe_amd_keygen_binary(int32_t *key, int n, uint64_t seed, uint64_t stream);
int tfhe_amd_lwe_encrypt32(int32_t *ct, int32_t mess, double stdev, const int32_t *key, int n,
                           uint64_t seed, uint64_t stream);
int32_t tfhe_amd_lwe_phase32(const int32_t *ct, const int32_t *key, int n);
/* bootstrapping key in coefficient form [n][(k+1)l][k+1][N] (feed to tfhe_amd_gsw_from_torus) */
int tfhe_amd_keygen_bk_torus32(int32_t *bk, const int32_t *lwe_key, int n, const int32_t *tlwe_key, int N,
                               int l, int Bgbit, double stdev, uint64_t seed, uint64_t stream);
int tfhe_amd_keygen_bk_torus64(int64_t *bk, const int32_t *lwe_key, int n, const int32_t *tlwe_key, int N,
                               int l, int Bgbit, double stdev, uint64_t seed, uint64_t stream);
int tfhe_amd_keygen_ks32(int32_t *ks, const int32_t *in_key, int n_in, const int32_t *out_key, int n_out,
                         int t, int basebit, double stdev, uint64_t seed, uint64_t stream);

#ifdef __cplusplus
}
#endif
#pragma pop_macro("k")
#endif /* TFHE_AMD_H */
