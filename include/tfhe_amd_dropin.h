// tfhe_amd_dropin.h -- the reference's entry points at GLOBAL scope, with the reference's full
// signatures: what an existing driver declares (or would get from the upstream headers the reference
// ships without) and links against libtfhe_amd_dropin.so instead of the CPU objects.
//
// Library form (declared EXPORT = extern "C" upstream):
//     tfhe_blindRotate_FFT            CB/lwe_functions.cpp:337-341
//     tfhe_blindRotateAndExtract_FFT  CB/lwe_functions.cpp:366-372
//     tfhe_bootstrap_woKS_FFT         CB/lwe_functions.cpp:399-402
//     tfhe_bootstrap_FFT              CB/lwe_functions.cpp:434-437
//     tGswFFTExternMulToTLwe          CB/tgsw_functions.cpp:424
//     tfhe_MuxRotate_FFT              CB/lwe_functions.cpp:328
//     lweKeySwitch                    CB/lwe_functions.cpp:163
//   The struct definitions below carry the fields those files use; a driver that has the
//   upstream headers defines TFHE_AMD_DROPIN_NO_LIBRARY_TYPES before including this file.
// PoC form (C++ linkage, types of CB/poc_types.h, `const Globals* env` kept):
//     preKeySwitch :437   preModSwitch :472   circuitBootstrapWoKS :530   circuitPrivKS :667
//     tfhe_CircuitBootstrapFFT :823   (+ tfhe_CircuitBootstrapFFT_array: the driver loop poc:1009-1013 as one launch)
//   Declared here when TFHE_AMD_DROPIN_POC is defined AFTER poc_types.h has been included; they are
//   defined by experimental-tfhe_amd/csrc/dropin_poc.cpp, which a PoC build compiles next to its own
//   poc_types.h (C++ linkage on user types cannot be pre-built).  The engine behind an `env` is created
//   on first use (keys uploaded once) and kept until tfhe_amd_dropin_release(env).  release may be called while other host
//   threads are inside tfhe_CircuitBootstrapFFT(env) (the one entry point that runs outside the shims' lock): those calls
//   finish on the engine they started on, which is destroyed when the last of them returns.
#ifndef TFHE_AMD_DROPIN_H
#define TFHE_AMD_DROPIN_H

#include <stdint.h>

#ifndef TFHE_AMD_DROPIN_POC
#ifndef TFHE_AMD_DROPIN_NO_LIBRARY_TYPES
typedef int32_t Torus32;
typedef int64_t Torus64;
#include "tfhe_amd_library_types.inc"
#endif

extern "C" {
void tfhe_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bkFFT, const int *bara, const int n,
                          const TGswParams *bk_params);
void tfhe_blindRotateAndExtract_FFT(LweSample *result, const TorusPolynomial *v, const TGswSampleFFT *bk, const int barb,
                                    const int *bara, const int n, const TGswParams *bk_params);
void tfhe_bootstrap_woKS_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, Torus32 mu, const LweSample *x);
void tfhe_bootstrap_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, Torus32 mu, const LweSample *x);
void tGswFFTExternMulToTLwe(TLweSample *accum, const TGswSampleFFT *gsw, const TGswParams *params);
void tfhe_MuxRotate_FFT(TLweSample *result, const TLweSample *accum, const TGswSampleFFT *bki, const int barai,
                        const TGswParams *bk_params);
void lweKeySwitch(LweSample *result, const LweKeySwitchKey *ks, const LweSample *sample);
/* ARRAY FORMS: the caller's loop over `count` independent samples (the reference's drivers loop over one-sample calls,
 * poc:1009-1013; parallel/src/test_parallel_multiplications.cpp:62) as one gather, ONE launch, one scatter.
 * results[c] / xs[c]: the loop's own caller-allocated objects; same results as `count` one-sample calls, bit for bit. */
void tfhe_bootstrap_woKS_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                                   const LweSample *const *xs, int count);
void tfhe_bootstrap_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                              const LweSample *const *xs, int count);
void lweKeySwitch_array(LweSample *const *results, const LweKeySwitchKey *ks, const LweSample *const *samples, int count);
/* frees every GPU-resident copy made for this key object (bk, bkFFT array or ks); NULL: all of them */
void tfhe_amd_dropin_release(const void *key_object);
/* GPU ordinal used by engines created from now on (default 0) */
void tfhe_amd_dropin_set_device(int device);
/* several GPUs: with n > 1 the ARRAY forms cut the caller's loop into contiguous slices over the named devices (one
 * context, host thread and pinned staging buffer per device, the caller's key uploaded once to each: tfhe_amd_pool in
 * tfhe_amd.h) -- the reference's `#pragma omp parallel for` over independent items
 * (parallel/src/test_parallel_multiplications.cpp:62) with GPUs as the workers; one-sample calls stay on devices[0] */
void tfhe_amd_dropin_set_devices(const int *devices, int n);
}
#else  /* TFHE_AMD_DROPIN_POC: poc_types.h has been included */
void preKeySwitch(LweSample32 *result, const LweSample32 *x, const Globals *env);
void preModSwitch(int *result, const LweSample32 *x, const Globals *env);
void circuitBootstrapWoKS(LweSample64 *result, const Torus64 mu, const int *abar, const Globals *env);
void circuitPrivKS(TLweSample32 *result, const int u, const LweSample64 *x, const Globals *env);
void tfhe_CircuitBootstrapFFT(TGswSample32 *result, const LweSample32 *sample, const Globals *env);
/* array form of the driver loop poc:1009-1013: `count` circuit bootstraps as one launch */
void tfhe_CircuitBootstrapFFT_array(TGswSample32 *const *results, const LweSample32 *const *samples, const Globals *env, int count);
/* the reference declares CMux and leaves its body empty (poc:877-879): out = c ? in1 : in0 */
void CMux(TLweSample32 *out, const TGswSample32 *c, const TLweSample32 *in0, const TLweSample32 *in1, const Globals *env);
void tfhe_amd_dropin_release(const Globals *env);
/* several GPUs behind tfhe_CircuitBootstrapFFT_array (see the library form above); defined by dropin_poc.cpp too, so that a
 * PoC build needs no other object */
void tfhe_amd_dropin_poc_set_devices(const int *devices, int n);
#endif
#endif
