// dropin_library.cpp -- libtfhe_amd_dropin.so: the reference's library-form entry points at global
// scope, extern "C" as upstream's EXPORT (CB/lwe_functions.cpp:337-341,366-372,399-402,434-437,163;
// CB/tgsw_functions.cpp:424), forwarding to the shims of include/tfhe_amd_compat.hpp (which marshal the
// pointer-rich structs into the flat C ABI of include/tfhe_amd.h and keep the keys GPU-resident).
// Host C++ only; links against libtfhe_amd.so.
#include "tfhe_amd_dropin.h"

#define TFHE_AMD_COMPAT_NO_TYPES  // the shims then work on the global-scope structs declared above
#include "tfhe_amd_compat.hpp"

extern "C" {
void tfhe_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bkFFT, const int *bara, const int n,
                          const TGswParams *bk_params) {
    tfhe_amd_compat::tfhe_blindRotate_FFT(accum, bkFFT, bara, n, bk_params);
}
void tfhe_blindRotateAndExtract_FFT(LweSample *result, const TorusPolynomial *v, const TGswSampleFFT *bk, const int barb,
                                    const int *bara, const int n, const TGswParams *bk_params) {
    tfhe_amd_compat::tfhe_blindRotateAndExtract_FFT(result, v, bk, barb, bara, n, bk_params);
}
void tfhe_bootstrap_woKS_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, Torus32 mu, const LweSample *x) {
    tfhe_amd_compat::tfhe_bootstrap_woKS_FFT(result, bk, mu, x);
}
void tfhe_bootstrap_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, Torus32 mu, const LweSample *x) {
    tfhe_amd_compat::tfhe_bootstrap_FFT(result, bk, mu, x);
}
void tGswFFTExternMulToTLwe(TLweSample *accum, const TGswSampleFFT *gsw, const TGswParams *params) {
    tfhe_amd_compat::tGswFFTExternMulToTLwe(accum, gsw, params);
}
void tfhe_MuxRotate_FFT(TLweSample *result, const TLweSample *accum, const TGswSampleFFT *bki, const int barai,
                        const TGswParams *bk_params) {
    tfhe_amd_compat::tfhe_MuxRotate_FFT(result, accum, bki, barai, bk_params);
}
void lweKeySwitch(LweSample *result, const LweKeySwitchKey *ks, const LweSample *sample) {
    tfhe_amd_compat::lweKeySwitch(result, ks, sample);
}
void tfhe_bootstrap_woKS_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                                   const LweSample *const *xs, int count) {
    tfhe_amd_compat::tfhe_bootstrap_woKS_FFT_array(results, bk, mu, xs, count);
}
void tfhe_bootstrap_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                              const LweSample *const *xs, int count) {
    tfhe_amd_compat::tfhe_bootstrap_FFT_array(results, bk, mu, xs, count);
}
void lweKeySwitch_array(LweSample *const *results, const LweKeySwitchKey *ks, const LweSample *const *samples, int count) {
    tfhe_amd_compat::lweKeySwitch_array(results, ks, samples, count);
}
void tfhe_amd_dropin_release(const void *key_object) {
    if (key_object)
        tfhe_amd_compat::release(key_object);
    else
        tfhe_amd_compat::release_all();
}
void tfhe_amd_dropin_set_device(int device) { tfhe_amd_compat::set_device(device); }
void tfhe_amd_dropin_set_devices(const int *devices, int n) { tfhe_amd_compat::set_devices(devices, n); }
}
