// Compile-only check (tests/test_compat.py): every PoC-form shim instantiated against the REFERENCE'S OWN
// poc_types.h, which this translation unit includes from /root/reference (never copied).  Catches drift
// between the duck-typed template and the real structs -- e.g. the PoC's `#define k 1` (poc_types.h:10).
#include "poc_types.h"
#include "tfhe_amd_compat.hpp"
using namespace tfhe_amd_compat;
// instantiate every PoC-form shim against the reference's own structs (compile-only)
void use(const Globals *env, TGswSample32 *tg, LweSample32 *l32, LweSample64 *l64, TLweSample32 *tl, int *ab) {
    PocEngine<Globals> e(env);
    e.tfhe_CircuitBootstrapFFT(tg, l32);
    e.circuitBootstrapWoKS(l64, (Torus64)1, ab);
    e.circuitPrivKS(tl, 0, l64);
    e.preKeySwitch(l32, l32);
    e.preModSwitch(ab, l32);
    e.CMux(tl, tg, tl, tl);
}
