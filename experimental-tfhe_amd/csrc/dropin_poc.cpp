// dropin_poc.cpp -- the PoC's five entry points (and the CMux it leaves empty) with the reference's
// full signatures, `const Globals* env` included, served by the MI355X engine:
//     preKeySwitch           CB/poc_CircuitBootstrapping.cpp:437-465
//     preModSwitch           :472-484
//     circuitBootstrapWoKS   :530-659  (library rotation semantics)
//     circuitPrivKS          :667-698
//     tfhe_CircuitBootstrapFFT :823-873  (+ _array: the driver loop :1009-1013 over `count` samples as one launch)
//     CMux                   :877-879 (empty upstream)
// These have C++ linkage on the PoC's own types, so this file is compiled NEXT TO the PoC's poc_types.h
// (add it to the PoC's build in place of the bodies above) and linked with
// libtfhe_amd.so.  One engine per `env` pointer, created on first use: Globals::preKS, bkFFT and privKS
// are flattened and uploaded once.
#include <map>

#include "poc_types.h"

#define TFHE_AMD_DROPIN_POC
#include "tfhe_amd_dropin.h"
#include "tfhe_amd_compat.hpp"

namespace {
typedef tfhe_amd_compat::PocEngine<Globals> Engine;
std::map<const Globals *, Engine *> &engines() {
    static std::map<const Globals *, Engine *> m;
    return m;
}
Engine &engine_of(const Globals *env) {
    auto &m = engines();
    auto it = m.find(env);
    if (it == m.end()) it = m.emplace(env, new Engine(env, tfhe_amd_compat::device_ordinal())).first;
    return *it->second;
}
}  // namespace

void preKeySwitch(LweSample32 *result, const LweSample32 *x, const Globals *env) { engine_of(env).preKeySwitch(result, x); }
void preModSwitch(int *result, const LweSample32 *x, const Globals *env) { engine_of(env).preModSwitch(result, x); }
void circuitBootstrapWoKS(LweSample64 *result, const Torus64 mu, const int *abar, const Globals *env) {
    engine_of(env).circuitBootstrapWoKS(result, mu, abar);
}
void circuitPrivKS(TLweSample32 *result, const int u, const LweSample64 *x, const Globals *env) {
    engine_of(env).circuitPrivKS(result, u, x);
}
void tfhe_CircuitBootstrapFFT(TGswSample32 *result, const LweSample32 *sample, const Globals *env) {
    engine_of(env).tfhe_CircuitBootstrapFFT(result, sample);
}
void tfhe_CircuitBootstrapFFT_array(TGswSample32 *const *results, const LweSample32 *const *samples, const Globals *env, int count) {
    engine_of(env).tfhe_CircuitBootstrapFFT_array(results, samples, count);
}
void CMux(TLweSample32 *out, const TGswSample32 *c, const TLweSample32 *in0, const TLweSample32 *in1, const Globals *env) {
    engine_of(env).CMux(out, c, in0, in1);
}
void tfhe_amd_dropin_release(const Globals *env) {
    auto &m = engines();
    auto it = m.find(env);
    if (it == m.end()) return;
    delete it->second;
    m.erase(it);
}
