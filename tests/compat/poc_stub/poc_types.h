// poc_types.h (TEST STAND-IN) -- the type names, public data members and constructor signatures of the
// PoC's header (CB/poc_types.h:40-312) that tests/compat/dropin_driver_poc.cpp and
// experimental-tfhe_amd/csrc/dropin_poc.cpp touch, so that the literal drop-in test can also run where
// /root/reference does not exist (the GPU box).  In the build container the same two files are ALSO
// compiled against the reference's real header (tests/test_dropin.py).  Not part of the product.
#ifndef TFHE_AMD_TEST_POC_TYPES_STUB
#define TFHE_AMD_TEST_POC_TYPES_STUB
#include <stddef.h>
#include <stdint.h>

#include <new>
#define k 1  // the PoC's macro for the TLWE dimension (poc_types.h:10); the shims must survive it
typedef int32_t Torus32;
typedef int64_t Torus64;
template <class E>
struct OwnedArray {  // members named `coefs` / `values` below own a heap array, as the PoC's do
    static E *make(int n) { return new E[n]; }
};
struct Torus32Polynomial {
    Torus32 *const coefs;
    explicit Torus32Polynomial(int N) : coefs(OwnedArray<Torus32>::make(N)) {}
    ~Torus32Polynomial() { delete[] coefs; }
};
struct LagrangeHalfCPolynomial {
    double *const values;
    explicit LagrangeHalfCPolynomial(int N) : values(OwnedArray<double>::make(N)) {}
    ~LagrangeHalfCPolynomial() { delete[] values; }
};
struct LweSample32 {
    Torus32 *const a;
    Torus32 *const b;
    explicit LweSample32(int n) : a(OwnedArray<Torus32>::make(n + 1)), b(a + n) {}
    ~LweSample32() { delete[] a; }
};
struct LweSample64 {
    Torus64 *const a;
    Torus64 *const b;
    explicit LweSample64(int n) : a(OwnedArray<Torus64>::make(n + 1)), b(a + n) {}
    ~LweSample64() { delete[] a; }
};
template <class P>
inline P *stub_new_polys(int count, int N) {  // `count` polynomials built in place in one block
    P *p = static_cast<P *>(operator new[](sizeof(P) * (size_t)count));
    for (int i = 0; i < count; i++) new (p + i) P(N);
    return p;
}
struct TLweSample32 {
    Torus32Polynomial *const a;
    Torus32Polynomial *const b;
    explicit TLweSample32(int N) : a(stub_new_polys<Torus32Polynomial>(k + 1, N)), b(a + k) {}
    ~TLweSample32() {
        for (int i = 0; i <= k; i++) a[i].~Torus32Polynomial();
        operator delete[](a);
    }
};
struct TLweSampleFFT {
    LagrangeHalfCPolynomial *const a;
    LagrangeHalfCPolynomial *const b;
    explicit TLweSampleFFT(int N) : a(stub_new_polys<LagrangeHalfCPolynomial>(k + 1, N)), b(a + k) {}
    ~TLweSampleFFT() {
        for (int i = 0; i <= k; i++) a[i].~LagrangeHalfCPolynomial();
        operator delete[](a);
    }
};
template <class Row>
inline Row **stub_new_rows(int blocs, int l, int N) {  // samples[bloc][i] over one contiguous block of rows
    Row *all = static_cast<Row *>(operator new[](sizeof(Row) * (size_t)(blocs * l)));
    for (int i = 0; i < blocs * l; i++) new (all + i) Row(N);
    Row **idx = new Row *[blocs];
    for (int b = 0; b < blocs; b++) idx[b] = all + b * l;
    return idx;
}
struct TGswSample32 {
    TLweSample32 **const samples;
    TLweSample32 *const allsamples;
    TGswSample32(int l, int N) : samples(stub_new_rows<TLweSample32>(k + 1, l, N)), allsamples(samples[0]) {}
};
struct TGswSampleFFT {
    TLweSampleFFT **const samples;
    TLweSampleFFT *const allsamples;
    TGswSampleFFT(int l, int N) : samples(stub_new_rows<TLweSampleFFT>(k + 1, l, N)), allsamples(samples[0]) {}
};
class Globals {
   public:
    static const int n_lvl0, n_lvl1, n_lvl2, bgbit_lvl1, ell_lvl1, bgbit_lvl2, ell_lvl2;
    static const int kslength_lvl10, ksbasebit_lvl10, kslength_lvl21, ksbasebit_lvl21;
    LweSample32 ***preKS;      // [n_lvl1][kslength_lvl10][base]
    TGswSampleFFT *bkFFT;      // [n_lvl0]
    TLweSample32 ****privKS;   // [k+1][n_lvl2+1][kslength_lvl21][base]
    Globals();
};
#endif
