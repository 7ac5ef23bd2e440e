/*
 * tfhe_amd_spqlios.h -- the reference's FFT plugin seam BY ITS OWN SYMBOLS, served by the MI355X engine.
 *
 * libtfhe_amd_spqlios.so exports exactly what the five spqlios objects of the reference export to their callers
 * (CB/ = circuit-bootstrapping/src/):
 *
 *   class FFT_Processor_Spqlios  + fftp1024, fftp2048        CB/spqlios/lagrangehalfc_impl.h:8-34
 *                                                             (defined in CB/spqlios/fft_processor_spqlios.cpp:19-170)
 *   extern "C" LagrangeHalfCPolynomialAddMulASM              CB/spqlios/lagrangehalfc_impl.h:36
 *                                                             (CB/spqlios/lagrangehalfc_impl_fma.s:78-135)
 *   extern "C" new_fft_table, new_ifft_table, fft_table_get_buffer, ifft_table_get_buffer, fft, ifft,
 *              fft_model, ifft_model                          CB/spqlios/spqlios-fft.h:46-53
 *
 * so the reference's UNMODIFIED objects (poc_CircuitBootstrapping.o and its callers at poc:248-283) link against it:
 * replace the `spqlios/` objects on the link line by `-ltfhe_amd_spqlios`.  A caller that includes
 * the reference's own lagrangehalfc_impl.h / spqlios-fft.h needs nothing from this file; it exists so that code
 * without the reference tree can compile against the same declarations.
 *
 * The class layout below (three const ints, four double*, two void*) IS the ABI: the reference's objects hold
 * `fftp1024` / `fftp2048` by value (copy relocation) and read N / Ns2 from them.  The engine behind an object is
 * created at its first execute_* call, never in a static constructor (a process that links the library and never
 * transforms anything does not touch the GPU).  Like the reference (per-instance scratch buffers,
 * fft_processor_spqlios.cpp:21-24) one object serves one thread at a time; calls are synchronous, one polynomial
 * per call: h2d, one kernel, d2h.  Device ordinal: environment TFHE_AMD_DEVICE (default 0).  Errors abort with a
 * message, as the reference's `require` does (spqlios-fft-impl.cpp:92-97).
 */
#ifndef TFHE_AMD_SPQLIOS_H
#define TFHE_AMD_SPQLIOS_H

#include <cstdint>

class FFT_Processor_Spqlios {
   public:
    const int _2N;
    const int N;
    const int Ns2;

   private:
    /* same storage as the reference's object (its scratch buffers and tables); all of it stays null here: the engine
     * of this ring degree lives in the library (created at the first execute_* call, calls serialised per ring degree). */
    double* real_inout_direct;
    double* imag_inout_direct;
    double* real_inout_rev;
    double* imag_inout_rev;
    void* tables_direct;
    void* tables_reverse;

   public:
    FFT_Processor_Spqlios(const int N);

    void execute_reverse_int(double* res, const int* a);
    void execute_reverse_torus32(double* res, const int32_t* a);
    void execute_direct_torus32(int32_t* res, const double* a);
    void execute_reverse_torus64(double* res, const int64_t* a);
    void execute_direct_torus64(int64_t* res, const double* a);

    ~FFT_Processor_Spqlios();
};

extern FFT_Processor_Spqlios fftp1024;
extern FFT_Processor_Spqlios fftp2048;

extern "C" void LagrangeHalfCPolynomialAddMulASM(double* res, double* a, double* b, long Ns2);

extern "C" {
void* new_fft_table(int nn);
double* fft_table_get_buffer(const void* tables);
void* new_ifft_table(int nn);
double* ifft_table_get_buffer(const void* tables);
void fft_model(const void* tables);
void ifft_model(void* tables);
void fft(const void* tables, double* data);
void ifft(const void* tables, double* data);
}

#endif /* TFHE_AMD_SPQLIOS_H */
