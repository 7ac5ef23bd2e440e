// ubench_mfma_f64.hip -- which rounding sequence does v_mfma_f64_16x16x4_f64 follow?  (run ON THE GPU BOX)
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/ubench/ubench_mfma_f64.hip -o tools/ubench/ubench_mfma_f64 && tools/ubench/ubench_mfma_f64
// D = A(16x4) * B(4x16) + C(16x16) on one wave; every element of D is compared with four CPU candidates:
//   up    d = c; for k = 0..3: d = fma(a_k, b_k, d)         (a sequential fma chain, one rounding per step, k ascending)
//   down  the same chain, k descending
//   exact the exactly rounded value of c + sum a_k b_k (long double / compensated, one rounding)
//   pair  fma(a0,b0, fma(a1,b1,.)) style two-level trees are not tried: "up" or "down" matching everywhere answers the question
// The blind rotation's MAC (lagrangehalfc_impl_fma.s:96-107) is such a chain; DESIGN.md section 3 / docs/experiments.md state why the matrix
// cores are not used for it -- this program is the measurement behind that sentence.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k(const double *A, const double *B, const double *C, double *D) {
    const int lane = threadIdx.x;
    const int i = lane & 15, kk = lane >> 4;     // A[i][k], B[k][j]: lane = 16 k + i (resp. j)
    const double a = A[i * 4 + kk], b = B[kk * 16 + i];
    v4d c;
    for (int r = 0; r < 4; r++) c[r] = C[(4 * r + kk) * 16 + i];  // D/C: row 4 r + lane / 16, column lane % 16
    const v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * r + kk) * 16 + i] = d[r];
}

int main() {
    double hA[64], hB[64], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
    long up = 0, down = 0, exact = 0, total = 0, layout_bad = 0;
    srand(7);
    for (int trial = 0; trial < 200; trial++) {
        // magnitudes spread over ~2^20 so that the order of the additions changes the roundings
        auto rnd = [&]() { return ldexp((double)rand() / RAND_MAX - 0.5, rand() % 21 - 10); };
        for (double &x : hA) x = rnd();
        for (double &x : hB) x = rnd();
        for (double &x : hC) x = rnd();
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
        hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double u = hC[i * 16 + j], d = hC[i * 16 + j];
                long double e = hC[i * 16 + j];
                for (int q = 0; q < 4; q++) u = fma(hA[i * 4 + q], hB[q * 16 + j], u);
                for (int q = 3; q >= 0; q--) d = fma(hA[i * 4 + q], hB[q * 16 + j], d);
                for (int q = 0; q < 4; q++) e += (long double)hA[i * 4 + q] * (long double)hB[q * 16 + j];
                const double got = hD[i * 16 + j];
                total++;
                up += got == u;
                down += got == d;
                exact += got == (double)e;
                if (fabs(got - u) > 1e-6 * (fabs(u) + 1e-300)) layout_bad++;
            }
    }
    printf("v_mfma_f64_16x16x4_f64 vs CPU over %ld elements: chain k ascending %ld, chain k descending %ld, "
           "single rounding of the long-double sum %ld, far off (operand layout wrong) %ld\n", total, up, down, exact, layout_bad);
    return 0;
}
