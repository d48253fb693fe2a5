#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// A: [16][128] fp8 bytes row-major, B: [16][128] fp8 (B^T), C[16][16] = A * B^T
__global__ void k(const unsigned char* A, const unsigned char* B, float* C) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + r * 128 + g * 32);
    i32x8 b = *reinterpret_cast<const i32x8*>(B + r * 128 + g * 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    // assume: c[i] = C[row = g*4 + i][col = r]  (A rows index M)
    for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];
}
static float e4m3(unsigned char v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
    if (e == 15 && m == 7) f = NAN;
    return s ? -f : f;
}
int main() {
    unsigned char hA[16 * 128], hB[16 * 128];
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { do { hA[i] = rand() & 0xff; } while ((hA[i] & 0x7f) == 0x7f); do { hB[i] = rand() & 0xff; } while ((hB[i] & 0x7f) == 0x7f); hA[i] &= 0xBF; hB[i] &= 0xBF; }
    unsigned char *dA, *dB; float* dC; float hC[256];
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0, maxerrT = 0, maxref = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double ref = 0; for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3(hA[m * 128 + kk]) * e4m3(hB[n * 128 + kk]);
        maxerr = fmax(maxerr, fabs(ref - hC[m * 16 + n])); maxerrT = fmax(maxerrT, fabs(ref - hC[n * 16 + m])); maxref = fmax(maxref, fabs(ref));
    }
    printf("f8f6f4 16x16x128: max |err| %.4g (as C[m][n]), %.4g (transposed); max |ref| %.4g\n", maxerr, maxerrT, maxref);
    return 0;
}
