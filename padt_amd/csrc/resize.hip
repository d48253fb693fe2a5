// Image resize on the GPU, bit-exact with Pillow's ImagingResample for 8-bit images (libImaging/Resample.c): the resampler the
// reference gets from `AutoProcessor` (HF Qwen2-VL image processor: PIL BICUBIC) and calls itself with LANCZOS in
// eval/test_demo.py:67-73 / eval/evaluation_scripts/utils.py:205-218.  Pillow resamples in two separable passes (horizontal, then
// vertical) with an 8-bit intermediate image, fixed-point coefficients of 22 fractional bits and
//     out = clip8((1 << 21) + Σ_k in[xmin + k] * kk[k]) >> 22)
// — pure integer arithmetic, so a kernel that is handed the same coefficient tables (built on the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do, padt_amd/preprocess.py) reproduces every byte.
#include "common.h"

extern "C" void padt_set_error(const char* msg);

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

// one thread per output byte; horizontal: out[y][x][c] over in[y][xmin..][c]; vertical: out[y][x][c] over in[ymin..][x][c]
__global__ __launch_bounds__(256) void resample_pass_kernel(const unsigned char* __restrict__ in, int in_h, int in_w, int C,
                                                            unsigned char* __restrict__ out, int out_h, int out_w,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                            int horizontal) {
    const long total = (long)out_h * out_w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % out_w);
        const int y = (int)(i / ((long)C * out_w));
        const int o = horizontal ? x : y;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (long)o * ksize;
        int ss = 1 << (PRECISION_BITS - 1);
        if (horizontal) {
            const unsigned char* p = in + ((long)y * in_w + lo) * C + c;
            for (int t = 0; t < n; ++t) ss += (int)p[(long)t * C] * k[t];
        } else {
            const unsigned char* p = in + ((long)lo * in_w + x) * C + c;
            for (int t = 0; t < n; ++t) ss += (int)p[(long)t * in_w * C] * k[t];
        }
        int v = ss >> PRECISION_BITS;                              // arithmetic shift, then Pillow's clip8 lookup = clamp to [0, 255]
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        out[i] = (unsigned char)v;
    }
}
}  // namespace

// One separable pass of Pillow's 8-bit resample: horizontal (in_h == out_h) or vertical (in_w == out_w).  bounds int32 [out][2] =
// (first source index, tap count), kk int32 [out][ksize] fixed-point taps (22 fractional bits).  Interleaved channels (H, W, C).
extern "C" int padt_resample_pass_u8(void* stream, const void* in_u8, int in_h, int in_w, int channels, void* out_u8, int out_h,
                                     int out_w, const int* bounds, const int* kk, int ksize, int horizontal) {
    if (out_h <= 0 || out_w <= 0) return 0;
    if (channels <= 0 || ksize <= 0 || (horizontal ? in_h != out_h : in_w != out_w)) {
        padt_set_error("padt_resample_pass_u8: a horizontal pass keeps the height, a vertical pass the width; channels, ksize > 0");
        return -1;
    }
    const long total = (long)out_h * out_w * channels;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(resample_pass_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)in_u8, in_h,
                       in_w, channels, (unsigned char*)out_u8, out_h, out_w, bounds, kk, ksize, horizontal);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}
