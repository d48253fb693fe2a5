// micro-benchmark: how fast can ONE launch stream N bytes from HBM on gfx950 (fixed cost vs asymptotic bandwidth)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U, bool NT>
__global__ void stream_read(const u32x4* __restrict__ p, long n_vec, unsigned* out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n_vec; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const long maxb = 1L << 30;
    char* buf; unsigned* out;
    (void)hipMalloc(&buf, maxb * 2 + (1 << 20)); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 1, maxb * 2 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    long sizes[] = {8L << 20, 16L << 20, 45L << 20, 90L << 20, 256L << 20, 1L << 30};
    int grids[] = {128, 256, 512, 1024, 2048};
    int threads[] = {256, 1024};
    printf("bytes_MB grid threads U nt  us  GB/s\n");
    for (long sz : sizes) for (int g : grids) for (int t : threads) {
        for (int variant = 0; variant < 2; ++variant) {
            const int reps = 20;
            float best = 1e9;
            for (int r = 0; r < reps; ++r) {
                const u32x4* p = (const u32x4*)(buf + ((r & 1) ? maxb : 0) + (long)(r % 7) * 4096 * 16);   // alternate regions: no cache reuse
                hipEventRecord(e0, 0);
                if (variant == 0) hipLaunchKernelGGL((stream_read<8, false>), dim3(g), dim3(t), 0, 0, p, sz / 16, out);
                else hipLaunchKernelGGL((stream_read<8, true>), dim3(g), dim3(t), 0, 0, p, sz / 16, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%5ld %5d %5d 8 %d %8.2f %8.1f\n", sz >> 20, g, t, variant, best * 1e3, sz / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
