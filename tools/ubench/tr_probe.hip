// Probe of ds_read_b64_tr_b16 lane semantics: every lane supplies the address of its own 4-element (8-byte) chunk, chunk
// index = lane; LDS holds lds[i] = i.  Prints, per lane, the 4 values it received.
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int stride) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + l * stride));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    for (int stride : {4, 40}) {
        probe<<<1, 64>>>(d, stride);
        short h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("chunk stride %d elements:\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
