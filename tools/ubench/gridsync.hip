// micro-benchmark for a persistent decode-step kernel on gfx950: what does a grid-wide barrier across the 8 XCDs cost, and how must
// activations written before it be read after it?
//   1. barrier latency: NB co-resident blocks, K barriers each = one agent-scope atomic add + polling of the same counter
//      (a) bare   (b) with agent-scope release / acquire fences (buffer_wbl2 sc1 / buffer_inv sc1) as cooperative groups' grid.sync() does
//   2. visibility: every block writes a generation number before each barrier and reads ANOTHER XCD's value after it, with
//      plain loads / sc1 loads, after plain stores / sc1 stores: counts stale reads
//   3. bandwidth of 16-byte sc1 loads (inline asm) against plain loads on an L2-resident 256 KB buffer read by every block
// All spin loops are bounded (a hung barrier returns an error instead of hanging the GPU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, bool fences) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) { ok = false; break; }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void barrier_bench(unsigned* counter, int K, int fences, int* err) {
    for (int k = 0; k < K; ++k)
        if (!grid_barrier(counter, (unsigned)(k + 1) * gridDim.x, fences != 0)) { if (threadIdx.x == 0) atomicExch(err, 1); return; }
}

__device__ __forceinline__ u32x4 load_sc1_x4(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1_x4(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// mode bit 0: sc1 stores, bit 1: sc1 loads, bit 2: fences in the barrier
__global__ __launch_bounds__(256) void visibility(unsigned* counter, u32x4* slots, int K, int mode, unsigned* stale, int* err) {
    const int nb = gridDim.x;
    unsigned bad = 0;
    for (int k = 0; k < K; ++k) {
        // each block owns 256 slots (one per thread): write generation k + 1
        u32x4 v = {(unsigned)(k + 1), blockIdx.x, threadIdx.x, 0u};
        u32x4* mine = slots + (long)blockIdx.x * 256 + threadIdx.x;
        if (mode & 1) store_sc1_x4(mine, v); else *mine = v;
        if (!grid_barrier(counter, (unsigned)(2 * k + 1) * nb, (mode & 4) != 0)) { if (threadIdx.x == 0) atomicExch(err, 1); return; }
        // read the slot of a block on another XCD (block ids round-robin over XCDs: +1 = next XCD) and of a far block
        for (int d : {1, 3, 129}) {
            const u32x4* other = slots + (long)((blockIdx.x + d) % nb) * 256 + threadIdx.x;
            u32x4 r = (mode & 2) ? load_sc1_x4(other) : *reinterpret_cast<const volatile u32x4*>(other);
            if (r[0] != (unsigned)(k + 1)) ++bad;
        }
        if (!grid_barrier(counter, (unsigned)(2 * k + 2) * nb, (mode & 4) != 0)) { if (threadIdx.x == 0) atomicExch(err, 1); return; }
    }
    if (bad) atomicAdd(stale, bad);
}

template <bool SC1>
__global__ __launch_bounds__(256) void read_shared(const u32x4* buf, int n_vec, int reps, unsigned* out) {
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n_vec; i += 256 * 4) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4* p = buf + ((i + u * 256) % n_vec);
                if (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
                else v[u] = *p;
            }
            if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u][0] ^ v[u][3];
        }
    if (acc == 0x12345u) out[0] = acc;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    unsigned* counter; int* err; unsigned* stale; u32x4* slots; u32x4* buf; unsigned* out;
    (void)hipMalloc(&counter, 256); (void)hipMalloc(&err, 4); (void)hipMalloc(&stale, 4); (void)hipMalloc(&slots, 1024L * 256 * 16);
    (void)hipMalloc(&buf, 4 << 20); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 1, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int herr = 0;
    if (only == 0 || only == 1) for (int nb : {64, 128, 256, 512}) for (int fences : {0, 1}) {
        const int K = 2000;
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            (void)hipMemset(counter, 0, 256); (void)hipMemset(err, 0, 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(barrier_bench, dim3(nb), dim3(256), 0, 0, counter, K, fences, err);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("barrier: %3d blocks fences=%d : %7.3f us per barrier%s\n", nb, fences, best * 1e3 / K, herr ? "  (TIMED OUT)" : "");
    }
    if (only == 0 || only == 2) for (int mode = 0; mode < 8; ++mode) {
        const int K = 300;
        unsigned hstale = 0;
        (void)hipMemset(counter, 0, 256); (void)hipMemset(err, 0, 4); (void)hipMemset(stale, 0, 4); (void)hipMemset(slots, 0, 1024L * 256 * 16);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(visibility, dim3(256), dim3(256), 0, 0, counter, slots, K, mode, stale, err);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&hstale, stale, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("visibility: stores %-5s loads %-5s fences %d : stale reads %u of %d  (%.2f us per round)%s\n", (mode & 1) ? "sc1" : "plain",
               (mode & 2) ? "sc1" : "plain", (mode >> 2) & 1, hstale, K * 256 * 256 * 3, ms * 1e3 / K, herr ? "  (TIMED OUT)" : "");
    }
    if (only == 0 || only == 3) for (int sc1 = 0; sc1 < 2; ++sc1) {
        const int n_vec = (256 << 10) / 16, reps = 50;
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0, 0);
            if (sc1) hipLaunchKernelGGL((read_shared<true>), dim3(256), dim3(256), 0, 0, buf, n_vec, reps, out);
            else hipLaunchKernelGGL((read_shared<false>), dim3(256), dim3(256), 0, 0, buf, n_vec, reps, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("256 blocks each reading the same 256 KB x %d, %s loads: %.1f us per pass, %.1f GB/s per block\n", reps, sc1 ? "sc1" : "plain",
               best * 1e3 / reps, 262144.0 * reps / (best * 1e-3) / 1e9);
    }
    return 0;
}
