// COCO RLE of the binarised masks on the device (round 5): the end state of the reference's eval loop (eval/evaluation_scripts/utils.py:262-265
//     mask = (...).sigmoid() > 0.5;  rle = cocomask.encode(np.asfortranarray(mask));  rle['counts'].decode()
// ) without the masks crossing PCIe.  pycocotools is a third-party dependency absent from the container; its published
// algorithm (cocoapi common/maskApi.c rleEncode + rleToString) is what padt_amd/postprocess.py restates on the host and what this file does
// per object in one thread block:
//   1. run lengths of the COLUMN-MAJOR pixel stream (i = x * h + y), first run = zeros (length 0 when the first pixel is set): the image is
//      walked in strips of SW columns staged through LDS (row-major coalesced reads, column-major walk out of LDS); a transition at stream
//      position i closes run k = (transitions before i) of length i - (previous transition) — one block-wide sum scan (run index) and one
//      max scan (previous transition) per chunk of 4096 pixels;
//   2. rleToString: count j (from the fourth on: minus count j - 2) as little-endian 5-bit groups, bit 5 = "more follows", + 48; the byte
//      offset of a count is a prefix sum of its group number.
// Integer work: results are byte-identical to the host statement (tests/test_kernels_gpu.py, tests/golden/postprocess.npz).
#include "common.h"

extern "C" void padt_set_error(const char* msg);

namespace {
constexpr int NT = 1024, NW = NT / 64, ITEMS = 4;

struct RleArgs {
    const unsigned char* mask; long ld_obj, ld_row;
    const int* dst_h; const int* dst_w;
    int* counts; long cap_counts;
    unsigned char* str; long cap_str;
    int* n_counts; int* str_len;
    int sw;                        // strip width (columns staged per pass); LDS pitch = sw + 4
    int max_h; long max_w;         // the LDS strip holds max_h rows, a mask row ld_row bytes: an object beyond either is reported, not clamped
};

__device__ __forceinline__ int wave_incl_sum(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v = max(v, t); }
    return v;
}
// exclusive block scans over one value per thread; `total` = reduction over the block.  sm: NW + 1 ints, free again on return.
__device__ __forceinline__ int block_excl_sum(int v, int* sm, int tid, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    const int incl = wave_incl_sum(v, lane);
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int w = lane < NW ? sm[lane] : 0;
        const int wi = wave_incl_sum(w, lane);
        if (lane < NW) sm[lane] = wi - w;
        if (lane == NW - 1) sm[NW] = wi;
    }
    __syncthreads();
    const int r = incl - v + sm[wave];
    total = sm[NW];
    __syncthreads();
    return r;
}
__device__ __forceinline__ int block_excl_max(int v, int* sm, int tid, int& total) {      // identity -1
    const int lane = tid & 63, wave = tid >> 6;
    const int incl = wave_incl_max(v, lane);
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int w = lane < NW ? sm[lane] : -1;
        const int wi = wave_incl_max(w, lane);
        int ex = __shfl_up(wi, 1, 64);
        if (lane == 0) ex = -1;
        if (lane < NW) sm[lane] = ex;
        if (lane == NW - 1) sm[NW] = wi;
    }
    __syncthreads();
    int ex = __shfl_up(incl, 1, 64);
    if (lane == 0) ex = -1;
    const int r = max(ex, sm[wave]);
    total = sm[NW];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(NT) void mask_rle_kernel(RleArgs p) {
    extern __shared__ unsigned char tile[];                       // [h][sw + 4]
    __shared__ int sm[NW + 1];
    __shared__ int s_carry;
    const int o = blockIdx.x, tid = threadIdx.x;
    const int h = p.dst_h[o], w = p.dst_w[o];
    const long n = (long)h * w;
    int* cnt = p.counts + (long)o * p.cap_counts;
    unsigned char* str = p.str + (long)o * p.cap_str;
    if (n <= 0) {                                                 // rleEncode of an empty mask: no counts, empty string
        if (tid == 0) { p.n_counts[o] = 0; p.str_len[o] = 0; }
        return;
    }
    if (h > p.max_h || w > p.max_w) {                             // taller than the LDS strip / wider than a mask row: failed object, never an overrun
        if (tid == 0) { p.n_counts[o] = -1; p.str_len[o] = -1; }
        return;
    }
    const unsigned char* img = p.mask + (long)o * p.ld_obj;
    const int SW = p.sw, pitch = SW + 4;
    int R = 0;                                                    // transitions so far = index of the run being counted
    int last = 0;                                                 // stream position of the last transition (start of the current run)
    int carry = 0;                                                // value in front of the stream: a leading 1 is a transition at i = 0 (zero-length first run)
    for (int x0 = 0; x0 < w; x0 += SW) {
        const int sw = min(SW, w - x0);
        for (int e = tid; e < h * sw; e += NT) {                  // stage the strip: row-major, consecutive threads = consecutive bytes of a row
            const int y = e / sw, xx = e - y * sw;
            tile[y * pitch + xx] = img[(long)y * p.ld_row + x0 + xx] ? 1 : 0;
        }
        __syncthreads();
        const int m = sw * h;
        for (int base = 0; base < m; base += NT * ITEMS) {
            const int j0 = base + tid * ITEMS;
            int pos[ITEMS];
            int nf = 0, loc_last = -1;
            int pv = 0;
            if (j0 < m) {
                if (j0 == 0) pv = carry;
                else { const int jp = j0 - 1, xp = jp / h; pv = tile[(jp - xp * h) * pitch + xp]; }
            }
            int xx = j0 / h, y = j0 - xx * h;
#pragma unroll
            for (int e = 0; e < ITEMS; ++e) {
                pos[e] = -1;
                if (j0 + e < m) {
                    const int v = tile[y * pitch + xx];
                    if (v != pv) { pos[e] = x0 * h + j0 + e; ++nf; loc_last = pos[e]; }
                    pv = v;
                    if (++y == h) { y = 0; ++xx; }
                }
            }
            int tot_f, tot_m;
            const int r0 = block_excl_sum(nf, sm, tid, tot_f);
            int prev = block_excl_max(loc_last, sm, tid, tot_m);
            if (prev < 0) prev = last;
            int k = R + r0;
#pragma unroll
            for (int e = 0; e < ITEMS; ++e)
                if (pos[e] >= 0) {
                    if (k < p.cap_counts) cnt[k] = pos[e] - prev;
                    prev = pos[e];
                    ++k;
                }
            R += tot_f;
            if (tot_m >= 0) last = tot_m;
        }
        if (tid == 0) s_carry = tile[(h - 1) * pitch + sw - 1];
        __syncthreads();
        carry = s_carry;
        __syncthreads();
    }
    if (tid == 0 && R < p.cap_counts) cnt[R] = (int)(n - last);
    const int M = R + 1;
    if (M > p.cap_counts) {
        if (tid == 0) { p.n_counts[o] = -M; p.str_len[o] = -1; }
        return;
    }
    __syncthreads();                                              // counts of this block are visible to the whole block
    // ---- rleToString
    long S = 0;
    for (int base = 0; base < M; base += NT) {
        const int j = base + tid;
        int g = 0;
        long x = 0;
        if (j < M) {
            x = cnt[j];
            if (j > 2) x -= cnt[j - 2];
            long t = x;
            bool more = true;
            while (more) {
                const int ch = (int)(t & 0x1f);
                t >>= 5;
                more = (ch & 0x10) ? (t != -1) : (t != 0);
                ++g;
            }
        }
        int tot;
        const int off = block_excl_sum(g, sm, tid, tot);
        if (j < M) {
            long t = x;
            for (int q = 0; q < g; ++q) {
                int ch = (int)(t & 0x1f);
                t >>= 5;
                if (q + 1 < g) ch |= 0x20;
                const long at = S + off + q;
                if (at < p.cap_str) str[at] = (unsigned char)(ch + 48);
            }
        }
        S += tot;
    }
    if (tid == 0) { p.n_counts[o] = M; p.str_len[o] = S <= p.cap_str ? (int)S : -1; }
}

// strings of all objects back to back behind an offset table: ONE device-to-host copy of offsets[n_obj] bytes after the table was read
__global__ __launch_bounds__(NT) void rle_pack_kernel(const unsigned char* __restrict__ str, long cap_str, const int* __restrict__ str_len, int n_obj,
                                                      unsigned char* __restrict__ packed, long cap_packed, int* __restrict__ offsets) {
    long off = 0;
    bool bad = false;
    for (int o = 0; o < n_obj; ++o) {
        const int len = str_len[o];
        if (len < 0 || off + len > cap_packed) { bad = true; break; }
        for (int i = threadIdx.x; i < len; i += NT) packed[off + i] = str[(long)o * cap_str + i];
        if (threadIdx.x == 0) offsets[o] = (int)off;
        off += len;
    }
    if (threadIdx.x == 0) offsets[n_obj] = bad ? -1 : (int)off;
}
}  // namespace

static long rle_lds_bytes(int max_h, int* strip_w) {
    int sw = 64;
    while (sw > 4 && (long)(sw + 4) * max_h > 60 * 1024) sw >>= 1;
    if ((long)(sw + 4) * max_h > 60 * 1024) return -1;
    if (strip_w) *strip_w = sw;
    return (long)(sw + 4) * max_h;
}

extern "C" int padt_mask_rle(void* stream, const void* mask_u8, long ld_obj, long ld_row, const int* dst_h, const int* dst_w, int n_obj, int max_h,
                             int* counts, long cap_counts, void* str_u8, long cap_str, int* n_counts, int* str_len, void* packed_u8, long cap_packed,
                             int* offsets) {
    if (n_obj <= 0) return 0;
    int sw = 0;
    const long lds = rle_lds_bytes(max_h, &sw);
    if (lds < 0 || mask_u8 == nullptr || counts == nullptr || str_u8 == nullptr || n_counts == nullptr || str_len == nullptr || cap_counts < 2 || cap_str < 1 ||
        (packed_u8 != nullptr && offsets == nullptr)) {
        padt_set_error("padt_mask_rle: max_h <= 7680, counts / string buffers with their capacities and (with packed) the offset table required");
        return -1;
    }
    static PerDeviceOnce once;                                    // per device, safe against two threads making the first call (common.h)
    once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_rle_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 61 * 1024); });
    RleArgs a{(const unsigned char*)mask_u8, ld_obj, ld_row, dst_h, dst_w, counts, cap_counts, (unsigned char*)str_u8, cap_str, n_counts, str_len, sw, max_h, ld_row};
    hipLaunchKernelGGL(mask_rle_kernel, dim3(n_obj), dim3(NT), (size_t)lds, (hipStream_t)stream, a);
    if (packed_u8 != nullptr)
        hipLaunchKernelGGL(rle_pack_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, (const unsigned char*)str_u8, cap_str, str_len, n_obj,
                           (unsigned char*)packed_u8, cap_packed, offsets);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}
