// Device-side counterparts of the reference's training-loop glue:
// the batch gather of dataio.py:104-120 with the video resident in HBM, and image_mse
// (loss_functions.py:1-3 after the (x-127.5)/127.5 normalisation of training.py:47-48)
// fused with its own gradient.
#include "grid_math.h"

// (__fmul_rn / __fadd_rn are plain operators in this HIP and fuse after inlining.)  Also forbid FMA contraction for the whole TU so
// index and interpolation arithmetic keeps the reference's separately rounded multiply and add.
#pragma clang fp contract(off)

namespace {

// coords = (tcoord_tab[ti], row/(H-1), col/(W-1)) with pi = row*W + col   (dataio.py:11-20,106-118)
__global__ __launch_bounds__(256) void sample_gather_kernel(const uint8_t* __restrict__ video, const int64_t* __restrict__ ti,
                                                            const int64_t* __restrict__ pi, const int64_t* __restrict__ order,
                                                            const float* __restrict__ tcoord_tab,
                                                            const float* __restrict__ tstep_tab, float* __restrict__ coords,
                                                            float* __restrict__ steps, uint8_t* __restrict__ gt,
                                                            int64_t n, int height, int width) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int64_t src_k = order ? order[k] : k;          // optional delivery order (e.g. ascending image column)
    const int64_t t = ti[src_k], p = pi[src_k];
    const int row = (int)(p / width), col = (int)(p - (int64_t)row * width);
    coords[k * 3 + 0] = tcoord_tab[t];
    coords[k * 3 + 1] = nvp_div_rn((float)row, (float)(height - 1));
    coords[k * 3 + 2] = nvp_div_rn((float)col, (float)(width - 1));
    steps[k] = tstep_tab[t];
    const uint8_t* src = video + (t * (int64_t)height * width + p) * 3;
    gt[k * 3 + 0] = src[0];
    gt[k * 3 + 1] = src[1];
    gt[k * 3 + 2] = src[2];
}

// ---- delivery order of a batch: ascending image column, stable -----------------------------------------------------------------
// order[k] = index of the sample that comes k-th when the batch is sorted by (pi % width), ties in drawing order - exactly what a
// stable sort of the column keys gives (and what torch.argsort's radix sort delivered), as ONE counting sort on a key of
// log2(width) bits: per-chunk histograms (key-major), one exclusive scan over (key, chunk), and a scatter in which one wave walks
// its chunk in drawing order (rank among equal keys inside a 64-sample group from ballots, running per-key cursors in LDS).
// Three small launches instead of the vendor sort's six (two radix passes, histogram, index initialisation, copies): the sampler
// runs on a side stream underneath the forward kernel and every microsecond of it is taken from that kernel's workgroup slots.
constexpr int kSortChunk = 4096;                 // samples per chunk (one histogram / scatter workgroup each)
constexpr int kSortScanTile = 4096;              // scan: entries per workgroup (256 threads x 16)

// key of sample i: its image column (pi % width), or - `keys` given - a precomputed key in [0, width)
__device__ __forceinline__ unsigned sort_key(const int64_t* __restrict__ pi, const unsigned* __restrict__ keys, int64_t i, int width) {
    return keys ? min(keys[i], (unsigned)(width - 1)) : (unsigned)(pi[i] % width);
}

__global__ __launch_bounds__(256) void sort_hist_kernel(const int64_t* __restrict__ pi, const unsigned* __restrict__ keys, unsigned* __restrict__ cnt, int64_t n, int width, int nchunks) {
    extern __shared__ unsigned hist[];
    for (int k = threadIdx.x; k < width; k += 256) hist[k] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortChunk;
    for (int j = 0; j < kSortChunk / 256; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&hist[sort_key(pi, keys, i, width)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < width; k += 256) cnt[(int64_t)k * nchunks + blockIdx.x] = hist[k];
}

__global__ __launch_bounds__(256) void sort_tilesum_kernel(const unsigned* __restrict__ cnt, unsigned* __restrict__ tsum, int64_t total) {
    const int64_t k0 = (int64_t)blockIdx.x * kSortScanTile + threadIdx.x * 16;
    unsigned sum = 0u;
    for (int e = 0; e < 16; ++e) if (k0 + e < total) sum += cnt[k0 + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// in place: cnt[i] <- number of samples in (key, chunk) cells before cell i
__global__ __launch_bounds__(256) void sort_scan_kernel(unsigned* __restrict__ cnt, const unsigned* __restrict__ tsum, int64_t total) {
    __shared__ unsigned part[256];
    __shared__ unsigned tile_base;
    unsigned b = 0u;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) b += tsum[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o);
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) tile_base = red[0] + red[1] + red[2] + red[3];
    const int64_t k0 = (int64_t)blockIdx.x * kSortScanTile + threadIdx.x * 16;
    unsigned v[16], sum = 0u;
    for (int e = 0; e < 16; ++e) { v[e] = k0 + e < total ? cnt[k0 + e] : 0u; sum += v[e]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {                       // 256 partial sums: a serial scan is a microsecond
        unsigned run = tile_base;
        for (int t = 0; t < 256; ++t) { const unsigned x = part[t]; part[t] = run; run += x; }
    }
    __syncthreads();
    unsigned run = part[threadIdx.x];
    for (int e = 0; e < 16; ++e) if (k0 + e < total) { cnt[k0 + e] = run; run += v[e]; }
}

// one wave per chunk, in drawing order
__global__ __launch_bounds__(64) void sort_scatter_kernel(const int64_t* __restrict__ pi, const unsigned* __restrict__ keys, const unsigned* __restrict__ cnt, int64_t* __restrict__ order,
                                                          int64_t n, int width, int nchunks, int key_bits) {
    extern __shared__ unsigned cursor[];
    const int lane = threadIdx.x;
    for (int k = lane; k < width; k += 64) cursor[k] = cnt[(int64_t)k * nchunks + blockIdx.x];
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * kSortChunk;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int g = 0; g < kSortChunk / 64; ++g) {
        const int64_t i = base + g * 64 + lane;
        const bool ok = i < n;                                       // wave-uniform except in the batch's last group
        const unsigned key = ok ? sort_key(pi, keys, i, width) : 0xffffffffu;
        unsigned long long same = __ballot(ok);
        for (int b = 0; b < key_bits; ++b) {
            const unsigned long long m = __ballot((key >> b) & 1u);
            same &= ((key >> b) & 1u) ? m : ~m;
        }
        if (ok) {
            const unsigned rank = (unsigned)__popcll(same & lt);
            const unsigned pos = cursor[key] + rank;
            order[pos] = i;
            // the lane that comes last among its equals moves the key's cursor past all of them (every lane of the group has read it)
            if ((same >> lane) == 1ull) cursor[key] = pos + 1u;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// key(y) of every sample: sum over the levels of the (clamped) grid row index, as encode_bwd.hip's keys_kernel computes it
__global__ __launch_bounds__(256) void row_keys_kernel(const float* __restrict__ coords, unsigned* __restrict__ keys, nvp_levels lv0, nvp_levels lv1, int both, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float y = coords[i * 3 + 2];
    unsigned key = 0u;
    for (int l = 0; l < lv0.n_levels; ++l) key += (unsigned)min(max((int)floorf(nvp_grid_pos(y, lv0.scale[l], lv0.flags)), 0), lv0.res[l] + 1);
    if (both)
        for (int l = 0; l < lv1.n_levels; ++l) key += (unsigned)min(max((int)floorf(nvp_grid_pos(y, lv1.scale[l], lv1.flags)), 0), lv1.res[l] + 1);
    keys[i] = key;
}

__global__ __launch_bounds__(1024) void mse_u8_kernel(const float* __restrict__ rgb, const uint8_t* __restrict__ gt,
                                                     float* __restrict__ drgb, float* __restrict__ loss_sum,
                                                     int64_t n3, float gscale) {
    float local = 0.f;
    // four elements per thread and trip where the pointers allow it (16-B rgb / drgb, 4-B gt): a scalar grid-stride loop of seven
    // dependent trips made this 34-MB pass take 33 us; the per-element arithmetic (and so every gradient bit) is unchanged
    const bool vec = ((reinterpret_cast<uintptr_t>(rgb) | reinterpret_cast<uintptr_t>(drgb)) & 15) == 0 && (reinterpret_cast<uintptr_t>(gt) & 3) == 0;
    const int64_t n4 = vec ? (n3 >> 2) : 0;
    const int64_t tid = (int64_t)blockIdx.x * 1024 + threadIdx.x, nthr = (int64_t)gridDim.x * 1024;
    for (int64_t q = tid; q < n4; q += nthr) {
        const float4 r = reinterpret_cast<const float4*>(rgb)[q];
        const uchar4 u = reinterpret_cast<const uchar4*>(gt)[q];
        const float rv[4] = {r.x, r.y, r.z, r.w};
        const float uv[4] = {(float)u.x, (float)u.y, (float)u.z, (float)u.w};
        float dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g = nvp_div_rn(nvp_sub_rn(uv[e], 127.5f), 127.5f);
            const float df = rv[e] - g;
            local = __fmaf_rn(df, df, local);
            dv[e] = df * gscale;
        }
        if (drgb) reinterpret_cast<float4*>(drgb)[q] = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
    for (int64_t k = 4 * n4 + tid; k < n3; k += nthr) {
        const float g = nvp_div_rn(nvp_sub_rn((float)gt[k], 127.5f), 127.5f);
        const float df = rgb[k] - g;
        local = __fmaf_rn(df, df, local);
        if (drgb) drgb[k] = df * gscale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {                      // one atomic per 1024-thread block: 2048 same-address atomics cost ~25 us at the kernel's end
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        nvp_atomic_add(loss_sum, t);
    }
}

}  // namespace

extern "C" {

int nvp_sample_gather(const uint8_t* video, const int64_t* ti, const int64_t* pi, const int64_t* order,
                      const float* tcoord_tab, const float* tstep_tab,
                      float* coords, float* steps, uint8_t* gt_u8,
                      int64_t n, int32_t t_frames, int32_t height, int32_t width, void* stream) {
    if (!video || !ti || !pi || !tcoord_tab || !tstep_tab || !coords || !steps || !gt_u8 || n < 0 || t_frames < 1 || height < 2 || width < 2)
        return NVP_ERR_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sample_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       video, ti, pi, order, tcoord_tab, tstep_tab, coords, steps, gt_u8, n, height, width);
    NVP_LAUNCH_CHECK();
    return 0;
}

int64_t nvp_sample_order_workspace_bytes(int64_t n, int32_t width) {
    if (n < 0 || width < 1) return NVP_ERR_BADARG;
    const int64_t nchunks = (n + kSortChunk - 1) / kSortChunk, total = nchunks * width;
    return (total + (total + kSortScanTile - 1) / kSortScanTile + 64) * 4;
}

// stable counting sort of n samples by key_of(i) in [0, width): order[k] = index of the k-th sample, ties in input order
static int stable_order(const int64_t* pi, const unsigned* keys, int64_t* order, int64_t n, int32_t width, void* workspace, hipStream_t stream) {
    const int nchunks = (int)((n + kSortChunk - 1) / kSortChunk);
    const int64_t total = (int64_t)nchunks * width;
    const unsigned ntiles = (unsigned)((total + kSortScanTile - 1) / kSortScanTile);
    unsigned* cnt = (unsigned*)workspace;
    unsigned* tsum = cnt + total;
    int key_bits = 1;
    while ((1 << key_bits) < width) ++key_bits;
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nchunks), dim3(256), (size_t)width * 4, stream, pi, keys, cnt, n, width, nchunks);
    hipLaunchKernelGGL(sort_tilesum_kernel, dim3(ntiles), dim3(256), 0, stream, (const unsigned*)cnt, tsum, total);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(ntiles), dim3(256), 0, stream, cnt, (const unsigned*)tsum, total);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nchunks), dim3(64), (size_t)width * 4, stream, pi, keys, (const unsigned*)cnt, order, n, width, nchunks, key_bits);
    NVP_LAUNCH_CHECK();
    return 0;
}

// order[k] (int64, as torch.argsort returns) = index of the k-th sample in ascending image column (pi % width), ties in drawing order
int nvp_sample_order_by_column(const int64_t* pi, int64_t* order, int64_t n, int32_t width, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!pi || !order || !workspace || n < 0 || width < 1) return NVP_ERR_BADARG;
    if (width > 12288 || n >= ((int64_t)1 << 31)) return NVP_ERR_UNSUPPORTED;          // the per-chunk table lives in LDS
    if (workspace_bytes < nvp_sample_order_workspace_bytes(n, width)) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    return stable_order(pi, nullptr, order, n, width, workspace, (hipStream_t)stream);
}

// ---- row order of an arbitrary batch (a drop-in caller's: the reference sampler's raw order, dataio.py:104-120) ------------------
// What the gradient scatter needs from a "y-sorted" batch is that the grid ROW index of the xy and of the yt plane (both indexed
// by the y coordinate, modules.py:61,63) is non-decreasing at EVERY level - not that y itself is.  key(y) = sum over the levels of
// row_l(y) is a non-decreasing step function of y that steps exactly where some level's row steps (the scatter's own keys_kernel,
// encode_bwd.hip, uses the same key): 13-14 bits, ONE stable counting sort instead of a four-pass float radix sort.  When both
// planes have the same level geometry (the reference's configs) the key is taken over one of them.
static bool same_levels(const nvp_levels* a, const nvp_levels* b) {
    if (a->n_levels != b->n_levels || a->flags != b->flags) return false;
    for (int l = 0; l < a->n_levels; ++l) if (a->scale[l] != b->scale[l] || a->res[l] != b->res[l]) return false;
    return true;
}
static int row_key_count(const nvp_levels* lv_xy, const nvp_levels* lv_yt) {
    int64_t k = 1;
    for (int l = 0; l < lv_xy->n_levels; ++l) k += lv_xy->res[l] + 1;
    if (!same_levels(lv_xy, lv_yt)) for (int l = 0; l < lv_yt->n_levels; ++l) k += lv_yt->res[l] + 1;
    return k > (1 << 30) ? (1 << 30) : (int)k;
}

int64_t nvp_order_by_rows_workspace_bytes(int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt) {
    if (n < 0 || !lv_xy || !lv_yt) return NVP_ERR_BADARG;
    const int nk = row_key_count(lv_xy, lv_yt);
    if (nk > 12288 || n >= ((int64_t)1 << 31)) return NVP_ERR_UNSUPPORTED;             // what nvp_order_by_rows refuses: say so BEFORE the caller allocates nchunks * nk * 4 bytes
    return nvp_sample_order_workspace_bytes(n, nk) + ((n + 63) / 64) * 256;             // counting-sort tables + the keys
}

int nvp_order_by_rows(const float* coords, int64_t* order, int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt,
                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (!coords || !order || !workspace || !lv_xy || !lv_yt || n < 0) return NVP_ERR_BADARG;
    if (lv_xy->n_levels < 1 || lv_xy->n_levels > NVP_MAX_LEVELS || lv_yt->n_levels < 1 || lv_yt->n_levels > NVP_MAX_LEVELS) return NVP_ERR_BADARG;
    const int nk = row_key_count(lv_xy, lv_yt);
    if (nk > 12288 || n >= ((int64_t)1 << 31)) return NVP_ERR_UNSUPPORTED;
    if (workspace_bytes < nvp_order_by_rows_workspace_bytes(n, lv_xy, lv_yt)) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    unsigned* keys = (unsigned*)((char*)workspace + nvp_sample_order_workspace_bytes(n, nk));
    hipLaunchKernelGGL(row_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords, keys, *lv_xy, *lv_yt,
                       same_levels(lv_xy, lv_yt) ? 0 : 1, n);
    return stable_order(nullptr, keys, order, n, nk, workspace, (hipStream_t)stream);
}

int nvp_mse_u8(const float* rgb, const uint8_t* gt_u8, float* drgb, float* loss_sum, int64_t n, void* stream) {
    if (!rgb || !gt_u8 || !loss_sum || n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    const int64_t n3 = n * 3;
    int64_t blocks = (n3 / 4 + 1023) / 1024;
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(mse_u8_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream,
                       rgb, gt_u8, drgb, loss_sum, n3, 2.0f / (float)n3);
    NVP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
