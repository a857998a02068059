// Device-side counterparts of the reference's training-loop glue:
// the batch gather of dataio.py:104-120 with the video resident in HBM, and image_mse
// (loss_functions.py:1-3 after the (x-127.5)/127.5 normalisation of training.py:47-48)
// fused with its own gradient.
#include "nvp_common.h"

// __fmul_rn/__fadd_rn are plain operators in this HIP: forbid FMA contraction for the whole TU so
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
    coords[k * 3 + 1] = __fdiv_rn((float)row, (float)(height - 1));
    coords[k * 3 + 2] = __fdiv_rn((float)col, (float)(width - 1));
    steps[k] = tstep_tab[t];
    const uint8_t* src = video + (t * (int64_t)height * width + p) * 3;
    gt[k * 3 + 0] = src[0];
    gt[k * 3 + 1] = src[1];
    gt[k * 3 + 2] = src[2];
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
            const float g = __fdiv_rn(__fsub_rn(uv[e], 127.5f), 127.5f);
            const float df = rv[e] - g;
            local = __fmaf_rn(df, df, local);
            dv[e] = df * gscale;
        }
        if (drgb) reinterpret_cast<float4*>(drgb)[q] = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
    for (int64_t k = 4 * n4 + tid; k < n3; k += nthr) {
        const float g = __fdiv_rn(__fsub_rn((float)gt[k], 127.5f), 127.5f);
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
