// Forward of the modulator MLP + modulated SIREN (R8-R10) on **split-operand 16-bit MFMA** (mlp_b3.h: fp16 x 2 scaled split,
// three products - or bf16 x 3, six products): same structure as mlp_fwd.hip (one wave = one 32-pixel tile, pixel on the
// MFMA column axis, a layer's 64 D registers are the next layer's B operand, latent tile in the wave's LDS region, five
// saved streams), but every GEMM runs on v_mfma_f32_32x32x16_{f16,bf16}: weights are split once per step by
// pack_fwd_b3_kernel, activations on the fly from the fp32 registers (fp16 x 2: scaled per PIXEL by a power of two taken
// from the pixel's largest input, the accumulator is un-scaled before the activation), products accumulated in fp32.
// The result carries an error BELOW that of an fp32 fma chain of the same length at 5x (fp16 x 2) / 2.5x (bf16 x 3)
// fewer matrix-pipe cycles.  Used for latents of <= 256 rows when the library is built with NVP_FWD_B3=1 (beyond 144
// rows the rest is read from the tensor); everything else (element-wise stages, saved streams, RGB layout) is identical
// to the fp32 kernel.
#include <cstdlib>
#include "mlp_fwd_b3_tile.h"
#if NVP_EXPERIMENTS
#include "mlp_fwd_b3x2_tile.h"     // two tiles per wave, one wave per SIMD: built, bit-identical, measured SLOWER (profiles/r05_ab_fwd_pair_per_wave.txt)
#endif

namespace {

#ifndef NVP_FWD_WAVES
#define NVP_FWD_WAVES 4        // waves (= 32-pixel tiles) per workgroup; the waves are independent (no barrier, per-wave LDS tiles)
#endif
constexpr int kWaves = NVP_FWD_WAVES;

// two workgroups per CU = two waves per SIMD (the one-wave-per-SIMD build with 512 registers and deeper weight prefetch: 2.19 vs 1.82 ms)
template <bool SAVE, int GF, bool INTER = false>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_fwd_b3_kernel(float* __restrict__ zt, const float* __restrict__ steps,
                                                                    nvp_mlp_params p, const unsigned* __restrict__ packed,
                                                                    float* __restrict__ rgb, float* __restrict__ saved,
                                                                    int64_t n, int64_t ntiles, int d, NvpTileEnc enc) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * kWaves + wv;
    if (tile >= ntiles) return;                       // wave-uniform (the waves are independent: no barrier)
    const bool active = true;
    nvp_stagger_start();
    extern __shared__ __attribute__((aligned(16))) float4 zlds[];
    const NvpFwdLayoutB3 Lz = nvp_fwd_layout_b3(d);
    float4* z = zlds + wv * (min(Lz.zs, kB3ZLdsSteps) * 4 * 32);       // this wave's latent tile
    float rgb_px[3];
    fwd_b3_tile<SAVE, GF, INTER>(zt, steps, p, packed, rgb, saved, n, ntiles, d, enc, tile, active, z, lane, rgb_px);
}

#if NVP_EXPERIMENTS
// ---- EXPERIMENT: two tiles per wave, one wave per SIMD (mlp_fwd_b3x2_tile.h), for the fused gather + forward of
// config_nvp_s-sized latents (8 latent k-steps).  Bit-identical RGB / saved streams / latent; 2.85 ms alone against 1.80 ms for the
// one-tile kernel at two waves per SIMD (profiles/r05_ab_fwd_pair_per_wave.txt: a lone wave per SIMD is parked on s_waitcnt for half of
// its life and issues its ~23 K non-MFMA instructions at ~10 cycles each: 2.14 ms with the MFMAs removed).  Experiments library only, NVP_FWD_X2=1.
constexpr int kWavesX2 = 4;    // one wave per SIMD
constexpr int kZsX2 = 8;       // latent k-steps the pair kernel is compiled for (rows <= 128)

template <bool SAVE, int GF>
__global__ __launch_bounds__(kWavesX2 * 64, 1) void mlp_fwd_b3x2_kernel(float* __restrict__ zt, const float* __restrict__ steps, nvp_mlp_params p,
                                                                        const unsigned* __restrict__ packed, float* __restrict__ rgb, float* __restrict__ saved,
                                                                        int64_t n, int64_t ntiles, int d, NvpTileEnc enc) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t pair = (int64_t)blockIdx.x * kWavesX2 + wv;
    const int64_t tileA = 2 * pair;
    if (tileA >= ntiles) return;                      // wave-uniform
    extern __shared__ __attribute__((aligned(16))) float4 zlds[];
    float4* zA = zlds + (2 * wv) * (kZsX2 * 4 * 32);
    float4* zB = zA + kZsX2 * 4 * 32;
    fwd_b3_pair<SAVE, GF, kZsX2>(zt, steps, p, packed, rgb, saved, n, ntiles, d, enc, tileA, tileA + 1 < ntiles, zA, zB, lane);
}
#endif

}  // namespace

// called by nvp_mlp_fwd (mlp_fwd.hip) when NVP_FWD_B3 is on and the latent has <= 256 rows
int nvp_mlp_fwd_b3_launch(const float* zt, const float* steps, const nvp_mlp_params* p, const float* packed_fwd,
                          float* rgb, float* saved, int64_t n, int32_t d, void* stream) {
    const int64_t ntiles = nvp_ntiles(n);
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    const int zs = nvp_fwd_layout_b3(d).zs;
    const size_t lds = (size_t)kWaves * (zs < kB3ZLdsSteps ? zs : kB3ZLdsSteps) * 4 * 32 * sizeof(float4);      // 64 KB (nvp_s), 72 KB (nvp_l)
    const unsigned* pk = reinterpret_cast<const unsigned*>(packed_fwd);
    NvpTileEnc none;
    memset(&none, 0, sizeof(none));
    float* z = const_cast<float*>(zt);
    if (saved)
        hipLaunchKernelGGL((mlp_fwd_b3_kernel<true, 0>), grid, dim3(kWaves * 64), lds, (hipStream_t)stream, z, steps, *p, pk, rgb, saved, n, ntiles, d, none);
    else
        hipLaunchKernelGGL((mlp_fwd_b3_kernel<false, 0>), grid, dim3(kWaves * 64), lds, (hipStream_t)stream, z, steps, *p, pk, rgb, saved, n, ntiles, d, none);
    NVP_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t nvp_encode_mlp_fwd_supported(const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh) {
    bool wide = false;
    if (!fused_ok(lv_xy, lv_yt, lv_xt, sh, nullptr, &wide)) return 0;
    return wide ? 2 : 1;          // 2: supported, and the latent tensor `zt` is required for inference too (rows beyond the wave's LDS tile are parked there)
}

extern "C" int nvp_encode_mlp_fwd(const float* coords, const float* steps, const float* kf_xy, const float* kf_yt, const float* kf_xt,
                                  const float* emb, const nvp_mlp_params* p, const float* packed_fwd, float* rgb, float* saved, float* zt,
                                  int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                                  const nvp_sparse_shape* sh, int temporal_interp, void* stream) {
    int d = 0;
    bool wide = false;
    if (!fused_ok(lv_xy, lv_yt, lv_xt, sh, &d, &wide)) return NVP_ERR_UNSUPPORTED;
    if (temporal_interp && saved) return NVP_ERR_UNSUPPORTED;          // SparseGrid.forward_inter is an inference path (eval.py --t_interp): no backward exists for it
    if (!coords || !steps || !kf_xy || !kf_yt || !kf_xt || !emb || !p || !packed_fwd || !rgb || n < 0) return NVP_ERR_BADARG;
    if ((saved || wide) && !zt) return NVP_ERR_BADARG;          // training: the latent is an output too (the dW GEMMs read it); wide latents: it is the kernel's workspace
    if (n == 0) return 0;
    NvpTileEnc e;
    e.lv[0] = *lv_xy; e.lv[1] = *lv_yt; e.lv[2] = *lv_xt; e.sh = *sh;
    e.kf[0] = kf_xy; e.kf[1] = kf_yt; e.kf[2] = kf_xt; e.emb = emb; e.coords = coords;
    int col = 0;
    for (int q = 0; q < 3; ++q) { e.col0[q] = col; col += e.lv[q].n_levels * e.lv[q].n_features; }
    e.col0[3] = col;
    e.rows = nvp_rows4(d);
    const int64_t ntiles = nvp_ntiles(n);
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    const int zs_fused = nvp_fwd_layout_b3(d).zs;
    const size_t lds = (size_t)kWaves * (zs_fused < kB3ZLdsSteps ? zs_fused : kB3ZLdsSteps) * 4 * 32 * sizeof(float4);
    const unsigned* pk = reinterpret_cast<const unsigned*>(packed_fwd);
    const int F = lv_xy->n_features;
#if NVP_EXPERIMENTS
    static const bool x2_on = [] { const char* e_ = getenv("NVP_FWD_X2"); return e_ && e_[0] == '1'; }();
    if (x2_on && NVP_SPLIT_H2 && F == 2 && nvp_fwd_layout_b3(d).zs == kZsX2) {
        // pair kernel: 2 tiles per wave, 4 waves per workgroup, one workgroup per CU (128 KiB of LDS: 2 x 16 KiB latent tiles per wave)
        const int64_t npairs = (ntiles + 1) / 2;
        dim3 grid2((unsigned)((npairs + kWavesX2 - 1) / kWavesX2));
        const size_t lds2 = (size_t)kWavesX2 * 2 * kZsX2 * 4 * 32 * sizeof(float4);
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_b3x2_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_b3x2_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                (void)hipGetLastError();              // (HIP on gfx950 accepts > 64 KiB of dynamic LDS without the attribute; a refusal here must not fail the call)
            attr_set = true;
        }
        if (saved) hipLaunchKernelGGL((mlp_fwd_b3x2_kernel<true, 2>), grid2, dim3(kWavesX2 * 64), lds2, (hipStream_t)stream, zt, steps, *p, pk, rgb, saved, n, ntiles, d, e);
        else hipLaunchKernelGGL((mlp_fwd_b3x2_kernel<false, 2>), grid2, dim3(kWavesX2 * 64), lds2, (hipStream_t)stream, zt, steps, *p, pk, rgb, saved, n, ntiles, d, e);
        NVP_LAUNCH_CHECK();
        return 0;
    }
#endif
#define NVP_FUSED_LAUNCH(SV, GF, ...) hipLaunchKernelGGL((mlp_fwd_b3_kernel<SV, GF, ##__VA_ARGS__>), grid, dim3(kWaves * 64), lds, (hipStream_t)stream, zt, steps, *p, pk, rgb, saved, n, ntiles, d, e)
    if (temporal_interp) { if (F == 2) NVP_FUSED_LAUNCH(false, 2, true); else NVP_FUSED_LAUNCH(false, 4, true); }
    else if (saved) { if (F == 2) NVP_FUSED_LAUNCH(true, 2); else NVP_FUSED_LAUNCH(true, 4); }
    else { if (F == 2) NVP_FUSED_LAUNCH(false, 2); else NVP_FUSED_LAUNCH(false, 4); }
#undef NVP_FUSED_LAUNCH
    NVP_LAUNCH_CHECK();
    return 0;
}
