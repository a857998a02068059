/* C ABI of the EXPERIMENTS build of the library (nvp_amd/csrc/build.sh: libnvp_hip_experiments.so, -DNVP_EXPERIMENTS=1): entry points
 * of designs that were built, verified bit-identical and measured NOT faster than the product path.  The product library
 * (libnvp_hip.so, include/nvp_hip.h) exports none of them. */
#ifndef NVP_HIP_EXPERIMENTS_H
#define NVP_HIP_EXPERIMENTS_H
#include "nvp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- R11 + R13 + R12 (chain) in ONE launch: the tile-fused training step ---------------------------------------------------------------
 * nvp_encode_mlp_fwd (forward incl. the grid lookups, saved streams, latent), image_mse's gradient d/d rgb = 2 (rgb - gt) / (3 N) on
 * the normalised ground truth (training.py:47-48, loss_functions.py:1-3: per pixel, so no grid-wide reduction is needed) and
 * nvp_mlp_bwd_dx (backward chain: dY streams, tile records, latent gradient, optionally level-major via `lm`) for the same 32-pixel
 * tile back to back in the same wave: the backward chain reads the streams its wave has just written.  Outputs are bit-identical to
 * the three separate calls nvp_encode_mlp_fwd -> nvp_mse_u8 -> nvp_mlp_bwd_dx on the same buffers; rgb is still written (the caller
 * computes the loss VALUE from it).  gt_u8 [N,3] uint8.  NVP_ERR_UNSUPPORTED (nothing enqueued) outside config_nvp_s-sized latents
 * (nvp_encode_mlp_fwd_bwd_supported tells).
 * Measured (configs[1], round 4): 4.14-4.18 ms against 2.09 + 1.75 ms for the two separate kernels on the same box - the backward chain does
 * run 16 % faster on cache-hot inputs, but 168 KB of straight-line code per tile, the extra spills and the lock step the backward ring
 * imposes on the forward halves cost more (profiles/r04_ab_tile_fused.txt). */
int32_t nvp_encode_mlp_fwd_bwd_supported(const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh);
int nvp_encode_mlp_fwd_bwd(const float* coords, const float* steps, const uint8_t* gt_u8, const float* kf_xy, const float* kf_yt, const float* kf_xt,
                           const float* emb, const nvp_mlp_params* p, const float* packed_fwd, const float* packed_bwd, float* rgb, float* saved,
                           float* zt, float* dy, float* dz_rows, const nvp_scatter_lm* lm /* NULL or nvp_encode_bwd_prepare's output */, int64_t n,
                           const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh, void* stream);

#ifdef __cplusplus
}
#endif
#endif
