/*
 * nvp_hip.h - C ABI of libnvp_hip.so: the MI355X (gfx950) implementation of NVP's
 * per-coordinate encoding path.
 *
 * The reference has no FFI of its own: its boundary is the Python module surface
 * (modules.NVP, tinycudann.Encoding, sparsegrid.SparseGrid, modulation.*).  The
 * native half of that surface - what the reference gets from the tiny-cuda-nn CUDA
 * extension and from ATen kernels - is replaced by the entry points below.  Each one
 * names the reference call site it stands in for.  nvp_amd/ (Python) binds them with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *    enqueues work on it, never synchronises, never allocates;
 *  - return value: 0 on success, otherwise a hipError_t value (launch/config error) or
 *    NVP_ERR_* (<0) for argument errors; nothing throws;
 *  - "tiles" are groups of 32 consecutive pixels.  Pixel-tile-major tensors use the PTM4
 *    layout [ntiles][rows/4][32 px][4]: rows (= features / hidden units) come in groups of
 *    4 that are contiguous per pixel, the 32 pixels of a tile are contiguous per row-group,
 *    so element (tile t, row r, pixel j) sits at ((t*(rows/4) + r/4)*32 + j)*4 + r%4 and a
 *    lane's natural access is one 16-B piece.  rows is a multiple of 4 (the latent: D rounded
 *    up; activations: 128), ntiles = ceil(N/32); pixels >= N inside the last tile are
 *    zero-filled by the producers and ignored by the consumers.
 *  - `accumulate` outputs (the stand-alone nvp_dense2d_bwd / nvp_sparse3x3_bwd gradients) are
 *    ADDED into; the caller zero-fills them (autograd's dense-grad contract).
 *    The fused nvp_encode_bwd OVERWRITES every gradient element instead (no zero-fill).
 */
#ifndef NVP_HIP_H
#define NVP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVP_MAX_LEVELS 16
#define NVP_HIDDEN 128          /* network.n_neurons; the kernels are specialised for 128 x 3 layers */
#define NVP_TILE 32             /* pixels per tile */

#define NVP_ERR_BADARG (-1)
#define NVP_ERR_UNSUPPORTED (-2)

/* nvp_encode_fwd / nvp_encode_bwd flags */
#define NVP_DZ_PLANES_READY 2      /* nvp_encode_bwd: the xy / yt planes' latent gradients already sit level-major in the workspace and
                                      the sparse columns' max|dz| in its slots (both written by nvp_mlp_bwd_dx through
                                      nvp_encode_bwd_prepare's pointers); needs NVP_COORDS_SORTED_BY_Y */
#define NVP_SCATTER_SPARSE_ONLY 8  /* nvp_encode_bwd (with NVP_DZ_PLANES_READY): only d_emb is produced by this call ... */
#define NVP_SCATTER_PRESORTED 32   /* nvp_encode_bwd: nvp_encode_bwd_presort already ran on this workspace (same n, levels, flags) */
#define NVP_SCATTER_DENSE_ONLY 16  /* ... only the three keyframe gradients.  Two calls (sparse first: 80 % of the gradient bytes) let a
                                      data-parallel host start exchanging the sparse grid's gradient while the dense planes scatter. */
#define NVP_COORDS_SORTED_BY_Y 1   /* caller guarantees coords[:,2] is non-decreasing: the scatter skips the y counting sort (the batch
                                      order IS the xy / yt planes' sorted order); the gather gets row locality from it, and - only in
                                      builds / runs with NVP_ENCODE_LDS=1, off by default because it measured slower - stages the
                                      xy / yt grid rows of a pixel run in LDS */

/* nvp_levels.flags: arithmetic variant of the dense grid (the tiny-cuda-nn fork the reference installs,
 * README.md:30-32, is absent from /root/reference, so the points where published tiny-cuda-nn and a plain
 * restatement differ are switchable; nvp_amd/csrc/grid_math.h).  Default = all of POS_FMA | INTERP_FMA. */
#define NVP_GRID_POS_FMA 1      /* pos = fmaf(scale, x, 0.5f) (upstream pos_fract); else fl(fl(x*scale)+0.5f) */
#define NVP_GRID_INTERP_FMA 2   /* corner blend is an fma chain (upstream); else separately rounded mul + add */
#define NVP_GRID_CLAMP 4        /* corner i+1 clamps to res-1; else the cell index wraps mod res^2 (upstream) */

/* Geometry of one 2D multi-resolution dense grid (one "learnable keyframe" plane).
 * Built on the host exactly as the reference restates it, eval.py:28-35:
 *   a = exp(l*log(per_level_scale))*base - 1 (double); res = ceil(a)+1; offset += res^2.
 * res/offset always follow that formula (the reference pins them).  `scale` is what the kernels multiply
 * with: by default upstream's fp32 `exp2f(l*log2f(per_level_scale))*base - 1` (grid_scale), or `a` rounded
 * once to fp32 (encoding_config "scale_mode": "double").  offset[] counts CELLS (multiply by n_features). */
typedef struct nvp_levels {
    int32_t n_levels;
    int32_t n_features;                   /* F: 1, 2, 4 or 8 */
    float scale[NVP_MAX_LEVELS];
    int32_t res[NVP_MAX_LEVELS];
    int32_t offset[NVP_MAX_LEVELS + 1];
    int32_t flags;                        /* NVP_GRID_* */
} nvp_levels;

/* Shape of the 3D sparse positional-feature grid, embeddings[T][X][Y][F]
 * (reference sparsegrid.py:13). */
typedef struct nvp_sparse_shape {
    int32_t t_res, x_res, y_res, n_features;
} nvp_sparse_shape;

/* The 14 MLP tensors in their natural (state_dict) layouts.
 * mod_w[k]: [128, D] (k=0) / [128, 128+D] (k=1,2)   reference modulation.py:97-107
 * sir_w[0]: [128,1]; sir_w[1,2]: [128,128]; last_w: [3,128]   modulation.py:61-81 */
typedef struct nvp_mlp_params {
    const float* mod_w[3];
    const float* mod_b[3];
    const float* sir_w[3];
    const float* sir_b[3];
    const float* last_w;
    const float* last_b;
} nvp_mlp_params;

/* Same 14 tensors, writable (gradient destinations, natural layouts, OVERWRITTEN). */
typedef struct nvp_mlp_grads {
    float* mod_w[3];
    float* mod_b[3];
    float* sir_w[3];
    float* sir_b[3];
    float* last_w;
    float* last_b;
} nvp_mlp_grads;

/* Library / device info ----------------------------------------------------------- */
const char* nvp_version(void);
/* How this build issues one fp32 product of the MLP GEMMs on the matrix cores: 3 = fp16 x 2 scaled operand split (three
 * v_mfma_f32_32x32x16_f16 products, the default), 6 = bf16 x 3 split (six bf16 products, -DNVP_SPLIT_H2=0), 1 = fp32 MFMA
 * (-DNVP_FWD_B3=0 -DNVP_BWD_B3=0 -DNVP_DW_B3=0).  Results of all three agree within the parity tolerances; only the rate differs. */
int32_t nvp_mlp_mfma_products(void);
/* Number of floats of each workspace, so the host can allocate with torch.empty. */
int64_t nvp_packed_fwd_floats(int32_t latent_dim);
int64_t nvp_packed_bwd_floats(int32_t latent_dim);
int64_t nvp_dw_partial_floats(int32_t latent_dim, int32_t n_chunks);
int64_t nvp_mlp_param_floats(int32_t latent_dim);
int32_t nvp_latent_rows(int32_t latent_dim);   /* rows of a PTM4 latent tensor (D rounded up to a multiple of 4) */
int32_t nvp_dz_stride(int32_t latent_dim);     /* row stride of the row-major latent gradient (D rounded up to 4) */

/* ---- R2/R3: tinycudann.Encoding forward / backward (reference modules.py:65-67) ---
 * x [N,2] row-major, out/dout [N, n_levels*F] row-major, dparams accumulate. */
int nvp_dense2d_fwd(const float* params, const float* x, float* out, int64_t n,
                    const nvp_levels* lv, void* stream);
int nvp_dense2d_bwd(const float* x, const float* dout, float* dparams, int64_t n,
                    const nvp_levels* lv, void* stream);

/* ---- R5/R6/R7: SparseGrid.forward / its autograd / forward_inter
 * (reference sparsegrid.py:23-72, autograd index_put_, :76-156).
 * coords [N,3]=(t,x,y) row-major, out/dout [N,9F] row-major, demb accumulate. */
int nvp_sparse3x3_fwd(const float* emb, const float* coords, float* out, int64_t n,
                      const nvp_sparse_shape* sh, void* stream);
int nvp_sparse3x3_bwd(const float* coords, const float* dout, float* demb, int64_t n,
                      const nvp_sparse_shape* sh, void* stream);
int nvp_sparse3x3_inter_fwd(const float* emb, const float* coords, float* out, int64_t n,
                            const nvp_sparse_shape* sh, void* stream);
/* SparseGrid(upsample=True) (reference sparsegrid.py:26-34): the x2 bilinear pre-upsample of the (x, y) axes of the WHOLE grid that the
 * reference runs on every call as permute -> F.interpolate(scale_factor=2, mode='bilinear') -> permute.  emb [T,X,Y,F] (sh) -> out
 * [T,2X,2Y,F], one pass, no permutes, ATen's formula (align_corners=False); the gather / scatter above then run on `out` with the
 * doubled resolutions.  _bwd is its adjoint as a deterministic gather: dout [T,2X,2Y,F] -> demb [T,X,Y,F] (every element written). */
int nvp_sparse_upsample2x_fwd(const float* emb, float* out, const nvp_sparse_shape* sh, void* stream);
int nvp_sparse_upsample2x_bwd(const float* dout, float* demb, const nvp_sparse_shape* sh, void* stream);

/* ---- R11 (encoding half of NVP.forward, modules.py:57-78), fused ------------------
 * coords [N,3] -> PTM4 latent zt [ntiles][rows/4][32][4], rows = nvp_latent_rows(D),
 * D = sum_p L_p*F_p + 9*F_s, row order xy | yt | xt | sparse (modules.py:69,78).
 * temporal_interp != 0 selects forward_inter for the sparse part (modules.py:72-73).
 * flags: NVP_COORDS_SORTED_BY_Y if coords[:,2] is non-decreasing.  The default gather kernel only profits from the locality; the
 * opt-in LDS-staged variant (environment NVP_ENCODE_LDS=1, nvp_amd/csrc/encode_fwd_lds.hip: each 256-pixel run stages the <= 3
 * grid rows per level it touches with coalesced loads) needs the flag.  Results are bit-identical with and without it. */
int nvp_encode_fwd(const float* coords, const float* kf_xy, const float* kf_yt, const float* kf_xt,
                   const float* emb, float* zt, int64_t n,
                   const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                   const nvp_sparse_shape* sh, int temporal_interp, int32_t flags, void* stream);
/* ---- R11 in ONE launch: coordinates -> RGB (modules.py:57-84; nvp_amd/csrc/mlp_fwd_b3.hip + encode_tile.h) ----------------
 * The grid lookups run INSIDE the forward MLP's waves: each wave gathers the latent tile of its 32 pixels straight into LDS (same
 * arithmetic as nvp_encode_fwd: the latent is bit-identical) and runs the seven layers on it, so the latent is never read back from
 * HBM and the separate gather launch disappears.  `saved` != NULL (training): the five activation streams are saved as by nvp_mlp_fwd
 * AND the latent is written once to `zt` (PTM4, as nvp_encode_fwd would have) because the weight-gradient GEMMs read it; `saved` ==
 * NULL (inference): `zt` may be NULL, nothing but RGB is written.  nvp_encode_mlp_fwd_supported() returns
 *   0  not supported: call nvp_encode_fwd + nvp_mlp_fwd (grids with other than 2 or 4 features per level, a plane that does not
 *      contribute a multiple of 8 latent rows, a non-default arithmetic variant, F = 2 latents of more than 144 rows);
 *   1  supported, the whole latent tile lives in the wave's LDS region (<= 144 rows: config_nvp_s);
 *   2  supported, and `zt` is REQUIRED for inference too: the latent is wider than the wave's LDS tile (config_nvp_l: F = 4, 228 rows)
 *      - rows beyond the first 144 are stored straight into `zt` by the gather and read back from there by the modulator chains
 *      (each wave reads only what it has just written); with `saved` == NULL the rest of `zt` is left unwritten.
 * temporal_interp != 0 (round 6): the sparse grid's rows are SparseGrid.forward_inter's (sparsegrid.py:76-156: the 3x3 patches of the t_lo and
 * t_hi slices blended with the reference's weights, NaN at t == 1 as in the reference; eval.py --t_interp, modules.py:72-73) - INFERENCE
 * only: with `saved` != NULL the call returns NVP_ERR_UNSUPPORTED (the reference never differentiates through forward_inter either).
 * packed_fwd: nvp_mlp_pack_fwd's output. */
int32_t nvp_encode_mlp_fwd_supported(const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh);
int nvp_encode_mlp_fwd(const float* coords, const float* steps, const float* kf_xy, const float* kf_yt, const float* kf_xt,
                       const float* emb, const nvp_mlp_params* p, const float* packed_fwd, float* rgb, float* saved, float* zt,
                       int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                       const nvp_sparse_shape* sh, int temporal_interp, void* stream);
/* R3 + R6 fused: latent gradient -> gradients of the four grids (nvp_amd/csrc/encode_bwd.hip).
 * dz: ROW-MAJOR [>= N][dz_stride] latent gradient as written by nvp_mlp_bwd_dx (columns
 * xy | yt | xt | sparse).  d_kf_* and d_emb: every element is OVERWRITTEN (deterministic
 * sorted-band fixed-point accumulation: no atomics, no zero-fill needed, bit-reproducible).
 * workspace: device scratch of nvp_encode_bwd_workspace_bytes() bytes.
 * flags: NVP_COORDS_SORTED_BY_Y if the batch is already ordered by its y coordinate (the loss is
 * permutation-invariant, so a sampler may deliver its batch that way; results are identical). */
int64_t nvp_encode_bwd_workspace_bytes(int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt,
                                       const nvp_levels* lv_xt, const nvp_sparse_shape* sh);
/* Hand-over of the latent gradient in the scatter's own layout (y-sorted batches only).  For a batch sorted by y the scatter's
 * sorted order of the xy and yt planes IS the batch order, so its permute pass would merely transpose those planes' columns of
 * dz into level-major [level][pixel][F] arrays.  nvp_encode_bwd_prepare zeroes the max|dz| slots on `stream` and returns device
 * pointers into `workspace`; pass them to nvp_mlp_bwd_dx (which then writes those two planes there instead of into dz_rows) and
 * set NVP_DZ_PLANES_READY | NVP_COORDS_SORTED_BY_Y for nvp_encode_bwd on the SAME workspace.  Gradients are bit-identical. */
typedef struct nvp_scatter_lm {
    float* dzs[2];          /* xy, yt: [n_levels][n][F] */
    uint32_t* dzmax;        /* 256 slots, bit patterns of max|dz| over the two planes */
    uint32_t* sdzmax;       /* 256 slots, the same for the sparse grid's columns [scol0, scol0 + scols) of the latent gradient */
    int32_t scol0, scols;
} nvp_scatter_lm;
int nvp_encode_bwd_prepare(int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                           const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, nvp_scatter_lm* out, void* stream);
/* Optional early start: everything the scatter derives from the coordinates alone (sort keys, the counting sorts that give the
 * planes' sorted orders, every plane's per-level row tables, the sparse grid's order and row table: a dozen small latency-bound
 * kernels) on `stream` - typically a side stream, underneath the gather kernel.  Order the streams with an event, then call nvp_encode_bwd with NVP_SCATTER_PRESORTED on the same
 * workspace; `flags` as for nvp_encode_bwd.  Bit-identical gradients. */
int nvp_encode_bwd_presort(const float* coords, int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                           const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags, void* stream);
/* The sparse grid's half of the split scatter (NVP_SCATTER_SPARSE_ONLY semantics; flags must carry NVP_COORDS_SORTED_BY_Y |
 * NVP_DZ_PLANES_READY) with the OPTIMIZER STEP applied in the flush: the gradient of a sparse-grid element goes from the kernel's
 * LDS table straight into the AdamW update of (emb, exp_avg, exp_avg_sq) - same arithmetic, same bits as nvp_adamw_step on the
 * gradient nvp_encode_bwd would have written (reference: sparsegrid.py autograd + training.py:13-14,75), but the gradient never
 * touches HBM (432 MB written + 432 MB read per step less for config_nvp_s).  One GPU only: a data-parallel step must exchange the
 * gradient first.  Returns NVP_ERR_UNSUPPORTED with nothing enqueued when y_res * n_features is not a multiple of 4, a tensor is
 * not 16-B aligned or one band table exceeds 4 096 entries: take nvp_encode_bwd + nvp_adamw_step then. */
int nvp_encode_bwd_sparse_adamw(const float* coords, const float* dz, int32_t dz_stride, int64_t n,
                                const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                                const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags,
                                float* emb, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                                double weight_decay, int64_t step, void* stream);

/* The dense planes' half of the split scatter (NVP_SCATTER_DENSE_ONLY semantics; flags must carry NVP_COORDS_SORTED_BY_Y |
 * NVP_DZ_PLANES_READY) with torch.optim.AdamW's step (training.py:13-14) applied in the scatter's flushes: no gradient tensor is
 * produced for the three planes; params[q] / exp_avg[q] / exp_avg_sq[q] (q = 0 xy, 1 yt, 2 xt; each Sum res_l^2 * F floats) are updated in
 * place, bit-identical to nvp_encode_bwd(NVP_SCATTER_DENSE_ONLY) followed by nvp_adamw_step on its output.  step[q]: 1-based step count. */
int nvp_encode_bwd_dense_adamw(const float* coords, const float* dz, int32_t dz_stride, int64_t n,
                               const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                               const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags,
                               float* const* params, float* const* exp_avg, float* const* exp_avg_sq, double lr, double beta1, double beta2,
                               double eps, double weight_decay, const int64_t* step, void* stream);
int32_t nvp_dz_lm_supported(int32_t latent_dim);   /* does nvp_mlp_bwd_dx honour `lm` for this latent width in this build? */
int nvp_encode_bwd(const float* coords, const float* dz, int32_t dz_stride,
                   float* d_kf_xy, float* d_kf_yt, float* d_kf_xt, float* d_emb, int64_t n,
                   const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                   const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags,
                   void* stream);

/* Row-major [N,D] <-> PTM4 [ntiles][rows/4][32][4] (used by the stand-alone SirenWrapper). */
int nvp_rows_to_ptm(const float* src, float* dst, int64_t n, int32_t d, int32_t rows, void* stream);
int nvp_ptm_to_rows(const float* src, float* dst, int64_t n, int32_t d, int32_t rows, void* stream);

/* ---- R8-R10: SirenWrapper.forward = Modulator + modulated SirenNet
 * (reference modulation.py:138-145, 112-121, 83-92).
 * pack: re-lays the 14 tensors out as MFMA A-operand streams (call after every
 *       optimizer step; 0.4-0.8 MB).  The packed buffers are opaque: their size is
 *       nvp_packed_{fwd,bwd}_floats(latent_dim) and their format is whatever the fwd /
 *       bwd_dx kernels of this build consume (fp32 k-steps, or hi/mid/lo bf16 operand
 *       quads for the bf16x3 split-MFMA kernels used when the latent has <= 128 rows).
 * fwd : zt PTM4 latent, steps [N] -> rgb [N,3] row-major; `saved` receives the five
 *       activations backward needs (h0,h1,h2 post-LeakyReLU, q1,q2 pre-sine), each PTM4
 *       [ntiles][128/4][32][4]; pass NULL for inference (nothing is stored). */
int nvp_mlp_pack_fwd(const nvp_mlp_params* p, float* packed, int32_t latent_dim, void* stream);
int nvp_mlp_pack_bwd(const nvp_mlp_params* p, float* packed, int32_t latent_dim, void* stream);
int nvp_mlp_fwd(const float* zt, const float* steps, const nvp_mlp_params* p, const float* packed_fwd,
                float* rgb, float* saved, int64_t n, int32_t latent_dim, void* stream);

/* ---- R12: autograd of R8-R10 (reference training.py:74).
 * bwd_dx: drgb [N,3] -> dz_rows (latent gradient, ROW-MAJOR [ntiles*32][nvp_dz_stride(D)])
 *         + `dy`, six slots of ntiles*128*32 floats: slots 0,1,2,4,5 are the PTM4 streams
 *         dp0,dp1,dp2 (modulator pre-activation grads), dq1, dq2 the weight-gradient GEMMs
 *         consume; slot 3 holds, per 32-pixel tile, a 644-float record of the last layer's and
 *         SIREN layer 0's gradients already summed over the tile (layout kRec* in mlp_layout.h).
 *         (The modulated sine outputs x_k are NOT written: bwd_dw rebuilds x_k = sin(q_k) h_k
 *         from `saved`, which removes 1.5 KB/px of HBM writes from a write-bound kernel.)
 * bwd_dw: all 14 parameter gradients = split-K GEMMs over the pixel axis into
 *         `partials` [n_chunks][nvp_mlp_param_floats], then a deterministic reduction
 *         into `g` (overwritten). */
int nvp_mlp_bwd_dx(const float* drgb, const float* steps, const float* saved,
                   const nvp_mlp_params* p, const float* packed_bwd,
                   float* dy, float* dz_rows, const nvp_scatter_lm* lm /* NULL or nvp_encode_bwd_prepare's output */,
                   int64_t n, int32_t latent_dim, void* stream);
int nvp_mlp_bwd_dw(const float* drgb, const float* steps, const float* zt, const float* saved,
                   const float* dy, const nvp_mlp_params* p, float* partials, int32_t n_chunks,
                   const nvp_mlp_grads* g, int64_t n, int32_t latent_dim, void* stream);


/* ---- R13: image_mse (reference loss_functions.py:1-3 with training.py:47-48) ---------
 * gt_u8 [N,3] uint8.  loss_sum[0] += sum((rgb-gt)^2) (caller zeroes, divides by 3N);
 * drgb = 2*(rgb-gt)/(3N) i.e. d(mean)/d(rgb).  drgb may be NULL. */
int nvp_mse_u8(const float* rgb, const uint8_t* gt_u8, float* drgb, float* loss_sum,
               int64_t n, void* stream);


/* Delivery order of a batch (harness: counterpart of sorting the reference sampler's draws, dataio.py:104-120, by image column):
 * order[k] (int64, what torch.argsort returns) = index of the sample that comes k-th in ascending (pi % width), ties in drawing
 * order (a stable sort).  One counting sort on the log2(width)-bit key; workspace from nvp_sample_order_workspace_bytes.
 * NVP_ERR_UNSUPPORTED for width > 12288 (the per-chunk table lives in LDS): use a library sort then. */
int64_t nvp_sample_order_workspace_bytes(int64_t n, int32_t width);
int nvp_sample_order_by_column(const int64_t* pi, int64_t* order, int64_t n, int32_t width, void* workspace, int64_t workspace_bytes, void* stream);

/* Row order of an ARBITRARY batch - what a drop-in caller delivers (the reference sampler's raw order, dataio.py:104-120; replaces
 * the library argsort nvp_amd.functional.NVPFused used for such batches).  order[k] (int64) = index of the sample that comes k-th in
 * ascending key(y) = sum over the levels of the xy / yt planes' grid row index of coords[:,2] (both planes are indexed by y,
 * modules.py:61,63), ties in input order (stable, hence deterministic).  A batch gathered in this order has non-decreasing grid rows
 * at every level, which is all NVP_COORDS_SORTED_BY_Y promises the scatter.  One counting sort; NVP_ERR_UNSUPPORTED when the key
 * space exceeds 12 288 (level geometries far beyond the reference's configs): use a library sort of coords[:,2] then.  The workspace
 * query returns the same NVP_ERR_UNSUPPORTED (negative) for such geometries, so a caller never sizes a buffer for a call that would refuse. */
int64_t nvp_order_by_rows_workspace_bytes(int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt);
int nvp_order_by_rows(const float* coords, int64_t* order, int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- row H helpers: on-device sampler (reference dataio.py:104-120) -------------------
 * ti [N], pi [N] int64 indices (drawn by the caller with torch.randint, temporal first,
 * same order as the reference); video u8 [T][H*W][3] resident on the device;
 * tcoord_tab = linspace(0,1,T), tstep_tab = linspace(.5/T, 1-.5/T, T) (dataio.py:93-99).
 * Writes coords [N,3] = (tcoord[ti], row/(H-1), col/(W-1)), steps [N], gt_u8 [N,3].
 * order (may be NULL): int64 [N] permutation - output row k is draw order[k] (the sampler delivers
 * its batch sorted by image column without separate index passes). */
int nvp_sample_gather(const uint8_t* video, const int64_t* ti, const int64_t* pi, const int64_t* order,
                      const float* tcoord_tab, const float* tstep_tab,
                      float* coords, float* steps, uint8_t* gt_u8,
                      int64_t n, int32_t t_frames, int32_t height, int32_t width, void* stream);

/* ---- dense AdamW (reference training.py:13-14 `torch.optim.AdamW(lr=1e-2,
 * weight_decay=1e-3)`, stepped at training.py:75) over a list of fp32 tensors in one launch.
 * torch.optim.AdamW's update rule (decoupled decay, no amsgrad), `step` = 1-based iteration
 * count (bias corrections 1-beta^step), grad_scale multiplies the gradient first (1/world
 * after a SUM all-reduce; pass 1.0 otherwise).  Updates param, exp_avg, exp_avg_sq in place. */
typedef struct nvp_adamw_seg {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
} nvp_adamw_seg;
int nvp_adamw_step(const nvp_adamw_seg* segs, int32_t n_segs, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int64_t step, double grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NVP_HIP_H */
