// Grid encoders of the NVP hot path for gfx950: the three 2D multi-resolution dense
// grids ("learnable keyframes", R1-R3) and the 3D sparse positional-feature grid
// (R4-R7), forward gathers and gradient scatter-adds.
//
// These are HBM/L2-bound random-access kernels: no MFMA here.  Lanes are
// pixels; each thread owns one (pixel, slot) pair where a slot is a group of 4 grid
// levels of one plane (or the sparse 3x3 patch), so a wavefront works on one small set of
// levels at a time (coarse levels stay L1/L2 resident) and has 16 independent gathers
// in flight per lane.  Outputs go either to the reference's row-major [N, C] layout
// (stand-alone modules) or to the pixel-tile-major (PTM) layout the MFMA MLP consumes,
// where the 32 pixels of a tile are contiguous -> every store/load is a full 128-B line.
//
// Index arithmetic reproduces the reference's separately rounded fp32 ops
// (sparsegrid.py:44-46): products and sums go through nvp_mul_rn / nvp_add_rn (nvp_common.h) so hipcc cannot
// contract them into an FMA and flip a cell at a .5 boundary.
#include <cstdlib>
#include "encode_gather.h"

// (__fmul_rn / __fadd_rn are plain operators in this HIP and fuse after inlining.)  Also forbid FMA contraction for the whole TU so
// index and interpolation arithmetic keeps the reference's separately rounded multiply and add.
#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kLevelsPerSlot = 4;

// Output addressing: row-major [N, C] or PTM4 [ntiles][rows/4][32][4] (rows a multiple of 4;
// the 4 rows of a row-group are contiguous per pixel, so a lane's natural access is 16 B).
template <bool PTM>
__device__ __forceinline__ int64_t out_addr(int64_t px, int col, int ncols_or_rows) {
    if constexpr (PTM) return (((px >> 5) * (ncols_or_rows >> 2) + (col >> 2)) * 32 + (px & 31)) * 4 + (col & 3);
    else return px * ncols_or_rows + col;
}

// store `count` consecutive columns starting at col0 for one pixel; 16-B stores when aligned
template <bool PTM>
__device__ __forceinline__ void store_cols(float* __restrict__ out, int64_t px, int col0, int ncols, const float* v, int count) {
    if constexpr (PTM) {
        if ((col0 & 3) == 0 && (count & 3) == 0) {
            for (int c = 0; c < count; c += 4)
                *reinterpret_cast<float4*>(out + out_addr<true>(px, col0 + c, ncols)) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
            return;
        }
    }
    for (int c = 0; c < count; ++c) out[out_addr<PTM>(px, col0 + c, ncols)] = v[c];
}

// ---- dense 2D grid: one slot (<= 4 levels) of one plane for one pixel -------------
template <int F, bool PTM>
__device__ __forceinline__ void dense_slot_fwd(const float* __restrict__ params, const nvp_levels& lv,
                                               int lvl0, float x0, float x1, bool valid,
                                               float* __restrict__ out, int64_t px, int col0, int ncols) {
    float res[kLevelsPerSlot * F];
    int nl = 0;
#pragma unroll
    for (int dl = 0; dl < kLevelsPerSlot; ++dl) {
        int l = lvl0 + dl;
        if (l >= lv.n_levels) break;
        nl = dl + 1;
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.f;
        if (valid) {
            NvpBilerp b = nvp_bilerp_setup(x0, x1, lv.scale[l], lv.res[l], lv.flags);
            const float* base = params + (int64_t)lv.offset[l] * F;
            Vec<F> v0, v1, v2, v3;
            if constexpr (F == 2) {
                // corners (0,b) and (1,b) are neighbours in memory unless the column wrapped / clamped: one 16-byte gather
                // per grid row instead of two 8-byte ones (the TA cost of a random gather is per lane-request, not per byte)
                load_pair2(v0, v1, base, b.cell[0], b.cell[1]);
                load_pair2(v2, v3, base, b.cell[2], b.cell[3]);
            } else {
                v0 = load_vec<F>(base + (int64_t)b.cell[0] * F);
                v1 = load_vec<F>(base + (int64_t)b.cell[1] * F);
                v2 = load_vec<F>(base + (int64_t)b.cell[2] * F);
                v3 = load_vec<F>(base + (int64_t)b.cell[3] * F);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = nvp_blend4(b.w, v0.v[f], v1.v[f], v2.v[f], v3.v[f], lv.flags);
        }
#pragma unroll
        for (int f = 0; f < F; ++f) res[dl * F + f] = acc[f];
    }
    store_cols<PTM>(out, px, col0 + lvl0 * F, ncols, res, nl * F);
}

template <int F, bool PTM>
__device__ __forceinline__ void dense_slot_bwd(float* __restrict__ dparams, const nvp_levels& lv,
                                               int lvl0, float x0, float x1,
                                               const float* __restrict__ dout, int64_t px, int col0, int ncols) {
#pragma unroll
    for (int dl = 0; dl < kLevelsPerSlot; ++dl) {
        int l = lvl0 + dl;
        if (l >= lv.n_levels) break;
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { g[f] = dout[out_addr<PTM>(px, col0 + l * F + f, ncols)]; any |= (g[f] != 0.f); }
        if (!any) continue;
        NvpBilerp b = nvp_bilerp_setup(x0, x1, lv.scale[l], lv.res[l], lv.flags);
        float* base = dparams + (int64_t)lv.offset[l] * F;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* p = base + (int64_t)b.cell[c] * F;
#pragma unroll
            for (int f = 0; f < F; ++f) nvp_atomic_add(p + f, b.w[c] * g[f]);
        }
    }
}

template <bool PTM>
__device__ __forceinline__ void sparse_fwd(const float* __restrict__ emb, const nvp_sparse_shape& sh, bool inter,
                                           float t, float x, float y, bool valid,
                                           float* __restrict__ out, int64_t px, int col0, int ncols) {
    const int F = sh.n_features;
    if (!valid) {
        for (int c = 0; c < 9 * F; ++c) out[out_addr<PTM>(px, col0 + c, ncols)] = 0.f;
        return;
    }
    Patch p = patch_setup(t, x, y, sh, inter);
    const int64_t plane = (int64_t)sh.x_res * sh.y_res;
    for (int i = 0; i < 3; ++i) {
        float row[3 * 8];                                    // one x-row of the patch: 3 cells x F <= 8
        for (int j = 0; j < 3; ++j) {
            int64_t cell = (int64_t)p.vx[i] * sh.y_res + p.vy[j];
            const float* lo = emb + ((int64_t)p.t_lo * plane + cell) * F;
            const float* hi = emb + ((int64_t)p.t_hi * plane + cell) * F;
            for (int f = 0; f < F && f < 8; ++f) {
                float v;
                if (!inter) v = lo[f];
                else v = nvp_add_rn(nvp_mul_rn(lo[f], p.w_lo), nvp_mul_rn(hi[f], p.w_hi));
                row[j * F + f] = v;
            }
        }
        for (int c = 0; c < 3 * F; ++c) out[out_addr<PTM>(px, col0 + i * 3 * F + c, ncols)] = row[c];
    }
}

// Fused-path variant with F known at compile time: the three cells of a patch row are contiguous in memory (y is the
// fastest grid axis), so each row is ONE contiguous 3F-float read ([y0, y0+2] with y0 = clamp(yi-1, 0, Y-3); the clamped
// border duplicates are picked from it by index) instead of 3F scalar loads, and the 9F (+ pad) latent rows leave as
// 16-byte PTM4 stores.  Needs Y >= 3 and 16-byte aligned row-groups (col0 % 4 == 0); otherwise the generic code runs.
template <int F>
__device__ __forceinline__ void sparse_fwd_ptm(const float* __restrict__ emb, const nvp_sparse_shape& sh, bool inter,
                                               float t, float x, float y, bool valid,
                                               float* __restrict__ out, int64_t px, int col0, int rows, int d) {
    constexpr int NV = 9 * F;
    constexpr int NVP4 = (NV + 3) & ~3;
    float v[NVP4];
#pragma unroll
    for (int c = 0; c < NVP4; ++c) v[c] = 0.f;
    if (valid) {
        const Patch p = patch_setup(t, x, y, sh, inter);
        const int y0 = min(max(p.vy[1] - 1, 0), sh.y_res - 3);
        const int64_t plane = (int64_t)sh.x_res * sh.y_res;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int64_t cell = (int64_t)p.vx[i] * sh.y_res + y0;
            float lo[3 * F], hi[3 * F];
            const float* pl = emb + ((int64_t)p.t_lo * plane + cell) * F;
            if constexpr (F == 2) {
                const Hex8 q = *reinterpret_cast<const Hex8*>(pl);
#pragma unroll
                for (int c = 0; c < 6; ++c) lo[c] = q.v[c];
            } else {
#pragma unroll
                for (int c = 0; c < 3 * F; ++c) lo[c] = pl[c];
            }
            if (inter) {
                const float* ph = emb + ((int64_t)p.t_hi * plane + cell) * F;
#pragma unroll
                for (int c = 0; c < 3 * F; ++c) hi[c] = ph[c];
#pragma unroll
                for (int c = 0; c < 3 * F; ++c) lo[c] = nvp_add_rn(nvp_mul_rn(lo[c], p.w_lo), nvp_mul_rn(hi[c], p.w_hi));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sel = p.vy[j] - y0;                 // 0, 1 or 2
#pragma unroll
                for (int f = 0; f < F; ++f)
                    v[(i * 3 + j) * F + f] = sel == 0 ? lo[f] : (sel == 1 ? lo[F + f] : lo[2 * F + f]);
            }
        }
    }
    // rows col0 .. col0 + 9F - 1, then zero pad rows up to `rows` (d = col0 + 9F is the latent width)
    float4* o4 = reinterpret_cast<float4*>(out);
    const int64_t zbase = ((px >> 5) * (int64_t)(rows >> 2)) * 32 + (px & 31);
#pragma unroll
    for (int q = 0; q < NVP4 / 4; ++q)
        if (col0 + 4 * q < rows) o4[zbase + (int64_t)((col0 >> 2) + q) * 32] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    (void)d;
}

template <bool PTM>
__device__ __forceinline__ void sparse_bwd(float* __restrict__ demb, const nvp_sparse_shape& sh,
                                           float t, float x, float y,
                                           const float* __restrict__ dout, int64_t px, int col0, int ncols) {
    const int F = sh.n_features;
    Patch p = patch_setup(t, x, y, sh, false);
    const int64_t plane = (int64_t)sh.x_res * sh.y_res;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            int64_t cell = (int64_t)p.vx[i] * sh.y_res + p.vy[j];
            float* dst = demb + ((int64_t)p.t_lo * plane + cell) * F;
            for (int f = 0; f < F; ++f) {
                float g = dout[out_addr<PTM>(px, col0 + (i * 3 + j) * F + f, ncols)];
                if (g != 0.f) nvp_atomic_add(dst + f, g);
            }
        }
    }
}

// ---- kernels ---------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(kThreads) void dense2d_fwd_kernel(const float* __restrict__ params, const float* __restrict__ x,
                                                               float* __restrict__ out, int64_t n, nvp_levels lv) {
    int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (px >= n) return;
    float2 c = *reinterpret_cast<const float2*>(x + px * 2);
    dense_slot_fwd<F, false>(params, lv, blockIdx.y * kLevelsPerSlot, c.x, c.y, true, out, px, 0, lv.n_levels * F);
}

template <int F>
__global__ __launch_bounds__(kThreads) void dense2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                               float* __restrict__ dparams, int64_t n, nvp_levels lv) {
    int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (px >= n) return;
    float2 c = *reinterpret_cast<const float2*>(x + px * 2);
    dense_slot_bwd<F, false>(dparams, lv, blockIdx.y * kLevelsPerSlot, c.x, c.y, dout, px, 0, lv.n_levels * F);
}

__global__ __launch_bounds__(kThreads) void sparse_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ coords,
                                                              float* __restrict__ out, int64_t n, nvp_sparse_shape sh, int inter) {
    int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (px >= n) return;
    const float* c = coords + px * 3;
    sparse_fwd<false>(emb, sh, inter != 0, c[0], c[1], c[2], true, out, px, 0, 9 * sh.n_features);
}

__global__ __launch_bounds__(kThreads) void sparse_bwd_kernel(const float* __restrict__ coords, const float* __restrict__ dout,
                                                              float* __restrict__ demb, int64_t n, nvp_sparse_shape sh) {
    int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (px >= n) return;
    const float* c = coords + px * 3;
    sparse_bwd<false>(demb, sh, c[0], c[1], c[2], dout, px, 0, 9 * sh.n_features);
}

struct EncodeArgs {
    nvp_levels lv[3];          // latent order: xy, yt, xt
    nvp_sparse_shape sh;
    int col0[4];               // first latent row of xy, yt, xt, sparse
    int slots[3];              // slots per plane
    int rows;                  // PTM4 rows (D rounded up to a multiple of 4)
    int d;                     // latent dim
    int s_first;               // first slot this launch covers (the LDS-staged kernel takes the xy / yt slots of y-sorted batches)
};

// slot decode: blockIdx.y in [0, slots0+slots1+slots2] ; last = sparse (+ pad rows)
template <int F>
__global__ __launch_bounds__(kThreads) void encode_fwd_kernel(const float* __restrict__ coords,
                                                              const float* __restrict__ kf0, const float* __restrict__ kf1,
                                                              const float* __restrict__ kf2, const float* __restrict__ emb,
                                                              float* __restrict__ zt, int64_t n, int64_t npad, EncodeArgs a, int inter) {
    int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (px >= npad) return;
    bool valid = px < n;
    float t = 0.f, x = 0.f, y = 0.f;
    if (valid) { const float* c = coords + px * 3; t = c[0]; x = c[1]; y = c[2]; }
    int s = blockIdx.y + a.s_first;
    if (s < a.slots[0]) {          // xy plane <- (x, y)            reference modules.py:61
        dense_slot_fwd<F, true>(kf0, a.lv[0], s * kLevelsPerSlot, x, y, valid, zt, px, a.col0[0], a.rows);
    } else if ((s -= a.slots[0]) < a.slots[1]) {   // yt plane <- (t, y)   modules.py:63
        dense_slot_fwd<F, true>(kf1, a.lv[1], s * kLevelsPerSlot, t, y, valid, zt, px, a.col0[1], a.rows);
    } else if ((s -= a.slots[1]) < a.slots[2]) {   // xt plane <- (t, x)   modules.py:62
        dense_slot_fwd<F, true>(kf2, a.lv[2], s * kLevelsPerSlot, t, x, valid, zt, px, a.col0[2], a.rows);
    } else {
        if (a.sh.n_features == F && a.sh.y_res >= 3 && (a.col0[3] & 3) == 0) {
            sparse_fwd_ptm<F>(emb, a.sh, inter != 0, t, x, y, valid, zt, px, a.col0[3], a.rows, a.d);     // writes the pad rows too
        } else {
            sparse_fwd<true>(emb, a.sh, inter != 0, t, x, y, valid, zt, px, a.col0[3], a.rows);
            for (int r = a.d; r < a.rows; ++r) zt[out_addr<true>(px, r, a.rows)] = 0.f;   // pad rows (rows = D rounded up to 4)
        }
    }
}

// row-major [N,D] <-> PTM4 [ntiles][rows/4][32][4]: one thread per (tile, row-group, pixel) moves
// 4 consecutive features of one pixel (a 16-B piece on both sides).
__global__ __launch_bounds__(256) void rows_to_ptm_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          int64_t n, int d, int rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index in dst
    const int r4 = rows >> 2;
    const int64_t total = nvp_ntiles(n) * (int64_t)r4 * 32;
    if (idx >= total) return;
    const int j = (int)(idx & 31);
    const int rg = (int)((idx >> 5) % r4);
    const int64_t t = (idx >> 5) / r4;
    const int64_t px = t * 32 + j;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * rg + e;
        v[e] = (px < n && c < d) ? src[px * d + c] : 0.f;
    }
    reinterpret_cast<float4*>(dst)[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void ptm_to_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          int64_t n, int d, int rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int r4 = rows >> 2;
    const int64_t total = nvp_ntiles(n) * (int64_t)r4 * 32;
    if (idx >= total) return;
    const int j = (int)(idx & 31);
    const int rg = (int)((idx >> 5) % r4);
    const int64_t t = (idx >> 5) / r4;
    const int64_t px = t * 32 + j;
    if (px >= n) return;
    const float4 v = reinterpret_cast<const float4*>(src)[idx];
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * rg + e;
        if (c < d) dst[px * d + c] = vv[e];
    }
}

bool levels_ok(const nvp_levels* lv) {
    if (!lv) return false;
    if (lv->n_levels < 1 || lv->n_levels > NVP_MAX_LEVELS) return false;
    int f = lv->n_features;
    return f == 1 || f == 2 || f == 4 || f == 8;
}

bool shape_ok(const nvp_sparse_shape* sh) {
    return sh && sh->t_res >= 1 && sh->x_res >= 1 && sh->y_res >= 1 && sh->n_features >= 1 && sh->n_features <= 8 &&
           (int64_t)sh->t_res * sh->x_res * sh->y_res * sh->n_features < (int64_t)1 << 40;
}

template <typename Fn>
int dispatch_f(int f, Fn&& fn) {
    switch (f) {
        case 1: fn(std::integral_constant<int, 1>{}); return 0;
        case 2: fn(std::integral_constant<int, 2>{}); return 0;
        case 4: fn(std::integral_constant<int, 4>{}); return 0;
        case 8: fn(std::integral_constant<int, 8>{}); return 0;
    }
    return NVP_ERR_UNSUPPORTED;
}

int make_args(EncodeArgs& a, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
              const nvp_sparse_shape* sh) {
    if (!levels_ok(lv_xy) || !levels_ok(lv_yt) || !levels_ok(lv_xt) || !shape_ok(sh)) return NVP_ERR_BADARG;
    if (lv_xy->n_features != lv_yt->n_features || lv_xy->n_features != lv_xt->n_features) return NVP_ERR_UNSUPPORTED;
    a.lv[0] = *lv_xy; a.lv[1] = *lv_yt; a.lv[2] = *lv_xt; a.sh = *sh;
    int col = 0;
    for (int p = 0; p < 3; ++p) {
        a.col0[p] = col;
        col += a.lv[p].n_levels * a.lv[p].n_features;
        a.slots[p] = (a.lv[p].n_levels + kLevelsPerSlot - 1) / kLevelsPerSlot;
    }
    a.col0[3] = col;
    a.d = col + 9 * sh->n_features;
    a.rows = nvp_rows4(a.d);
    a.s_first = 0;
    return 0;
}

}  // namespace

// encode_fwd_lds.hip
int nvp_encode_fwd_lds_launch(const float* coords, const float* kf_xy, const float* kf_yt, float* zt, int64_t n, int64_t npad,
                              const nvp_levels* lv_xy, const nvp_levels* lv_yt, int col0_xy, int col0_yt, int rows, hipStream_t stream);


// ---- SparseGrid(upsample=True): the x2 bilinear pre-upsample of the (x, y) axes (sparsegrid.py:26-34) --------------------------------
// The reference runs F.interpolate(scale_factor=2, mode='bilinear') over [F, T, X, Y] on every call (align_corners=False; ATen's
// area_pixel_compute_source_index with scale 1/2): destination index u reads source position max(0, u / 2 - 0.25), i.e.
//   u = 0: cell 0 alone;  u = 2k (k >= 1): 0.25 * cell k-1 + 0.75 * cell min(k, R-1);  u = 2k + 1: 0.75 * cell k + 0.25 * cell min(k+1, R-1)
// and the output is h0 * (w0 v00 + w1 v01) + h1 * (w0 v10 + w1 v11) with h along x, w along y (ATen's own formula, multiply and add rounded
// separately: this file is compiled with FMA contraction off).  Layouts stay [T][X][Y][F] / [T][2X][2Y][F]: no permute passes.
struct Up1 { int i0, i1; float l0, l1; };
__device__ __forceinline__ Up1 up_src(int u, int R) {
    const float src = fmaxf(0.5f * ((float)u + 0.5f) - 0.5f, 0.f);
    const int i0 = (int)src;
    Up1 r;
    r.i0 = i0; r.i1 = min(i0 + 1, R - 1);
    r.l1 = src - (float)i0; r.l0 = 1.0f - r.l1;
    return r;
}

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ emb, float* __restrict__ out, int T, int X, int Y, int F) {
    const int64_t total = (int64_t)T * (2 * X) * (2 * Y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int uy = (int)(idx % (2 * Y));
        const int ux = (int)((idx / (2 * Y)) % (2 * X));
        const int64_t t = idx / ((int64_t)4 * X * Y);
        const Up1 hx = up_src(ux, X), wy = up_src(uy, Y);
        const float* b = emb + t * (int64_t)X * Y * F;
        const float* v00 = b + ((int64_t)hx.i0 * Y + wy.i0) * F;
        const float* v01 = b + ((int64_t)hx.i0 * Y + wy.i1) * F;
        const float* v10 = b + ((int64_t)hx.i1 * Y + wy.i0) * F;
        const float* v11 = b + ((int64_t)hx.i1 * Y + wy.i1) * F;
        float* o = out + idx * F;
        for (int f = 0; f < F; ++f)
            o[f] = hx.l0 * (wy.l0 * v00[f] + wy.l1 * v01[f]) + hx.l1 * (wy.l0 * v10[f] + wy.l1 * v11[f]);
    }
}

// adjoint, as a GATHER (deterministic, no atomics): source cell (x, y) collects every upsampled cell that read it.  In one dimension cell c is
// read by u in {2c-1, 2c, 2c+1, 2c+2} (where they exist), with the forward's own (i0, i1, l0, l1) of that u - at the upper border both
// references of u = 2R-1 land on cell R-1 and both weights count.
__device__ __forceinline__ float up_weight(int u, int c, int R) {
    if (u < 0 || u >= 2 * R) return 0.f;
    const Up1 s = up_src(u, R);
    return (s.i0 == c ? s.l0 : 0.f) + (s.i1 == c ? s.l1 : 0.f);
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dout, float* __restrict__ demb, int T, int X, int Y, int F) {
    const int64_t total = (int64_t)T * X * Y;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int y = (int)(idx % Y);
        const int x = (int)((idx / Y) % X);
        const int64_t t = idx / ((int64_t)X * Y);
        const float* d = dout + t * (int64_t)4 * X * Y * F;
        float acc[16];
        for (int f = 0; f < F; ++f) acc[f] = 0.f;
        for (int a = -1; a <= 2; ++a) {
            const int ux = 2 * x + a;
            const float wx = up_weight(ux, x, X);
            if (wx == 0.f) continue;
            for (int b = -1; b <= 2; ++b) {
                const int uy = 2 * y + b;
                const float wyv = up_weight(uy, y, Y);
                if (wyv == 0.f) continue;
                const float w = wx * wyv;
                const float* g = d + ((int64_t)ux * (2 * Y) + uy) * F;
                for (int f = 0; f < F; ++f) acc[f] += w * g[f];
            }
        }
        for (int f = 0; f < F; ++f) demb[idx * F + f] = acc[f];
    }
}

extern "C" {

int nvp_dense2d_fwd(const float* params, const float* x, float* out, int64_t n, const nvp_levels* lv, void* stream) {
    if (!levels_ok(lv) || n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!params || !x || !out) return NVP_ERR_BADARG;          // (argument errors are decided on the host: a C caller must never get a GPU fault for a NULL)
    dim3 grid((unsigned)((n + kThreads - 1) / kThreads), (lv->n_levels + kLevelsPerSlot - 1) / kLevelsPerSlot);
    int rc = dispatch_f(lv->n_features, [&](auto f) {
        hipLaunchKernelGGL((dense2d_fwd_kernel<decltype(f)::value>), grid, dim3(kThreads), 0, (hipStream_t)stream, params, x, out, n, *lv);
    });
    if (rc) return rc;
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_dense2d_bwd(const float* x, const float* dout, float* dparams, int64_t n, const nvp_levels* lv, void* stream) {
    if (!levels_ok(lv) || n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!x || !dout || !dparams) return NVP_ERR_BADARG;
    dim3 grid((unsigned)((n + kThreads - 1) / kThreads), (lv->n_levels + kLevelsPerSlot - 1) / kLevelsPerSlot);
    int rc = dispatch_f(lv->n_features, [&](auto f) {
        hipLaunchKernelGGL((dense2d_bwd_kernel<decltype(f)::value>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, dout, dparams, n, *lv);
    });
    if (rc) return rc;
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_sparse3x3_fwd(const float* emb, const float* coords, float* out, int64_t n, const nvp_sparse_shape* sh, void* stream) {
    if (!shape_ok(sh) || n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!emb || !coords || !out) return NVP_ERR_BADARG;
    hipLaunchKernelGGL(sparse_fwd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       emb, coords, out, n, *sh, 0);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_sparse3x3_inter_fwd(const float* emb, const float* coords, float* out, int64_t n, const nvp_sparse_shape* sh, void* stream) {
    if (!shape_ok(sh) || n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!emb || !coords || !out) return NVP_ERR_BADARG;
    hipLaunchKernelGGL(sparse_fwd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       emb, coords, out, n, *sh, 1);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_sparse3x3_bwd(const float* coords, const float* dout, float* demb, int64_t n, const nvp_sparse_shape* sh, void* stream) {
    if (!shape_ok(sh) || n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!coords || !dout || !demb) return NVP_ERR_BADARG;
    hipLaunchKernelGGL(sparse_bwd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       coords, dout, demb, n, *sh);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_encode_fwd(const float* coords, const float* kf_xy, const float* kf_yt, const float* kf_xt, const float* emb,
                   float* zt, int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                   const nvp_sparse_shape* sh, int temporal_interp, int32_t flags, void* stream) {
    EncodeArgs a;
    int rc = make_args(a, lv_xy, lv_yt, lv_xt, sh);
    if (rc) return rc;
    if (n < 0) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!coords || !kf_xy || !kf_yt || !kf_xt || !emb || !zt) return NVP_ERR_BADARG;
    int64_t npad = nvp_ntiles(n) * NVP_T;
    // NVP_ENCODE_LDS=1 (environment, read once) + a y-sorted batch: the xy and yt planes (row coordinate = y) go through the
    // LDS-staged kernel (encode_fwd_lds.hip), this kernel keeps the xt plane and the sparse grid.  Bit-identical results.
    // OFF by default: measured on MI355X at N = 1 245 184 the staged kernel takes 0.41 ms for the two planes the global gather
    // below does in ~0.25 ms - with y-sorted batches those planes' rows already sit in L1/L2 and both kernels are bound by the
    // per-level index arithmetic (~100 VALU instructions per pixel and level), not by the fetches the staging removes.
#if NVP_EXPERIMENTS
    static const bool lds_on = [] { const char* e = getenv("NVP_ENCODE_LDS"); return e && e[0] == '1'; }();
    const int F = a.lv[0].n_features;
    const bool lds = lds_on && (flags & NVP_COORDS_SORTED_BY_Y) && a.lv[1].n_features == F && (F == 1 || ((a.col0[0] | a.col0[1]) & 3) == 0);
    if (lds) {
        rc = nvp_encode_fwd_lds_launch(coords, kf_xy, kf_yt, zt, n, npad, &a.lv[0], &a.lv[1], a.col0[0], a.col0[1], a.rows, (hipStream_t)stream);
        if (rc) return rc;
        a.s_first = a.slots[0] + a.slots[1];
    }
#else
    (void)flags;
#endif
    dim3 grid((unsigned)((npad + kThreads - 1) / kThreads), a.slots[0] + a.slots[1] + a.slots[2] + 1 - a.s_first);
    rc = dispatch_f(a.lv[0].n_features, [&](auto f) {
        hipLaunchKernelGGL((encode_fwd_kernel<decltype(f)::value>), grid, dim3(kThreads), 0, (hipStream_t)stream,
                           coords, kf_xy, kf_yt, kf_xt, emb, zt, n, npad, a, temporal_interp);
    });
    if (rc) return rc;
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_rows_to_ptm(const float* src, float* dst, int64_t n, int32_t d, int32_t rows, void* stream) {
    if (n < 0 || d < 1 || rows < d || (rows & 3)) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!src || !dst) return NVP_ERR_BADARG;
    const int64_t total = nvp_ntiles(n) * (int64_t)(rows >> 2) * 32;
    hipLaunchKernelGGL(rows_to_ptm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n, d, rows);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_ptm_to_rows(const float* src, float* dst, int64_t n, int32_t d, int32_t rows, void* stream) {
    if (n < 0 || d < 1 || rows < d || (rows & 3)) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    if (!src || !dst) return NVP_ERR_BADARG;
    const int64_t total = nvp_ntiles(n) * (int64_t)(rows >> 2) * 32;
    hipLaunchKernelGGL(ptm_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n, d, rows);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_sparse_upsample2x_fwd(const float* emb, float* out, const nvp_sparse_shape* sh, void* stream) {
    if (!emb || !out || !sh || sh->t_res < 1 || sh->x_res < 1 || sh->y_res < 1 || sh->n_features < 1 || sh->n_features > 16) return NVP_ERR_BADARG;
    const int64_t total = (int64_t)sh->t_res * 4 * sh->x_res * sh->y_res;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, emb, out, sh->t_res, sh->x_res, sh->y_res, sh->n_features);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_sparse_upsample2x_bwd(const float* dout, float* demb, const nvp_sparse_shape* sh, void* stream) {
    if (!dout || !demb || !sh || sh->t_res < 1 || sh->x_res < 1 || sh->y_res < 1 || sh->n_features < 1 || sh->n_features > 16) return NVP_ERR_BADARG;
    const int64_t total = (int64_t)sh->t_res * sh->x_res * sh->y_res;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, demb, sh->t_res, sh->x_res, sh->y_res, sh->n_features);
    NVP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
