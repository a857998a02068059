// Dense-grid / sparse-grid index arithmetic shared by the gather (encode.hip, encode_fwd_lds.hip) and the
// scatter (encode_bwd.hip).  Include only from translation units compiled with -ffp-contract=off and
// `#pragma clang fp contract(off)`: products and sums below must keep the rounding the comments state.
//
// R2 (tinycudann.Encoding, reference call sites modules.py:65-67; level geometry eval.py:28-35).  The
// tiny-cuda-nn fork is absent from /root/reference (parity unpinned), so the two places where published
// tiny-cuda-nn and a "torch-natural" restatement differ are switchable per encoding (nvp_levels.flags):
//   NVP_GRID_POS_FMA    pos = fmaf(scale, x, 0.5f)          (one rounding; upstream pos_fract)   [default]
//                       else pos = fl(fl(x*scale) + 0.5f)   (two roundings; the round-1 form)
//   NVP_GRID_INTERP_FMA out = fma(w_c, v_c, out) over the corners, out starting at 0 (upstream)  [default]
//                       else out = fl(out + fl(w_c*v_c))
//   NVP_GRID_CLAMP      corner coordinate i+1 clamps to res-1 instead of the cell index wrapping mod res^2
//                       (upstream dense grids wrap: `index % hashmap_size`; default off)
#pragma once
#include "nvp_common.h"

// pos along one axis; floor / fract are taken by the caller
__device__ __forceinline__ float nvp_grid_pos(float x, float scale, int flags) {
    if (flags & NVP_GRID_POS_FMA) return __builtin_fmaf(scale, x, 0.5f);
    return nvp_add_rn(nvp_mul_rn(x, scale), 0.5f);
}

struct NvpBilerp {
    int cell[4];     // corner cells (level-local), order (0,0),(1,0),(0,1),(1,1)
    float w[4];
    int i0, i1;      // floor(pos) per axis
};

// level-local cell of corner (a, b) of the pixel whose floor coordinates are (i0, i1)
__device__ __forceinline__ int nvp_grid_cell(int i0, int i1, int a, int b, int res, int flags) {
    if (flags & NVP_GRID_CLAMP) {
        const int c0 = min(max(i0 + a, 0), res - 1), c1 = min(max(i1 + b, 0), res - 1);
        return c0 + c1 * res;
    }
    const int size = res * res;
    int c = (i0 + a) + (i1 + b) * res;          // wrapped into the level like tcnn's `index % hashmap_size`
    if ((unsigned)c >= (unsigned)size) { c %= size; if (c < 0) c += size; }
    return c;
}

__device__ __forceinline__ NvpBilerp nvp_bilerp_setup(float x0, float x1, float scale, int res, int flags) {
    const float p0 = nvp_grid_pos(x0, scale, flags), p1 = nvp_grid_pos(x1, scale, flags);
    const float f0 = floorf(p0), f1 = floorf(p1);
    const float w0 = nvp_sub_rn(p0, f0), w1 = nvp_sub_rn(p1, f1);
    const float u0 = nvp_sub_rn(1.0f, w0), u1 = nvp_sub_rn(1.0f, w1);
    NvpBilerp b;
    b.i0 = (int)f0; b.i1 = (int)f1;
    b.w[0] = nvp_mul_rn(u0, u1);
    b.w[1] = nvp_mul_rn(w0, u1);
    b.w[2] = nvp_mul_rn(u0, w1);
    b.w[3] = nvp_mul_rn(w0, w1);
#pragma unroll
    for (int c = 0; c < 4; ++c) b.cell[c] = nvp_grid_cell(b.i0, b.i1, c & 1, c >> 1, res, flags);
    return b;
}

// one feature of the 4-corner blend, corners in order, in the arithmetic `flags` selects
__device__ __forceinline__ float nvp_blend4(const float (&w)[4], float v0, float v1, float v2, float v3, int flags) {
    if (flags & NVP_GRID_INTERP_FMA) {
        float a = nvp_mul_rn(w[0], v0);           // fma(w, v, 0) == fl(w*v)
        a = __builtin_fmaf(w[1], v1, a);
        a = __builtin_fmaf(w[2], v2, a);
        a = __builtin_fmaf(w[3], v3, a);
        return a;
    }
    float a = nvp_mul_rn(w[0], v0);
    a = nvp_add_rn(a, nvp_mul_rn(w[1], v1));
    a = nvp_add_rn(a, nvp_mul_rn(w[2], v2));
    a = nvp_add_rn(a, nvp_mul_rn(w[3], v3));
    return a;
}

// clamp(int64(fp32((res-1)*c) + 0.5), 0, res-1)   reference sparsegrid.py:44-46 (mul and add rounded separately,
// conversion truncates toward zero like .type(int64))
__device__ __forceinline__ int nvp_nearest_idx(float c, int res) {
    const float f = nvp_mul_rn((float)(res - 1), c);
    const int i = (int)nvp_add_rn(f, 0.5f);
    return min(max(i, 0), res - 1);
}
