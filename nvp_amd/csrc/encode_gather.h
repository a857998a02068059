// Gather helpers shared by the stand-alone encoders (encode.hip) and the forward kernel's in-wave tile gather (encode_tile.h):
// corner fetches of the dense 2D grids and the 3x3 patch set-up of the sparse grid.  Index / interpolation arithmetic: grid_math.h.
// Include only where FMA contraction is off (see grid_math.h).
#pragma once
#include "grid_math.h"

// index / interpolation arithmetic: grid_math.h (switchable dense-grid variant, nvp_levels.flags)
template <int F>
struct Vec { float v[F]; };

template <int F>
__device__ __forceinline__ Vec<F> load_vec(const float* p) {
    Vec<F> r;
    if constexpr (F == 2) { float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
    else if constexpr (F == 4) { float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else if constexpr (F == 8) {
        float4 t = *reinterpret_cast<const float4*>(p); float4 u = *reinterpret_cast<const float4*>(p + 4);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; r.v[4] = u.x; r.v[5] = u.y; r.v[6] = u.z; r.v[7] = u.w;
    } else {
#pragma unroll
        for (int f = 0; f < F; ++f) r.v[f] = p[f];
    }
    return r;
}

// 16 bytes at an 8-byte aligned address (gfx950 global loads only need dword alignment; hipcc then emits global_load_dwordx4)
struct __attribute__((packed, aligned(8))) Quad8 { float v[4]; };
struct __attribute__((packed, aligned(8))) Hex8 { float v[6]; };

__device__ __forceinline__ void load_pair2(Vec<2>& a, Vec<2>& b, const float* __restrict__ base, int ca, int cb) {
    if (cb == ca + 1) {
        const Quad8 q = *reinterpret_cast<const Quad8*>(base + (int64_t)ca * 2);
        a.v[0] = q.v[0]; a.v[1] = q.v[1]; b.v[0] = q.v[2]; b.v[1] = q.v[3];
    } else {
        a = load_vec<2>(base + (int64_t)ca * 2);
        b = load_vec<2>(base + (int64_t)cb * 2);
    }
}

// ---- sparse 3x3 patch ---------------------------------------------------------------
struct Patch {
    int t_lo, t_hi;        // t_hi used by forward_inter only
    float w_lo, w_hi;
    int vx[3], vy[3];
};

__device__ __forceinline__ Patch patch_setup(float t, float x, float y, const nvp_sparse_shape& sh, bool inter) {
    Patch p;
    int xi = nvp_nearest_idx(x, sh.x_res), yi = nvp_nearest_idx(y, sh.y_res);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        p.vx[d] = min(max(xi + d - 1, 0), sh.x_res - 1);
        p.vy[d] = min(max(yi + d - 1, 0), sh.y_res - 1);
    }
    if (!inter) {
        p.t_lo = p.t_hi = nvp_nearest_idx(t, sh.t_res);
        p.w_lo = 1.f; p.w_hi = 0.f;
    } else {
        // reference sparsegrid.py:98-109 (note: lc is divided by the UPDATED uc + lc)
        float tf = nvp_mul_rn((float)(sh.t_res - 1), t);
        int lo = (int)tf;
        int hi = min(max((int)nvp_add_rn(tf, 1.0f), 0), sh.t_res - 1);
        float uc = nvp_sub_rn(tf, (float)lo);
        float lc = nvp_sub_rn((float)hi, tf);
        uc = nvp_div_rn(uc, nvp_add_rn(uc, lc));
        lc = nvp_div_rn(lc, nvp_add_rn(uc, lc));
        p.t_lo = min(max(lo, 0), sh.t_res - 1);   // reference indexes E[lo] unclamped; lo is in range for t in [0,1]
        p.t_hi = hi;
        p.w_lo = lc; p.w_hi = uc;
    }
    return p;
}

