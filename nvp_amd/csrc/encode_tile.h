// In-wave tile gather: the encoding half of NVP.forward (modules.py:57-78: three tinycudann.Encoding planes + SparseGrid.forward /
// concatenated xy | yt | xt | sparse; INTER: SparseGrid.forward_inter, sparsegrid.py:76-156, eval.py --t_interp - inference kernels only) for the 32 pixels of ONE MLP tile, written straight into the wave's LDS latent
// tile - the forward MLP kernel (mlp_fwd_b3.hip) calls it instead of staging a latent that a separate gather kernel had to write to
// HBM first ("grid lookups fused with the modulated coordinate MLP").  Same helpers, same arithmetic, same order of operations as
// encode.hip's encode_fwd_kernel: the latent is BIT-identical.
//
// Work split: lane (j, h) owns pixel j of the tile and, of every plane, the row-groups (4 latent rows = one float4 per pixel in
// PTM4) with parity h: for F = 2 the level pairs {4q + 2h, 4q + 2h + 1}, for F = 4 the levels 2q + h.  The sparse 3x3 patch is
// split by patch x-row: h = 0 takes rows 0 and 1, h = 1 row 2 and the zero padding.  Requirements (checked on the host, otherwise
// the two-kernel path runs): F in {2, 4} for all four grids, n_levels a multiple of 4 / F * 2, sparse y_res >= 3.
//
// FMA contraction is OFF inside this header (index arithmetic must keep the reference's separately rounded multiply / add,
// grid_math.h) and restored to the HIP default afterwards.
#pragma once
#pragma clang fp contract(off)
#include "encode_gather.h"

struct NvpTileEnc {
    nvp_levels lv[3];          // latent order: xy, yt, xt
    nvp_sparse_shape sh;
    const float* kf[3];
    const float* emb;
    const float* coords;       // [n][3] (t, x, y)
    int col0[4];               // first latent row of xy, yt, xt, sparse
    int rows;                  // PTM4 rows of the latent (D rounded up to 4)
};

// a uniform table entry through a scalar register (a per-lane select between two kernel-argument entries otherwise becomes ONE
// load with a per-lane address: a dependent memory access in front of every gather)
__device__ __forceinline__ float sgpr_f(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ int sgpr_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The arithmetic variant the fused path is compiled for: the default ("tcnn": fmaf position, fma-chain blend, wrapping border).
// Encodings configured otherwise take the two-kernel path (nvp_encode_mlp_fwd_supported).
constexpr int kTileFlags = NVP_GRID_POS_FMA | NVP_GRID_INTERP_FMA;

// Where a row-group of the tile goes: the wave's LDS tile holds row-groups [0, rg_lds) (kB3ZLdsSteps k-steps = 144 rows: all of a config_nvp_s
// latent); row-groups beyond it - config_nvp_l's 228-row latent - are stored straight into the latent tensor's tile `zg`, from where the MLP
// chains read them back (chain_zg_b3), so that the gather can run inside the MLP's waves for wide latents too.
// WIDE false (config_nvp_s): every row-group goes to LDS - no test, the code of the narrow kernels is unchanged.
template <bool WIDE>
struct TileDst {
    float4* __restrict__ zl;       // LDS tile
    float4* __restrict__ zg;       // this tile of the latent tensor (WIDE only)
    int rg_lds;
    __device__ __forceinline__ void put(int rg, int j, const float4 v) const {
        if (!WIDE || rg < rg_lds) zl[rg * 32 + j] = v;
        else zg[rg * 32 + j] = v;
    }
};

// one plane, this lane's row-groups -> tile; returns the largest |value| written.
// Straight-line code for memory-level parallelism (a wave of the MLP kernel has one partner on its SIMD, not seven, to hide a
// miss behind): level geometry through scalar registers with compile-time indices, the variant flags folded at compile time,
// the cell wrap as two selects (exact for cells within one level size of the level - any coordinate in [-1, 2]), the two corners
// of a grid row as ONE 16-byte fetch; all 16 fetches of the plane are issued before the first blend.  The rare cases the
// selects cannot express (a corner pair split by the wrap, far-away coordinates) are redone exactly, for the whole wave, behind one
// wave-uniform branch.  Pixels past the end of the batch gather at coordinate 0 and write zeros: no load sits behind a branch.
template <int F, bool WIDE>
__device__ __forceinline__ float tile_plane(const TileDst<WIDE>& dst, const float* __restrict__ params, const nvp_levels& lv,
                                            int col0, float x0, float x1, bool valid, int j, int h) {
    constexpr int LPG = 4 / F;                       // levels per row-group
    constexpr int NQ = NVP_MAX_LEVELS / LPG / 2;     // row-groups per lane and plane at 16 levels
    constexpr int NL = NQ * LPG;                     // levels per lane (8)
    const int nlev = lv.n_levels;
    float w[NL][4];
    Vec<F> v[NL][4];
    bool redo = false;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int q = k / LPG, dl = k % LPG;
        const int lA = min(min((2 * q) * LPG + dl, NVP_MAX_LEVELS - 1), nlev - 1);          // lane half 0 (uniform)
        const int lB = min(min((2 * q + 1) * LPG + dl, NVP_MAX_LEVELS - 1), nlev - 1);      // lane half 1
        // both halves' entries in scalar registers FIRST (uniform code), then one select each: a select whose arms contain the
        // reads turns into divergent control flow with a scalar-load wait in every arm
        const float sA = sgpr_f(lv.scale[lA]), sB = sgpr_f(lv.scale[lB]);
        const int rA = sgpr_i(lv.res[lA]), rB = sgpr_i(lv.res[lB]);
        const int oA = sgpr_i(lv.offset[lA]), oB = sgpr_i(lv.offset[lB]);
        const float scale = h ? sB : sA;
        const int res = h ? rB : rA;
        const int off = h ? oB : oA;
        // == nvp_bilerp_setup(x0, x1, scale, res, kTileFlags), with the cell wrap as selects
        const float p0 = nvp_grid_pos(x0, scale, kTileFlags), p1 = nvp_grid_pos(x1, scale, kTileFlags);
        const float f0 = floorf(p0), f1 = floorf(p1);
        const float w0 = nvp_sub_rn(p0, f0), w1 = nvp_sub_rn(p1, f1);
        const float u0 = nvp_sub_rn(1.0f, w0), u1 = nvp_sub_rn(1.0f, w1);
        w[k][0] = nvp_mul_rn(u0, u1); w[k][1] = nvp_mul_rn(w0, u1); w[k][2] = nvp_mul_rn(u0, w1); w[k][3] = nvp_mul_rn(w0, w1);
        const int i0 = (int)f0, i1 = (int)f1;
        const int size = res * res;
        int c[4];
        const int c00 = i0 + i1 * res;
        c[0] = c00; c[1] = c00 + 1; c[2] = c00 + res; c[3] = c00 + res + 1;
        bool far = false;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            far |= c[cc] < -size || c[cc] >= 2 * size;
            c[cc] -= c[cc] >= size ? size : 0;
            c[cc] += c[cc] < 0 ? size : 0;
        }
        const float* base = params + (int64_t)off * F;
        if constexpr (F == 2) {
            const bool nb = (c[1] == c[0] + 1) && (c[3] == c[2] + 1);
            redo |= far || !nb;
            const int a0 = far ? 0 : min(c[0], size - 2), a1 = far ? 0 : min(c[2], size - 2);      // in bounds whatever happens
            const Quad8 t0 = *reinterpret_cast<const Quad8*>(base + (int64_t)a0 * 2);
            const Quad8 t1 = *reinterpret_cast<const Quad8*>(base + (int64_t)a1 * 2);
            v[k][0].v[0] = t0.v[0]; v[k][0].v[1] = t0.v[1]; v[k][1].v[0] = t0.v[2]; v[k][1].v[1] = t0.v[3];
            v[k][2].v[0] = t1.v[0]; v[k][2].v[1] = t1.v[1]; v[k][3].v[0] = t1.v[2]; v[k][3].v[1] = t1.v[3];
        } else {
            redo |= far;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v[k][cc] = load_vec<F>(base + (int64_t)(far ? 0 : c[cc]) * F);
        }
    }
    if (__any(redo)) {                                // wave-uniform, rare (border pixels): the generic, exact path for every level
#pragma unroll 1
        for (int k = 0; k < NL; ++k) {
            const int q = k / LPG, dl = k % LPG;
            const int l = min(min((2 * q + h) * LPG + dl, NVP_MAX_LEVELS - 1), nlev - 1);
            const NvpBilerp b = nvp_bilerp_setup(x0, x1, lv.scale[l], lv.res[l], kTileFlags);
            const float* base = params + (int64_t)lv.offset[l] * F;
            Vec<F> t[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) t[cc] = load_vec<F>(base + (int64_t)b.cell[cc] * F);
#pragma unroll
            for (int kk = 0; kk < NL; ++kk)          // (static register indices)
                if (kk == k) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { v[kk][cc] = t[cc]; w[kk][cc] = b.w[cc]; }
                }
        }
    }
    // ---- blend + write
    float m = 0.f;
    const int ngroups = nlev / LPG;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = 2 * q + h;
        float r[4];
#pragma unroll
        for (int dl = 0; dl < LPG; ++dl) {
            const int k = q * LPG + dl;
            const bool live = valid && (g * LPG + dl) < nlev;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float t = nvp_blend4(w[k], v[k][0].v[f], v[k][1].v[f], v[k][2].v[f], v[k][3].v[f], kTileFlags);
                r[dl * F + f] = live ? t : 0.f;
            }
        }
        if (2 * q < ngroups) {                        // (uniform: an even number of row-groups per plane)
            const float4 o = make_float4(r[0], r[1], r[2], r[3]);
            dst.put((col0 >> 2) + g, j, o);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
        }
    }
    return m;
}

// the sparse 3x3 patch (sparse_fwd_ptm of encode.hip, split over the two lanes of a pixel), in two phases so that its fetches
// - the longest misses of the tile: a 432 MB grid - are in flight while the three planes are gathered.  Lane half 0 owns patch
// x-rows 0 and 1, lane half 1 row 2 (its second slot repeats row 2: same address, no branch).
template <int F>
struct SparseFetch { float lo[2][3 * F]; int sel[3]; };

// INTER (SparseGrid.forward_inter): the same rows of the t_hi slice too, blended as sparse_fwd_ptm (encode.hip) does - lo * w_lo + hi * w_hi,
// separately rounded; at t == 1 the reference's weights are 0 / 0 and the row is NaN (the quirk eval.py --t_interp's last frame hits): kept.
template <int F, bool INTER = false>
__device__ __forceinline__ void tile_sparse_fetch(SparseFetch<F>& sf, const NvpTileEnc& a, float t, float x, float y, int h) {
    const Patch p = patch_setup(t, x, y, a.sh, INTER);
    const int y0 = min(max(p.vy[1] - 1, 0), a.sh.y_res - 3);
    const int64_t plane = (int64_t)a.sh.x_res * a.sh.y_res;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) sf.sel[jj] = p.vy[jj] - y0;          // 0, 1 or 2
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int vx = h ? p.vx[2] : p.vx[s2];
        const int64_t cell = (int64_t)vx * a.sh.y_res + y0;
        const float* pl = a.emb + ((int64_t)p.t_lo * plane + cell) * F;
        if constexpr (F == 2) {
            const Hex8 q = *reinterpret_cast<const Hex8*>(pl);
#pragma unroll
            for (int c = 0; c < 6; ++c) sf.lo[s2][c] = q.v[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3 * F; ++c) sf.lo[s2][c] = pl[c];
        }
        if constexpr (INTER) {
            const float* ph = a.emb + ((int64_t)p.t_hi * plane + cell) * F;
            float hi[3 * F];
#pragma unroll
            for (int c = 0; c < 3 * F; ++c) hi[c] = ph[c];
#pragma unroll
            for (int c = 0; c < 3 * F; ++c) sf.lo[s2][c] = nvp_add_rn(nvp_mul_rn(sf.lo[s2][c], p.w_lo), nvp_mul_rn(hi[c], p.w_hi));
        }
    }
}

template <int F, bool WIDE>
__device__ __forceinline__ float tile_sparse_write(const TileDst<WIDE>& dst, const SparseFetch<F>& sf, const NvpTileEnc& a, bool valid, int j, int h) {
    constexpr int NV = 9 * F;
    constexpr int NVP4 = (NV + 3) & ~3;
    constexpr int Q0 = (6 * F) / 4;                  // float4 of patch rows 0 and 1 (12 or 24 floats): lane half 0
    constexpr int Q1 = NVP4 / 4 - Q0;                // float4 of patch row 2 + padding: lane half 1
    // this lane's values in patch order: half 0: rows 0, 1 (2 x 3F floats); half 1: row 2 (3F floats) then zeros
    float u[2 * 3 * F + 4];
#pragma unroll
    for (int c = 0; c < 2 * 3 * F + 4; ++c) u[c] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float val = sf.sel[jj] == 0 ? sf.lo[s2][f] : (sf.sel[jj] == 1 ? sf.lo[s2][F + f] : sf.lo[s2][2 * F + f]);
                u[(s2 * 3 + jj) * F + f] = (s2 == 1 && h) ? 0.f : val;              // half 1 has only one row
            }
    float m = 0.f;
    const int rg0 = a.col0[3] >> 2;
#pragma unroll
    for (int q = 0; q < Q0; ++q) {                   // half 0 writes row-groups [0, Q0), half 1 [Q0, Q0 + Q1)
        const bool mine = h == 0 || q < Q1;
        const int rg = rg0 + (h ? Q0 + q : q);
        const float4 o = valid ? make_float4(u[4 * q], u[4 * q + 1], u[4 * q + 2], u[4 * q + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine && 4 * rg < a.rows) {
            dst.put(rg, j, o);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    return m;
}

// Fill the wave's latent tile for tile `tile`: row-groups [0, rg_lds) into zl ([.][32] float4, PTM4), row-groups beyond it into the latent
// tensor's tile zg (TileDst).  copy_lds: also write the LDS part to zg (training: the dW GEMMs read the whole latent) - from LDS, after the
// gather, so that no fetch waits for a store's data registers.  zg may be null when rg_lds covers the tile and copy_lds is false.
// Returns this lane's largest |z| (combine the two lane halves for the pixel's).
template <int F, bool WIDE = false, bool INTER = false>
__device__ __forceinline__ float nvp_gather_tile(float4* __restrict__ zl, float4* __restrict__ zg, const NvpTileEnc& a, int64_t tile, int64_t n, int lane,
                                                 int rg_lds = 1 << 20, bool copy_lds = true) {
    const int j = lane & 31, h = lane >> 5;
    const int64_t px = tile * 32 + j;
    const bool valid = px < n;
    float t = 0.f, x = 0.f, y = 0.f;
    if (valid) { const float* c = a.coords + px * 3; t = c[0]; x = c[1]; y = c[2]; }
    const TileDst<WIDE> dst = {zl, zg, rg_lds};
    SparseFetch<F> sf;
    tile_sparse_fetch<F, INTER>(sf, a, t, x, y, h);
    float m = tile_plane<F, WIDE>(dst, a.kf[2], a.lv[2], a.col0[2], t, x, valid, j, h);          // xt plane <- (t, x)   modules.py:62
    m = fmaxf(m, tile_plane<F, WIDE>(dst, a.kf[0], a.lv[0], a.col0[0], x, y, valid, j, h));      // xy plane <- (x, y)   modules.py:61
    m = fmaxf(m, tile_plane<F, WIDE>(dst, a.kf[1], a.lv[1], a.col0[1], t, y, valid, j, h));      // yt plane <- (t, y)   modules.py:63
    m = fmaxf(m, tile_sparse_write<F, WIDE>(dst, sf, a, valid, j, h));
    if (zg && copy_lds) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n4 = (WIDE ? min(a.rows >> 2, rg_lds) : (a.rows >> 2)) * 32;
        for (int idx = lane; idx < n4; idx += 64) zg[idx] = zl[idx];
    }
    return m;
}

#pragma clang fp contract(fast)
