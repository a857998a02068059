// The forward MLP's per-tile body and its latent-chain helpers, shared by mlp_fwd_b3.hip (the forward kernels) and mlp_bwd_b3r.hip (the
// tile-fused forward + backward-chain kernel, nvp_encode_mlp_fwd_bwd).  See mlp_fwd_b3.hip for the description.
#pragma once
#ifndef NVP_SPLIT_ASM
#define NVP_SPLIT_ASM 2        // chain kernels: residuals of the fp16 x 2 split as v_fma_mix with op_sel (mlp_b3.h); -0.02 ms each, same bits
#endif
#include <cstring>
#include "mlp_b3.h"
#include "encode_tile.h"      // in-wave tile gather (FMA contraction off inside, restored after)

namespace {


// `ns` k-steps over the latent tile in LDS (PTM4: row-group rg = rows 4rg..4rg+3 of pixel j at zl[rg*32 + j]);
// step s, lane half h consumes rows 16 s + 8 h .. + 7 = row-groups 4s + 2h, 4s + 2h + 1
__device__ __forceinline__ void chain_z_b3_step(f32x16 (&acc)[4], const float4* __restrict__ zl, int s, const float sc, const u32x4* __restrict__ w, int j, int h, int lane) {
    const float4 t0 = zl[(4 * s + 2 * h) * 32 + j];
    const float4 t1 = zl[(4 * s + 2 * h + 1) * 32 + j];
    const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    BOp b;
    split8(x, sc, b);
    step_b3(acc, w + NVP_WSTRIDE(s * kB3StepQuads), b, lane);
}

// sc: the pixel's operand scale (mlp_b3.h; 1 for bf16 x 3).  post(s) (optional) runs after k-step s of the straight-line 8-step path only -
// independent VALU work the caller wants issued in the shadow of that step's MFMAs; returns true when post ran for every step.
template <typename Post>
__device__ __forceinline__ bool chain_z_b3(f32x16 (&acc)[4], const float4* __restrict__ zl, int ns, const float sc, const u32x4* __restrict__ w, int lane, Post post) {
    const int j = lane & 31, h = lane >> 5;
    if (ns == 8) {                               // nvp_s: straight-line code (wave-uniform branch): 1.854 vs 1.879 ms for the rolled loop
        NVP_CHAIN_ENTER();
#pragma unroll
        for (int s = 0; s < 8; ++s) { chain_z_b3_step(acc, zl, s, sc, w, j, h, lane); post(s); }
        NVP_CHAIN_LEAVE();
        return true;
    }
    NVP_CHAIN_ENTER();
#pragma unroll 1
    for (int s = 0; s < ns; ++s) chain_z_b3_step(acc, zl, s, sc, w, j, h, lane);
    NVP_CHAIN_LEAVE();
    return false;
}
__device__ __forceinline__ void chain_z_b3(f32x16 (&acc)[4], const float4* __restrict__ zl, int ns, const float sc, const u32x4* __restrict__ w, int lane) {
    chain_z_b3(acc, zl, ns, sc, w, lane, [](int) {});
}

// Stage this wave's latent tile into its LDS region (as stage_z, mlp_chain.h) and return the largest |z| this lane saw: every
// float4 a lane moves belongs to pixel lane & 31 (the tile is [row-group][32 px] and 64 divides every chunk offset), so the
// two lane halves' maxima combine to the pixel's.
__device__ __forceinline__ float stage_z_absmax(float4* __restrict__ zl, const float4* __restrict__ z4, int n4, int lane) {
    float m = 0.f;
    for (int base = 0; base < n4; base += 16 * 64) {          // <= 2 passes (rows <= 256)
        float4 tmp[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {                        // 16 independent 1-KiB wave loads in flight
            const int idx = base + k * 64 + lane;
            tmp[k] = idx < n4 ? z4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = base + k * 64 + lane;
            if (idx < n4) zl[idx] = tmp[k];
            m = absmax_f4(m, tmp[k]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return m;
}

// k-steps [s0, s1) straight from the latent tensor (wide latents: rows the LDS tile does not hold); row-groups at or
// beyond rg_end (the tensor's rows / 4) read as zero - the tile of the LAST pixels is followed by nothing
struct ZgRows { float4 t0, t1; };
__device__ __forceinline__ ZgRows zg_rows(const float4* __restrict__ zg, int s, int rg_end, int j, int h) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rg = 4 * s + 2 * h;
    ZgRows r;
    r.t0 = rg < rg_end ? zg[rg * 32 + j] : zero;
    r.t1 = rg + 1 < rg_end ? zg[(rg + 1) * 32 + j] : zero;
    return r;
}
__device__ __forceinline__ void chain_zg_b3(f32x16 (&acc)[4], const float4* __restrict__ zg, int s0, int s1, int rg_end, const float sc,
                                            const u32x4* __restrict__ w, int lane) {
    const int j = lane & 31, h = lane >> 5;
    NVP_CHAIN_ENTER();
#pragma unroll 1
    for (int s = s0; s < s1; ++s) {
        const ZgRows cur = zg_rows(zg, s, rg_end, j, h);
        const float x[8] = {cur.t0.x, cur.t0.y, cur.t0.z, cur.t0.w, cur.t1.x, cur.t1.y, cur.t1.z, cur.t1.w};
        BOp b;
        split8(x, sc, b);
        step_b3(acc, w + s * kB3StepQuads, b, lane);
    }
    NVP_CHAIN_LEAVE();
}

// GF = 0: the latent tile is staged from the tensor `zt` a gather kernel wrote.  GF = 2 / 4 (= features per level): the wave
// GATHERS its tile itself (encode_tile.h) - `zt` is then an OUTPUT, written only when SAVE (the dW GEMMs of the backward pass
// read it) and may be null otherwise.
// One 32-pixel tile, all seven layers.  `z`: this wave's LDS latent region; `active` false: a duplicate walk whose results are not stored.
// rgb_out: the tile pixel lane & 31's RGB (the tile-fused kernel derives the loss gradient from it).
// INTER (inference kernels only): the sparse grid's rows are SparseGrid.forward_inter's (encode_tile.h).
template <bool SAVE, int GF, bool INTER = false>
__device__ __forceinline__ void fwd_b3_tile(float* __restrict__ zt, const float* __restrict__ steps, const nvp_mlp_params& p, const unsigned* __restrict__ packed,
                                            float* __restrict__ rgb, float* __restrict__ saved, int64_t n, int64_t ntiles, int d, const NvpTileEnc& enc,
                                            int64_t tile, bool active, float4* __restrict__ z, int lane, float (&rgb_out)[3]) {
    const int j = lane & 31, h = lane >> 5;
    const NvpFwdLayoutB3 L = nvp_fwd_layout_b3(d);
    // latent tile -> this wave's LDS region: the first zs_l k-steps' rows, zero-padded to whole k-steps (the packed
    // weights are zero there, but 0 x garbage could be NaN); wide latents (nvp_l) read the remaining rows from the tensor
    const int zs_l = min(L.zs, kB3ZLdsSteps);
    const int zl4 = zs_l * 4 * 32;                           // float4 per wave in LDS (the caller's `z` region holds at least that)
    const int z4 = (nvp_rows4(d) / 4) * 32;
    const float4* zg = reinterpret_cast<const float4*>(zt) + tile * (int64_t)z4;
    float mz;                                                 // per-pixel max |z|: the latent's share of the operand scale
    // GF == 4 (config_nvp_l: 228 rows): the tile is WIDER than the wave's LDS region - its row-groups beyond kB3ZLdsSteps k-steps go straight into the
    // latent tensor (which the caller then always provides) and are read back from there by the chains, like the staged path's tail
    constexpr bool kWide = GF == 4;
    if (GF == 0) mz = stage_z_absmax(z, zg, min(z4, zl4), lane);
    else if (kWide) mz = nvp_gather_tile<(GF == 0 ? 2 : GF), kWide, INTER>(z, active ? reinterpret_cast<float4*>(zt) + tile * (int64_t)z4 : nullptr, enc, tile, n, lane, zs_l * 4, SAVE);
    else mz = nvp_gather_tile<(GF == 0 ? 2 : GF), false, INTER>(z, (SAVE && active) ? reinterpret_cast<float4*>(zt) + tile * (int64_t)z4 : nullptr, enc, tile, n, lane);
    if (kWide && zs_l < L.zs) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail rows this wave has just stored are read back below (a wave reading its own stores: the count suffices)
    for (int idx = z4 + lane; idx < zl4; idx += 64) z[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NVP_SPLIT_H2) {
        if (GF == 0) for (int idx = zl4 + lane; idx < z4; idx += 64) mz = absmax_f4(mz, zg[idx]);      // wide latents: the rows the LDS tile does not hold
        mz = fmaxf(mz, __shfl_xor(mz, 32));
    }
    const int rg_end = nvp_rows4(d) / 4;
    const u32x4* wp = reinterpret_cast<const u32x4*>(packed);
    const float* tab = reinterpret_cast<const float*>(packed + L.off[5]);      // sir_w0 / sir_b0 / last_w in D-register order
    const float* winv = tab + kB3ScaleOff + 8;                                 // 2^-e of each weight stream (mlp_layout.h)
    const int64_t px = tile * 32 + j;
    const float s = px < n ? steps[px] : 0.f;
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    float* sv = (SAVE && active) ? saved + tile * (int64_t)NVP_H * 32 : nullptr;

    f32x16 hm[4], x[4], acc[4];
    bool sir0_early = false;        // x already holds sin(30 (w s + c)) when the first chain could carry it (wave-uniform)

    // ---- modulator layer 0: h0 = lrelu(W0 z + b0)                 modulation.py:112-121
    {
        const u32x4* w = wp + NVP_WSTRIDE(L.off[0] / 4);
#pragma unroll
        for (int T = 0; T < 4; ++T) hm[T] = nvp_zero16();
        const PxScale ps = px_scale(fmaxf(mz, 1.0f));                // the bias (B = 1) shares the scale
        bias_b3(hm, w, ps.s, lane);
#ifndef NVP_FWD_SIR0_EARLY
#define NVP_FWD_SIR0_EARLY 1     // SIREN layer 0's sines - sin(30 (w s + c)), independent of everything the MLP has computed so far - are issued eight per
#endif                           // k-step in the shadow of this chain's MFMAs instead of as a serial 900-instruction VALU stage behind it (same bits)
        sir0_early = chain_z_b3(hm, z, zs_l, ps.s, w + NVP_WSTRIDE(kB3StepQuads), lane,
                   [&](int ks) {
                       if (!NVP_FWD_SIR0_EARLY) return;
                       const int T = ks >> 1, r0 = 8 * (ks & 1);
                       const float4* pw = reinterpret_cast<const float4*>(tab + ((0 * 2 + h) * 4 + T) * 16 + r0);
                       const float4* pc = reinterpret_cast<const float4*>(tab + ((1 * 2 + h) * 4 + T) * 16 + r0);
                       const float4 wa = pw[0], wb = pw[1], ca = pc[0], cb = pc[1];
                       const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                       const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
                       for (int q = 0; q < 8; ++q) x[T][r0 + q] = nvp_sin(30.0f * __fmaf_rn(s, wv[q], cv[q]));
                   }) && NVP_FWD_SIR0_EARLY;
        if (GF == 0 || GF == 4) chain_zg_b3(hm, zg, zs_l, L.zs, rg_end, ps.s, w + kB3StepQuads, lane);
        lrelu4_scaled(hm, ps.u * winv[0]);
#pragma unroll
        for (int T = 0; T < 4; ++T) nvp_pin(hm[T]);
        if (SAVE && active) store_ptm(sv + 0 * act, hm, lane);
    }
    // ---- SIREN layer 0: x0 = sin(30 (w s + c)) * h0                modulation.py:53-56,90
    {
        if (sir0_early) {
#pragma unroll
            for (int T = 0; T < 4; ++T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[T][r] = x[T][r] * hm[T][r];
                nvp_pin(x[T]);
            }
        } else {
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                float w0v[16], c0v[16];
                load_tab16(w0v, tab, 0, T, h);
                load_tab16(c0v, tab, 1, T, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float q = 30.0f * __fmaf_rn(s, w0v[r], c0v[r]);
                    x[T][r] = nvp_sin(q) * hm[T][r];
                }
                nvp_pin(x[T]);
                NVP_LOAD_FENCE();
            }
        }
    }
    // ---- layers 1 and 2
#pragma unroll
    for (int k = 1; k <= 2; ++k) {
        {   // modulator: h_k = lrelu(Wh h_{k-1} + Wz z + b)
            const u32x4* w = wp + NVP_WSTRIDE(L.off[k] / 4);
    #pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            const PxScale ps = px_scale(fmaxf(fmaxf(px_absmax(hm), mz), 1.0f));
            bias_b3(acc, w, ps.s, lane);
            chain_h_b3(acc, hm, ps.s, w + NVP_WSTRIDE(kB3StepQuads), lane);
            chain_z_b3(acc, z, zs_l, ps.s, w + NVP_WSTRIDE(9 * kB3StepQuads), lane);
            if (GF == 0 || GF == 4) chain_zg_b3(acc, zg, zs_l, L.zs, rg_end, ps.s, w + 9 * kB3StepQuads, lane);
            lrelu4_scaled(acc, ps.u * winv[k]);
#pragma unroll
            for (int T = 0; T < 4; ++T) { hm[T] = acc[T]; nvp_pin(hm[T]); }
            if (SAVE && active) store_ptm(sv + (int64_t)k * act, hm, lane);
        }
        {   // SIREN: q_k = V x_{k-1} + c ; x_k = sin(q_k) * h_k
            const u32x4* w = wp + NVP_WSTRIDE(L.off[2 + k] / 4);
    #pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            const PxScale ps = px_scale(fmaxf(px_absmax(x), 1.0f));
            bias_b3(acc, w, ps.s, lane);
            chain_h_b3(acc, x, ps.s, w + NVP_WSTRIDE(kB3StepQuads), lane);
            scale4(acc, ps.u * winv[2 + k]);
            if (SAVE && active) store_ptm(sv + (int64_t)(2 + k) * act, acc, lane);
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[T][r] = nvp_sin(acc[T][r]) * hm[T][r];
#pragma unroll
            for (int T = 0; T < 4; ++T) nvp_pin(x[T]);
        }
    }
    // ---- last layer (3 x 128, Identity): VALU dot products + cross-half add
    {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float v0[16], v1[16], v2[16];
            load_tab16(v0, tab, 2, T, h);
            load_tab16(v1, tab, 3, T, h);
            load_tab16(v2, tab, 4, T, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = x[T][r];
                o0 = __fmaf_rn(v0[r], v, o0);
                o1 = __fmaf_rn(v1[r], v, o1);
                o2 = __fmaf_rn(v2[r], v, o2);
            }
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
            NVP_LOAD_FENCE();
        }
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        rgb_out[0] = o0 + p.last_b[0]; rgb_out[1] = o1 + p.last_b[1]; rgb_out[2] = o2 + p.last_b[2];      // (both lane halves hold the pixel's RGB)
        if (active && h == 0 && px < n) {
            rgb[px * 3 + 0] = rgb_out[0];
            rgb[px * 3 + 1] = rgb_out[1];
            rgb[px * 3 + 2] = rgb_out[2];
        }
    }
}

// ---- R11 fused: coordinates -> RGB in ONE kernel (the gather runs inside the forward MLP's waves) --------------------------------
// Supported when the grids have 2 or 4 features per level; a latent wider than the wave's LDS tile (> 144 rows: config_nvp_l, F = 4) parks its tail in the latent tensor,
// every plane's rows start on a row-group boundary; nvp_encode_mlp_fwd_supported() tells a host.
// needs_tensor (optional): the latent is wider than the wave's LDS tile (config_nvp_l) - the caller must provide the latent tensor `zt` even
// for inference, the kernel parks the rows beyond kB3ZLdsSteps k-steps there.
bool fused_ok(const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh, int* d_out, bool* needs_tensor = nullptr) {
    if (!NVP_FWD_B3 || !lv_xy || !lv_yt || !lv_xt || !sh) return false;
    const int F = lv_xy->n_features;
    if (!(F == 2 || F == 4) || lv_yt->n_features != F || lv_xt->n_features != F || sh->n_features != F || sh->y_res < 3) return false;
    const nvp_levels* lv[3] = {lv_xy, lv_yt, lv_xt};
    int d = 0;
    for (int q = 0; q < 3; ++q) {
        if (lv[q]->flags != kTileFlags) return false;        // the in-wave gather is compiled for the default arithmetic variant
        if (lv[q]->n_levels < 1 || lv[q]->n_levels > NVP_MAX_LEVELS || (lv[q]->n_levels * F) % 8) return false;     // an even number of row-groups per plane
        d += lv[q]->n_levels * F;
    }
    d += 9 * F;
    const int zs = nvp_fwd_layout_b3(d).zs;
    if (!nvp_fwd_b3_ok(d)) return false;
    if (zs > kB3ZLdsSteps && F != 4) return false;           // only the F = 4 kernel is built with the tensor-parked tail
    if (needs_tensor) *needs_tensor = zs > kB3ZLdsSteps;
    if (d_out) *d_out = d;
    return true;
}


}  // namespace
