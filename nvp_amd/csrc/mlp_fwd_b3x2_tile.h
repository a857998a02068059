// Forward MLP for a PAIR of 32-pixel tiles per wave, ONE wave per SIMD: instruction-level instead of thread-level
// parallelism.  The two tiles walk the seven layers in lockstep and share every weight fetch - one (hi, lo) operand quad pair feeds
// 3 + 3 MFMAs - which halves the L2 -> L1 weight stream of the one-tile kernel (every wave re-streams all 442 KB of packed weights per
// tile: 17 GB per launch, ~0.5 ms of the CUs' 64 B/clk vector-memory return path) and doubles the MFMA time a fetch has to cover.  With
// 512 registers a wave holds both tiles' layer state (2 x 3 fragments of 64 registers) next to a weight pipeline that runs NVP_X2_DEPTH
// operand groups ahead ACROSS layer boundaries (the packed buffer is stored in consumption order: one linear walk), so the element-wise
// stage of a layer is covered by fetches already in flight.  Independent VALU work is placed in the shadow of the MFMA chains by hand:
// the operand split of k-step c + 1 under the MFMAs of k-step c, SIREN layer 0's sines under the first latent chain, the sines of q1
// under the second modulator layer-2 chain (h2 needs h1 and z only).
// Same arithmetic, same order of operations per pixel as fwd_b3_tile (mlp_fwd_b3_tile.h): RGB, saved streams and latent are BIT-identical.
#pragma once
#include "mlp_fwd_b3_tile.h"

#ifndef NVP_X2_DEPTH
#define NVP_X2_DEPTH 3          // weight groups (one output tile of one k-step = kP operand quads) in flight ahead of the group being multiplied
#endif

namespace {

// eight consecutive D registers of fragment tile c >> 1 -> split operand (k-step c of an h-chain)
__device__ __forceinline__ void x2_split_h(BOp& b, const f32x16 (&hin)[4], int c, float s) {
    float v[8];
    chain_in8(v, hin, c);
    split8(v, s, b);
}
// k-step s of the latent tile in LDS -> split operand
__device__ __forceinline__ void x2_split_z(BOp& b, const float4* __restrict__ zl, int s, float sc, int j, int h) {
    const float4 t0 = zl[(4 * s + 2 * h) * 32 + j];
    const float4 t1 = zl[(4 * s + 2 * h + 1) * 32 + j];
    const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    split8(v, sc, b);
}

// ZS: latent k-steps (8 for config_nvp_s: 114 rows); the whole latent of both tiles lives in the wave's LDS region (2 x ZS x 2 KiB).
// tileA / tileB: the pair (tileB == tileA with activeB false: a duplicate walk whose results are not stored).
template <bool SAVE, int GF, int ZS>
__device__ __forceinline__ void fwd_b3_pair(float* __restrict__ zt, const float* __restrict__ steps, const nvp_mlp_params& p, const unsigned* __restrict__ packed,
                                            float* __restrict__ rgb, float* __restrict__ saved, int64_t n, int64_t ntiles, int d, const NvpTileEnc& enc,
                                            int64_t tileA, bool activeB, float4* __restrict__ zA, float4* __restrict__ zB, int lane) {
    constexpr int DPT = NVP_X2_DEPTH;
    constexpr int NB = DPT + 1;
    // groups of the linear weight walk: layer order mod0, mod1, sir1, mod2, sir2 (nvp_fwd_layout_b3); a k-step is four groups
    constexpr int G_MOD0 = 0, G_MOD1 = 4 * (1 + ZS), G_SIR1 = G_MOD1 + 4 * (9 + ZS), G_MOD2 = G_SIR1 + 4 * 9, G_SIR2 = G_MOD2 + 4 * (9 + ZS), G_END = G_SIR2 + 4 * 9;
    const int j = lane & 31, h = lane >> 5;
    const NvpFwdLayoutB3 L = nvp_fwd_layout_b3(d);
    const int64_t tile[2] = {tileA, activeB ? tileA + 1 : tileA};
    const bool active[2] = {true, activeB};
    float4* const z[2] = {zA, zB};
    constexpr int zl4 = ZS * 4 * 32;
    const int z4 = (nvp_rows4(d) / 4) * 32;

    // ---- weight pipeline: group g -> operand quads wq[g % NB]; requested DPT groups ahead of their use
    u32x4 wq[NB][kP];
    const u32x4* wl = reinterpret_cast<const u32x4*>(packed) + (unsigned)lane;
    static_assert(ZS >= 8 && (4 * (1 + ZS)) % NB == 0 && (4 * (18 + ZS)) % NB == 0, "the pending sines ride in 16 k-steps of a modulator layer; the rolled layer loop needs its group count to be a multiple of the pipeline's buffers");
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int gl = g < G_END ? g : G_END - 1;          // beyond the end of the walk: the last group again (no branch; the registers are dead)
#pragma unroll
        for (int k = 0; k < kP; ++k) wq[g % NB][k] = wl[(NVP_WSTRIDE(gl) * kP + k) * 64];
    };
    // one group for both tiles: the group's weights times each tile's B operand
    auto grp = [&](int g, f32x16& a0, f32x16& a1, const BOp& b0, const BOp& b1) __attribute__((always_inline)) {
        issue(g + DPT);
        NVP_CHAIN_FENCE();
        mac_parts(a0, wq[g % NB], b0);
        mac_parts(a1, wq[g % NB], b1);
    };
    auto grp_bias = [&](int g, f32x16& a0, f32x16& a1, const u32x4 e0, const u32x4 e1) __attribute__((always_inline)) {
        issue(g + DPT);
        NVP_CHAIN_FENCE();
        bias_mac(a0, wq[g % NB], e0);
        bias_mac(a1, wq[g % NB], e1);
    };

    // ---- gather both latent tiles (encode_tile.h) into the wave's LDS regions; the first weight groups are requested behind the gathers' fetches
    float mz[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
        mz[t] = nvp_gather_tile<GF>(z[t], (SAVE && active[t]) ? reinterpret_cast<float4*>(zt) + tile[t] * (int64_t)z4 : nullptr, enc, tile[t], n, lane);
#pragma unroll
    for (int g = 0; g < DPT; ++g) issue(g);
#pragma unroll
    for (int t = 0; t < 2; ++t)
        for (int idx = z4 + lane; idx < zl4; idx += 64) z[t][idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t) mz[t] = fmaxf(mz[t], __shfl_xor(mz[t], 32));

    const float* tab = reinterpret_cast<const float*>(packed + L.off[5]);      // sir_w0 / sir_b0 / last_w in D-register order
    const float* winv = tab + kB3ScaleOff + 8;                                 // 2^-e of each weight stream (mlp_layout.h)
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    int64_t px[2];
    float st[2];
    float* sv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        px[t] = tile[t] * 32 + j;
        st[t] = px[t] < n ? steps[px[t]] : 0.f;
        sv[t] = (SAVE && active[t]) ? saved + tile[t] * (int64_t)NVP_H * 32 : nullptr;
    }

    f32x16 hm[2][4], x[2][4], acc[2][4];

    // A chain of NS k-steps for both tiles: `src(t, c, b)` produces tile t's operand of k-step c; the operands of k-step c + 1 are
    // produced under the MFMAs of k-step c (tile 0's behind group 0, tile 1's behind group 1), `fill(c, T)` is further independent work
    // for the shadow of group (c, T).
#define NVP_X2_CHAIN(G0, NS, OUT, SRC, FILL)                                                                     \
    {                                                                                                            \
        BOp b_[2], bn_[2];                                                                                       \
        SRC(0, 0, b_[0]);                                                                                        \
        SRC(1, 0, b_[1]);                                                                                        \
        _Pragma("unroll") for (int c = 0; c < (NS); ++c) {                                                       \
            _Pragma("unroll") for (int T = 0; T < 4; ++T) {                                                      \
                grp((G0) + 4 * c + T, OUT[0][T], OUT[1][T], b_[0], b_[1]);                                       \
                if (T < 2 && c + 1 < (NS)) { SRC(T, c + 1, bn_[T]); }                                            \
                FILL(c, T);                                                                                      \
            }                                                                                                    \
            if (c + 1 < (NS)) { b_[0] = bn_[0]; b_[1] = bn_[1]; }                                                \
        }                                                                                                        \
    }
#define NVP_X2_BIAS(G0, OUT, PS)                                                                                 \
    {                                                                                                            \
        const u32x4 e0_ = bias_bop(PS[0].s, lane), e1_ = bias_bop(PS[1].s, lane);                                \
        _Pragma("unroll") for (int T = 0; T < 4; ++T) grp_bias((G0) + T, OUT[0][T], OUT[1][T], e0_, e1_);        \
    }
#define NVP_X2_NOFILL(c, T)

#define NVP_X2_FILL_SIN(cc, T)                                                                                  \
    {   /* two pending sines of one tile per group: 16 k-steps x 4 groups cover 2 x 64 values */                 \
        const int Tp_ = (cc) >> 2, r0_ = 4 * ((cc) & 3) + ((T) & 1) * 2, tt_ = (T) >> 1;                         \
        x[tt_][Tp_][r0_] = nvp_sin(x[tt_][Tp_][r0_]);                                                            \
        x[tt_][Tp_][r0_ + 1] = nvp_sin(x[tt_][Tp_][r0_ + 1]);                                                    \
    }
#define NVP_X2_FILL_SIN_H(c, T) NVP_X2_FILL_SIN(c, T)
#define NVP_X2_FILL_SIN_Z(c, T) if ((c) < 8) NVP_X2_FILL_SIN(8 + (c), T)

    NVP_CHAIN_ENTER();
    // ---- modulator layer 0: h0 = lrelu(W0 z + b0)                                                  modulation.py:112-121
    {
        PxScale ps[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ps[t] = px_scale(fmaxf(mz[t], 1.0f));
#pragma unroll
            for (int T = 0; T < 4; ++T) hm[t][T] = nvp_zero16();
        }
        NVP_X2_BIAS(G_MOD0, hm, ps);
#define SRC_Z0(t, c, b) x2_split_z(b, z[t], c, ps[t].s, j, h)
        NVP_X2_CHAIN(G_MOD0 + 4, ZS, hm, SRC_Z0, NVP_X2_NOFILL);
#undef SRC_Z0
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            lrelu4_scaled(hm[t], ps[t].u * winv[0]);
            if (sv[t]) store_ptm(sv[t] + 0 * act, hm[t], lane);
        }
        // SIREN layer 0's ARGUMENTS 30 (w s + c) (modulation.py:53-56): their sines ride in the shadow of modulator layer 1's chain below
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float w0v[16], c0v[16];
            load_tab16(w0v, tab, 0, T, h);
            load_tab16(c0v, tab, 1, T, h);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[t][T][r] = 30.0f * __fmaf_rn(st[t], w0v[r], c0v[r]);
        }
    }
    // ---- layers 1 and 2: ONE loop body (kept rolled: half the code).  Entering iteration k, hm = h_{k-1} and x = the ARGUMENT of the sine of
    // SIREN layer k-1 (k = 1: 30 (w s + c); k = 2: q1); the sines are taken inside modulator layer k's chain - h_k needs h_{k-1} and z only -
    // and x_{k-1} = sin(.) * h_{k-1} (modulation.py:90) is completed behind it, just before SIREN layer k consumes it.
#pragma unroll 1
    for (int k = 1; k <= 2; ++k) {
        const int g_mod = k == 1 ? G_MOD1 : G_MOD2, g_sir = k == 1 ? G_SIR1 : G_SIR2;
        {   // modulator: h_k = lrelu(Wh h_{k-1} + Wz z + b)
            PxScale ps[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ps[t] = px_scale(fmaxf(fmaxf(px_absmax(hm[t]), mz[t]), 1.0f));
#pragma unroll
                for (int T = 0; T < 4; ++T) acc[t][T] = nvp_zero16();
            }
            NVP_X2_BIAS(g_mod, acc, ps);
#define SRC_H(t, c, b) x2_split_h(b, hm[t], c, ps[t].s)
#define SRC_Z(t, c, b) x2_split_z(b, z[t], c, ps[t].s, j, h)
            NVP_X2_CHAIN(g_mod + 4, 8, acc, SRC_H, NVP_X2_FILL_SIN_H);
            NVP_X2_CHAIN(g_mod + 4 * 9, ZS, acc, SRC_Z, NVP_X2_FILL_SIN_Z);
#undef SRC_H
#undef SRC_Z
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int T = 0; T < 4; ++T)
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[t][T][r] = x[t][T][r] * hm[t][T][r];          // x_{k-1} = sin(.) * h_{k-1}
                lrelu4_scaled(acc[t], ps[t].u * winv[k]);
#pragma unroll
                for (int T = 0; T < 4; ++T) hm[t][T] = acc[t][T];
                if (sv[t]) store_ptm(sv[t] + (int64_t)k * act, hm[t], lane);
            }
        }
        {   // SIREN: q_k = V x_{k-1} + c
            PxScale ps[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ps[t] = px_scale(fmaxf(px_absmax(x[t]), 1.0f));
#pragma unroll
                for (int T = 0; T < 4; ++T) acc[t][T] = nvp_zero16();
            }
            NVP_X2_BIAS(g_sir, acc, ps);
#define SRC_X(t, c, b) x2_split_h(b, x[t], c, ps[t].s)
            NVP_X2_CHAIN(g_sir + 4, 8, acc, SRC_X, NVP_X2_NOFILL);
#undef SRC_X
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                scale4(acc[t], ps[t].u * winv[2 + k]);
                if (sv[t]) store_ptm(sv[t] + (int64_t)(2 + k) * act, acc[t], lane);
#pragma unroll
                for (int T = 0; T < 4; ++T) x[t][T] = acc[t][T];                                  // q_k: the next sine's argument
            }
        }
    }
    NVP_CHAIN_LEAVE();
    // x2 = sin(q2) * h2: nothing left to hide these sines under
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[t][T][r] = nvp_sin(x[t][T][r]) * hm[t][T][r];
#undef NVP_X2_FILL_SIN
#undef NVP_X2_FILL_SIN_H
#undef NVP_X2_FILL_SIN_Z
#undef NVP_X2_CHAIN
#undef NVP_X2_BIAS
#undef NVP_X2_NOFILL
    // ---- last layer (3 x 128, Identity): VALU dot products + cross-half add
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float v0[16], v1[16], v2[16];
            load_tab16(v0, tab, 2, T, h);
            load_tab16(v1, tab, 3, T, h);
            load_tab16(v2, tab, 4, T, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = x[t][T][r];
                o0 = __fmaf_rn(v0[r], v, o0);
                o1 = __fmaf_rn(v1[r], v, o1);
                o2 = __fmaf_rn(v2[r], v, o2);
            }
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
        }
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        if (active[t] && h == 0 && px[t] < n) {
            rgb[px[t] * 3 + 0] = o0 + p.last_b[0];
            rgb[px[t] * 3 + 1] = o1 + p.last_b[1];
            rgb[px[t] * 3 + 2] = o2 + p.last_b[2];
        }
    }
}

}  // namespace
