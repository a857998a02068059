// MFMA chain helpers shared by the forward and backward MLP kernels (see mlp_fwd.hip for
// the design: pixel on the MFMA column axis, D registers of one layer = B operands of the next).
#pragma once
#include "mlp_layout.h"

#ifdef NVP_ABL_NOLOAD      // ablation: weight operand from a register instead of memory
#define NVP_WLOAD(ptr_expr) make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f)
#else
#define NVP_WLOAD(ptr_expr) (ptr_expr)
#endif
#if defined(NVP_ABL_NOLOAD) || defined(NVP_ABL_NOZ)   // ablation: latent operand from a register
#define NVP_ZLOAD(ptr_expr) 0.25f
#else
#define NVP_ZLOAD(ptr_expr) (ptr_expr)
#endif

__device__ __forceinline__ void mfma4(f32x16 (&acc)[4], const float4 a, const float b) {
    acc[0] = nvp_mfma(a.x, b, acc[0]);
    acc[1] = nvp_mfma(a.y, b, acc[1]);
    acc[2] = nvp_mfma(a.z, b, acc[2]);
    acc[3] = nvp_mfma(a.w, b, acc[3]);
}

// Operand loads are software-pipelined by hand in groups of G k-steps (double-buffered in
// registers); the empty asm with a memory clobber stops hipcc from hoisting every load of
// the unrolled chain to the top (which costs >100 VGPRs and spills).
#ifndef NVP_G
#define NVP_G 4
#endif
constexpr int G = NVP_G;
#define NVP_LOAD_FENCE() asm volatile("" ::: "memory")
// Inside the MFMA chains the fence must also pin the MFMAs: hipcc otherwise hoists the (register-only)
// MFMAs of group g above it and sinks the loads of group g+1 next to their consumers, which shrinks the
// prefetch distance to ONE k-step (256 cycles < L2 latency; seen as `s_waitcnt vmcnt(1)` before every
// MFMA quad, MFMA pipe 62 % busy).  sched_barrier(0) lets nothing cross.
#ifdef NVP_ABL_NOSB
#define NVP_CHAIN_FENCE() NVP_LOAD_FENCE()
#else
#define NVP_CHAIN_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// 64 chained steps: B operands are the previous layer's D registers.  GG = k-steps per
// prefetch group (register cost 8*GG for the double buffer).
template <int GG = G>
__device__ __forceinline__ void chain_h(f32x16 (&acc)[4], const f32x16 (&hin)[4], const float4* __restrict__ wp, int lane) {
    constexpr int G = GG;
    float4 a[2][G];
    // wp is wave-uniform (SGPR base); lane is the only per-lane part -> saddr + voffset + imm loads
    const unsigned ul = (unsigned)lane;
#pragma unroll
    for (int i = 0; i < G; ++i) a[0][i] = NVP_WLOAD((wp + i * 64)[ul]);
#pragma unroll
    for (int g = 0; g < 64 / G; ++g) {
        if (g + 1 < 64 / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) a[(g + 1) & 1][i] = NVP_WLOAD((wp + ((g + 1) * G + i) * 64)[ul]);
        }
        NVP_CHAIN_FENCE();
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int step = g * G + i;
            mfma4(acc, a[g & 1][i], hin[step >> 4][step & 15]);
        }
    }
}

// zs steps over the latent: step u consumes rows 2u (lane half 0) and 2u+1 (half 1).
__device__ __forceinline__ void chain_z(f32x16 (&acc)[4], const float* __restrict__ z, int zs,
                                        const float4* __restrict__ wp, int lane) {
    // z and wp are wave-uniform; row 2u+h of the PTM latent sits at (2u+h)*32 + j = u*64 + lane
    const unsigned ul = (unsigned)lane;
    const float* zl = z;
    const float4* w = wp;
    const int ng = zs / G;
    float4 a[G], an[G];
    float b[G], bn[G];
    if (ng > 0) {
#pragma unroll
        for (int i = 0; i < G; ++i) { a[i] = NVP_WLOAD((w + i * 64)[ul]); b[i] = NVP_ZLOAD((zl + i * 64)[ul]); }
    }
    for (int g = 0; g < ng; ++g) {
        if (g + 1 < ng) {
            const float4* wn = w + (g + 1) * G * 64;      // scalar pointer bumps
            const float* zn = zl + (g + 1) * G * 64;
#pragma unroll
            for (int i = 0; i < G; ++i) { an[i] = NVP_WLOAD((wn + i * 64)[ul]); bn[i] = NVP_ZLOAD((zn + i * 64)[ul]); }
        }
        NVP_CHAIN_FENCE();
#pragma unroll
        for (int i = 0; i < G; ++i) mfma4(acc, a[i], b[i]);
#pragma unroll
        for (int i = 0; i < G; ++i) { a[i] = an[i]; b[i] = bn[i]; }
    }
    for (int u = ng * G; u < zs; ++u) mfma4(acc, NVP_WLOAD((w + u * 64)[ul]), NVP_ZLOAD((zl + u * 64)[ul]));
}

__device__ __forceinline__ void lrelu4(f32x16 (&v)[4]) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[T][r] = v[T][r] > 0.f ? v[T][r] : v[T][r] * 0.01f;
}

// PTM addressing with a wave-uniform tile base: row 32T + 8g + 4h + e, pixel j sits at
// tile_base[(32T + 8g + e)*32 + (128h + j)] - a compile-time constant plus one 32-bit lane offset.
__device__ __forceinline__ unsigned nvp_lane_off(int lane) { return (unsigned)(((lane >> 5) << 7) + (lane & 31)); }

__device__ __forceinline__ void store_ptm(float* __restrict__ tile_base, const f32x16 (&v)[4], int lane) {
    const unsigned lo = nvp_lane_off(lane);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) (tile_base + (32 * T + 8 * (r >> 2) + (r & 3)) * 32)[lo] = v[T][r];
}


// One 64-step chain whose output has ZT 32-row tiles (latent gradient): the packed stream
// holds ZT floats per lane and step.
template <int ZT>
__device__ __forceinline__ void chain_hz(f32x16 (&acc)[ZT], const f32x16 (&hin)[4], const float* __restrict__ wp, int lane) {
    constexpr int Q = ZT / 4;                     // float4 per lane per step
    const float4* w = reinterpret_cast<const float4*>(wp);      // wave-uniform
    const unsigned ul = (unsigned)(lane * Q);
    constexpr int GZ = 2;
    float4 a[2][GZ][Q];
#pragma unroll
    for (int i = 0; i < GZ; ++i)
#pragma unroll
        for (int q = 0; q < Q; ++q) a[0][i][q] = (w + (i * 64 * Q + q))[ul];
#pragma unroll
    for (int g = 0; g < 64 / GZ; ++g) {
        if (g + 1 < 64 / GZ) {
#pragma unroll
            for (int i = 0; i < GZ; ++i)
#pragma unroll
                for (int q = 0; q < Q; ++q) a[(g + 1) & 1][i][q] = (w + (((g + 1) * GZ + i) * 64 * Q + q))[ul];
        }
        NVP_CHAIN_FENCE();
#pragma unroll
        for (int i = 0; i < GZ; ++i) {
            const int step = g * GZ + i;
            const float b = hin[step >> 4][step & 15];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                acc[4 * q + 0] = nvp_mfma(a[g & 1][i][q].x, b, acc[4 * q + 0]);
                acc[4 * q + 1] = nvp_mfma(a[g & 1][i][q].y, b, acc[4 * q + 1]);
                acc[4 * q + 2] = nvp_mfma(a[g & 1][i][q].z, b, acc[4 * q + 2]);
                acc[4 * q + 3] = nvp_mfma(a[g & 1][i][q].w, b, acc[4 * q + 3]);
            }
        }
    }
}

// Load one 16-row block (tile T) of a PTM activation into fragment registers.
__device__ __forceinline__ void load_ptm16(f32x16& v, const float* __restrict__ tile_base, int T, int lane) {
    const unsigned lo = nvp_lane_off(lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (tile_base + (32 * T + 8 * (r >> 2) + (r & 3)) * 32)[lo];
}

__device__ __forceinline__ void store_ptm16(float* __restrict__ tile_base, const f32x16& v, int T, int lane) {
    const unsigned lo = nvp_lane_off(lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) (tile_base + (32 * T + 8 * (r >> 2) + (r & 3)) * 32)[lo] = v[r];
}
