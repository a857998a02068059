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

// The two waves that share a SIMD run the same phase sequence (MFMA chain, element-wise stage, ...).
// If they start together they stay phase-aligned: both sit in VALU/memory phases at once (MFMA pipe
// idle) and then contend for it.  Delaying the workgroups that land in the second residency slot of a
// CU (observed dispatch: the first 256 blocks take one CU each) by a fraction of a phase, once, breaks
// the alignment for the rest of the launch.  Placement is only a speed heuristic, never correctness.
#ifndef NVP_STAGGER_SLEEPS
#define NVP_STAGGER_SLEEPS 2
#endif
__device__ __forceinline__ void nvp_stagger_start() {
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
#pragma unroll 1
        for (int i = 0; i < NVP_STAGGER_SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);      // ~8k cycles each
    }
}

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

// Stage this wave's latent tile (PTM4, contiguous n4 float4) into its private LDS region with
// coalesced 16-B loads.  Only the owning wave reads it back, so a wave barrier suffices.
__device__ __forceinline__ void stage_z(float4* __restrict__ zl, const float4* __restrict__ z4, int n4, int lane) {
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
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// zs k-steps over the latent: step u consumes rows 2u (lane half 0) and 2u+1 (half 1).  In
// PTM4 the float4 of row-group rg = u>>1 holds rows 4rg..4rg+3 of pixel j, i.e. the B operands
// of steps 2rg and 2rg+1 for both lane halves: one ds_read_b128 per two steps, no HBM traffic
// (the forward ablation showed re-streaming z per layer cost 17 % of the kernel).
__device__ __forceinline__ void chain_z(f32x16 (&acc)[4], const float4* __restrict__ zl, int zs,
                                        const float4* __restrict__ wp, int lane) {
    const unsigned ul = (unsigned)lane;
    const int j = lane & 31;
    const bool hi = lane >= 32;
    const int ng = zs / 4;
    float4 a[4], an[4];
    if (ng > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = NVP_WLOAD((wp + i * 64)[ul]);
    }
    for (int g = 0; g < ng; ++g) {
        if (g + 1 < ng) {
            const float4* wn = wp + (g + 1) * 4 * 64;      // scalar pointer bump
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = NVP_WLOAD((wn + i * 64)[ul]);
        }
        const float4 t0 = zl[(2 * g) * 32 + j];
        const float4 t1 = zl[(2 * g + 1) * 32 + j];
        NVP_CHAIN_FENCE();
        mfma4(acc, a[0], hi ? t0.y : t0.x);
        mfma4(acc, a[1], hi ? t0.w : t0.z);
        mfma4(acc, a[2], hi ? t1.y : t1.x);
        mfma4(acc, a[3], hi ? t1.w : t1.z);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = an[i];
    }
    for (int u = ng * 4; u < zs; ++u) {
        const float4 t = zl[(u >> 1) * 32 + j];
        const float b = (u & 1) ? (hi ? t.w : t.z) : (hi ? t.y : t.x);
        mfma4(acc, NVP_WLOAD((wp + u * 64)[ul]), b);
    }
}

// Same k-steps for latent rows that are NOT staged in LDS (wide latents, nvp_l: the tile would not leave room
// for two workgroups per CU): the B operand comes straight from the PTM4 tensor, one 16-B load per lane per
// two k-steps (both lane halves read the same 512 B), prefetched one row-group ahead.  u0 must be even.
__device__ __forceinline__ void chain_zg(f32x16 (&acc)[4], const float4* __restrict__ zg, int u0, int u1,
                                         const float4* __restrict__ wp, int lane) {
    const unsigned ul = (unsigned)lane;
    const int j = lane & 31;
    const bool hi = lane >= 32;
    if (u0 >= u1) return;
    float4 t = zg[(u0 >> 1) * 32 + j];
    for (int u = u0; u < u1; u += 2) {
        float4 tn = t;
        if (u + 2 < u1) tn = zg[((u + 2) >> 1) * 32 + j];
        const float4 a0 = NVP_WLOAD((wp + u * 64)[ul]);
        float4 a1 = a0;
        if (u + 1 < u1) a1 = NVP_WLOAD((wp + (u + 1) * 64)[ul]);
        NVP_CHAIN_FENCE();
        mfma4(acc, a0, hi ? t.y : t.x);
        if (u + 1 < u1) mfma4(acc, a1, hi ? t.w : t.z);
        t = tn;
    }
}

__device__ __forceinline__ void lrelu4(f32x16 (&v)[4]) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[T][r] = v[T][r] > 0.f ? v[T][r] : v[T][r] * 0.01f;
}

// Stream stores: the activations / gradient streams are written once and consumed by a LATER kernel,
// so they are stored non-temporally (no L2 allocation) unless NVP_NT_STORES=0.
#ifndef NVP_NT_STORES
#define NVP_NT_STORES 1
#endif
typedef float nvp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nvp_stream_store(float4* p, float a, float b, float c, float d) {
#if NVP_NT_STORES
    nvp_f4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<nvp_f4*>(p));
#else
    *p = make_float4(a, b, c, d);
#endif
}

// PTM4 addressing: a 128-row stream tile is 32 row-groups x 32 pixels x float4 (rows 4rg..4rg+3
// of one pixel are contiguous).  Lane (j, h) owns rows 8g+4h+{0..3} of each 32-row tile T, i.e.
// row-group 8T+2g+h, so its float4 sits at index (8T+2g)*32 + lane: every fragment load/store is
// one 16-B access per lane and one contiguous 1-KiB line set per wave instruction.
__device__ __forceinline__ void store_ptm(float* __restrict__ tile_base, const f32x16 (&v)[4], int lane) {
#ifdef NVP_ABL_NOSTORE          // ablation builds only
    if (v[0][0] != 123.456f) return;
#endif
    float4* b4 = reinterpret_cast<float4*>(tile_base);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            nvp_stream_store(&(b4 + (8 * T + 2 * g) * 32)[(unsigned)lane], v[T][4 * g], v[T][4 * g + 1], v[T][4 * g + 2], v[T][4 * g + 3]);
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

// Load / store one 32-row tile T of a PTM4 stream into fragment registers.
__device__ __forceinline__ void load_ptm16(f32x16& v, const float* __restrict__ tile_base, int T, int lane) {
    const float4* b4 = reinterpret_cast<const float4*>(tile_base);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = (b4 + (8 * T + 2 * g) * 32)[(unsigned)lane];
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
}

__device__ __forceinline__ void store_ptm16(float* __restrict__ tile_base, const f32x16& v, int T, int lane) {
#ifdef NVP_ABL_NOSTORE          // ablation builds only
    if (v[0] != 123.456f) return;
#endif
    float4* b4 = reinterpret_cast<float4*>(tile_base);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        nvp_stream_store(&(b4 + (8 * T + 2 * g) * 32)[(unsigned)lane], v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}


// ---- latent gradient handed to the scatter in ITS layout (encode_bwd.hip) ---------------------------------------------------
// For a y-sorted batch the scatter's sorted order of the xy and yt planes is the batch order itself, so its `permute` pass would
// only transpose those planes' columns of the row-major latent gradient into level-major [level][pixel][F] arrays.  The chain
// kernels write them there directly (and feed max|dz| for the fixed-point scale), `permute` then only handles the xt plane:
// -2/3 of that pass and of its traffic.  Device pointers into the scatter workspace (nvp_encode_bwd_prepare); dzs[0] == nullptr: off.
struct NvpDzLm {
    float* dzs[2];
    unsigned* dzmax;
    unsigned* sdzmax;          // max|dz| slots of the sparse grid's columns [scol0, scol0 + scols) (those rows stay row-major)
    int scol0, scols;
};
#define NVP_DZLM_OFF NvpDzLm{{nullptr, nullptr}, nullptr, nullptr, 0, 0}

// one float4 = latent rows base..base+3 of pixel px.  Returns true when the rows belong to plane 0 / 1 and were stored level-major.
// `ms` collects max|dz| of the sparse columns (which the caller then stores row-major as before).
__device__ __forceinline__ bool nvp_dz_store_lm(const NvpDzLm& lm, int F, int base, int64_t px, int64_t n, float a, float b, float c, float d4, unsigned& m, unsigned& ms) {
    if (lm.dzs[0] == nullptr) return false;
    const int plane_rows = 16 * F;
    if (base >= 2 * plane_rows) {
        if (px < n && base >= lm.scol0 && base < lm.scol0 + lm.scols)        // rows past scols inside the last float4 are padding: exact zeros
            ms = max(ms, max(max(__float_as_uint(fabsf(a)), __float_as_uint(fabsf(b))), max(__float_as_uint(fabsf(c)), __float_as_uint(fabsf(d4)))));
        return false;
    }
    const int plane = base >= plane_rows ? 1 : 0;
    const int rowin = base - plane * plane_rows;
    if (px < n) {
        float* dst = lm.dzs[plane];
        if (F == 2) {
            const int l0 = rowin >> 1;
            *reinterpret_cast<float2*>(dst + ((int64_t)l0 * n + px) * 2) = make_float2(a, b);
            *reinterpret_cast<float2*>(dst + ((int64_t)(l0 + 1) * n + px) * 2) = make_float2(c, d4);
        } else {            // F == 4: the float4 is exactly one level
            *reinterpret_cast<float4*>(dst + ((int64_t)(rowin >> 2) * n + px) * 4) = make_float4(a, b, c, d4);
        }
        m = max(m, max(max(__float_as_uint(fabsf(a)), __float_as_uint(fabsf(b))), max(__float_as_uint(fabsf(c)), __float_as_uint(fabsf(d4)))));
    }
    return true;
}

// after the stores: the wave's max|dz| (bit patterns: NaN / Inf win, see encode_bwd.hip) into one of the scatter's slots
__device__ __forceinline__ void nvp_dz_lm_finish(const NvpDzLm& lm, unsigned m, unsigned ms, int64_t tile, int lane) {
    if (lm.dzs[0] == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = max(m, (unsigned)__shfl_xor((int)m, o)); ms = max(ms, (unsigned)__shfl_xor((int)ms, o)); }
    if (lane == 0 && m > 0u) atomicMax(lm.dzmax + (int)(tile & 255), m);
    if (lane == 0 && ms > 0u) atomicMax(lm.sdzmax + (int)(tile & 255), ms);
}
