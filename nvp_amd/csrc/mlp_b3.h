// bf16 x 3 split-MFMA helpers shared by mlp_fwd_b3.hip and mlp_bwd_b3.hip (see mlp_layout.h, "b3").
#pragma once
#include "mlp_chain.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32, round to nearest even
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// 8 floats -> hi / mid / lo bf16x8 with x = hi + mid + lo to ~2^-25 |x|
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#ifdef NVP_ABL_NOSPLIT          // ablation builds only (tools/ablate_b3.sh): one conversion per pair, no residuals
#pragma unroll
    for (int p = 0; p < 4; ++p) { hi[p] = pk_bf16(x[2 * p], x[2 * p + 1]); mid[p] = hi[p]; lo[p] = hi[p]; }
    return;
#endif
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned h = pk_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        hi[p] = h; mid[p] = m; lo[p] = pk_bf16(sa, sb);
    }
}
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
#ifdef NVP_ABL_NOMFMA           // ablation builds only: one VALU op instead of the MFMA
    c[0] = __uint_as_float(__float_as_uint(c[0]) ^ a[0] ^ b[0]); return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one k-step (16 inputs) into the four output tiles; w points at the step's 12 operand quads.  PF: prefetch the next
// tile's three quads while the current tile's six MFMAs issue (24 instead of 12 operand registers).
#ifdef NVP_ABL_WFIXED            // ablation builds only: every k-step of every layer reads the SAME 12 KiB of weights (stays in L1)
#define NVP_WSTRIDE(x) 0
#else
#define NVP_WSTRIDE(x) (x)
#endif
template <bool PF = true>
__device__ __forceinline__ void step_b3(f32x16 (&acc)[4], const u32x4* __restrict__ w, const u32x4 bh, const u32x4 bm, const u32x4 bl, int lane) {
    const unsigned ul = (unsigned)lane;
#ifdef NVP_B3_INTERLEAVE        // experiment: consecutive MFMAs never share an accumulator (all 12 operand quads live)
    {
        u32x4 q[4][3];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int k = 0; k < 3; ++k) q[T][k] = (w + (T * 3 + k) * 64)[ul];
        NVP_CHAIN_FENCE();
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = mf(q[T][2], bh, acc[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = mf(q[T][0], bl, acc[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = mf(q[T][1], bm, acc[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = mf(q[T][1], bh, acc[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = mf(q[T][0], bm, acc[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = mf(q[T][0], bh, acc[T]);
        return;
    }
#endif
    u32x4 a[2][3];
    if (PF) {
#pragma unroll
        for (int q = 0; q < 3; ++q) a[0][q] = (w + q * 64)[ul];
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        if (PF) {
            if (T < 3) {
#pragma unroll
                for (int q = 0; q < 3; ++q) a[(T + 1) & 1][q] = (w + ((T + 1) * 3 + q) * 64)[ul];
            }
        } else {
            // register-lean order: lo first (one use), then mid, then hi; at most two operand quads are live
            const u32x4 al = (w + (T * 3 + 2) * 64)[ul];
            const u32x4 am = (w + (T * 3 + 1) * 64)[ul];
            NVP_CHAIN_FENCE();
            acc[T] = mf(al, bh, acc[T]);
            acc[T] = mf(am, bm, acc[T]);
            const u32x4 ah = (w + (T * 3 + 0) * 64)[ul];
            acc[T] = mf(am, bh, acc[T]);
            NVP_CHAIN_FENCE();
            acc[T] = mf(ah, bl, acc[T]);
            acc[T] = mf(ah, bm, acc[T]);
            acc[T] = mf(ah, bh, acc[T]);
            continue;
        }
        NVP_CHAIN_FENCE();
        const u32x4 ah = a[T & 1][0], am = a[T & 1][1], al = a[T & 1][2];
        acc[T] = mf(al, bh, acc[T]);              // smallest terms first
        acc[T] = mf(ah, bl, acc[T]);
        acc[T] = mf(am, bm, acc[T]);
        acc[T] = mf(am, bh, acc[T]);
        acc[T] = mf(ah, bm, acc[T]);
        acc[T] = mf(ah, bh, acc[T]);
    }
}

// the same k-step for four consecutive tiles of a longer accumulator array
template <bool PF = true>
__device__ __forceinline__ void step_b3_at(f32x16* acc, const u32x4* __restrict__ w, const u32x4 bh, const u32x4 bm, const u32x4 bl, int lane) {
    step_b3<PF>(*reinterpret_cast<f32x16(*)[4]>(acc), w, bh, bm, bl, lane);
}

// bias step: B = e_0 (1.0 at k = 0, exact in bf16), A[.][0] = hi/mid/lo of the bias
__device__ __forceinline__ void bias_b3(f32x16 (&acc)[4], const u32x4* __restrict__ w, int lane) {
    const unsigned ul = (unsigned)lane;
    const u32x4 e0 = {lane < 32 ? 0x00003f80u : 0u, 0u, 0u, 0u};
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const u32x4 ah = (w + (T * 3 + 0) * 64)[ul], am = (w + (T * 3 + 1) * 64)[ul], al = (w + (T * 3 + 2) * 64)[ul];
        acc[T] = mf(al, e0, acc[T]);
        acc[T] = mf(am, e0, acc[T]);
        acc[T] = mf(ah, e0, acc[T]);
    }
}

// 8 k-steps over the previous layer's D registers
template <bool PF>
__device__ __forceinline__ void chain_h_b3_swp(f32x16 (&acc)[4], const f32x16 (&hin)[4], const u32x4* __restrict__ w, int lane);
template <bool PF = true>
__device__ __forceinline__ void chain_h_b3(f32x16 (&acc)[4], const f32x16 (&hin)[4], const u32x4* __restrict__ w, int lane) {
#if NVP_B3_SWP
    if (PF) { chain_h_b3_swp<PF>(acc, hin, w, lane); return; }
#endif
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = hin[c >> 1][8 * (c & 1) + q];
        u32x4 bh, bm, bl;
        split8(x, bh, bm, bl);
        step_b3<PF>(acc, w + NVP_WSTRIDE(c * 12 * 64), bh, bm, bl, lane);
    }
}

// Two transposed GEMMs over the SAME input registers (the backward chain's dz and dh chains both consume dp): every k-step's
// operand split is done once and feeds both weight streams.
template <bool PF = true>
__device__ __forceinline__ void chain_h2_b3(f32x16 (&acc_a)[4], const u32x4* __restrict__ wa, f32x16 (&acc_b)[4], const u32x4* __restrict__ wb,
                                            const f32x16 (&hin)[4], int lane) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = hin[c >> 1][8 * (c & 1) + q];
        u32x4 bh, bm, bl;
        split8(x, bh, bm, bl);
        step_b3<PF>(acc_a, wa + c * 12 * 64, bh, bm, bl, lane);
        step_b3<PF>(acc_b, wb + c * 12 * 64, bh, bm, bl, lane);
    }
}

// ---- software-pipelined operand split (NVP_B3_SWP) ------------------------------------------------------------------
// A wave issues in order: with the split of k-step c placed in front of its 24 MFMAs, the wave's ~45 split instructions
// and its MFMAs never overlap (only the PARTNER wave's MFMAs can run underneath), and the ablations show the VALU pipe is
// as loaded as the matrix pipe.  Here the split of k-step c+1 is cut into four pair-splits and each is woven between the
// six MFMAs of one output tile of k-step c (sched_group_barrier: 1 MFMA, 2 VALU, ...): the VALU instructions issue in the
// shadow of the wave's own MFMAs (<= 5 issue slots are free per 32-cycle MFMA).
#ifndef NVP_B3_SWP
#define NVP_B3_SWP 0
#endif
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = pk_bf16(sa, sb);
}

// one k-step into four output tiles with the NEXT k-step's operand split woven in: xn = the next step's eight inputs
// (ignored when !more); nh/nm/nl receive its parts
__device__ __forceinline__ void step_b3_swp(f32x16 (&acc)[4], const u32x4* __restrict__ w, const u32x4 bh, const u32x4 bm, const u32x4 bl,
                                            const float (&xn)[8], bool more, u32x4& nh, u32x4& nm, u32x4& nl, int lane) {
    const unsigned ul = (unsigned)lane;
    u32x4 a[2][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[0][q] = (w + q * 64)[ul];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        if (T < 3) {
#pragma unroll
            for (int q = 0; q < 3; ++q) a[(T + 1) & 1][q] = (w + ((T + 1) * 3 + q) * 64)[ul];
        }
        NVP_CHAIN_FENCE();
        const u32x4 ah = a[T & 1][0], am = a[T & 1][1], al = a[T & 1][2];
        acc[T] = mf(al, bh, acc[T]);              // smallest terms first
        acc[T] = mf(ah, bl, acc[T]);
        acc[T] = mf(am, bm, acc[T]);
        acc[T] = mf(am, bh, acc[T]);
        acc[T] = mf(ah, bm, acc[T]);
        acc[T] = mf(ah, bh, acc[T]);
        if (more) {
            unsigned h_, m_, l_;
            split2(xn[2 * T], xn[2 * T + 1], h_, m_, l_);
            nh[T] = h_; nm[T] = m_; nl[T] = l_;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two VALU in its shadow
            }
        }
    }
}

template <bool PF = true>
__device__ __forceinline__ void chain_h_b3_swp(f32x16 (&acc)[4], const f32x16 (&hin)[4], const u32x4* __restrict__ w, int lane) {
    u32x4 bh, bm, bl;
    {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = hin[0][q];
        split8(x, bh, bm, bl);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float xn[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) xn[q] = hin[((c + 1) & 7) >> 1][8 * ((c + 1) & 1) + q];
        u32x4 nh = bh, nm = bm, nl = bl;
        step_b3_swp(acc, w + NVP_WSTRIDE(c * 12 * 64), bh, bm, bl, xn, c + 1 < 8, nh, nm, nl, lane);
        bh = nh; bm = nm; bl = nl;
    }
}

// the 16 values of table `t` for this lane's rows of tile T (see kB3TabFloats): four 16-B loads
__device__ __forceinline__ void load_tab16(float (&v)[16], const float* __restrict__ tab, int t, int T, int h) {
    const float4* p = reinterpret_cast<const float4*>(tab + ((t * 2 + h) * 4 + T) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 q = p[g];
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
}
