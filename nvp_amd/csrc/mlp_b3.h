// Split-operand MFMA helpers shared by the "b3" chain kernels (mlp_fwd_b3*.hip, mlp_bwd_b3*.hip) and mlp_dw.hip
// (see mlp_layout.h, "b3").  Two operand splits, selected at build time (NVP_SPLIT_H2):
//
//  * fp16 x 2 (default).  x * s = hi + lo with hi = fp16(x s), lo = fp16(x s - hi): 22+ significant bits, representation
//    error <= 2^-22 |x| (2^-24 rms).  s is a POWER OF TWO chosen so that the largest magnitude that shares the scale lands in
//    [2^13, 2^14): per packed weight stream (pack_b3_scales_kernel), per PIXEL for activations (the pixel is the MFMA column
//    = the lane, so the scale and its inverse are per-lane scalars).  Scaling by 2^e is exact, fp16 subnormal parts are kept
//    (v_mfma_f32_32x32x16_f16 honours them: tools/probes/f16_mfma_probe.hip), so an element 2^-k below its pixel's maximum
//    still carries an absolute error <= 2^-38 of that maximum.  THREE products - lo*hi, hi*lo, hi*hi, smallest first - are
//    accumulated in fp32; the accumulator is multiplied by 2^-(e_w + e_px) afterwards (exact).  Dropped: lo*lo <= 2^-22.
//  * bf16 x 3.  x = hi + mid + lo in bf16 (8 + 8 + 8 bits, fp32's exponent range: no scale), the SIX products >= 2^-24.
//
// Measured on K = 128 / 242 dot products (tools/split_accuracy.py, profiles/r02_probe_f16x2_split.txt): both splits end
// BELOW the error of an fp32 fma chain (the fp32 accumulation dominates); fp16 x 2 needs half the matrix-pipe cycles, two
// thirds of the weight bytes and about half the split instructions (v_fma_mix folds scale, residual and conversion).
#pragma once
#include "mlp_chain.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kP = kB3Parts;
// the 16-bit parts of eight consecutive k of one operand: p[0] = hi, then (mid,) lo
struct BOp { u32x4 p[kP]; };

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32, round to nearest even
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pk_f16(float a, float b) {       // v_cvt_pk_f16_f32, round to nearest even
    f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

// ---- per-pixel power-of-two scale (fp16 x 2) ----------------------------------------------------------------------------------
// s = 2^e with m * s in [2^13, 2^14) for the pixel's largest magnitude m, u = 1 / s.  m must be >= 2^-113 (callers clamp:
// 1.0 where a bias shares the scale, kTinyMax otherwise); m = +inf gives s = 2^-115, and the infinity itself still converts
// to an fp16 infinity, so non-finite inputs poison the pixel's outputs as they do in fp32.
struct PxScale { float s, u; };
constexpr float kTinyMax = 8.8817841970012523e-16f;      // 2^-50: pixels whose inputs are all smaller keep 2^-63 absolute precision
__device__ __forceinline__ PxScale px_scale(float m) {
#if NVP_SPLIT_H2
    const unsigned e = __float_as_uint(m) >> 23;
    return {__uint_as_float((267u - e) << 23), __uint_as_float((e - 13u) << 23)};
#else
    return {1.0f, 1.0f};
#endif
}
// largest magnitude among the 128 values of this lane's pixel held in four D-register tiles (both lane halves)
__device__ __forceinline__ float px_absmax(const f32x16 (&v)[4]) {
#if NVP_SPLIT_H2
    float m = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(fabsf(v[T][r]), fabsf(v[T][r + 1])));      // v_max3_f32 |.|
    return fmaxf(m, __shfl_xor(m, 32));
#else
    return 1.0f;
#endif
}

// 8 floats -> hi / mid / lo bf16x8 with x = hi + mid + lo to ~2^-25 |x| (always available: mlp_dw.hip)
__device__ __forceinline__ void split8_bf3(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], c = x[2 * p + 1];
        const unsigned h = pk_bf16(a, c);
        const float ra = a - __uint_as_float(h << 16), rb = c - __uint_as_float(h & 0xffff0000u);
        const unsigned m = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        hi[p] = h; mid[p] = m; lo[p] = pk_bf16(sa, sb);
    }
}
__device__ __forceinline__ f32x16 mf_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifndef NVP_SPLIT_ASM
#define NVP_SPLIT_ASM 0          // fp16 x 2 split, instruction selection (same bits): 0 = hipcc's (3 per value); 1 = four v_fma_mix per pair (2 per
                                 // value) - SLOWER, dW 1.80 vs 1.73 ms: v_fma_mix is not full rate; 2 = hi by hipcc, the two residuals as v_fma_mix
                                 // with op_sel (2 per value, two of them mix): chains -0.02 ms each (their default), dW +0.19 ms (register limit)
#endif
// 8 floats -> parts.  fp16 x 2: of x * s.
__device__ __forceinline__ void split8(const float (&x)[8], const float s, BOp& b) {
#ifdef NVP_ABL_NOSPLIT          // ablation builds only (tools/ablate_b3.sh): one conversion per pair, no residuals
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int k = 0; k < kP; ++k) b.p[k][p] = NVP_SPLIT_H2 ? pk_f16(x[2 * p], x[2 * p + 1]) : pk_bf16(x[2 * p], x[2 * p + 1]);
    return;
#endif
#if NVP_SPLIT_H2
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], c = x[2 * p + 1];
#if NVP_SPLIT_ASM == 1
        // Two instructions per value: v_fma_mix{lo,hi}_f16 computes fma(x, s, c) in fp32 and writes the fp16 result into one
        // half of the destination; for the residual the addend is the matching HALF of the packed hi register (op_sel),
        // so hi is never unpacked.  (hipcc's own selection of the C code below spends three: it converts hi twice.)
        unsigned hh, ll;
        asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
            "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(hh), "=&v"(ll) : "v"(a), "v"(c), "v"(s));
        b.p[0][p] = hh;
        b.p[1][p] = ll;
#elif NVP_SPLIT_ASM == 2
        // hi from hipcc's v_pk_mul_f32 + v_cvt_pk_f16_f32; the two residuals as v_fma_mix{lo,hi}_f16 whose addend is the matching
        // HALF of the packed hi register (op_sel) - hipcc's own selection converts hi a second time, once per half
        const f16x2 h = {(_Float16)(a * s), (_Float16)(c * s)};
        const unsigned hh = __builtin_bit_cast(unsigned, h);
        unsigned ll;
        asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(ll) : "v"(a), "v"(c), "v"(s), "v"(hh));
        b.p[0][p] = hh;
        b.p[1][p] = ll;
#else
        const f16x2 h = {(_Float16)(a * s), (_Float16)(c * s)};
        // residual x s - hi is exact in fp32 (hipcc: v_fma_mixlo/hi_f16 - scale, subtraction and conversion in one instruction)
        const f16x2 l = {(_Float16)__builtin_fmaf(a, s, -(float)h.x), (_Float16)__builtin_fmaf(c, s, -(float)h.y)};
        b.p[0][p] = __builtin_bit_cast(unsigned, h);
        b.p[1][p] = __builtin_bit_cast(unsigned, l);
#endif
    }
#else
    split8_bf3(x, b.p[0], b.p[1], b.p[2]);
#endif
}

__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
#ifdef NVP_ABL_NOMFMA           // ablation builds only: one VALU op instead of the MFMA
    c[0] = __uint_as_float(__float_as_uint(c[0]) ^ a[0] ^ b[0]); return c;
#endif
#if NVP_SPLIT_H2
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// all significant part products of one (A tile, B) pair into one accumulator, smallest terms first
__device__ __forceinline__ void mac_parts(f32x16& acc, const u32x4 (&a)[kP], const BOp& b) {
#if NVP_SPLIT_H2
    acc = mf(a[1], b.p[0], acc);
    acc = mf(a[0], b.p[1], acc);
    acc = mf(a[0], b.p[0], acc);
#else
    acc = mf(a[2], b.p[0], acc);
    acc = mf(a[0], b.p[2], acc);
    acc = mf(a[1], b.p[1], acc);
    acc = mf(a[1], b.p[0], acc);
    acc = mf(a[0], b.p[1], acc);
    acc = mf(a[0], b.p[0], acc);
#endif
}

// one k-step (16 inputs) into the four output tiles; w points at the step's 4 kP operand quads.  PF: prefetch the next
// tile's quads while the current tile's MFMAs issue (2 kP instead of kP operand quads live).
#ifdef NVP_ABL_WFIXED            // ablation builds only: every k-step of every layer reads the SAME block of weights (stays in L1)
#define NVP_WSTRIDE(x) 0
#else
#define NVP_WSTRIDE(x) (x)
#endif
template <bool PF = true>
__device__ __forceinline__ void step_b3(f32x16 (&acc)[4], const u32x4* __restrict__ w, const BOp& b, int lane) {
    const unsigned ul = (unsigned)lane;
    u32x4 a[2][kP];
    if (PF) {
#pragma unroll
        for (int q = 0; q < kP; ++q) a[0][q] = (w + q * 64)[ul];
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        if (PF) {
            if (T < 3) {
#pragma unroll
                for (int q = 0; q < kP; ++q) a[(T + 1) & 1][q] = (w + ((T + 1) * kP + q) * 64)[ul];
            }
        } else {
            // register-lean order: one tile's operand quads live at a time
#pragma unroll
            for (int q = kP - 1; q >= 0; --q) a[0][q] = (w + (T * kP + q) * 64)[ul];
            NVP_CHAIN_FENCE();
            mac_parts(acc[T], a[0], b);
            NVP_CHAIN_FENCE();
            continue;
        }
        NVP_CHAIN_FENCE();
        mac_parts(acc[T], a[T & 1], b);
    }
}

// the same k-step for four consecutive tiles of a longer accumulator array
template <bool PF = true>
__device__ __forceinline__ void step_b3_at(f32x16* acc, const u32x4* __restrict__ w, const BOp& b, int lane) {
    step_b3<PF>(*reinterpret_cast<f32x16(*)[4]>(acc), w, b, lane);
}

// the B operand of a bias step: e_0 times the pixel's scale (1.0 for bf16 x 3; a power of two, exact in fp16 down to 2^-24)
__device__ __forceinline__ u32x4 bias_bop(float s, int lane) {
#if NVP_SPLIT_H2
    const unsigned one = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)s);
#else
    const unsigned one = 0x00003f80u;
#endif
    return u32x4{lane < 32 ? one : 0u, 0u, 0u, 0u};
}
// the bias products of one tile: A[.][0] = the parts of the (scaled) bias
__device__ __forceinline__ void bias_mac(f32x16& acc, const u32x4 (&a)[kP], const u32x4 e0) {
#pragma unroll
    for (int q = kP - 1; q >= 0; --q) acc = mf(a[q], e0, acc);
}
// bias step: B = e_0 s
__device__ __forceinline__ void bias_b3(f32x16 (&acc)[4], const u32x4* __restrict__ w, float s, int lane) {
    const unsigned ul = (unsigned)lane;
    const u32x4 e0 = bias_bop(s, lane);
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        u32x4 a[kP];
#pragma unroll
        for (int q = 0; q < kP; ++q) a[q] = (w + (T * kP + q) * 64)[ul];
        bias_mac(acc[T], a, e0);
    }
}

// the eight inputs of chained k-step c: D registers 8 (c & 1) .. + 7 of tile c >> 1
__device__ __forceinline__ void chain_in8(float (&x)[8], const f32x16 (&hin)[4], int c) {
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = hin[c >> 1][8 * (c & 1) + q];
}

// experiment: wave priority around the MFMA chains (two waves share a SIMD; whose instructions win when both are ready?)
//   1: a wave inside a chain outranks its partner (keeps the matrix pipe fed), 2: the opposite (element-wise stages outrank chains)
#ifndef NVP_MFMA_PRIO
#define NVP_MFMA_PRIO 1          // measured (profiles/r04_ab_wave_priority.txt): backward chain 1.65 -> 1.60 ms, forward unchanged; bit-identical
#endif
#if NVP_MFMA_PRIO == 1
#define NVP_CHAIN_ENTER() __builtin_amdgcn_s_setprio(2)
#define NVP_CHAIN_LEAVE() __builtin_amdgcn_s_setprio(0)
#elif NVP_MFMA_PRIO == 2
#define NVP_CHAIN_ENTER() __builtin_amdgcn_s_setprio(0)
#define NVP_CHAIN_LEAVE() __builtin_amdgcn_s_setprio(2)
#else
#define NVP_CHAIN_ENTER()
#define NVP_CHAIN_LEAVE()
#endif

// 8 k-steps over the previous layer's D registers, scaled by s.  post(c) runs after k-step c's MFMAs have been issued (independent work
// a caller wants in the shadow of the chain).  Weight prefetch: step_b3's - the next output tile's operand quads are requested while the
// current tile's MFMAs issue.  The schemes measured against it in rounds 4-5 (a whole k-step one to three steps ahead, tile pairs, all
// quads of a step at once; all bit-identical, none faster: HISTORY.md / DESIGN.md 4.1) were removed from this header in round 6; the
// last commit that carries them is named in HISTORY.md.
template <bool PF = true, typename Post>
__device__ __forceinline__ void chain_h_b3(f32x16 (&acc)[4], const f32x16 (&hin)[4], const float s, const u32x4* __restrict__ w, int lane, Post post) {
    NVP_CHAIN_ENTER();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
        chain_in8(x, hin, c);
        BOp b;
        split8(x, s, b);
        step_b3<PF>(acc, w + NVP_WSTRIDE(c * kB3StepQuads), b, lane);
        post(c);
    }
    NVP_CHAIN_LEAVE();
}
template <bool PF = true>
__device__ __forceinline__ void chain_h_b3(f32x16 (&acc)[4], const f32x16 (&hin)[4], const float s, const u32x4* __restrict__ w, int lane) {
    chain_h_b3<PF>(acc, hin, s, w, lane, [](int) {});
}

// Two transposed GEMMs over the SAME input registers (the backward chain's dz and dh chains both consume dp): every k-step's
// operand split is done once and feeds both weight streams.
template <bool PF = true>
__device__ __forceinline__ void chain_h2_b3(f32x16 (&acc_a)[4], const u32x4* __restrict__ wa, f32x16 (&acc_b)[4], const u32x4* __restrict__ wb,
                                            const f32x16 (&hin)[4], const float s, int lane) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
        chain_in8(x, hin, c);
        BOp b;
        split8(x, s, b);
        step_b3<PF>(acc_a, wa + c * kB3StepQuads, b, lane);
        step_b3<PF>(acc_b, wb + c * kB3StepQuads, b, lane);
    }
}

// acc *= f for the four tiles (un-scaling an accumulator, or bringing a parked one into a chain's scaled units); f = 1 folds away
__device__ __forceinline__ void scale4(f32x16 (&v)[4], const float f) {
#if NVP_SPLIT_H2
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[T][r] *= f;
#endif
}

// LeakyReLU(0.01) of acc * f (modulation.py:112-121)
__device__ __forceinline__ void lrelu4_scaled(f32x16 (&v)[4], const float f) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = NVP_SPLIT_H2 ? v[T][r] * f : v[T][r];
            v[T][r] = t > 0.f ? t : t * 0.01f;
        }
}

// per-pixel largest magnitude of a float4 (used while staging / streaming the latent)
__device__ __forceinline__ float absmax_f4(float m, const float4 t) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
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
