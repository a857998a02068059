// Shared device helpers for the gfx950 kernels of libnvp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nvp_hip.h"

// NVP_EXPERIMENTS=1 (build.sh's libnvp_hip_experiments.so only): compile the measured-slower kernel variants kept as A/B evidence -
// LDS-staged gather, forward weight ring, per-wave backward chain, merged / paired / grouped / DMA-fed dW jobs - and the NVP_*
// environment switches that select them.  The product library (libnvp_hip.so) has neither the kernels nor the switches.
#ifndef NVP_EXPERIMENTS
#define NVP_EXPERIMENTS 0
#endif

// ---- separately rounded fp32 operations --------------------------------------------------------------------------------------------
// hipcc's __fmul_rn / __fadd_rn / __fsub_rn / __fdiv_rn are plain `*` `+` `-` `/` compiled INSIDE clang's own header, where contraction
// is allowed: once inlined, a product that feeds a sum fuses into ONE v_fma_f32 whatever `#pragma clang fp contract` the caller is
// under (round 6, seen in the ISA of the fused forward - a translation unit built with contraction on: SparseGrid.forward_inter's blend
// lo * w_lo + hi * w_hi came out as fma(lo, w_lo, hi * w_hi), one ulp away from the stand-alone gather kernel on 80 % of the pixels; the
// nearest index trunc(fl((res - 1) c) + 0.5) was a v_fma too - harmless there: rounding to the grid of ulp(p) commutes with adding 0.5,
// an exhaustive-ish host search over 24 M coordinates finds no index that differs).  The operations below are compiled under
// contract(off) INSIDE their own bodies: they carry no contract flag, so nothing fuses through them, in any translation unit.  Every
// index / weight / blend computation that must keep the reference's separately rounded operations uses these.
__device__ __forceinline__ float nvp_mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float nvp_add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float nvp_sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float nvp_div_rn(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

#define NVP_H NVP_HIDDEN
#define NVP_T NVP_TILE

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NVP_LAUNCH_CHECK()                       \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

__host__ __device__ inline int64_t nvp_ntiles(int64_t n) { return (n + NVP_T - 1) / NVP_T; }
__host__ __device__ inline int nvp_rows_even(int d) { return (d + 1) & ~1; }
__host__ __device__ inline int nvp_rows4(int d) { return (d + 3) & ~3; }      // PTM4 rows of the latent
__host__ __device__ inline int nvp_dz_stride_dev(int d) { return (d + 3) & ~3; }
__host__ __device__ inline int nvp_ztiles(int d) { return (nvp_rows_even(d) + 31) / 32; }

// ---------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 fragment geometry (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   D: lane l, register r holds D[row = 8*(r>>2) + 4*(l>>5) + (r&3)][col = l&31]
// All chained kernels put the OUTPUT FEATURE on i (A = weights), the PIXEL on j
// (B = activations), so a layer's D registers are directly the next layer's B
// operands: lane (j, h=l>>5) owns pixel j and the 16 rows {8g+4h+e} of each 32-row tile.
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int nvp_frag_row(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

// k-index consumed by lane-half h at chained step t (t = 16*T + r, T = source 32-row tile)
__host__ __device__ __forceinline__ int nvp_chain_k(int t, int h) { return 32 * (t >> 4) + nvp_frag_row(t & 15, h); }

__device__ __forceinline__ f32x16 nvp_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 nvp_zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// Pin a 16-register fragment at this point of the instruction stream: the scheduler can
// neither sink the computation of `v` below nor hoist its consumers above.
__device__ __forceinline__ void nvp_pin(f32x16& v) { asm volatile("" : "+v"(v)); }

// hardware fp32 atomic add (global_atomic_add_f32, no return)
__device__ __forceinline__ void nvp_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---------------------------------------------------------------------------------
#ifndef NVP_SINCOS_PI
#define NVP_SINCOS_PI 1        // 1: reduction by pi shared by a degree-9 sine and a degree-10 cosine polynomial (19 operations; 0: the pi/2 + Cephes form, 22)
#endif

// Branch-free fp32 sin/cos (no ocml slow path: a divergent branch inside the unrolled
// per-register epilogues forces whole-accumulator spills).  3-term Cody-Waite reduction
// by pi/2 with FMA + Cephes single-precision minimax polynomials on [-pi/4, pi/4].
// Measured max |err| vs float64 sin/cos: 9.3e-8 for |x| <= 1e5 (tests/test_oracle.py
// re-derives this on the host from the same constants).  The default (NVP_SINCOS_PI) instead reduces by pi once
// and evaluates a degree-9 sine and a degree-10 cosine polynomial on [-pi/2, pi/2]: 19 operations, max |err|
// 1.3e-7 / 1.4e-7 (same test), backward chain 2.48 -> 2.40 ms.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void nvp_sincos(float x, float& sn, float& cs) {
#ifdef NVP_ABL_NOSIN            // ablation builds only
    sn = x * 0.5f; cs = x * 0.25f; return;
#endif
#if NVP_SINCOS_PI               // A/B: reduction by pi shared by one odd (degree 9) and one even (degree 10) polynomial: 19 operations
    {
        const float n = __builtin_rintf(x * 0.318309886f);
        float r = __fmaf_rn(n, -3.14159274f, x);
        r = __fmaf_rn(n, 8.74227766e-08f, r);
        const float r2 = r * r;
        float p = __fmaf_rn(r2, 2.6000545605e-06f, -1.9806615092e-04f);
        p = __fmaf_rn(p, r2, 8.3330172897e-03f);
        p = __fmaf_rn(p, r2, -1.6666657096e-01f);
        const float s0 = __fmaf_rn(p * r2, r, r);
        float c = __fmaf_rn(r2, -2.6077104766e-07f, 2.4761886211e-05f);
        c = __fmaf_rn(c, r2, -1.3888403507e-03f);
        c = __fmaf_rn(c, r2, 4.1666640728e-02f);
        c = __fmaf_rn(c, r2, -4.9999999550e-01f);
        const float c0 = __fmaf_rn(c, r2, 1.0f);
        const unsigned flip = (unsigned)(int)n << 31;
        sn = __uint_as_float(__float_as_uint(s0) ^ flip);
        cs = __uint_as_float(__float_as_uint(c0) ^ flip);
        return;
    }
#endif
    const float n = __builtin_rintf(x * 0.636619747f);               // 2/pi
    float r = __fmaf_rn(n, -1.57079637e+00f, x);                       // 0x3fc90fdb
    r = __fmaf_rn(n, 4.37113883e-08f, r);                              // -(0xb33bbd2e)
    r = __fmaf_rn(n, 1.71512451e-15f, r);                              // -(0xa6f72ced)
    const int q = (int)n;
    const float r2 = r * r;
    float ps = __fmaf_rn(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __fmaf_rn(ps, r2, -1.6666654611e-1f);
    ps = __fmaf_rn(ps * r2, r, r);
    float pc = __fmaf_rn(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __fmaf_rn(pc, r2, 4.166664568298827e-2f);
    pc = __fmaf_rn(pc * r2, r2, __fmaf_rn(r2, -0.5f, 1.0f));
    const bool swap = q & 1;
    float s0 = swap ? pc : ps;
    float c0 = swap ? ps : pc;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// sin alone (forward pass, x_k rebuild in the dW staging): reduction by pi (2-term Cody-Waite) and ONE odd degree-9
// polynomial on [-pi/2, pi/2] (near-minimax fit, 4.6e-9 in exact arithmetic), sign from the parity of n: 13 VALU
// operations instead of the 22 of nvp_sincos.  Max |err| vs float64 sin: 1.3e-7 for |x| <= 1e5 (host emulation in
// tests/test_oracle.py).
__device__ __forceinline__ float nvp_sin(float x) {
#ifdef NVP_ABL_NOSIN            // ablation builds only (tools/ablate.sh): cheap stand-in activation
    return x * 0.5f;
#elif defined(NVP_SIN_VIA_SINCOS)  // A/B builds: the previous formulation
    float s_, c_;
    nvp_sincos(x, s_, c_);
    return s_;
#else
    const float n = __builtin_rintf(x * 0.318309886f);                 // 1/pi
    float r = __fmaf_rn(n, -3.14159274f, x);                           // 0x40490fdb
    r = __fmaf_rn(n, 8.74227766e-08f, r);                              // -(0xb3bbbd2e)
    const float r2 = r * r;
    float p = __fmaf_rn(r2, 2.6000545605e-06f, -1.9806615092e-04f);
    p = __fmaf_rn(p, r2, 8.3330172897e-03f);
    p = __fmaf_rn(p, r2, -1.6666657096e-01f);
    const float sn = __fmaf_rn(p * r2, r, r);
    return __uint_as_float(__float_as_uint(sn) ^ ((unsigned)(int)n << 31));
#endif
}
