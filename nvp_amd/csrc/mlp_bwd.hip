// Backward of the modulator MLP + modulated SIREN w.r.t. activations (the "dX chain" of
// R12): from dL/drgb produce the latent gradient and the nine per-pixel streams the
// weight-gradient GEMMs (mlp_dw.hip) contract over the pixel axis.
//
// Same structure as mlp_fwd.hip: one wavefront = one 32-pixel tile, every transposed GEMM
//   dX[in][pixel] = sum_out W[out][in] * dY[out][pixel]
// runs on v_mfma_f32_32x32x2_f32 with A = packed W^T stream and B = the dY registers that
// the previous stage just produced; nothing is staged through LDS.  Two kernels, both at two
// waves per SIMD so one wave's element-wise / memory phases hide behind the other's MFMAs:
// the chain kernel (dq, dp, one accumulator: <= 256 registers) and the latent-gradient
// kernel, which re-reads the three dp streams it needs (an extra 1.5 KB/px of warm reads).
//
// Chain (forward names: p_k modulator pre-activation, h_k = lrelu(p_k), q_k SIREN
// pre-sine, x_k = sin(q_k) h_k, q_0 = 30 (w s + c)):
//   dx2 = V3^T drgb
//   dq_k = dx_k h_k cos(q_k);  dh_k (+)= dx_k sin(q_k);  dp_k = dh_k lrelu'(p_k)
//   dx_{k-1} = V_k^T dq_k;     dh_{k-1} = W_k[:, :128]^T dp_k;   dz += W_k[:, 128:]^T dp_k
// Bound: fp32 MFMA nominally (219 392 FLOP/px, nvp_s), in practice HBM WRITE bandwidth: the six
// dY streams are 3 KB/px out on top of 2.5 KB/px in.  The modulated sine outputs x_k are therefore NOT
// written: the dW kernel rebuilds x_k = sin(q_k) h_k from the forward pass's saved streams.
#include "mlp_chain.h"

#ifndef NVP_BWD_STORES
#define NVP_BWD_STORES 0     // where the dq/dp stream stores of a layer are issued: 0 per 32-row block inside the element-wise
#endif                       // stage; 1 one burst after it; 2 one burst behind preloaded weights; 3 two bursts (one per chain)
#ifndef NVP_BWD_PRE
#define NVP_BWD_PRE 8
#endif

namespace {

constexpr int kWaves = 4;

// ------------------------------------------------------------------------------------------
// Kernel A: the dX chain without the latent gradient.  Register plan (2 waves/SIMD, <= 256):
//   dx[4], dh[4] (128) are rewritten in place into dq, dp by the element-wise stage, then
//   dx' = chain(dq) needs one 64-register accumulator (192 live), after which dq is dead and
//   dh' = chain(dp) reuses the space.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_bwd_dx_kernel(const float* __restrict__ drgb, const float* __restrict__ steps,
                                                                    const float* __restrict__ saved, nvp_mlp_params p,
                                                                    const float* __restrict__ packed,
                                                                    float* __restrict__ dy,
                                                                    int64_t n, int64_t ntiles, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform
    if (tile >= ntiles) return;                       // wave-uniform
    nvp_stagger_start();
    const int j = lane & 31, h = lane >> 5;
    const NvpBwdLayout L = nvp_bwd_layout(d);
    const int64_t px = tile * 32 + j;
    const bool valid = px < n;
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    const int64_t tb = tile * (int64_t)NVP_H * 32;
    const float* sv = saved + tb;          // h0,h1,h2,q1,q2 at +k*act
    float* dyt = dy + tb;                  // dp0,dp1,dp2,dq0s,dq1,dq2
    const float4* wp = reinterpret_cast<const float4*>(packed);

    f32x16 dx[4], dh[4];

    // ---- last layer: dx2 = V3^T drgb (VALU, 3 terms)
    {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (valid) { g0 = drgb[px * 3 + 0]; g1 = drgb[px * 3 + 1]; g2 = drgb[px * 3 + 2]; }
        const float* w3 = p.last_w;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * T + nvp_frag_row(r, h);
                dx[T][r] = __fmaf_rn(w3[2 * NVP_H + row], g2, __fmaf_rn(w3[NVP_H + row], g1, w3[row] * g0));
            }
            nvp_pin(dx[T]);
            NVP_LOAD_FENCE();
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) dh[T] = nvp_zero16();
    }

    // ---- layers 2, 1: element-wise stage (dx,dh -> dq,dp in place) then the two transposed GEMMs
#pragma unroll
    for (int k = 2; k >= 1; --k) {
        const float* hk = sv + (int64_t)k * act;
        const float* qk = sv + (int64_t)(2 + k) * act;
        // vmcnt retires in order: a wait for loads issued AFTER a store burst also waits for those stores
        // (an HBM write round trip).  So block T+1's loads are issued before block T's stores.
        f32x16 hv, qv;
        load_ptm16(hv, hk, 0, lane);
        load_ptm16(qv, qk, 0, lane);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x16 hn, qn;
            if (T < 3) {
                load_ptm16(hn, hk, T + 1, lane);
                load_ptm16(qn, qk, T + 1, lane);
            }
            NVP_LOAD_FENCE();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sn, cs;
                nvp_sincos(qv[r], sn, cs);
                const float dxv = dx[T][r];
                dx[T][r] = dxv * hv[r] * cs;                      // dq
                const float dhv = dh[T][r] + dxv * sn;
                dh[T][r] = hv[r] > 0.f ? dhv : dhv * 0.01f;       // dp
            }
            nvp_pin(dx[T]);
            nvp_pin(dh[T]);
#if NVP_BWD_STORES == 0
            store_ptm16(dyt + (int64_t)(3 + k) * act, dx[T], T, lane);
            store_ptm16(dyt + (int64_t)k * act, dh[T], T, lane);
#endif
            NVP_LOAD_FENCE();
            if (T < 3) { hv = hn; qv = qn; }
        }
        const float4* w1 = wp + L.off[2 - k] / 4;                  // streams 0 (sir2^T), 1 (sir1^T)
        const float4* w2 = wp + L.off[4 - k] / 4;                  // streams 2 (mod2h^T), 3 (mod1h^T)
        f32x16 acc[4];
#if NVP_BWD_STORES <= 1
#if NVP_BWD_STORES == 1
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            store_ptm16(dyt + (int64_t)(3 + k) * act, dx[T], T, lane);
            store_ptm16(dyt + (int64_t)k * act, dh[T], T, lane);
        }
        NVP_LOAD_FENCE();
#endif
        // dx_{k-1} = V_k^T dq_k
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
        chain_h<2>(acc, dx, w1, lane);
#pragma unroll
        for (int T = 0; T < 4; ++T) { dx[T] = acc[T]; nvp_pin(dx[T]); }
        // dh_{k-1} = W_k[:, :128]^T dp_k
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
        chain_h<2>(acc, dh, w2, lane);
#pragma unroll
        for (int T = 0; T < 4; ++T) { dh[T] = acc[T]; nvp_pin(dh[T]); }
#else
        // The dq / dp registers stay live through the chains, so their stream stores can be issued as ONE
        // burst right after the first weights of the following chain have been requested (see chain_h_pre).
        {
            float4 pre[NVP_BWD_PRE];
            chain_preload(pre, w1, lane);
            NVP_CHAIN_FENCE();
#pragma unroll
            for (int T = 0; T < 4; ++T) store_ptm16(dyt + (int64_t)(3 + k) * act, dx[T], T, lane);
#if NVP_BWD_STORES == 2
#pragma unroll
            for (int T = 0; T < 4; ++T) store_ptm16(dyt + (int64_t)k * act, dh[T], T, lane);
#endif
            NVP_CHAIN_FENCE();
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            chain_h_pre<2, NVP_BWD_PRE>(acc, dx, w1, lane, pre);
#pragma unroll
            for (int T = 0; T < 4; ++T) { dx[T] = acc[T]; nvp_pin(dx[T]); }
        }
        {
#if NVP_BWD_STORES == 3
            float4 pre[NVP_BWD_PRE];
            chain_preload(pre, w2, lane);
            NVP_CHAIN_FENCE();
#pragma unroll
            for (int T = 0; T < 4; ++T) store_ptm16(dyt + (int64_t)k * act, dh[T], T, lane);
            NVP_CHAIN_FENCE();
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            chain_h_pre<2, NVP_BWD_PRE>(acc, dh, w2, lane, pre);
#else
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            chain_h<2>(acc, dh, w2, lane);
#endif
#pragma unroll
            for (int T = 0; T < 4; ++T) { dh[T] = acc[T]; nvp_pin(dh[T]); }
        }
#endif
    }

    // ---- layer 0: q0 = 30 (w s + c) is recomputed
    {
        const float s = valid ? steps[px] : 0.f;
        const float* w0 = p.sir_w[0];
        const float* c0 = p.sir_b[0];
        const float* h0 = sv;
        f32x16 hv;
        load_ptm16(hv, h0, 0, lane);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x16 hn, dq0, dp0;
            if (T < 3) load_ptm16(hn, h0, T + 1, lane);
            NVP_LOAD_FENCE();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * T + nvp_frag_row(r, h);
                const float q = 30.0f * __fmaf_rn(s, w0[row], c0[row]);
                float sn, cs;
                nvp_sincos(q, sn, cs);
                const float dxv = dx[T][r];
                dq0[r] = 30.0f * (dxv * hv[r] * cs);          // gradient w.r.t. (w s + c)
                const float dhv = dh[T][r] + dxv * sn;
                dp0[r] = hv[r] > 0.f ? dhv : dhv * 0.01f;
            }
            store_ptm16(dyt + 3 * act, dq0, T, lane);
            store_ptm16(dyt, dp0, T, lane);
            NVP_LOAD_FENCE();
            if (T < 3) hv = hn;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Kernel Z: latent gradient  dz = W0^T dp0 + W1[:,128:]^T dp1 + W2[:,128:]^T dp2, reading the
// three dp streams the chain kernel just wrote (L2/MALL-warm).  Output ROW-MAJOR
// [pixel][stride] (stride = D rounded up to 4): a lane owns 4 consecutive features per
// register group -> one 16-B store; the scatter stage gathers a pixel's features as
// contiguous 128-B runs.
// ------------------------------------------------------------------------------------------
template <int ZT>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_bwd_dz_kernel(const float* __restrict__ dy, const float* __restrict__ packed,
                                                                    float* __restrict__ dzr, int64_t ntiles, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform
    if (tile >= ntiles) return;
    nvp_stagger_start();
    const int j = lane & 31, h = lane >> 5;
    const NvpBwdLayout L = nvp_bwd_layout(d);
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    const float* dyt = dy + tile * (int64_t)NVP_H * 32;

    f32x16 dz[ZT];
#pragma unroll
    for (int T = 0; T < ZT; ++T) dz[T] = nvp_zero16();

    f32x16 b[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) load_ptm16(b[T], dyt + 2 * act, T, lane);
#pragma unroll
    for (int k = 2; k >= 0; --k) {
        f32x16 nb[4];
        if (ZT == 4 && k > 0) {             // prefetch the next dp stream while this one is consumed
#pragma unroll
            for (int T = 0; T < 4; ++T) load_ptm16(nb[T], dyt + (int64_t)(k - 1) * act, T, lane);
        }
        NVP_LOAD_FENCE();
        chain_hz<ZT>(dz, b, packed + L.off[4 + k], lane);       // streams 6 (z2^T), 5 (z1^T), 4 (z0^T)
        if (k > 0) {
            if (ZT == 4) {
#pragma unroll
                for (int T = 0; T < 4; ++T) b[T] = nb[T];
            } else {
#pragma unroll
                for (int T = 0; T < 4; ++T) load_ptm16(b[T], dyt + (int64_t)(k - 1) * act, T, lane);
            }
        }
    }
    const int stride = nvp_dz_stride_dev(d);
    float* o = dzr + (tile * 32 + j) * stride;
#pragma unroll
    for (int T = 0; T < ZT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int base = 32 * T + 8 * g + 4 * h;
            if (base < stride)
                *reinterpret_cast<float4*>(o + base) = make_float4(dz[T][4 * g], dz[T][4 * g + 1], dz[T][4 * g + 2], dz[T][4 * g + 3]);
        }
}

}  // namespace

extern "C" int nvp_mlp_bwd_dx(const float* drgb, const float* steps, const float* saved, const nvp_mlp_params* p,
                              const float* packed_bwd, float* dy, float* dz_rows, int64_t n, int32_t d, void* stream) {
    if (!drgb || !steps || !saved || !p || !packed_bwd || !dy || !dz_rows || n < 0 || d < 1) return NVP_ERR_BADARG;
    if (n == 0) return 0;
    const int64_t ntiles = nvp_ntiles(n);
    const int zt = nvp_bwd_layout(d).zt;
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    if (zt != 4 && zt != 8) return NVP_ERR_UNSUPPORTED;       // latent wider than 256 rows (n_features_per_level = 8)
    hipLaunchKernelGGL(mlp_bwd_dx_kernel, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, drgb, steps, saved, *p, packed_bwd, dy, n, ntiles, d);
    if (zt == 4)
        hipLaunchKernelGGL(mlp_bwd_dz_kernel<4>, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, dy, packed_bwd, dz_rows, ntiles, d);
    else
        hipLaunchKernelGGL(mlp_bwd_dz_kernel<8>, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, dy, packed_bwd, dz_rows, ntiles, d);
    NVP_LAUNCH_CHECK();
    return 0;
}
