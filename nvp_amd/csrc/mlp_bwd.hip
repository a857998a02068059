#include <cstdlib>
// Backward of the modulator MLP + modulated SIREN w.r.t. activations (the "dX chain" of
// R12): from dL/drgb produce the latent gradient, the five per-pixel dY streams the
// weight-gradient GEMMs (mlp_dw.hip) contract over the pixel axis, and - already reduced over
// the tile's 32 pixels - the gradients of the two tiny layers (last layer, SIREN layer 0).
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

namespace {

constexpr int kWaves = 4;

// saved-activation loads of the element-wise stages (ablation hook: NVP_ABL_NOELOAD takes them from a register)
__device__ __forceinline__ void load_act16(f32x16& v, const float* __restrict__ tile_base, int T, int lane) {
#ifdef NVP_ABL_NOELOAD
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.25f + 0.001f * (float)(lane + T);
#else
    load_ptm16(v, tile_base, T, lane);
#endif
}

// ------------------------------------------------------------------------------------------
// Kernel A: the dX chain without the latent gradient.  Register plan (2 waves/SIMD, <= 256):
//   dx[4], dh[4] (128) are rewritten in place into dq, dp by the element-wise stage, then
//   dx' = chain(dq) needs one 64-register accumulator (192 live), after which dq is dead and
//   dh' = chain(dp) reuses the space.
// ------------------------------------------------------------------------------------------
// FUSE_DZ (latent <= 128 rows, i.e. one 64-register accumulator): the latent gradient
//   dz = W2[:,128:]^T dp2 + W1[:,128:]^T dp1 + W0^T dp0
// is accumulated by this kernel as a third chain per layer, straight from the dp registers, instead of by
// mlp_bwd_dz_kernel re-reading the three dp streams (1.5 KB/px).  Its accumulator does not fit next to
// dx/dh/acc, so between layers it is parked in the wave's LDS tile (free while the chains run: x2 is consumed
// before the layer-2 chains, dq0 is written after the accumulator has been fetched back for layer 0).
template <bool FUSE_DZ>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_bwd_dx_kernel(const float* __restrict__ drgb, const float* __restrict__ steps,
                                                                    const float* __restrict__ saved, nvp_mlp_params p,
                                                                    const float* __restrict__ packed,
                                                                    float* __restrict__ dy, float* __restrict__ dzr,
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
    float* dyt = dy + tb;                  // dp0,dp1,dp2,(records),dq1,dq2
    const float4* wp = reinterpret_cast<const float4*>(packed);

    // this wave's private LDS tile [128 features][32 px] (row stride 33): transposes x2 and dq0 so that a lane
    // can sum one feature row over the tile's pixels (the last layer's and SIREN layer 0's weight gradients)
    extern __shared__ __attribute__((aligned(16))) float xl_all[];
    float* xl = xl_all + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kRecTileFloats;
    float* rec = dy + 3 * act + tile * (int64_t)NVP_H * 32;      // this tile's record (stream-3 slot)

    f32x16 dx[4], dh[4];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (valid) { g0 = drgb[px * 3 + 0]; g1 = drgb[px * 3 + 1]; g2 = drgb[px * 3 + 2]; }

    // ---- last layer: dx2 = V3^T drgb (VALU, 3 terms)
    {
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
        load_act16(hv, hk, 0, lane);
        load_act16(qv, qk, 0, lane);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x16 hn, qn;
            if (T < 3) {
                load_act16(hn, hk, T + 1, lane);
                load_act16(qn, qk, T + 1, lane);
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
                if (k == 2) xl[(32 * T + nvp_frag_row(r, h)) * kRecRowStride + j] = sn * hv[r];     // x2 = sin(q2) h2
            }
            nvp_pin(dx[T]);
            nvp_pin(dh[T]);
            store_ptm16(dyt + (int64_t)(3 + k) * act, dx[T], T, lane);
            store_ptm16(dyt + (int64_t)k * act, dh[T], T, lane);
            NVP_LOAD_FENCE();
            if (T < 3) { hv = hn; qv = qn; }
        }
        if (k == 2) {
            // d last_w[c][f] = sum_px drgb[c][px] x2[f][px],  d last_b[c] = sum_px drgb[c][px]      (modulation.py:92)
            // lane l owns features l and l + 64; pixel px's drgb sits in lane px (readlane -> SGPR broadcast)
            NVP_LOAD_FENCE();
            float a[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            float b0 = 0.f, b1 = 0.f, b2 = 0.f;
            const float* r0 = xl + lane * kRecRowStride;
            const float* r1 = xl + (lane + 64) * kRecRowStride;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g0), q));
                const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g1), q));
                const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g2), q));
                const float x0 = r0[q], x1 = r1[q];
                a[0][0] = __fmaf_rn(c0, x0, a[0][0]); a[0][1] = __fmaf_rn(c1, x0, a[0][1]); a[0][2] = __fmaf_rn(c2, x0, a[0][2]);
                a[1][0] = __fmaf_rn(c0, x1, a[1][0]); a[1][1] = __fmaf_rn(c1, x1, a[1][1]); a[1][2] = __fmaf_rn(c2, x1, a[1][2]);
                b0 += c0; b1 += c1; b2 += c2;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rec[kRecLastW + c * NVP_H + lane] = a[0][c];
                rec[kRecLastW + c * NVP_H + 64 + lane] = a[1][c];
            }
            if (lane < 3) rec[kRecLastB + lane] = lane == 0 ? b0 : (lane == 1 ? b1 : b2);
            NVP_LOAD_FENCE();
        }
        // dx_{k-1} = V_k^T dq_k
        f32x16 acc[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
        chain_h<2>(acc, dx, wp + L.off[2 - k] / 4, lane);          // streams 0 (sir2^T), 1 (sir1^T)
#pragma unroll
        for (int T = 0; T < 4; ++T) { dx[T] = acc[T]; nvp_pin(dx[T]); }
        if (FUSE_DZ) {
            // dz += W_k[:, 128:]^T dp_k; the accumulator lives in LDS between layers (see above)
            float4* park = reinterpret_cast<float4*>(xl);
            if (k == 2) {
#pragma unroll
                for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            } else {
#pragma unroll
                for (int T = 0; T < 4; ++T)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 t = park[(T * 4 + g) * 64 + lane];
                        acc[T][4 * g] = t.x; acc[T][4 * g + 1] = t.y; acc[T][4 * g + 2] = t.z; acc[T][4 * g + 3] = t.w;
                    }
            }
            chain_hz<4>(acc, dh, packed + L.off[4 + k], lane);     // streams 6 (z2^T), 5 (z1^T)
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    park[(T * 4 + g) * 64 + lane] = make_float4(acc[T][4 * g], acc[T][4 * g + 1], acc[T][4 * g + 2], acc[T][4 * g + 3]);
            NVP_LOAD_FENCE();
        }
        // dh_{k-1} = W_k[:, :128]^T dp_k
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
        chain_h<2>(acc, dh, wp + L.off[4 - k] / 4, lane);          // streams 2 (mod2h^T), 3 (mod1h^T)
#pragma unroll
        for (int T = 0; T < 4; ++T) { dh[T] = acc[T]; nvp_pin(dh[T]); }
    }

    // ---- layer 0: q0 = 30 (w s + c) is recomputed
    {
        const float s = valid ? steps[px] : 0.f;
        const float* w0 = p.sir_w[0];
        const float* c0 = p.sir_b[0];
        const float* h0 = sv;
        f32x16 hv;
        load_act16(hv, h0, 0, lane);
        // The parked dz accumulator is fetched back quarter by quarter, just ahead of the dq0 rows that overwrite
        // its LDS region (quarter T of the accumulator sits in floats [1024 T, 1024 T + 1024), rows 32 T .. 32 T + 31 of
        // the dq0 tile in [1056 T, 1056 T + 1056)): that keeps the register peak of this stage below 256.
        f32x16 dzacc[4];
        auto fetch_dz = [&](int T) {
            const float4* park = reinterpret_cast<const float4*>(xl);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = park[(T * 4 + g) * 64 + lane];
                dzacc[T][4 * g] = t.x; dzacc[T][4 * g + 1] = t.y; dzacc[T][4 * g + 2] = t.z; dzacc[T][4 * g + 3] = t.w;
            }
            nvp_pin(dzacc[T]);
        };
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x16 hn;
            if (T < 3) load_act16(hn, h0, T + 1, lane);
            if (FUSE_DZ) {
                if (T == 0) fetch_dz(0);
                if (T < 3) fetch_dz(T + 1);
            }
            NVP_LOAD_FENCE();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * T + nvp_frag_row(r, h);
                const float q = 30.0f * __fmaf_rn(s, w0[row], c0[row]);
                float sn, cs;
                nvp_sincos(q, sn, cs);
                const float dxv = dx[T][r];
                dx[T][r] = 30.0f * (dxv * hv[r] * cs);        // dq0: gradient w.r.t. (w s + c)
                const float dhv = dh[T][r] + dxv * sn;
                dh[T][r] = hv[r] > 0.f ? dhv : dhv * 0.01f;   // dp0
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) xl[(32 * T + nvp_frag_row(r, h)) * kRecRowStride + j] = dx[T][r];
            nvp_pin(dh[T]);
            store_ptm16(dyt, dh[T], T, lane);
            NVP_LOAD_FENCE();
            if (T < 3) hv = hn;
        }
        // d sir_w0[f] = sum_px dq0[f][px] s[px],  d sir_b0[f] = sum_px dq0[f][px]        (modulation.py:53-56)
        float wl = 0.f, wh = 0.f, cl = 0.f, ch = 0.f;
        const float* r0 = xl + lane * kRecRowStride;
        const float* r1 = xl + (lane + 64) * kRecRowStride;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float sp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), q));
            const float d0 = r0[q], d1 = r1[q];
            wl = __fmaf_rn(d0, sp, wl); wh = __fmaf_rn(d1, sp, wh);
            cl += d0; ch += d1;
        }
        rec[kRecSir0W + lane] = wl; rec[kRecSir0W + 64 + lane] = wh;
        rec[kRecSir0B + lane] = cl; rec[kRecSir0B + 64 + lane] = ch;
        if (FUSE_DZ) {
            // dz += W_0^T dp_0, then the row-major store (same layout as mlp_bwd_dz_kernel)
            NVP_LOAD_FENCE();
            chain_hz<4>(dzacc, dh, packed + L.off[4], lane);       // stream 4 (z0^T)
            const int stride = nvp_dz_stride_dev(d);
            float* o = dzr + (tile * 32 + j) * stride;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int base = 32 * T + 8 * g + 4 * h;
                    if (base < stride)
                        *reinterpret_cast<float4*>(o + base) = make_float4(dzacc[T][4 * g], dzacc[T][4 * g + 1], dzacc[T][4 * g + 2], dzacc[T][4 * g + 3]);
                }
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

int nvp_mlp_bwd_b3r_launch(const float* drgb, const float* steps, const float* saved, const nvp_mlp_params* p,
                           const float* packed_bwd, float* dy, float* dz_rows, NvpDzLm lm, int64_t n, int32_t d, void* stream);      // mlp_bwd_b3r.hip
int nvp_mlp_bwd_b3_launch(const float* drgb, const float* steps, const float* saved, const nvp_mlp_params* p,
                          const float* packed_bwd, float* dy, float* dz_rows, NvpDzLm lm, int64_t n, int32_t d, void* stream);   // mlp_bwd_b3.hip

// 1 when nvp_mlp_bwd_dx honours `lm` for this latent width (bf16x3 chain kernels, F = 2 or 4): the host only hands the scatter's
// level-major buffers over (and tells nvp_encode_bwd so) when this says yes
extern "C" int32_t nvp_dz_lm_supported(int32_t d) {
    const int F = d / 57;
    return (NVP_BWD_B3 && nvp_bwd_b3_ok(d) && d == 57 * F && (F == 2 || F == 4)) ? 1 : 0;
}

extern "C" int nvp_mlp_bwd_dx(const float* drgb, const float* steps, const float* saved, const nvp_mlp_params* p,
                              const float* packed_bwd, float* dy, float* dz_rows, const nvp_scatter_lm* lm_host,
                              int64_t n, int32_t d, void* stream) {
    if (!drgb || !steps || !saved || !p || !packed_bwd || !dy || !dz_rows || n < 0 || d < 1) return NVP_ERR_BADARG;
    NvpDzLm lm = NVP_DZLM_OFF;
    if (lm_host && lm_host->dzs[0]) {
        if (!nvp_dz_lm_supported(d) || !lm_host->dzs[1] || !lm_host->dzmax || !lm_host->sdzmax) return NVP_ERR_UNSUPPORTED;
        if ((lm_host->scol0 & 3) || lm_host->scol0 < 0 || lm_host->scols < 0 || lm_host->scol0 + lm_host->scols > nvp_dz_stride_dev(d)) return NVP_ERR_BADARG;
        lm.dzs[0] = lm_host->dzs[0]; lm.dzs[1] = lm_host->dzs[1]; lm.dzmax = lm_host->dzmax;
        lm.sdzmax = lm_host->sdzmax; lm.scol0 = lm_host->scol0; lm.scols = lm_host->scols;
    }
    if (n == 0) return 0;
    if (NVP_BWD_B3 && nvp_bwd_b3_ok(d)) {
        // NVP_MLP_RING_BWD=0 (environment, read once): per-wave weight streaming (mlp_bwd_b3.hip) instead of the workgroup-shared LDS
        // weight ring (mlp_bwd_b3r.hip, default for fused-dz latents).  Bit-identical results; measured 2.28-2.35 ms vs 2.43-2.46 ms.
#if NVP_EXPERIMENTS
        static const bool ring = [] { const char* e = getenv("NVP_MLP_RING_BWD"); return !(e && e[0] == '0'); }();
        if (!ring) return nvp_mlp_bwd_b3_launch(drgb, steps, saved, p, packed_bwd, dy, dz_rows, lm, n, d, stream);
#endif
        return nvp_mlp_bwd_b3r_launch(drgb, steps, saved, p, packed_bwd, dy, dz_rows, lm, n, d, stream);
    }
    const int64_t ntiles = nvp_ntiles(n);
    const int zt = nvp_bwd_layout(d).zt;
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    if (zt != 4 && zt != 8) return NVP_ERR_UNSUPPORTED;       // latent wider than 256 rows (n_features_per_level = 8)
    const size_t lds = kWaves * kRecTileFloats * sizeof(float);
#ifndef NVP_BWD_FUSE_DZ
#define NVP_BWD_FUSE_DZ 1
#endif
    if (zt == 4 && NVP_BWD_FUSE_DZ) {
        hipLaunchKernelGGL(mlp_bwd_dx_kernel<true>, grid, dim3(kWaves * 64), lds, (hipStream_t)stream, drgb, steps, saved, *p, packed_bwd, dy, dz_rows, n, ntiles, d);
    } else {
        hipLaunchKernelGGL(mlp_bwd_dx_kernel<false>, grid, dim3(kWaves * 64), lds, (hipStream_t)stream, drgb, steps, saved, *p, packed_bwd, dy, dz_rows, n, ntiles, d);
        if (zt == 4)
            hipLaunchKernelGGL(mlp_bwd_dz_kernel<4>, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, dy, packed_bwd, dz_rows, ntiles, d);
        else
            hipLaunchKernelGGL(mlp_bwd_dz_kernel<8>, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, dy, packed_bwd, dz_rows, ntiles, d);
    }
    NVP_LAUNCH_CHECK();
    return 0;
}
