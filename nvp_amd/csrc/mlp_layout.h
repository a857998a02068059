// Layout of the packed MFMA A-operand streams shared by mlp_pack.hip, mlp_fwd.hip and
// mlp_bwd.hip.  Host+device constexpr-style helpers only.
//
// A "step" is one v_mfma_f32_32x32x2_f32 k-step: lane (i = l&31, h = l>>5) supplies
// A[out-or-in row i of tile T'][k(step, h)].  A packed stream stores, for every step, one
// float4 per lane holding the value for the four 32-row output tiles T' = 0..3, so one
// coalesced 1-KiB global_load_dwordx4 feeds four MFMAs:
//        stream[(step*64 + lane)*4 + T']
// Streams whose output has ZT tiles (the latent gradient, ZT = ceil(rows/32)) store ZT
// floats per lane and step instead: stream[(step*64 + lane)*ZT + T'].
//
// Forward streams (A = W[out][in], out = 32T' + i):
//   step 0           : bias step - lane half 0 carries b[out], half 1 carries 0, B = 1.0
//   h-part, 64 steps : in = nvp_chain_k(t, h)            (skipped for the first modulator layer)
//   z-part, ZS steps : in = zoff + 2u + h,  ZS = rows/2  (modulator layers only; 0 beyond D)
// Backward streams (A = W^T: row i = INPUT index 32T' + i, k = OUTPUT index nvp_chain_k(t,h)),
//   64 steps each, no bias step.
#pragma once
#include "nvp_common.h"

struct NvpFwdLayout {
    int zs;            // z-part steps = rows/2
    int steps[5];      // mod0, mod1, mod2, sir1, sir2
    int64_t off[6];    // float offsets of each stream, off[5] = total
};

__host__ __device__ inline NvpFwdLayout nvp_fwd_layout(int d) {
    NvpFwdLayout L;
    L.zs = ((d + 1) & ~1) / 2;
    L.steps[0] = 1 + L.zs;
    L.steps[1] = 1 + 64 + L.zs;
    L.steps[2] = 1 + 64 + L.zs;
    L.steps[3] = 1 + 64;
    L.steps[4] = 1 + 64;
    int64_t o = 0;
    for (int i = 0; i < 5; ++i) { L.off[i] = o; o += (int64_t)L.steps[i] * 64 * 4; }
    L.off[5] = o;
    return L;
}

struct NvpBwdLayout {
    int zt;            // latent-gradient output tiles, rounded up to a multiple of 4 (4 or 8 supported)
    // streams: 0 sir2^T, 1 sir1^T, 2 mod2h^T, 3 mod1h^T (4 floats/lane/step)
    //          4 z0^T, 5 z1^T, 6 z2^T (zt floats/lane/step)
    int64_t off[8];
};

__host__ __device__ inline NvpBwdLayout nvp_bwd_layout(int d) {
    NvpBwdLayout L;
    L.zt = ((((d + 1) & ~1) + 31) / 32 + 3) & ~3;
    int64_t o = 0;
    for (int i = 0; i < 4; ++i) { L.off[i] = o; o += 64 * 64 * 4; }
    for (int i = 4; i < 7; ++i) { L.off[i] = o; o += (int64_t)64 * 64 * L.zt; }
    L.off[7] = o;
    return L;
}

// Flat layout of the 14 parameter tensors inside a dW partial / gradient vector.
struct NvpParamLayout {
    int64_t mod_w[3], mod_b[3], sir_w[3], sir_b[3], last_w, last_b, total;
};

__host__ __device__ inline NvpParamLayout nvp_param_layout(int d) {
    NvpParamLayout P;
    int64_t o = 0;
    for (int k = 0; k < 3; ++k) {
        P.mod_w[k] = o; o += (int64_t)NVP_HIDDEN * (k == 0 ? d : NVP_HIDDEN + d);
        P.mod_b[k] = o; o += NVP_HIDDEN;
    }
    for (int k = 0; k < 3; ++k) {
        P.sir_w[k] = o; o += (int64_t)NVP_HIDDEN * (k == 0 ? 1 : NVP_HIDDEN);
        P.sir_b[k] = o; o += NVP_HIDDEN;
    }
    P.last_w = o; o += 3 * NVP_HIDDEN;
    P.last_b = o; o += 3;
    P.total = o;
    return P;
}

// Per-tile record of the "small" gradients (last layer + SIREN layer 0: 643 values) that the backward chain
// kernel reduces over its 32 pixels itself (mlp_bwd.hip) and the dW stage only has to sum over tiles
// (mlp_dw.hip).  The record of tile t occupies the first kRecFloats floats of tile t's slot in stream 3 of
// the `dy` buffer (the slot that used to hold the dq0 stream).
constexpr int kRecLastW = 0;          // [3][128]
constexpr int kRecLastB = 384;        // [3] (+1 pad)
constexpr int kRecSir0W = 388;        // [128]
constexpr int kRecSir0B = 516;        // [128]
constexpr int kRecFloats = 644;
constexpr int kRecRowStride = 33;     // LDS transpose tile [128 features][32 px], conflict-free both ways
constexpr int kRecTileFloats = 128 * kRecRowStride;
