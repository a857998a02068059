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

#ifndef NVP_BWD_B3
#define NVP_BWD_B3 1         // 1: the backward chain (dX + latent gradient) too (mlp_bwd_b3.hip)
#endif
#ifndef NVP_FWD_B3
#define NVP_FWD_B3 1         // 1: the forward MLP runs on bf16 x 3 split MFMA for latents of <= 128 rows (mlp_fwd_b3.hip)
#endif

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

// ---- split-operand forward ("b3" kernels, mlp_fwd_b3.hip) ---------------------------------------------------
// Every fp32 operand is written as a sum of 16-bit parts and the significant part products are accumulated in fp32 by a
// 16-bit MFMA (32x32x16: 16 inputs per k-step).  Two splits are built (NVP_SPLIT_H2, mlp_b3.h):
//   * fp16 x 2 (default): x * 2^e = hi + lo in fp16 (11 + 11 significant bits; the power-of-two scale 2^e - per weight
//     stream, per pixel for activations - keeps the parts inside fp16's range and is divided out of the accumulator),
//     THREE products (lo*hi, hi*lo, hi*hi) on v_mfma_f32_32x32x16_f16, two operand quads per tile;
//   * bf16 x 3: x = hi + mid + lo in bf16 (8 + 8 + 8 bits, no scale needed), SIX products >= 2^-24 on
//     v_mfma_f32_32x32x16_bf16, three operand quads per tile (profiles/r01_probe_bf16x3_split_chain.txt).
// Both carry an error below that of an fp32 fma chain of the same length (profiles/r02_probe_f16x2_split.txt); fp16 x 2
// needs half the matrix cycles, two thirds of the weight bytes and half the split instructions.
// One k-step covers 16 inputs: lane (i, h) supplies A[out 32T'+i][k = 8h + q], q = 0..7, as one 16-B register quad per
// part.  Packed stream, in u32x4 units (P = kB3Parts):
//        stream[((step*4 + T')*P + part)*64 + lane]          part 0 hi, then (mid,) lo
// Steps of a layer: bias step (A[.][k=0] = b, B = e_0), then 8 steps per 128 chained inputs with
//        in(c, h, q) = 32 (c>>1) + 8 (2 (c&1) + (q>>2)) + 4 h + (q&3)        (= the D registers 8(c&1)..+7 of tile c>>1)
// then, for the modulator layers, ceil(rows/16) latent steps with in(s, h, q) = 16 s + 8 h + q (zero beyond D).
struct NvpFwdLayoutB3 {
    int zs;            // latent k-steps = ceil(rows / 16)
    int steps[5];      // mod0, mod1, mod2, sir1, sir2
    int64_t off[6];    // u32 offsets, off[5] = total
};
#ifndef NVP_SPLIT_H2
#define NVP_SPLIT_H2 1       // 1: fp16 x 2 scaled split, three products; 0: bf16 x 3 split, six products
#endif
constexpr int kB3Parts = NVP_SPLIT_H2 ? 2 : 3;   // 16-bit parts per fp32 operand
constexpr int kB3TileQuads = kB3Parts * 64;      // u32x4 per output tile of a k-step (one quad per lane and part)
constexpr int kB3StepQuads = 4 * kB3TileQuads;   // u32x4 per k-step of a four-tile stream
constexpr int kB3StepU32 = kB3StepQuads * 4;     // u32 per k-step (8 KiB fp16 x 2, 12 KiB bf16 x 3)
constexpr int kB3ZLdsSteps = 9;                  // latent k-steps kept in LDS per wave (144 rows = 18 KiB); the rest is read from the tensor

__host__ __device__ inline bool nvp_fwd_b3_ok(int d) { return ((d + 3) & ~3) <= 256; }     // forward: latent up to 256 rows
__host__ __device__ inline bool nvp_bwd_b3_ok(int d) { return ((d + 3) & ~3) <= 256; }     // backward chain; <= 128 rows: latent gradient fused (mlp_bwd_b3.hip)
__host__ __device__ inline int nvp_bwd_b3_zt(int d) { return ((d + 3) & ~3) <= 128 ? 4 : 8; }   // output tiles of the latent-gradient streams

// Streams are stored in the order the forward kernel CONSUMES them - mod0, mod1, sir1, mod2, sir2 - so that the kernel's
// running k-step index addresses the packed buffer linearly (mlp_b3_ring.h); off[] is still indexed by layer id
// (0 mod0, 1 mod1, 2 mod2, 3 sir1, 4 sir2), off[5] = total.
__host__ __device__ inline NvpFwdLayoutB3 nvp_fwd_layout_b3(int d) {
    NvpFwdLayoutB3 L;
    L.zs = (((d + 3) & ~3) + 15) / 16;
    L.steps[0] = 1 + L.zs; L.steps[1] = 1 + 8 + L.zs; L.steps[2] = 1 + 8 + L.zs; L.steps[3] = 1 + 8; L.steps[4] = 1 + 8;
    const int order[5] = {0, 1, 3, 2, 4};
    int64_t o = 0;
    for (int i = 0; i < 5; ++i) { L.off[order[i]] = o; o += (int64_t)L.steps[order[i]] * kB3StepU32; }
    L.off[5] = o;
    return L;
}

// Small per-row tables in D-REGISTER order, appended to both b3 packed buffers so that a lane fetches the 16 values of its
// rows of one 32-row tile with four 16-B loads instead of 16 scalar ones: table id t (0 sir_w0, 1 sir_b0, 2..4 last_w rows
// 0..2), lane half h, tile T, register r -> value at row 32 T + 8 (r>>2) + 4 h + (r&3):   tab[((t * 2 + h) * 4 + T) * 16 + r]
// Behind the tables: the power-of-two scales of the packed weight streams (fp16 x 2 split; all 1.0 for bf16 x 3), written by
// pack_b3_scales_kernel BEFORE the streams are packed: tab[kB3ScaleOff + seg] = 2^e of stream `seg` (forward: layer id
// 0..4; backward: stream 0..6), tab[kB3ScaleOff + 8 + seg] = 2^-e.
constexpr int kB3ScaleOff = 5 * 2 * 64;
constexpr int kB3TabFloats = kB3ScaleOff + 16;

__host__ __device__ inline int nvp_b3_chain_in(int c, int h, int q) { return 32 * (c >> 1) + 8 * (2 * (c & 1) + (q >> 2)) + 4 * h + (q & 3); }

// Backward b3 streams (A = W^T: row i = INPUT index 32T' + i, k = OUTPUT index nvp_b3_chain_in(c, h, q)), 8 steps each,
// no bias step: 0 sir2^T, 1 sir1^T, 2 mod2h^T, 3 mod1h^T (4 output tiles), 4 z0^T, 5 z1^T, 6 z2^T (zt = 4 or 8 output tiles: a
// step then holds zt * kB3Parts operand quads per lane; latent rows beyond D are zero).  Offsets in u32.
__host__ __device__ inline int64_t nvp_bwd_b3_off(int stream, int zt) {
    const int64_t h = 8 * (int64_t)kB3StepU32;
    return stream <= 4 ? stream * h : 4 * h + (stream - 4) * h * (zt / 4);
}


// ---- consumption-ordered copy of the backward b3 streams for the workgroup-shared ring (mlp_bwd_b3r.hip, zt == 4) ------
// The ring kernel walks ONE linear sequence of k-steps: sir2^T (8), then z2^T / mod2h^T interleaved per k-step (16),
// sir1^T (8), z1^T / mod1h^T interleaved (16), z0^T (8) = 56 k-steps of kB3StepU32.  The copy sits behind the tables.
constexpr int kBwdRingSteps = 56;
__host__ __device__ inline int64_t nvp_bwd_b3_ring_off(int zt) { return nvp_bwd_b3_off(7, zt) + kB3TabFloats; }
// position (k-step index in the ring copy) of k-step c of stream `stream`
__host__ __device__ inline int nvp_bwd_b3_ring_pos(int stream, int c) {
    switch (stream) {
        case 0: return c;
        case 6: return 8 + 2 * c;
        case 2: return 9 + 2 * c;
        case 1: return 24 + c;
        case 5: return 32 + 2 * c;
        case 3: return 33 + 2 * c;
        default: return 48 + c;      // stream 4
    }
}
