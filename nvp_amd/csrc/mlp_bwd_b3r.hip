// Backward dX chain + latent gradient + per-tile records (mlp_bwd_b3.hip) with WORKGROUP-SHARED weight operands - latents of
// <= 128 rows: latent gradient fused (mlp_bwd_b3r_kernel<true>); 129-256 rows (nvp_l): mlp_bwd_b3r_kernel<false> followed by
// mlp_bwd_dz_b3r_kernel<8>, whose eight-tile weight steps go through a full-step ring (ZRing below): the A operands of every transposed GEMM come from a two-slot LDS ring that the four
// waves of the workgroup fill cooperatively (mlp_b3_ring.h, half-step ring: the per-wave transpose / parking tiles leave
// 12 KiB per workgroup with two workgroups per CU) instead of 8 / 12 KiB of per-wave global loads per k-step.  Arithmetic,
// streams, records and the latent gradient are those of mlp_bwd_b3_kernel<true>, bit for bit (same MFMA order per
// accumulator); the packed weights are read from the consumption-ordered copy behind the tables (mlp_layout.h).
#ifndef NVP_SPLIT_ASM
#define NVP_SPLIT_ASM 2        // chain kernels: residuals of the fp16 x 2 split as v_fma_mix with op_sel (mlp_b3.h); -0.02 ms each, same bits
#endif
#include "mlp_b3_ring.h"
#if NVP_EXPERIMENTS
#include "mlp_fwd_b3_tile.h"      // the forward tile body, for the tile-fused forward + backward-chain kernel (experiments build)
#include "../../include/nvp_hip_experiments.h"
#endif

#ifndef NVP_RING_MERGE_OPEN
#define NVP_RING_MERGE_OPEN 0     // experiment: a layer's dx chain and its shared dz / dh pass are fed by ONE ring fill (48 half-steps)
#endif

namespace {

constexpr int kWaves = 4;

__device__ __forceinline__ void load_act16(f32x16& v, const float* __restrict__ tile_base, int T, int lane) {
    load_ptm16(v, tile_base, T, lane);
}

// FUSE_DZ (latent <= 128 rows): the weights come from the consumption-ordered copy (z / mod-h streams interleaved per k-step);
// otherwise every chain walks ONE stream, whose ordinary packed layout already is its consumption order (a k-step's four tiles =
// two consecutive half-steps), so the ring reads the plain streams: half-step 16 s of stream s.
// One 32-pixel tile of the backward chain.  (g0, g1, g2): the loss gradient of this lane's pixel (0 beyond the batch); xl_all: the
// workgroup's LDS (four per-wave transpose / parking tiles, then the ring).  All four waves of the workgroup must call it (barriers).
template <bool FUSE_DZ>
__device__ __forceinline__ void bwd_b3r_tile(float g0, float g1, float g2, const float* __restrict__ steps,
                                             const float* __restrict__ saved, const nvp_mlp_params& p, const unsigned* __restrict__ packed,
                                             float* __restrict__ dy, float* __restrict__ dzr, const NvpDzLm& lm,
                                             int64_t n, int64_t ntiles, int d, int64_t tile, int wv, int lane, float* __restrict__ xl_all) {
    const int j = lane & 31, h = lane >> 5;
    const int64_t px = tile * 32 + j;
    const bool valid = px < n;
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    const int64_t tb = tile * (int64_t)NVP_H * 32;
    const float* sv = saved + tb;          // h0,h1,h2,q1,q2 at +k*act
    float* dyt = dy + tb;                  // dp0,dp1,dp2,(records),dq1,dq2
    const float* tab = reinterpret_cast<const float*>(packed + nvp_bwd_b3_off(7, nvp_bwd_b3_zt(d)));      // sir_w0 / sir_b0 / last_w in D-register order
    const float* wsc = tab + kB3ScaleOff;              // 2^e of each weight stream, 2^-e at + 8 (mlp_layout.h)

    // this wave's private LDS tile [128 features][32 px] (row stride 33): transposes x2 and dq0 so that a lane
    // can sum one feature row over the tile's pixels (the last layer's and SIREN layer 0's weight gradients)
    float* xl = xl_all + wv * kRecTileFloats;                               // xl_all: [4 waves][kRecTileFloats] | ring: 2 x kHalfQuads u32x4
    HRing R;
    R.lds = reinterpret_cast<u32x4*>(xl_all + kWaves * kRecTileFloats);
    R.g = reinterpret_cast<const u32x4*>(FUSE_DZ ? packed + nvp_bwd_b3_ring_off(4) : packed);
    R.chain_end = 0; R.wv = wv; R.lane = lane;
    int hs = 0;                                        // running half-step: the ring copy of the weights is in consumption order
    float* rec = dy + 3 * act + tile * (int64_t)NVP_H * 32;      // this tile's record (stream-3 slot)

    f32x16 dx[4], dh[4];

    // ---- last layer: dx2 = V3^T drgb (VALU, 3 terms)
    {
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float v0[16], v1[16], v2[16];
            load_tab16(v0, tab, 2, T, h);
            load_tab16(v1, tab, 3, T, h);
            load_tab16(v2, tab, 4, T, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[T][r] = __fmaf_rn(v2[r], g2, __fmaf_rn(v1[r], g1, v0[r] * g0));
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
        {
            const PxScale pq = px_scale(fmaxf(px_absmax(dx), kTinyMax));
            if (!FUSE_DZ) hs = 16 * (2 - k);
            chain_h_b3_hring(acc, dx, pq.s, R, hs, lane, (FUSE_DZ && NVP_RING_MERGE_OPEN) ? 48 : 16);        // streams 0 (sir2^T), 1 (sir1^T)
            scale4(acc, pq.u * wsc[8 + 2 - k]);
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) { dx[T] = acc[T]; nvp_pin(dx[T]); }
        const PxScale pp = px_scale(fmaxf(px_absmax(dh), kTinyMax));       // dp_k feeds the dz and the dh chain
        if (!FUSE_DZ) {
            // dh_{k-1} = W_k[:, :128]^T dp_k; the latent gradient is mlp_bwd_dz_b3r_kernel's
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            hs = 16 * (4 - k);
            chain_h_b3_hring(acc, dh, pp.s, R, hs, lane, 16);          // streams 2 (mod2h^T), 3 (mod1h^T)
            scale4(acc, pp.u * wsc[8 + 4 - k]);
#pragma unroll
            for (int T = 0; T < 4; ++T) { dh[T] = acc[T]; nvp_pin(dh[T]); }
            continue;
        } else {
            // dz += W_k[:, 128:]^T dp_k and dh_{k-1} = W_k[:, :128]^T dp_k in ONE pass over dp (one operand split per
            // k-step instead of two).  Both accumulators are live, so dx' waits in the wave's LDS tile meanwhile: the
            // parked dz accumulator is swapped out for it before the pass and back in after it.
            float4* park = reinterpret_cast<float4*>(xl);
            f32x16 dzacc[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                if (k == 2) {
                    dzacc[T] = nvp_zero16();
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 t = park[(T * 4 + g) * 64 + lane];
                        dzacc[T][4 * g] = t.x; dzacc[T][4 * g + 1] = t.y; dzacc[T][4 * g + 2] = t.z; dzacc[T][4 * g + 3] = t.w;
                    }
                    if (NVP_SPLIT_H2) dzacc[T] *= pp.s * wsc[4 + k];      // into this chain's scaled units (exact)
                }
                NVP_LOAD_FENCE();
#pragma unroll
                for (int g = 0; g < 4; ++g)                        // dx' (tile T) takes the slot its dz quarter just left
                    park[(T * 4 + g) * 64 + lane] = make_float4(dx[T][4 * g], dx[T][4 * g + 1], dx[T][4 * g + 2], dx[T][4 * g + 3]);
                NVP_LOAD_FENCE();
            }
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            chain_h2_b3_hring(dzacc, acc, dh, pp.s, R, hs, lane, NVP_RING_MERGE_OPEN ? 0 : 32);    // streams 6/2 (z2^T, mod2h^T), 5/3 (z1^T, mod1h^T), interleaved per k-step
            scale4(acc, pp.u * wsc[8 + 4 - k]);
            scale4(dzacc, pp.u * wsc[8 + 4 + k]);
#pragma unroll
            for (int T = 0; T < 4; ++T) { dh[T] = acc[T]; nvp_pin(dh[T]); }
#pragma unroll
            for (int T = 0; T < 4; ++T) {                          // swap back: dx' into registers, dz accumulator into LDS
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = park[(T * 4 + g) * 64 + lane];
                    dx[T][4 * g] = t.x; dx[T][4 * g + 1] = t.y; dx[T][4 * g + 2] = t.z; dx[T][4 * g + 3] = t.w;
                }
                NVP_LOAD_FENCE();
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    park[(T * 4 + g) * 64 + lane] = make_float4(dzacc[T][4 * g], dzacc[T][4 * g + 1], dzacc[T][4 * g + 2], dzacc[T][4 * g + 3]);
                nvp_pin(dx[T]);
                NVP_LOAD_FENCE();
            }
            continue;
        }
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
            float w0v[16], c0v[16];
            load_tab16(w0v, tab, 0, T, h);
            load_tab16(c0v, tab, 1, T, h);
            NVP_LOAD_FENCE();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float q = 30.0f * __fmaf_rn(s, w0v[r], c0v[r]);
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
            const PxScale pp = px_scale(fmaxf(px_absmax(dh), kTinyMax));
            scale4(dzacc, pp.s * wsc[4]);
            chain_h_b3_hring(dzacc, dh, pp.s, R, hs, lane);     // stream 4 (z0^T)
            scale4(dzacc, pp.u * wsc[8 + 4]);
            const int stride = nvp_dz_stride_dev(d);
            float* o = dzr + (tile * 32 + j) * stride;
            const int F = d / 57;                       // latent = 57 F columns (modules.py:42-45)
            unsigned mx = 0u, msx = 0u;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int base = 32 * T + 8 * g + 4 * h;
                    if (base < stride && !nvp_dz_store_lm(lm, F, base, px, n, dzacc[T][4 * g], dzacc[T][4 * g + 1], dzacc[T][4 * g + 2], dzacc[T][4 * g + 3], mx, msx))
                        *reinterpret_cast<float4*>(o + base) = make_float4(dzacc[T][4 * g], dzacc[T][4 * g + 1], dzacc[T][4 * g + 2], dzacc[T][4 * g + 3]);
                }
            nvp_dz_lm_finish(lm, mx, msx, tile, lane);
        }
    }
}


template <bool FUSE_DZ>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_bwd_b3r_kernel(const float* __restrict__ drgb, const float* __restrict__ steps,
                                                                    const float* __restrict__ saved, nvp_mlp_params p,
                                                                    const unsigned* __restrict__ packed,
                                                                    float* __restrict__ dy, float* __restrict__ dzr, NvpDzLm lm,
                                                                    int64_t n, int64_t ntiles, int d) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t tile = (int64_t)blockIdx.x * kWaves + wv;   // provably wave-uniform
    // Every wave of the workgroup walks the ring (barriers).  A wave beyond the last tile recomputes the last tile and
    // rewrites that tile's outputs with the very same values (benign: identical bits), so no store needs a predicate.
    if (tile >= ntiles) tile = ntiles - 1;
    nvp_stagger_start();
    extern __shared__ __attribute__((aligned(16))) float xl_all[];          // [4 waves][kRecTileFloats] | ring: 2 x kHalfQuads u32x4
    const int64_t px = tile * 32 + (lane & 31);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (px < n) { g0 = drgb[px * 3 + 0]; g1 = drgb[px * 3 + 1]; g2 = drgb[px * 3 + 2]; }
    bwd_b3r_tile<FUSE_DZ>(g0, g1, g2, steps, saved, p, packed, dy, dzr, lm, n, ntiles, d, tile, wv, lane, xl_all);
}


// ---- latent gradient of a wide latent through a full-step ring ---------------------------------------------------------
// mlp_bwd_dz_b3_kernel<8> (mlp_bwd_b3.hip) pulls the three eight-tile z^T streams - 384 KiB per 32-pixel tile - through every
// wave's own vector-memory path.  Here the four waves of a workgroup share them: a ring slot holds one whole k-step (ZT tiles x
// kP parts = 16 KiB fp16 x 2), every wave moves a quarter of it (fetched into registers two k-steps ahead, written to the free
// slot one k-step ahead), one barrier per k-step.  This kernel parks nothing in LDS, so two 16-KiB slots per workgroup fit
// easily with two workgroups per CU.  The 24 k-steps of a tile (z2^T, z1^T, z0^T) are ONE ring sequence: no refill at the layer
// boundaries.  Same MFMA order per accumulator as the per-wave kernel: bit-identical.
template <int ZT>
struct ZRing {
    static constexpr int kPieces = ZT * kP;        // 1-KiB pieces per k-step
    static constexpr int kMine = kPieces / 4;      // per wave
    static constexpr int kQuads = kPieces * 64;    // u32x4 per k-step
    u32x4* lds;                // two slots of kQuads
    const u32x4* g;            // packed buffer
    int wv, lane;
    u32x4 sg[kMine];
    // ring step s = k-step (s & 7) of stream 6 - (s >> 3): z2^T, z1^T, z0^T in the order the layers are walked
    __device__ __forceinline__ const u32x4* src(int s) const { return g + nvp_bwd_b3_off(6 - (s >> 3), ZT) / 4 + (int64_t)(s & 7) * kQuads; }
    __device__ __forceinline__ void fetch(int s) {
        const u32x4* p = src(s) + (kMine * wv) * 64;
#pragma unroll
        for (int q = 0; q < kMine; ++q) sg[q] = (p + q * 64)[(unsigned)lane];
    }
    __device__ __forceinline__ void publish(int s) {
        u32x4* d = lds + (s & 1) * kQuads + (kMine * wv) * 64;
#pragma unroll
        for (int q = 0; q < kMine; ++q) (d + q * 64)[(unsigned)lane] = sg[q];
    }
    __device__ __forceinline__ void open() { fetch(0); publish(0); fetch(1); end(); }
    __device__ __forceinline__ const u32x4* begin(int s) {
        if (s + 1 < 24) publish(s + 1);
        if (s + 2 < 24) fetch(s + 2);
        return lds + (s & 1) * kQuads;
    }
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
};

template <int ZT>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_bwd_dz_b3r_kernel(const float* __restrict__ dy, const unsigned* __restrict__ packed,
                                                                        float* __restrict__ dzr, NvpDzLm lm, int64_t n, int64_t ntiles, int d) {
    static_assert((ZT * kP) % 4 == 0, "a k-step must split into four equal shares");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t tile = (int64_t)blockIdx.x * kWaves + wv;
    if (tile >= ntiles) tile = ntiles - 1;             // every wave walks the ring; a surplus wave rewrites the last tile's values (identical bits)
    nvp_stagger_start();
    const int j = lane & 31, h = lane >> 5;
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    const float* dyt = dy + tile * (int64_t)NVP_H * 32;
    const float* wsc = reinterpret_cast<const float*>(packed + nvp_bwd_b3_off(7, ZT)) + kB3ScaleOff;
    extern __shared__ __attribute__((aligned(16))) float xl_all[];
    ZRing<ZT> R;
    R.lds = reinterpret_cast<u32x4*>(xl_all);
    R.g = reinterpret_cast<const u32x4*>(packed);
    R.wv = wv; R.lane = lane;
    const unsigned ul = (unsigned)lane;
    f32x16 dz[ZT];
#pragma unroll
    for (int T = 0; T < ZT; ++T) dz[T] = nvp_zero16();
    f32x16 b[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) load_ptm16(b[T], dyt + 2 * act, T, lane);
    R.open();
    int s = 0;
#pragma unroll 1
    for (int k = 2; k >= 0; --k) {
        NVP_LOAD_FENCE();
        const PxScale pp = px_scale(fmaxf(px_absmax(b), kTinyMax));
        if (NVP_SPLIT_H2) {                                         // the running sum into this layer's scaled units (exact)
            const float f = pp.s * wsc[4 + k];
#pragma unroll
            for (int T = 0; T < ZT; ++T) dz[T] *= f;
        }
        NVP_CHAIN_ENTER();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const u32x4* w = R.begin(s);
            float x[8];
            chain_in8(x, b, c);
            BOp bo;
            split8(x, pp.s, bo);
            u32x4 a[2][kP];
#pragma unroll
            for (int q = 0; q < kP; ++q) a[0][q] = (w + q * 64)[ul];
#pragma unroll
            for (int T = 0; T < ZT; ++T) {
                if (T + 1 < ZT) {
#pragma unroll
                    for (int q = 0; q < kP; ++q) a[(T + 1) & 1][q] = (w + ((T + 1) * kP + q) * 64)[ul];
                }
                NVP_CHAIN_FENCE();
                mac_parts(dz[T], a[T & 1], bo);
            }
            R.end();
            ++s;
        }
        NVP_CHAIN_LEAVE();
#ifndef NVP_ABL_DZ_NOLOAD        // ablation builds only (wrong results): what the exposed dp loads of layers 1 and 0 cost
        if (k > 0) {
#pragma unroll
            for (int T = 0; T < 4; ++T) load_ptm16(b[T], dyt + (int64_t)(k - 1) * act, T, lane);
        }
#endif
        if (NVP_SPLIT_H2) {
            const float f = pp.u * wsc[8 + 4 + k];
#pragma unroll
            for (int T = 0; T < ZT; ++T) dz[T] *= f;
        }
    }
    const int stride = nvp_dz_stride_dev(d);
    float* o = dzr + (tile * 32 + j) * stride;
    const int F = d / 57;
    const int64_t px = tile * 32 + j;
    unsigned mx = 0u, msx = 0u;
#pragma unroll
    for (int T = 0; T < ZT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int base = 32 * T + 8 * g + 4 * h;
            if (base < stride && !nvp_dz_store_lm(lm, F, base, px, n, dz[T][4 * g], dz[T][4 * g + 1], dz[T][4 * g + 2], dz[T][4 * g + 3], mx, msx))
                *reinterpret_cast<float4*>(o + base) = make_float4(dz[T][4 * g], dz[T][4 * g + 1], dz[T][4 * g + 2], dz[T][4 * g + 3]);
        }
    nvp_dz_lm_finish(lm, mx, msx, tile, lane);
}

#if NVP_EXPERIMENTS
// ---- tile-fused forward + backward chain (nvp_encode_mlp_fwd_bwd) -----------------------------------------------------------------------
// The three-kernel step writes the five saved streams (2 560 B per pixel) in the forward kernel and reads them back in the backward
// chain a full kernel later, from HBM.  image_mse's gradient is per pixel - d/d rgb = 2 (rgb - gt) / (3 N), loss_functions.py:1-3 - so
// a wave can run the backward chain of its tile right after the forward of the SAME tile: the streams it reads are the ones it has just
// written (memory-side-cache hits instead of HBM reads: the backward chain alone runs 16 % faster on cache-hot inputs,
// profiles/r04_probe_hot_streams.txt).  Both bodies are the ones of the separate kernels (fwd_b3_tile, bwd_b3r_tile): same arithmetic,
// same buffers, bit-identical outputs; only the order of the work changes.  The wave's LDS region serves first as the forward's latent
// tile, then as the backward's transpose / parking tile.
template <int GF>
__global__ __launch_bounds__(kWaves * 64, 2) void step_b3_kernel(float* __restrict__ zt, const float* __restrict__ steps, const uint8_t* __restrict__ gt, float gscale,
                                                                nvp_mlp_params p, const unsigned* __restrict__ packed_f, const unsigned* __restrict__ packed_b,
                                                                float* __restrict__ rgb, float* __restrict__ saved, float* __restrict__ dy, float* __restrict__ dzr,
                                                                NvpDzLm lm, int64_t n, int64_t ntiles, int d, NvpTileEnc enc) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t tile = (int64_t)blockIdx.x * kWaves + wv;
    if (tile >= ntiles) tile = ntiles - 1;              // duplicate walk of the last tile (the ring's barriers need all four waves): identical stores
    nvp_stagger_start();
    extern __shared__ __attribute__((aligned(16))) float xl_all[];          // [4 waves][kRecTileFloats] | ring
    float rgb_px[3];
    fwd_b3_tile<true, GF>(zt, steps, p, packed_f, rgb, saved, n, ntiles, d, enc, tile, true, reinterpret_cast<float4*>(xl_all + wv * kRecTileFloats), lane, rgb_px);
    // the loss gradient of this lane's pixel, exactly as mse_u8_kernel (harness.hip) computes it
    const int64_t px = tile * 32 + (lane & 31);
    float g[3] = {0.f, 0.f, 0.f};
    if (px < n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q = nvp_div_rn(nvp_sub_rn((float)gt[px * 3 + c], 127.5f), 127.5f);
            const float df = rgb_px[c] - q;
            g[c] = df * gscale;
        }
    }
    // The backward chain reads what THIS wave has just stored: the stores only have to have left the wave (vector L1 is write-through,
    // the lines were never read before, so no stale copy exists).  An agent-scope release / acquire pair here costs an L2 write-back and
    // an L1 invalidate per wave: the kernel took 6.35 ms with it (profiles/r04_ab_tile_fused.txt).
#ifndef NVP_STEP_AGENT_FENCE
#define NVP_STEP_AGENT_FENCE 0
#endif
#if NVP_STEP_AGENT_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    bwd_b3r_tile<true>(g[0], g[1], g[2], steps, saved, p, packed_b, dy, dzr, lm, n, ntiles, d, tile, wv, lane, xl_all);
}

#endif  // NVP_EXPERIMENTS

}  // namespace

// called by nvp_mlp_bwd_dx (mlp_bwd.hip) when NVP_BWD_B3 is on and NVP_MLP_RING_BWD != 0
int nvp_mlp_bwd_b3r_launch(const float* drgb, const float* steps, const float* saved, const nvp_mlp_params* p,
                           const float* packed_bwd, float* dy, float* dz_rows, NvpDzLm lm, int64_t n, int32_t d, void* stream) {
    const int64_t ntiles = nvp_ntiles(n);
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    const size_t lds = kWaves * kRecTileFloats * sizeof(float) + 2 * kHalfQuads * sizeof(u32x4);        // 67 584 + 8 192 / 12 288 B: two workgroups per CU
    const unsigned* pk = reinterpret_cast<const unsigned*>(packed_bwd);
    if (nvp_bwd_b3_zt(d) == 4) {
        hipLaunchKernelGGL(mlp_bwd_b3r_kernel<true>, grid, dim3(kWaves * 64), lds, (hipStream_t)stream, drgb, steps, saved, *p, pk, dy, dz_rows, lm, n, ntiles, d);
    } else {
        hipLaunchKernelGGL(mlp_bwd_b3r_kernel<false>, grid, dim3(kWaves * 64), lds, (hipStream_t)stream, drgb, steps, saved, *p, pk, dy, dz_rows, NVP_DZLM_OFF, n, ntiles, d);
        const size_t zlds = 2 * (size_t)ZRing<8>::kQuads * sizeof(u32x4);                                 // 32 / 48 KiB
        hipLaunchKernelGGL(mlp_bwd_dz_b3r_kernel<8>, grid, dim3(kWaves * 64), zlds, (hipStream_t)stream, dy, pk, dz_rows, lm, n, ntiles, d);
    }
    NVP_LAUNCH_CHECK();
    return 0;
}

#if NVP_EXPERIMENTS
// ---- C ABI of the tile-fused step -------------------------------------------------------------------------------------------------
static bool step_fused_ok(const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh, int* d_out) {
    int d = 0;
    bool wide = false;
    if (!NVP_BWD_B3 || !fused_ok(lv_xy, lv_yt, lv_xt, sh, &d, &wide) || wide) return false;      // (the tile-fused step is built for latents that fit the LDS tile)
    if (nvp_bwd_b3_zt(d) != 4 || !nvp_dz_lm_supported(d)) return false;                          // fused latent gradient, level-major hand-over
    if (nvp_fwd_layout_b3(d).zs * 4 * 32 * 4 > kRecTileFloats) return false;                     // the latent tile must fit the backward's per-wave LDS tile
    if (d_out) *d_out = d;
    return true;
}

extern "C" int32_t nvp_encode_mlp_fwd_bwd_supported(const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh) {
    return step_fused_ok(lv_xy, lv_yt, lv_xt, sh, nullptr) ? 1 : 0;
}

extern "C" int nvp_encode_mlp_fwd_bwd(const float* coords, const float* steps, const uint8_t* gt_u8, const float* kf_xy, const float* kf_yt, const float* kf_xt,
                                      const float* emb, const nvp_mlp_params* p, const float* packed_fwd, const float* packed_bwd, float* rgb, float* saved,
                                      float* zt, float* dy, float* dz_rows, const nvp_scatter_lm* lm_host, int64_t n,
                                      const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt, const nvp_sparse_shape* sh, void* stream) {
    int d = 0;
    if (!step_fused_ok(lv_xy, lv_yt, lv_xt, sh, &d)) return NVP_ERR_UNSUPPORTED;
    if (!coords || !steps || !gt_u8 || !kf_xy || !kf_yt || !kf_xt || !emb || !p || !packed_fwd || !packed_bwd || !rgb || !saved || !zt || !dy || !dz_rows || n < 0)
        return NVP_ERR_BADARG;
    NvpDzLm lm = NVP_DZLM_OFF;
    if (lm_host && lm_host->dzs[0]) {
        if (!lm_host->dzs[1] || !lm_host->dzmax || !lm_host->sdzmax) return NVP_ERR_BADARG;
        if ((lm_host->scol0 & 3) || lm_host->scol0 < 0 || lm_host->scols < 0 || lm_host->scol0 + lm_host->scols > nvp_dz_stride_dev(d)) return NVP_ERR_BADARG;
        lm.dzs[0] = lm_host->dzs[0]; lm.dzs[1] = lm_host->dzs[1]; lm.dzmax = lm_host->dzmax;
        lm.sdzmax = lm_host->sdzmax; lm.scol0 = lm_host->scol0; lm.scols = lm_host->scols;
    }
    if (n == 0) return 0;
    NvpTileEnc e;
    e.lv[0] = *lv_xy; e.lv[1] = *lv_yt; e.lv[2] = *lv_xt; e.sh = *sh;
    e.kf[0] = kf_xy; e.kf[1] = kf_yt; e.kf[2] = kf_xt; e.emb = emb; e.coords = coords;
    int col = 0;
    for (int q = 0; q < 3; ++q) { e.col0[q] = col; col += e.lv[q].n_levels * e.lv[q].n_features; }
    e.col0[3] = col;
    e.rows = nvp_rows4(d);
    const int64_t ntiles = nvp_ntiles(n);
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    const size_t lds = kWaves * kRecTileFloats * sizeof(float) + 2 * kHalfQuads * sizeof(u32x4);
    const float gscale = 2.0f / (float)(n * 3);                   // == nvp_mse_u8's
    const unsigned* pf = reinterpret_cast<const unsigned*>(packed_fwd);
    const unsigned* pb = reinterpret_cast<const unsigned*>(packed_bwd);
    if (lv_xy->n_features == 2)
        hipLaunchKernelGGL((step_b3_kernel<2>), grid, dim3(kWaves * 64), lds, (hipStream_t)stream, zt, steps, gt_u8, gscale, *p, pf, pb, rgb, saved, dy, dz_rows, lm, n, ntiles, d, e);
    else
        hipLaunchKernelGGL((step_b3_kernel<4>), grid, dim3(kWaves * 64), lds, (hipStream_t)stream, zt, steps, gt_u8, gscale, *p, pf, pb, rgb, saved, dy, dz_rows, lm, n, ntiles, d, e);
    NVP_LAUNCH_CHECK();
    return 0;
}
#endif  // NVP_EXPERIMENTS
