// Forward of the modulator MLP + modulated SIREN (R8-R10) on split-operand MFMA (mlp_b3.h) with WORKGROUP-SHARED weight operands
// (mlp_b3_ring.h): same arithmetic, same saved streams and the same RGB as mlp_fwd_b3.hip, bit for bit - the MFMA
// sequence per accumulator is unchanged; what changes is where the A operands come from (a two-slot LDS ring filled
// cooperatively by the workgroup's four waves instead of 8 / 12 KiB of per-wave global loads per k-step) and where the
// latent comes from (the B operands of the latent k-steps are read from the PTM4 tensor one step ahead - 2 x 16 B per
// lane and step - which frees the 64 KiB of LDS the per-wave latent tiles used, so two workgroups still share a CU).
#ifndef NVP_SPLIT_ASM
#define NVP_SPLIT_ASM 2        // chain kernels: residuals of the fp16 x 2 split as v_fma_mix with op_sel (mlp_b3.h); -0.02 ms each, same bits
#endif
#include "mlp_b3_ring.h"

#ifndef NVP_RING_FLAGS
#define NVP_RING_FLAGS 0        // 1: flag-synchronised ring of kRingSlots slots (WRingF) instead of the two-slot barrier ring
#endif

namespace {

constexpr int kWaves = 4;
#if NVP_RING_FLAGS
constexpr int kRingSlots = 4;
typedef WRingF<kRingSlots> Ring;
#else
constexpr int kRingSlots = 2;
typedef WRing Ring;
#endif

// B operands of latent k-step s (rows 16 s + 8 h .. + 7 of pixel j), fetched one step ahead
struct ZFeed {
    const float4* zg;          // this tile's PTM4 latent (row-group rg of pixel j at zg[rg * 32 + j])
    int rg_end;                // row-groups the tensor holds (rows / 4); beyond: zero
    int j, h;
    float4 t0, t1;
    __device__ __forceinline__ void fetch(int s) {
        const int rg = 4 * s + 2 * h;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        t0 = rg < rg_end ? zg[rg * 32 + j] : zero;
        t1 = rg + 1 < rg_end ? zg[(rg + 1) * 32 + j] : zero;
    }
};

// `ns` latent k-steps; Z holds step 0's rows on entry (fetched one k-step earlier by the caller)
__device__ __forceinline__ void chain_z_b3_ring(f32x16 (&acc)[4], ZFeed& Z, int ns, const float sc, Ring& R, int& s, int lane) {
#pragma unroll 1
    for (int u = 0; u < ns; ++u) {
        const u32x4* w = R.begin(s);
        const float x[8] = {Z.t0.x, Z.t0.y, Z.t0.z, Z.t0.w, Z.t1.x, Z.t1.y, Z.t1.z, Z.t1.w};
        if (u + 1 < ns) Z.fetch(u + 1);
        BOp b;
        split8(x, sc, b);
        step_b3_ring(acc, w, b, lane);
        R.end(s);
        ++s;
    }
}

template <bool SAVE>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_fwd_b3r_kernel(const float* __restrict__ zt, const float* __restrict__ steps,
                                                                     nvp_mlp_params p, const unsigned* __restrict__ packed,
                                                                     float* __restrict__ rgb, float* __restrict__ saved,
                                                                     int64_t n, int64_t ntiles, int d) {
    __shared__ __attribute__((aligned(16))) u32x4 ring[kRingSlots * kRingQuads + 4];          // 16 / 24 KiB (+ the flag ring's counters)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t tile = (int64_t)blockIdx.x * kWaves + wv;
    const bool active = tile < ntiles;                // wave-uniform; inactive waves still walk the ring (barriers), on the last tile
    if (!active) tile = ntiles - 1;
    nvp_stagger_start();
    const int j = lane & 31, h = lane >> 5;
    const NvpFwdLayoutB3 L = nvp_fwd_layout_b3(d);
    const int z4 = (nvp_rows4(d) / 4) * 32;
    const float* tab = reinterpret_cast<const float*>(packed + L.off[5]);      // sir_w0 / sir_b0 / last_w in D-register order
    const float* winv = tab + kB3ScaleOff + 8;                                 // 2^-e of each weight stream (mlp_layout.h)
    const int64_t px = tile * 32 + j;
    const float s = px < n ? steps[px] : 0.f;
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    float* sv = (SAVE && active) ? saved + tile * (int64_t)NVP_H * 32 : nullptr;

    Ring R;
    R.lds = ring; R.g = reinterpret_cast<const u32x4*>(packed); R.total = (int)(L.off[5] / kB3StepU32); R.wv = wv; R.lane = lane;
    ZFeed Z;
    Z.zg = reinterpret_cast<const float4*>(zt) + tile * (int64_t)z4; Z.rg_end = nvp_rows4(d) / 4; Z.j = j; Z.h = h;
    float mz = 0.f;                                   // per-pixel max |z|: the latent's share of the operand scale (same value as mlp_fwd_b3's)
    if (NVP_SPLIT_H2) {
        for (int idx = lane; idx < z4; idx += 64) mz = absmax_f4(mz, Z.zg[idx]);
        mz = fmaxf(mz, __shfl_xor(mz, 32));
    }
    Z.fetch(0);
    R.prologue();
    int ks = 0;                                       // running k-step: the packed stream is in consumption order

    f32x16 hm[4], x[4], acc[4];

    // ---- modulator layer 0: h0 = lrelu(W0 z + b0)                 modulation.py:112-121
    {
#pragma unroll
        for (int T = 0; T < 4; ++T) hm[T] = nvp_zero16();
        const PxScale ps = px_scale(fmaxf(mz, 1.0f));
        { const u32x4* w = R.begin(ks); bias_b3_ring(hm, w, ps.s, lane); R.end(ks); ++ks; }
        chain_z_b3_ring(hm, Z, L.zs, ps.s, R, ks, lane);
        lrelu4_scaled(hm, ps.u * winv[0]);
#pragma unroll
        for (int T = 0; T < 4; ++T) nvp_pin(hm[T]);
        if (SAVE && active) store_ptm(sv + 0 * act, hm, lane);
    }
    // ---- SIREN layer 0: x0 = sin(30 (w s + c)) * h0                modulation.py:53-56,90
    {
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float w0v[16], c0v[16];
            load_tab16(w0v, tab, 0, T, h);
            load_tab16(c0v, tab, 1, T, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float q = 30.0f * __fmaf_rn(s, w0v[r], c0v[r]);
                x[T][r] = nvp_sin(q) * hm[T][r];
            }
            nvp_pin(x[T]);
            NVP_LOAD_FENCE();
        }
    }
    // ---- layers 1 and 2
#pragma unroll
    for (int k = 1; k <= 2; ++k) {
        {   // modulator: h_k = lrelu(Wh h_{k-1} + Wz z + b)
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            const PxScale ps = px_scale(fmaxf(fmaxf(px_absmax(hm), mz), 1.0f));
            { const u32x4* w = R.begin(ks); bias_b3_ring(acc, w, ps.s, lane); R.end(ks); ++ks; }
            chain_h_b3_ring(acc, hm, ps.s, R, ks, lane, [&] { Z.fetch(0); });
            chain_z_b3_ring(acc, Z, L.zs, ps.s, R, ks, lane);
            lrelu4_scaled(acc, ps.u * winv[k]);
#pragma unroll
            for (int T = 0; T < 4; ++T) { hm[T] = acc[T]; nvp_pin(hm[T]); }
            if (SAVE && active) store_ptm(sv + (int64_t)k * act, hm, lane);
        }
        {   // SIREN: q_k = V x_{k-1} + c ; x_k = sin(q_k) * h_k
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = nvp_zero16();
            const PxScale ps = px_scale(fmaxf(px_absmax(x), 1.0f));
            { const u32x4* w = R.begin(ks); bias_b3_ring(acc, w, ps.s, lane); R.end(ks); ++ks; }
            chain_h_b3_ring(acc, x, ps.s, R, ks, lane);
            scale4(acc, ps.u * winv[2 + k]);
            if (SAVE && active) store_ptm(sv + (int64_t)(2 + k) * act, acc, lane);
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[T][r] = nvp_sin(acc[T][r]) * hm[T][r];
#pragma unroll
            for (int T = 0; T < 4; ++T) nvp_pin(x[T]);
        }
    }
    // ---- last layer (3 x 128, Identity): VALU dot products + cross-half add
    {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float v0[16], v1[16], v2[16];
            load_tab16(v0, tab, 2, T, h);
            load_tab16(v1, tab, 3, T, h);
            load_tab16(v2, tab, 4, T, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = x[T][r];
                o0 = __fmaf_rn(v0[r], v, o0);
                o1 = __fmaf_rn(v1[r], v, o1);
                o2 = __fmaf_rn(v2[r], v, o2);
            }
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
            NVP_LOAD_FENCE();
        }
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        if (active && h == 0 && px < n) {
            rgb[px * 3 + 0] = o0 + p.last_b[0];
            rgb[px * 3 + 1] = o1 + p.last_b[1];
            rgb[px * 3 + 2] = o2 + p.last_b[2];
        }
    }
}

}  // namespace

// called by nvp_mlp_fwd (mlp_fwd.hip) when NVP_FWD_B3 is on, the latent has <= 256 rows and NVP_MLP_RING_FWD=1
int nvp_mlp_fwd_b3r_launch(const float* zt, const float* steps, const nvp_mlp_params* p, const float* packed_fwd,
                           float* rgb, float* saved, int64_t n, int32_t d, void* stream) {
    const int64_t ntiles = nvp_ntiles(n);
    dim3 grid((unsigned)((ntiles + kWaves - 1) / kWaves));
    const unsigned* pk = reinterpret_cast<const unsigned*>(packed_fwd);
    if (saved)
        hipLaunchKernelGGL(mlp_fwd_b3r_kernel<true>, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, zt, steps, *p, pk, rgb, saved, n, ntiles, d);
    else
        hipLaunchKernelGGL(mlp_fwd_b3r_kernel<false>, grid, dim3(kWaves * 64), 0, (hipStream_t)stream, zt, steps, *p, pk, rgb, saved, n, ntiles, d);
    NVP_LAUNCH_CHECK();
    return 0;
}
