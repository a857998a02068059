// Re-lay the 14 MLP tensors out as MFMA A-operand streams (see mlp_layout.h).
// ~0.5 MB per call, one thread per packed float; runs once per optimizer step.
#include "mlp_layout.h"
#include <cstdlib>

namespace {

__global__ __launch_bounds__(256) void pack_fwd_kernel(nvp_mlp_params p, float* __restrict__ out, int d) {
    const NvpFwdLayout L = nvp_fwd_layout(d);
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L.off[5]) return;
    int seg = 0;
    while (idx >= L.off[seg + 1]) ++seg;
    int64_t loc = idx - L.off[seg];
    int tp = (int)(loc & 3);
    int lane = (int)((loc >> 2) & 63);
    int step = (int)(loc >> 8);
    int i = lane & 31, h = lane >> 5;
    int out_row = 32 * tp + i;

    const float* W; const float* b; int ld; bool has_h, has_z;
    switch (seg) {
        case 0: W = p.mod_w[0]; b = p.mod_b[0]; ld = d; has_h = false; has_z = true; break;
        case 1: W = p.mod_w[1]; b = p.mod_b[1]; ld = NVP_H + d; has_h = true; has_z = true; break;
        case 2: W = p.mod_w[2]; b = p.mod_b[2]; ld = NVP_H + d; has_h = true; has_z = true; break;
        case 3: W = p.sir_w[1]; b = p.sir_b[1]; ld = NVP_H; has_h = true; has_z = false; break;
        default: W = p.sir_w[2]; b = p.sir_b[2]; ld = NVP_H; has_h = true; has_z = false; break;
    }
    float v = 0.f;
    if (step == 0) {
        v = (h == 0) ? b[out_row] : 0.f;
    } else {
        int s = step - 1;
        if (has_h && s < 64) {
            v = W[(int64_t)out_row * ld + nvp_chain_k(s, h)];
        } else {
            int u = has_h ? s - 64 : s;
            int in = 2 * u + h;
            if (has_z && in < d) v = W[(int64_t)out_row * ld + (has_h ? NVP_H : 0) + in];
        }
    }
    out[idx] = v;
}

__device__ __forceinline__ unsigned nvp_bf16_rne(float f) {           // bf16 bits of f, round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// the D-register-ordered tables behind a b3 packed buffer (mlp_layout.h, kB3TabFloats)
__global__ __launch_bounds__(256) void pack_b3_tables_kernel(nvp_mlp_params p, float* __restrict__ tab) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kB3ScaleOff) return;
    const int r = idx & 15, T = (idx >> 4) & 3, h = (idx >> 6) & 1, t = idx >> 7;
    const int row = 32 * T + 8 * (r >> 2) + 4 * h + (r & 3);
    tab[idx] = t == 0 ? p.sir_w[0][row] : (t == 1 ? p.sir_b[0][row] : p.last_w[(t - 2) * NVP_H + row]);
}

// ---- operand split of one packed weight (the same rounding the kernels apply to activations, mlp_b3.h) -----------------------
// fp16 x 2: v * scale = hi + lo in fp16 (round to nearest even; subnormal parts are kept - the MFMA honours them);
// bf16 x 3: v = hi + mid + lo in bf16, scale unused (1.0).  Returns the 16 bits of `part`.
__device__ __forceinline__ unsigned nvp_split_part(float v, float scale, int part) {
#if NVP_SPLIT_H2
    const float t = v * scale;                                   // exact: power-of-two scale
    const _Float16 hi = (_Float16)t;
    if (part == 0) return (unsigned)__builtin_bit_cast(unsigned short, hi);
    const _Float16 lo = (_Float16)(t - (float)hi);               // the residual is exact in fp32
    return (unsigned)__builtin_bit_cast(unsigned short, lo);
#else
    const unsigned hi = nvp_bf16_rne(v);
    const float r1 = v - __uint_as_float(hi << 16);
    const unsigned mid = nvp_bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(mid << 16);
    const unsigned lo = nvp_bf16_rne(r2);
    return (part == 0 ? hi : (part == 1 ? mid : lo)) & 0xffffu;
#endif
}

// Power-of-two scale per packed weight stream (fp16 x 2 split): 2^e with max|w| * 2^e in [2^13, 2^14), so that hi and lo stay
// inside fp16's range (max 65504) and lo is a normal fp16 for every weight within 2^-16 of the largest.  One block per
// stream; `bwd` selects the backward streams (mlp_layout.h).  A stream's maximum is taken over the whole tensor(s) it is cut
// from (and the bias for the forward streams).  Non-finite tensors get scale 1.
__global__ __launch_bounds__(1024) void pack_b3_scales_kernel(nvp_mlp_params p, float* __restrict__ tab, int d, int bwd) {
    const int seg = blockIdx.x;
    const float* W; const float* b = nullptr; int64_t nw;
    if (!bwd) {
        switch (seg) {
            case 0: W = p.mod_w[0]; b = p.mod_b[0]; nw = (int64_t)NVP_H * d; break;
            case 1: W = p.mod_w[1]; b = p.mod_b[1]; nw = (int64_t)NVP_H * (NVP_H + d); break;
            case 2: W = p.mod_w[2]; b = p.mod_b[2]; nw = (int64_t)NVP_H * (NVP_H + d); break;
            case 3: W = p.sir_w[1]; b = p.sir_b[1]; nw = (int64_t)NVP_H * NVP_H; break;
            default: W = p.sir_w[2]; b = p.sir_b[2]; nw = (int64_t)NVP_H * NVP_H; break;
        }
    } else {
        switch (seg) {
            case 0: W = p.sir_w[2]; nw = (int64_t)NVP_H * NVP_H; break;
            case 1: W = p.sir_w[1]; nw = (int64_t)NVP_H * NVP_H; break;
            case 2: case 6: W = p.mod_w[2]; nw = (int64_t)NVP_H * (NVP_H + d); break;
            case 3: case 5: W = p.mod_w[1]; nw = (int64_t)NVP_H * (NVP_H + d); break;
            default: W = p.mod_w[0]; nw = (int64_t)NVP_H * d; break;
        }
    }
    // 1024 threads, 8 independent loads in flight per thread: the largest tensor (128 x 356) is four rounds
    unsigned m = 0u;
    for (int64_t i0 = threadIdx.x; i0 < nw; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * 1024 < nw ? W[i0 + u * 1024] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) m = max(m, __float_as_uint(v[u]) & 0x7fffffffu);
    }
    if (b && threadIdx.x < NVP_H) m = max(m, __float_as_uint(b[threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w) red[0] = max(red[0], red[w]);
    }
    if (threadIdx.x == 0) {
        const unsigned e = red[0] >> 23;                         // biased exponent of the largest magnitude
        float sc = 1.0f, inv = 1.0f;
#if NVP_SPLIT_H2
        if (e < 255u) {                                          // tensors whose largest weight is below 2^-40 share the scale 2^53:
            const unsigned ec = max(e, 87u);                     // the kernels multiply it with a per-pixel scale <= 2^63 (mlp_b3.h)
            sc = __uint_as_float((267u - ec) << 23); inv = __uint_as_float((ec - 13u) << 23);
        }
#endif
        tab[kB3ScaleOff + seg] = sc;
        tab[kB3ScaleOff + 8 + seg] = inv;
    }
}

// split forward stream (mlp_layout.h "b3"): one thread per packed u32 = two consecutive k of one part
__global__ __launch_bounds__(256) void pack_fwd_b3_kernel(nvp_mlp_params p, unsigned* __restrict__ out, int d) {
    const NvpFwdLayoutB3 L = nvp_fwd_layout_b3(d);
    const float* scales = reinterpret_cast<const float*>(out + L.off[5]) + kB3ScaleOff;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L.off[5]) return;
    int seg = 0;                                   // layer whose stream holds idx (streams are stored in consumption order, not by id)
    for (int k = 0; k < 5; ++k)
        if (idx >= L.off[k] && idx < L.off[k] + (int64_t)L.steps[k] * kB3StepU32) seg = k;
    const int64_t loc = idx - L.off[seg];
    const int pr = (int)(loc & 3);                 // which pair of the lane's eight k
    const int lane = (int)((loc >> 2) & 63);
    const int part = (int)((loc >> 8) % kB3Parts);
    const int tp = (int)(((loc >> 8) / kB3Parts) & 3);
    const int step = (int)((loc >> 8) / (4 * kB3Parts));
    const int i = lane & 31, h = lane >> 5;
    const int out_row = 32 * tp + i;
    const float* W; const float* b; int ld; bool has_h, has_z;
    switch (seg) {
        case 0: W = p.mod_w[0]; b = p.mod_b[0]; ld = d; has_h = false; has_z = true; break;
        case 1: W = p.mod_w[1]; b = p.mod_b[1]; ld = NVP_H + d; has_h = true; has_z = true; break;
        case 2: W = p.mod_w[2]; b = p.mod_b[2]; ld = NVP_H + d; has_h = true; has_z = true; break;
        case 3: W = p.sir_w[1]; b = p.sir_b[1]; ld = NVP_H; has_h = true; has_z = false; break;
        default: W = p.sir_w[2]; b = p.sir_b[2]; ld = NVP_H; has_h = true; has_z = false; break;
    }
    unsigned packed = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = 2 * pr + e;
        float v = 0.f;
        if (step == 0) {
            if (h == 0 && q == 0) v = b[out_row];
        } else {
            const int s = step - 1;
            if (has_h && s < 8) {
                v = W[(int64_t)out_row * ld + nvp_b3_chain_in(s, h, q)];
            } else if (has_z) {
                const int in = 16 * (has_h ? s - 8 : s) + 8 * h + q;
                if (in < d) v = W[(int64_t)out_row * ld + (has_h ? NVP_H : 0) + in];
            }
        }
        packed |= nvp_split_part(v, scales[seg], part) << (16 * e);
    }
    out[idx] = packed;
}

// The forward pack in ONE launch (scales + streams + tables): it sits on the critical path of every training step - the
// forward kernel needs it and it needs the optimizer's last update - so three dependent launches (15 + 28 + 5 us) are one
// of ~15 us.  One 1024-thread block per k-step (kB3StepU32 packed words, two per thread): the block first takes the maximum
// over the tensor(s) its stream is cut from - redundantly with the other blocks of the stream, 45 L2-resident loads per
// thread - and derives the stream's scale exactly as pack_b3_scales_kernel does, so the packed words are the same bits.
__device__ __forceinline__ void b3_scale_from_max(unsigned mbits, float& sc, float& inv) {
    const unsigned e = mbits >> 23;
    sc = 1.0f; inv = 1.0f;
#if NVP_SPLIT_H2
    if (e < 255u) {
        const unsigned ec = max(e, 87u);
        sc = __uint_as_float((267u - ec) << 23); inv = __uint_as_float((ec - 13u) << 23);
    }
#endif
}

__global__ __launch_bounds__(1024) void pack_fwd_b3_all_kernel(nvp_mlp_params p, unsigned* __restrict__ out, int d) {
    const NvpFwdLayoutB3 L = nvp_fwd_layout_b3(d);
    float* tab = reinterpret_cast<float*>(out + L.off[5]);
    const int64_t base = (int64_t)blockIdx.x * kB3StepU32;
    if (base >= L.off[5]) {                         // the block behind the streams: the D-register-ordered tables
        for (int idx = threadIdx.x; idx < kB3ScaleOff; idx += 1024) {
            const int r = idx & 15, T = (idx >> 4) & 3, h = (idx >> 6) & 1, t = idx >> 7;
            const int row = 32 * T + 8 * (r >> 2) + 4 * h + (r & 3);
            tab[idx] = t == 0 ? p.sir_w[0][row] : (t == 1 ? p.sir_b[0][row] : p.last_w[(t - 2) * NVP_H + row]);
        }
        return;
    }
    int seg = 0;
    for (int k = 0; k < 5; ++k)
        if (base >= L.off[k] && base < L.off[k] + (int64_t)L.steps[k] * kB3StepU32) seg = k;
    const float* W; const float* b; int ld; bool has_h, has_z;
    switch (seg) {
        case 0: W = p.mod_w[0]; b = p.mod_b[0]; ld = d; has_h = false; has_z = true; break;
        case 1: W = p.mod_w[1]; b = p.mod_b[1]; ld = NVP_H + d; has_h = true; has_z = true; break;
        case 2: W = p.mod_w[2]; b = p.mod_b[2]; ld = NVP_H + d; has_h = true; has_z = true; break;
        case 3: W = p.sir_w[1]; b = p.sir_b[1]; ld = NVP_H; has_h = true; has_z = false; break;
        default: W = p.sir_w[2]; b = p.sir_b[2]; ld = NVP_H; has_h = true; has_z = false; break;
    }
    const int64_t nw = (int64_t)NVP_H * ld;
    unsigned m = 0u;
    for (int64_t i0 = threadIdx.x; i0 < nw; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * 1024 < nw ? W[i0 + u * 1024] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) m = max(m, __float_as_uint(v[u]) & 0x7fffffffu);
    }
    if (threadIdx.x < NVP_H) m = max(m, __float_as_uint(b[threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) m = max(m, red[w]);
    float sc, inv;
    b3_scale_from_max(m, sc, inv);
    const int step = (int)((base - L.off[seg]) / kB3StepU32);
    if (step == 0 && threadIdx.x == 0) { tab[kB3ScaleOff + seg] = sc; tab[kB3ScaleOff + 8 + seg] = inv; }
#pragma unroll
    for (int u = 0; u < kB3StepU32 / 1024; ++u) {
        const int loc = u * 1024 + threadIdx.x;                  // word within the k-step: ((tile * kP + part) * 64 + lane) * 4 + pair
        const int pr = loc & 3;
        const int lane = (loc >> 2) & 63;
        const int part = (loc >> 8) % kB3Parts;
        const int tp = ((loc >> 8) / kB3Parts) & 3;
        const int i = lane & 31, h = lane >> 5;
        const int out_row = 32 * tp + i;
        unsigned packed = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int q = 2 * pr + e;
            float v = 0.f;
            if (step == 0) {
                if (h == 0 && q == 0) v = b[out_row];
            } else {
                const int s = step - 1;
                if (has_h && s < 8) {
                    v = W[(int64_t)out_row * ld + nvp_b3_chain_in(s, h, q)];
                } else if (has_z) {
                    const int in = 16 * (has_h ? s - 8 : s) + 8 * h + q;
                    if (in < d) v = W[(int64_t)out_row * ld + (has_h ? NVP_H : 0) + in];
                }
            }
            packed |= nvp_split_part(v, sc, part) << (16 * e);
        }
        out[base + loc] = packed;
    }
}

// split backward streams (mlp_layout.h): one thread per packed u32
__global__ __launch_bounds__(256) void pack_bwd_b3_kernel(nvp_mlp_params p, unsigned* __restrict__ out, int d) {
    const int zt = nvp_bwd_b3_zt(d);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nvp_bwd_b3_off(7, zt)) return;
    const float* scales = reinterpret_cast<const float*>(out + nvp_bwd_b3_off(7, zt)) + kB3ScaleOff;
    int seg = 0;
    while (idx >= nvp_bwd_b3_off(seg + 1, zt)) ++seg;
    const int64_t loc = idx - nvp_bwd_b3_off(seg, zt);
    const int tiles = seg < 4 ? 4 : zt;            // output tiles per k-step of this stream
    const int pr = (int)(loc & 3);
    const int lane = (int)((loc >> 2) & 63);
    const int quad = (int)(loc >> 8);              // (step * tiles + tile) * kB3Parts + part
    const int part = quad % kB3Parts;
    const int tp = (quad / kB3Parts) % tiles;
    const int step = quad / (kB3Parts * tiles);
    const int i = lane & 31, h = lane >> 5;
    const int in = 32 * tp + i;                    // A row = input index
    unsigned packed = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int o = nvp_b3_chain_in(step, h, 2 * pr + e);        // k = output index
        float v = 0.f;
        switch (seg) {
            case 0: v = p.sir_w[2][(int64_t)o * NVP_H + in]; break;
            case 1: v = p.sir_w[1][(int64_t)o * NVP_H + in]; break;
            case 2: v = p.mod_w[2][(int64_t)o * (NVP_H + d) + in]; break;
            case 3: v = p.mod_w[1][(int64_t)o * (NVP_H + d) + in]; break;
            case 4: if (in < d) v = p.mod_w[0][(int64_t)o * d + in]; break;
            case 5: if (in < d) v = p.mod_w[1][(int64_t)o * (NVP_H + d) + NVP_H + in]; break;
            default: if (in < d) v = p.mod_w[2][(int64_t)o * (NVP_H + d) + NVP_H + in]; break;
        }
        packed |= nvp_split_part(v, scales[seg], part) << (16 * e);
    }
    out[idx] = packed;
    if (zt == 4)        // the consumption-ordered copy the workgroup-shared ring kernel reads (mlp_layout.h)
        out[nvp_bwd_b3_ring_off(zt) + (int64_t)nvp_bwd_b3_ring_pos(seg, step) * kB3StepU32 + (loc - (int64_t)step * kB3StepU32)] = packed;
}

__global__ __launch_bounds__(256) void pack_bwd_kernel(nvp_mlp_params p, float* __restrict__ out, int d) {
    const NvpBwdLayout L = nvp_bwd_layout(d);
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L.off[7]) return;
    int seg = 0;
    while (idx >= L.off[seg + 1]) ++seg;
    int64_t loc = idx - L.off[seg];
    int per = (seg < 4) ? 4 : L.zt;
    int tp = (int)(loc % per);
    int64_t rest = loc / per;
    int lane = (int)(rest & 63);
    int step = (int)(rest >> 6);
    int i = lane & 31, h = lane >> 5;
    int in = 32 * tp + i;                 // A row = input index
    int o = nvp_chain_k(step, h);         // k = output index
    float v = 0.f;
    switch (seg) {
        case 0: v = p.sir_w[2][(int64_t)o * NVP_H + in]; break;
        case 1: v = p.sir_w[1][(int64_t)o * NVP_H + in]; break;
        case 2: v = p.mod_w[2][(int64_t)o * (NVP_H + d) + in]; break;
        case 3: v = p.mod_w[1][(int64_t)o * (NVP_H + d) + in]; break;
        case 4: if (in < d) v = p.mod_w[0][(int64_t)o * d + in]; break;
        case 5: if (in < d) v = p.mod_w[1][(int64_t)o * (NVP_H + d) + NVP_H + in]; break;
        default: if (in < d) v = p.mod_w[2][(int64_t)o * (NVP_H + d) + NVP_H + in]; break;
    }
    out[idx] = v;
}

}  // namespace

extern "C" {

int64_t nvp_packed_fwd_floats(int32_t d) { return (NVP_FWD_B3 && nvp_fwd_b3_ok(d)) ? nvp_fwd_layout_b3(d).off[5] + kB3TabFloats : nvp_fwd_layout(d).off[5]; }
int64_t nvp_packed_bwd_floats(int32_t d) {
    if (!(NVP_BWD_B3 && nvp_bwd_b3_ok(d))) return nvp_bwd_layout(d).off[7];
    const int zt = nvp_bwd_b3_zt(d);
    return nvp_bwd_b3_off(7, zt) + kB3TabFloats + (zt == 4 ? (int64_t)kBwdRingSteps * kB3StepU32 : 0);      // + the ring-ordered copy
}
int64_t nvp_mlp_param_floats(int32_t d) { return nvp_param_layout(d).total; }
int64_t nvp_dw_partial_floats(int32_t d, int32_t n_chunks) {
    return nvp_param_layout(d).total * (int64_t)n_chunks;       // one full gradient record per pixel chunk
}
int32_t nvp_latent_rows(int32_t d) { return nvp_rows4(d); }
int32_t nvp_dz_stride(int32_t d) { return nvp_dz_stride_dev(d); }
const char* nvp_version(void) { return "nvp_hip 0.2 (gfx950)"; }
int32_t nvp_mlp_mfma_products(void) { return !(NVP_FWD_B3 && NVP_BWD_B3) ? 1 : (NVP_SPLIT_H2 ? 3 : 6); }

int nvp_mlp_pack_fwd(const nvp_mlp_params* p, float* packed, int32_t d, void* stream) {
    if (!p || !packed || d < 1) return NVP_ERR_BADARG;
    if (NVP_FWD_B3 && nvp_fwd_b3_ok(d)) {
        const int64_t nb = nvp_fwd_layout_b3(d).off[5];
        // NVP_PACK_ONE_LAUNCH=0 (environment, read once): the three-launch version (same bits)
#if NVP_EXPERIMENTS
        static const bool one = [] { const char* e = getenv("NVP_PACK_ONE_LAUNCH"); return !(e && e[0] == '0'); }();
#else
        constexpr bool one = true;
#endif
        if (one) {
            hipLaunchKernelGGL(pack_fwd_b3_all_kernel, dim3((unsigned)(nb / kB3StepU32 + 1)), dim3(1024), 0, (hipStream_t)stream, *p, reinterpret_cast<unsigned*>(packed), d);
            NVP_LAUNCH_CHECK();
            return 0;
        }
        hipLaunchKernelGGL(pack_b3_scales_kernel, dim3(5), dim3(1024), 0, (hipStream_t)stream, *p, packed + nb, d, 0);
        hipLaunchKernelGGL(pack_fwd_b3_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, reinterpret_cast<unsigned*>(packed), d);
        hipLaunchKernelGGL(pack_b3_tables_kernel, dim3((kB3ScaleOff + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, packed + nb);
        NVP_LAUNCH_CHECK();
        return 0;
    }
    int64_t n = nvp_fwd_layout(d).off[5];
    hipLaunchKernelGGL(pack_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, packed, d);
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_mlp_pack_bwd(const nvp_mlp_params* p, float* packed, int32_t d, void* stream) {
    if (!p || !packed || d < 1) return NVP_ERR_BADARG;
    if (NVP_BWD_B3 && nvp_bwd_b3_ok(d)) {
        const int64_t nb = nvp_bwd_b3_off(7, nvp_bwd_b3_zt(d));
        hipLaunchKernelGGL(pack_b3_scales_kernel, dim3(7), dim3(1024), 0, (hipStream_t)stream, *p, packed + nb, d, 1);
        hipLaunchKernelGGL(pack_bwd_b3_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, reinterpret_cast<unsigned*>(packed), d);
        hipLaunchKernelGGL(pack_b3_tables_kernel, dim3((kB3ScaleOff + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, packed + nb);
        NVP_LAUNCH_CHECK();
        return 0;
    }
    int64_t n = nvp_bwd_layout(d).off[7];
    hipLaunchKernelGGL(pack_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, packed, d);
    NVP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
