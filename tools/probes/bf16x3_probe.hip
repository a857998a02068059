// Microbenchmark 3: is a bf16 x 3 split ("fp32-accurate bf16 MFMA") chain worth building?
// One wave = one 32-pixel tile, D[feature][pixel] = W[128x128] * act[128][pixel], the 64 D registers of a chain are
// the next chain's B operand (as in mlp_fwd.hip).  Variants:
//   0  fp32          v_mfma_f32_32x32x2_f32, 64 k-steps x 4 row tiles, weights from registers        (today's skeleton)
//   1  bf16x3 reg    v_mfma_f32_32x32x16_bf16, 8 k-steps x 4 row tiles x 6 products, activations split into
//                    hi/mid/lo bf16 on the fly, weight parts from registers                          (matrix-pipe ceiling)
//   2  bf16x3 glob   same, the 3 weight parts streamed per wave from an L2-resident packed tensor (96 KB per chain)
//   3  bf16x3 lds    same, weight parts read from LDS (one layer staged once per 8-wave workgroup)    (LDS read cost)
//   4 / 5            as 1, but each product is issued for 2 / 4 output tiles in turn, so back-to-back MFMAs never share an accumulator
// Reports fp32-equivalent TFLOP/s (2*128*128*32 FLOP per chain per wave) and, for one tile, the error of the
// 6-product split against float64 next to the error of an fp32 fma chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ unsigned pk(float a, float b) {      // two floats -> packed bf16 pair (RNE)
    typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// 8 floats -> hi/mid/lo bf16x8 with x = hi + mid + lo to ~2^-25 |x|
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned h = pk(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = pk(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        hi[p] = h; mid[p] = m; lo[p] = pk(sa, sb);
    }
}
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a), as_bf(b), c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(MODE == 3 ? 512 : 256, MODE == 3 ? 1 : 2) void chain_probe(const u32x4* __restrict__ wpk, float* __restrict__ out, int nchain) {
    extern __shared__ __attribute__((aligned(16))) u32x4 wl[];
    const int lane = threadIdx.x & 63;
    f32x16 act[4], acc[4];
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[T][r] = 0.01f * (float)((lane * 7 + r * 3 + T) % 17 - 8);
    if (MODE == 3) {            // stage one layer's packed weights: 8 k-steps x 4 tiles x 3 parts x 64 lanes x 16 B = 96 KB
        for (int i = threadIdx.x; i < 8 * 4 * 3 * 64; i += blockDim.x) wl[i] = wpk[i];
        __syncthreads();
    }
    for (int c = 0; c < nchain; ++c) {
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = (f32x16)(0.f);
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                const float w = 0.003f * (float)((s + lane) & 7);
#pragma unroll
                for (int T = 0; T < 4; ++T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w + 0.001f * T, act[s >> 4][s & 15], acc[T], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {                 // k-step s: registers 8u..8u+7 of tile T' = s >> 1, u = s & 1
                float x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = act[s >> 1][8 * (s & 1) + q];
                u32x4 bh, bm, bl;
                split8(x, bh, bm, bl);
                if (MODE == 4 || MODE == 5) {
                    const int W = MODE == 4 ? 2 : 4;           // tiles interleaved per product
#pragma unroll
                    for (int T0 = 0; T0 < 4; T0 += W) {
                        u32x4 ah[4], am[4], al[4];
#pragma unroll
                        for (int t = 0; t < W; ++t) {
                            const unsigned k = 0x3c003c00u + (unsigned)(s * 4 + T0 + t);
                            ah[t] = (u32x4)(k); am[t] = (u32x4)(k ^ 0x00400040u); al[t] = (u32x4)(k ^ 0x01000100u);
                        }
#pragma unroll
                        for (int t = 0; t < W; ++t) acc[T0 + t] = mf(al[t], bh, acc[T0 + t]);
#pragma unroll
                        for (int t = 0; t < W; ++t) acc[T0 + t] = mf(ah[t], bl, acc[T0 + t]);
#pragma unroll
                        for (int t = 0; t < W; ++t) acc[T0 + t] = mf(am[t], bm, acc[T0 + t]);
#pragma unroll
                        for (int t = 0; t < W; ++t) acc[T0 + t] = mf(am[t], bh, acc[T0 + t]);
#pragma unroll
                        for (int t = 0; t < W; ++t) acc[T0 + t] = mf(ah[t], bm, acc[T0 + t]);
#pragma unroll
                        for (int t = 0; t < W; ++t) acc[T0 + t] = mf(ah[t], bh, acc[T0 + t]);
                    }
                    continue;
                }
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    u32x4 ah, am, al;
                    if (MODE == 1) {
                        const unsigned k = 0x3c003c00u + (unsigned)(s * 4 + T);
                        ah = (u32x4)(k); am = (u32x4)(k ^ 0x00400040u); al = (u32x4)(k ^ 0x01000100u);
                    } else {
                        const u32x4* src = (MODE == 2 ? wpk + (size_t)(c & 3) * (8 * 4 * 3 * 64) : wl) + ((s * 4 + T) * 3) * 64 + lane;
                        ah = src[0]; am = src[64]; al = src[128];
                    }
                    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);      // keep the operand loads of later tiles where they are
                    acc[T] = mf(al, bh, acc[T]);          // smallest terms first
                    acc[T] = mf(ah, bl, acc[T]);
                    acc[T] = mf(am, bm, acc[T]);
                    acc[T] = mf(am, bh, acc[T]);
                    acc[T] = mf(ah, bm, acc[T]);
                    acc[T] = mf(ah, bh, acc[T]);
                }
            }
        }
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) act[T][r] = acc[T][r] * 0.25f + 0.001f;      // keep magnitudes bounded
    }
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += act[T][r];
    if (s == 123.456f) out[0] = s;
}

// accuracy: one 32x32 output tile, K = 128, random fp32 A and B; device computes it with the split (B split on the
// fly, A parts prepared on the host with the same rounding), the host with float64 and with an fp32 fma chain
__global__ void acc_probe(const u32x4* __restrict__ apk, const float* __restrict__ b, float* __restrict__ d) {
    const int lane = threadIdx.x;
    f32x16 acc = (f32x16)(0.f);
    for (int s = 0; s < 8; ++s) {
        float x[8];
        for (int q = 0; q < 8; ++q) x[q] = b[(s * 16 + 8 * (lane >> 5) + q) * 32 + (lane & 31)];     // B[k][j]
        u32x4 bh, bm, bl;
        split8(x, bh, bm, bl);
        const u32x4 ah = apk[(s * 3 + 0) * 64 + lane], am = apk[(s * 3 + 1) * 64 + lane], al = apk[(s * 3 + 2) * 64 + lane];
        acc = mf(al, bh, acc); acc = mf(ah, bl, acc); acc = mf(am, bm, acc); acc = mf(am, bh, acc); acc = mf(ah, bm, acc); acc = mf(ah, bh, acc);
    }
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

static unsigned short bf_rne(float f) { unsigned u; memcpy(&u, &f, 4); unsigned r = u + 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(r >> 16); }
static float bf_to_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u32x4* wpk; float* out;
    const size_t wn = (size_t)4 * 8 * 4 * 3 * 64;           // four layers' worth of packed weight parts (384 KB, L2-resident)
    CK(hipMalloc(&wpk, wn * 16)); CK(hipMalloc(&out, 1024));
    { std::vector<unsigned> h(wn * 4); for (size_t i = 0; i < h.size(); ++i) { unsigned short a = bf_rne(0.02f * (float)((int)(i % 13) - 6)); h[i] = a | ((unsigned)a << 16); } CK(hipMemcpy(wpk, h.data(), wn * 16, hipMemcpyHostToDevice)); }
    const int nchain = 24, waves_blocks = 256 * 2 * 8;      // 8 rounds of 2 workgroups per CU
    const double flop = 2.0 * 128 * 128 * 32 * nchain;
    auto run = [&](int mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(chain_probe<0>, dim3(waves_blocks), dim3(256), 0, 0, wpk, out, nchain);
            if (mode == 1) hipLaunchKernelGGL(chain_probe<1>, dim3(waves_blocks), dim3(256), 0, 0, wpk, out, nchain);
            if (mode == 2) hipLaunchKernelGGL(chain_probe<2>, dim3(waves_blocks), dim3(256), 0, 0, wpk, out, nchain);
            if (mode == 4) hipLaunchKernelGGL(chain_probe<4>, dim3(waves_blocks), dim3(256), 0, 0, wpk, out, nchain);
            if (mode == 5) hipLaunchKernelGGL(chain_probe<5>, dim3(waves_blocks), dim3(256), 0, 0, wpk, out, nchain);
            if (mode == 3) hipLaunchKernelGGL(chain_probe<3>, dim3(waves_blocks / 2), dim3(512), 8 * 4 * 3 * 64 * 16, 0, wpk, out, nchain);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double waves = (double)waves_blocks * 4;
        printf("mode %d: %.3f ms  %.1f TFLOP/s fp32-equivalent  (%.0f cycles/chain/SIMD at 2.4 GHz incl. 2 waves)\n", mode, best,
               flop * waves / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (nchain * (waves / 1024.0)));
    };
    for (int m = 0; m < 6; ++m) run(m);

    // accuracy of the 6-product split
    std::vector<float> A(32 * 128), B(128 * 32); srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.7f;
    std::vector<unsigned> apk(8 * 3 * 64 * 4);
    for (int s = 0; s < 8; ++s) for (int lane = 0; lane < 64; ++lane) for (int p = 0; p < 4; ++p) {
        unsigned short part[3][2];
        for (int e = 0; e < 2; ++e) {
            float x = A[(lane & 31) * 128 + s * 16 + 8 * (lane >> 5) + 2 * p + e];        // A[i][k]
            unsigned short h = bf_rne(x); float r = x - bf_to_f(h); unsigned short m = bf_rne(r); float r2 = r - bf_to_f(m);
            part[0][e] = h; part[1][e] = m; part[2][e] = bf_rne(r2);
        }
        for (int q = 0; q < 3; ++q) apk[((s * 3 + q) * 64 + lane) * 4 + p] = part[q][0] | ((unsigned)part[q][1] << 16);
    }
    u32x4* dapk; float *db, *dd; CK(hipMalloc(&dapk, apk.size() * 4)); CK(hipMalloc(&db, B.size() * 4)); CK(hipMalloc(&dd, 32 * 32 * 4));
    CK(hipMemcpy(dapk, apk.data(), apk.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(acc_probe, dim3(1), dim3(64), 0, 0, dapk, db, dd);
    std::vector<float> D(32 * 32); CK(hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost));
    double e_split = 0, e_f32 = 0, ref_max = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double r = 0; float f = 0.f;
        for (int k = 0; k < 128; ++k) { r += (double)A[i * 128 + k] * (double)B[k * 32 + j]; f = fmaf(A[i * 128 + k], B[k * 32 + j], f); }
        e_split = fmax(e_split, fabs(D[i * 32 + j] - r)); e_f32 = fmax(e_f32, fabs((double)f - r)); ref_max = fmax(ref_max, fabs(r));
    }
    printf("accuracy, K=128 dot products of O(1) values (max |ref| %.2f): bf16x3 6-product max abs err %.3e, fp32 fma chain %.3e\n", ref_max, e_split, e_f32);
    return 0;
}
