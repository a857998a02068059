// Microbenchmark 4: what does the matrix pipe deliver for the MFMA patterns of the bf16 x 3 chains (2 waves / SIMD)?
//   A  24 MFMAs per step: 4 accumulators x 6 back-to-back dependent products (the step_b3 pattern), operands fixed registers
//   B  A + the on-the-fly operand split of 8 fresh values (split8, ~45 VALU) in front of every step
//   C  B with the split woven between the MFMAs (sched_group_barrier 1 MFMA : 2 VALU)
//   D  24 MFMAs per step round-robin over 8 accumulators (no back-to-back dependency at all)
//   E  A with ONE wave per SIMD
//   F  A with the 12 operand quads re-read from LDS every step (ds_read_b128, lane-contiguous)
//   G  A with the 12 operand quads re-read from an L1/L2-resident global buffer every step
// Output: cycles per MFMA per SIMD (32 = the pipe's rate) at the device clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ __forceinline__ unsigned pk(float a, float b) {
    typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pk(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = pk(sa, sb);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const u32x4* __restrict__ wg, float* __restrict__ out, int nsteps) {
    __shared__ __attribute__((aligned(16))) u32x4 wl[12 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 12 * 64; i += 256) wl[i] = wg[i];
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = (f32x16)(0.f);
    u32x4 a[4][3];
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[T][q] = wg[(T * 3 + q) * 64 + lane];
    u32x4 bh = wg[lane], bm = wg[64 + lane], bl = wg[128 + lane];
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = 0.001f * (float)(lane + q);
    for (int s = 0; s < nsteps; ++s) {
        if (MODE == 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { unsigned h, m, l; split2(x[2 * p], x[2 * p + 1], h, m, l); bh[p] = h; bm[p] = m; bl[p] = l; }
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = x[q] * 1.0001f + 0.5f;
            asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[T][q] = (MODE == 5 ? (const u32x4*)wl : wg)[(T * 3 + q) * 64 + lane];
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 24; ++i) acc[i & 7] = mf(a[i & 3][i % 3], bh, acc[i & 7]);
        } else {
            u32x4 nh = bh, nm = bm, nl = bl;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                acc[T] = mf(a[T][2], bh, acc[T]);
                acc[T] = mf(a[T][0], bl, acc[T]);
                acc[T] = mf(a[T][1], bm, acc[T]);
                acc[T] = mf(a[T][1], bh, acc[T]);
                acc[T] = mf(a[T][0], bm, acc[T]);
                acc[T] = mf(a[T][0], bh, acc[T]);
                if (MODE == 2) {
                    unsigned h, m, l;
                    split2(x[2 * T], x[2 * T + 1], h, m, l);
                    nh[T] = h; nm[T] = m; nl[T] = l;
                    x[2 * T] = x[2 * T] * 1.0001f + 0.5f; x[2 * T + 1] = x[2 * T + 1] * 1.0001f + 0.5f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
                    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 2) { bh = nh; bm = nm; bl = nl; }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[T][r];
    if (sum == 123.456f) out[0] = sum;
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u32x4* wg; float* out;
    CK(hipMalloc(&wg, 12 * 64 * 16)); CK(hipMemset(wg, 0x3c, 12 * 64 * 16)); CK(hipMalloc(&out, 64));
    const int nsteps = 2000;
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const char* names[] = {"A 4 acc x 6 dependent, fixed operands", "B A + split8 in front of every step", "C split woven between the MFMAs", "D round-robin over 8 accumulators",
                           "E A, one wave per SIMD", "F A, operands re-read from LDS", "G A, operands re-read from global (L1)"};
    for (int mode = 0; mode < 7; ++mode) {
        const int blocks = mode == 4 ? 256 : 512;          // 256-thread blocks: 2 per CU = 2 waves / SIMD
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: case 4: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, wg, out, nsteps); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, wg, out, nsteps); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, wg, out, nsteps); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, wg, out, nsteps); break;
                case 5: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, wg, out, nsteps); break;
                case 6: hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(256), 0, 0, wg, out, nsteps); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double waves_per_simd = blocks * 4 / 1024.0;
        const double mfma_per_simd = waves_per_simd * nsteps * 24.0;
        printf("%-42s %.3f ms  %.1f cycles/MFMA/SIMD at %.2f GHz  (%.0f %% of the pipe)\n", names[mode], best,
               best * 1e-3 * clk_khz * 1e3 / mfma_per_simd, clk_khz / 1e6, 100.0 * 32.0 / (best * 1e-3 * clk_khz * 1e3 / mfma_per_simd));
    }
    return 0;
}
