// Probe for the fp16 x 2 split: (1) does v_mfma_f32_32x32x16_f16 honour SUBNORMAL f16 inputs, (2) its issue rate against the
// bf16 form, (3) what v_fma_mixlo_f16-style residuals give.   hipcc --offload-arch=gfx950 -O3 f16_mfma_probe.hip -o f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void denorm_kernel(float a, float b, float* out) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)0.f; B[i] = (_Float16)0.f; }
    const int lane = threadIdx.x;
    if (lane < 32) { A[0] = (_Float16)a; B[0] = (_Float16)b; }       // k = 0 of every row / column
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (lane == 0) { out[0] = acc[0]; out[1] = (float)A[0]; out[2] = (float)B[0]; }
}

template <bool F16>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 A, B; b8 Ab, Bb;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(lane * 0.001f + i); B[i] = (_Float16)(1.f / (lane + 1 + i)); Ab[i] = (__bf16)(float)A[i]; Bb[i] = (__bf16)(float)B[i]; }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int T = 0; T < 4; ++T)
                acc[T] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[T], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, Bb, acc[T], 0, 0, 0);
    }
    float s = 0.f;
    for (int T = 0; T < 4; ++T) for (int i = 0; i < 16; ++i) s += acc[T][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 1 << 24);
    float h[3];
    const float cases[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /*2^-20*/, 1024.f}, {5.9604644775390625e-08f /*2^-24*/, 16384.f}, {1024.f, 9.5367431640625e-07f}, {9.5367431640625e-07f, 9.5367431640625e-07f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("a=%g (as f16 %g) b=%g (as f16 %g): MFMA d=%g expected %g\n", c[0], h[1], c[1], h[2], h[0], (double)h[1] * h[2]);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 2;           // 2 workgroups x 4 waves per CU: two waves per SIMD
    for (int f16 = 0; f16 < 2; ++f16) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) hipLaunchKernelGGL(rate_kernel<true>, dim3(blocks), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(rate_kernel<false>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * iters * 24 * 32768.0;
            if (rep) printf("%s 32x32x16: %.3f ms, %.1f TFLOP/s\n", f16 ? "f16 " : "bf16", ms, fl / ms * 1e-9);
        }
    }
    return 0;
}
