// Microbenchmark: does a stream written by one kernel come back faster when it is re-read SOON (memory-side 256 MB cache) than when
// it is re-read after more than the cache's capacity has gone by?  Sizes the fwd -> bwd hand-over of the saved streams:
//   (1) write X MB with kernel A, read the same X MB with kernel B right after: read GB/s vs X;
//   (2) the same with a 1 GB "polluter" stream between A and B (what the three-kernel step does today);
//   (3) per-workgroup scratch: every workgroup writes S KB of its own scratch then re-reads it, R rounds (the tile-fused shape), GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_write(float4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(256) void k_read(const float4* p, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 t = p[i]; s += t.x + t.y + t.z + t.w; }
    if (s == 12345.678f) out[0] = s;
}
// every workgroup: `rounds` times { write its scratch (kb KB), fence, read it back }
__global__ __launch_bounds__(256) void k_scratch(float4* scratch, int kb, int rounds, float* out) {
    float4* mine = scratch + (size_t)blockIdx.x * kb * 64;
    const int n4 = kb * 64;
    float s = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < n4; i += 256) mine[i] = make_float4(r, i, s, 1.f);
        __threadfence();
        __syncthreads();
        for (int i = threadIdx.x; i < n4; i += 256) { const float4 t = mine[(i + 64) % n4]; s += t.x + t.w; }
        __syncthreads();
    }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const size_t maxb = (size_t)4 << 30;
    float4 *buf, *pol; float* out;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&pol, (size_t)1 << 30)); CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    printf("# write X then read X (GB/s of the READ); with / without a 1 GB polluter stream in between\n");
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024, 2048, 3200}) {
        const size_t n4 = mb * 1024 * 1024 / 16;
        for (int pollute = 0; pollute < 2; ++pollute) {
            float best = 1e9f, bestw = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, buf, n4, (float)rep);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float msw; CK(hipEventElapsedTime(&msw, a, b));
                if (pollute) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, pol, ((size_t)1 << 30) / 16, 1.f);
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const float4*)buf, n4, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
                if (rep && msw < bestw) bestw = msw;
            }
            printf("X = %5zu MB  polluter %d  write %7.1f GB/s  read-back %7.1f GB/s\n", mb, pollute, mb / 1024.0 / (bestw * 1e-3), mb / 1024.0 / (best * 1e-3));
        }
    }
    printf("# per-workgroup scratch written then re-read, 2048 workgroups x 16 rounds (GB/s = bytes written + bytes read)\n");
    for (int kb : {16, 32, 48, 80, 128, 256}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_scratch, dim3(2048), dim3(256), 0, 0, buf, kb, 16, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        const double bytes = 2048.0 * kb * 1024 * 16 * 2;
        printf("scratch %4d KB per workgroup (%6.1f MB in flight)  %8.1f GB/s  (%.3f ms)\n", kb, 2048.0 * kb / 1024, bytes / 1e9 / (best * 1e-3), best);
    }
    return 0;
}
