// Microbenchmark 2: LDS atomic flavours (f32 / u32 / u64), conflict-free vs random, and the
// random 8-byte global gather rate at several footprints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE, bool RANDOM>
__global__ void k_lds(float* out, int per_thread, unsigned mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lf = (float*)smem; unsigned* lu = (unsigned*)smem; unsigned long long* l64 = (unsigned long long*)smem;
    for (int i = threadIdx.x; i <= (int)mask; i += blockDim.x) lu[i] = 0;
    __syncthreads();
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        unsigned a = RANDOM ? (hash32(tid * 977u + i * 0x9e3779b9u) & mask) : ((threadIdx.x + i * 67u) & mask);
        if (MODE == 0) atomicAdd(&lf[a], 1.0f);
        if (MODE == 1) atomicAdd(&lu[a], 1u);
        if (MODE == 2) atomicAdd(&l64[a >> 1], 1ull);
        if (MODE == 3) { float v = lf[a]; lf[a] = v + 1.0f; }          // plain RMW (racy), rate reference
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lf[0];
}

__global__ void k_gather(const float2* __restrict__ src, float* out, unsigned mask, int per_thread) {
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < per_thread; ++i) {
        unsigned a = hash32(tid * 977u + i * 0x9e3779b9u) & mask;
        float2 v = src[a]; acc += v.x + v.y;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int threads = 256, blocks = 256 * 16, per = 1024;
    double total = (double)threads * blocks * per;
    float* out; CK(hipMalloc(&out, blocks * 4));
    const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
    for (int mode = 0; mode < 4; ++mode)
        for (int rnd = 0; rnd < 2; ++rnd) {
            unsigned mask = (1u << 13) - 1;   // 32 KB table
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
#define L(M, R) hipLaunchKernelGGL((k_lds<M, R>), dim3(blocks), dim3(threads), (mask + 1) * 4, 0, out, per, mask)
                if (mode == 0) { if (rnd) L(0, true); else L(0, false); }
                if (mode == 1) { if (rnd) L(1, true); else L(1, false); }
                if (mode == 2) { if (rnd) L(2, true); else L(2, false); }
                if (mode == 3) { if (rnd) L(3, true); else L(3, false); }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            }
            printf("LDS %-10s %-13s %8.1f Gops/s\n", names[mode], rnd ? "random" : "conflict-free", total / ms / 1e6);
        }
    float2* src; size_t maxe = (size_t)1 << 27;   // 1 GiB of float2
    CK(hipMalloc(&src, maxe * 8)); CK(hipMemset(src, 0, maxe * 8));
    for (int lg = 14; lg <= 27; lg += 3) {
        unsigned mask = (1u << lg) - 1; float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, src, out, mask, 64);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("gather 8B random over %9.1f KB: %8.1f Gops/s\n", (mask + 1) * 8.0 / 1024, (double)threads * blocks * 64 / ms / 1e6);
    }
    return 0;
}
