// Microbenchmark: fp32 atomic-add throughput on MI355X vs footprint, contention and scope.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void k_atomic(float* buf, unsigned mask, int per_thread, int vec) {
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        unsigned a = hash32(tid * 977u + i * 0x9e3779b9u) & mask;
        a &= ~(unsigned)(vec - 1);
        for (int v = 0; v < vec; ++v) {
            if (MODE == 0) unsafeAtomicAdd(buf + a + v, 1.0f);
            else if (MODE == 1) __hip_atomic_fetch_add(buf + a + v, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) buf[a + v] += 1.0f;                    // plain RMW (racy) for reference
        }
    }
}

__global__ void k_lds(float* out, int per_thread, unsigned mask) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i <= (int)mask; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        unsigned a = hash32(tid * 977u + i * 0x9e3779b9u) & mask;
        atomicAdd(&lds[a], 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}

int main() {
    float* buf; size_t maxf = (size_t)1 << 28;   // 1 GiB
    CK(hipMalloc(&buf, maxf * 4)); CK(hipMemset(buf, 0, maxf * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int threads = 256, blocks = 256 * 16, per = 64;
    double total = (double)threads * blocks * per;
    printf("mode(0=agent atomic,1=wg-scope atomic,2=plain rmw) footprint vec Gops/s\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int vec = 1; vec <= 2; vec *= 2)
            for (int lg = 9; lg <= 28; lg += (lg < 20 ? 4 : 3)) {
                unsigned mask = (1u << lg) - 1;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(a));
                    if (mode == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(threads), 0, 0, buf, mask, per, vec);
                    if (mode == 1) hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, buf, mask, per, vec);
                    if (mode == 2) hipLaunchKernelGGL(k_atomic<2>, dim3(blocks), dim3(threads), 0, 0, buf, mask, per, vec);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (rep == 1) printf("%d %8.1f KB vec%d %8.2f\n", mode, (mask + 1) * 4.0 / 1024, vec, total * vec / ms / 1e6);
                }
            }
    float* out; CK(hipMalloc(&out, blocks * 4));
    for (int lg = 9; lg <= 15; lg += 2) {
        unsigned mask = (1u << lg) - 1;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), (mask + 1) * 4, 0, out, per * 16, mask);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 1) printf("LDS atomic %6.1f KB %8.2f Gops/s\n", (mask + 1) * 4.0 / 1024, total * 16 / ms / 1e6);
        }
    }
    return 0;
}
