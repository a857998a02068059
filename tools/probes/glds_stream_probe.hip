// Probe: can ONE 768-thread workgroup per CU stream the grouped dW operand set (dp, dq, h, q: 128 rows; z: 116 rows; PTM4) at
// HBM speed if the tiles go global -> LDS by DMA (global_load_lds, 16 B per lane) through a ring of S stages of 16-pixel half
// tiles?  The register-staged grouped kernel (mlp_dw_group_kernel) holds 64-80 KB per CU in flight only part of the time and
// reaches 2.8 TB/s; the per-job kernels hold 96 KB and reach 4.9 TB/s.  This skeleton moves the same bytes with S - 1 stages
// (40 KB each) continuously in flight and optionally issues the real kernel's MFMA count per step on dummy operands.
//   hipcc --offload-arch=gfx950 -O3 -o glds_stream_probe glds_stream_probe.hip && ./glds_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 768, kWavesPerWg = 12;
constexpr int kStreamUnits = 512;                // 16-B units of a 128-row stream per 16-px half tile (32 row-groups x 16 px)
constexpr int kZUnits = 29 * 16;                 // 116-row latent
constexpr int kStageUnits = 4 * kStreamUnits + kZUnits;      // 2512 units = 40 192 B
constexpr int kStageInstr = (kStageUnits + 63) / 64;        // 40 wave instructions per stage

struct Args { const float* s[5]; int rows[5]; float* out; int64_t ntiles; int tiles_per_chunk; int mfma; };

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int S>
__global__ __launch_bounds__(kThreads, 1) void probe_kernel(Args A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t h0 = (int64_t)blockIdx.x * A.tiles_per_chunk * 2;                 // half tiles of this chunk
    const int64_t h1 = min(A.ntiles * 2, h0 + (int64_t)A.tiles_per_chunk * 2);
    const int nsteps = (int)(h1 - h0);
    constexpr int kStageBytes = kStageUnits * 16;

    auto issue = [&](int step) {                 // DMA of half tile h0 + step into stage step % S
        const int64_t ht = h0 + step;
        const int64_t tile = ht >> 1;
        const int half = (int)(ht & 1);
        char* stage = lds + (step % S) * kStageBytes;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = wv + kWavesPerWg * k;               // wave-instruction index within the stage
            if (q >= kStageInstr) break;
            const int u = q * 64 + lane;                      // 16-B unit within the stage
            int sid = u / kStreamUnits;
            if (sid > 4) sid = 4;
            const int f = u - sid * kStreamUnits;             // unit within the stream's half tile: rg * 16 + px
            const int rg = f >> 4, px = f & 15;
            const int rgs = A.rows[sid] >> 2;
            const bool ok = rg < rgs;
            // source: PTM4 tile [rg][32 px][4]; this half's pixels 16 half .. + 15, rotated by 2 rg inside the row-group (bank swizzle)
            const int spx = (px - 2 * rg) & 15;
            const float4* src = reinterpret_cast<const float4*>(A.s[sid]) + (tile * rgs + (ok ? rg : 0)) * 32 + 16 * half + spx;
            if (u < kStageUnits) glds16(src, stage + q * 1024);     // LDS: wave-uniform base, lane x 16 B added by the hardware
        }
    };

    float acc = 0.f;
    f32x16 c0 = {0}, c1 = {0};
    for (int s = 0; s < S - 1 && s < nsteps; ++s) issue(s);
    for (int s = 0; s < nsteps; ++s) {
        if (s + S - 1 < nsteps) issue(s + S - 1);
        // wait for stage s: everything issued after it may stay in flight
        const int ahead = min(S - 1, nsteps - 1 - s);
        const bool four = wv < (kStageInstr - 3 * kWavesPerWg);          // waves that issue 4 instructions per stage
        if (ahead >= 3) { if (four) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
        else if (ahead == 2) { if (four) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else if (ahead == 1) { if (four) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // consume: every thread reads a few 16-B units of the stage
        const float4* st = reinterpret_cast<const float4*>(lds + (s % S) * kStageBytes);
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float4 v = st[(tid + kThreads * k) % kStageUnits]; acc += v.x + v.w; }
        for (int m = 0; m < A.mfma; ++m) {
            f16x8 a, b;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)acc; b[e] = (_Float16)(acc + 1.f); }
            if (m & 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); else c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // the stage may be overwritten by the next issue
    }
    A.out[blockIdx.x * kThreads + tid] = acc + c0[0] + c1[3];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int S>
void run(Args A, int chunks, double bytes) {
    const size_t ldsb = (size_t)S * kStageUnits * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mf : {0, 12, 24}) {
        A.mfma = mf;
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(probe_kernel<S>, dim3(chunks), dim3(kThreads), ldsb, 0, A);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(probe_kernel<S>, dim3(chunks), dim3(kThreads), ldsb, 0, A);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("stages %d  lds %zu B  mfma/step/wave %2d : %.3f ms  %.2f TB/s\n", S, ldsb, mf, ms, bytes / ms * 1e-9);
    }
}

int main() {
    const int64_t n = 1245184, ntiles = n / 32;
    Args A;
    const int rows[5] = {128, 128, 128, 128, 116};
    double bytes = 0;
    for (int i = 0; i < 5; ++i) {
        float* p; const size_t b = (size_t)ntiles * rows[i] * 32 * 4;
        CK(hipMalloc(&p, b)); CK(hipMemset(p, 0x11 * (i + 1), b));
        A.s[i] = p; A.rows[i] = rows[i]; bytes += b;
    }
    const int chunks = 256;
    CK(hipMalloc(&A.out, (size_t)chunks * kThreads * 4));
    A.ntiles = ntiles; A.tiles_per_chunk = (int)((ntiles + chunks - 1) / chunks);
    printf("grouped dW operand set (k = 2): %.2f GB per launch, %d chunks x %d threads\n", bytes * 1e-9, chunks, kThreads);
    run<2>(A, chunks, bytes);
    run<3>(A, chunks, bytes);
    run<4>(A, chunks, bytes);
    return 0;
}
