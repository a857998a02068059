// Probe for VERDICT r5 item 6: does a BLOCKED cell layout cut the forward gather's over-fetch?
//
// The fused forward fetches 3.97 GB per launch against 2.02 GB of coordinates + cells (profiles/r05s_pmc_summary.txt).  The two access
// patterns without locality under y-sorted batches: the xt plane's fine levels (a pixel's (t, x) is random: 2 grid rows x 16 B per
// level) and the sparse grid's 3 x 3 patch (three 24-B runs 2.4 KB apart).  This program gathers exactly those cells for
// N = 1 245 184 random pixels from (a) the state_dict layouts and (b) 4 x 4-cell blocks of 128 B (one L2 line), with the same arithmetic,
// and prints the time of each kernel; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` for the bytes (tools/README.md).
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_layout_probe gather_layout_probe.hip && ./gather_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kLevels = 8;                 // levels 8..15 of config_nvp_s (16 * 1.35^l): 176 .. 1439 cells per side
struct Lv { int res[kLevels]; float scale[kLevels]; long long off[kLevels]; };

__device__ __forceinline__ unsigned hash_u(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd01(unsigned i, unsigned salt) { return (hash_u(i * 2654435761u + salt) >> 8) * (1.0f / 16777216.0f); }

// ---- dense plane, F = 2: row-major (cell = i0 + i1 * res: the state_dict layout, dim 0 fastest) vs 4 x 4 blocks
template <bool BLOCKED>
__global__ __launch_bounds__(256) void dense_kernel(const float2* __restrict__ tab, Lv lv, float* __restrict__ sink, int n) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= n) return;
    const float x0 = rnd01(px, 1u), x1 = rnd01(px, 2u);
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < kLevels; ++l) {
        const int res = lv.res[l];
        const float p0 = fmaf(x0, lv.scale[l], 0.5f), p1 = fmaf(x1, lv.scale[l], 0.5f);
        const int i0 = (int)p0, i1 = (int)p1;
        const float w0 = p0 - i0, w1 = p1 - i1;
        const float2* base = tab + lv.off[l];
        float2 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int a = min(i0 + (c & 1), res - 1), b = min(i1 + (c >> 1), res - 1);
            long long cell;
            if (BLOCKED) { const int nb = (res + 3) >> 2; cell = ((long long)(b >> 2) * nb + (a >> 2)) * 16 + (b & 3) * 4 + (a & 3); }
            else cell = (long long)b * res + a;
            v[c] = base[cell];
        }
        acc += (1 - w0) * (1 - w1) * (v[0].x + v[0].y) + w0 * (1 - w1) * (v[1].x + v[1].y) + (1 - w0) * w1 * (v[2].x + v[2].y) + w0 * w1 * (v[3].x + v[3].y);
    }
    if (acc == 123.456f) sink[0] = acc;
}

// ---- sparse grid [T][X][Y][2]: 3 x 3 patch; blocked: (x, y) in 4 x 4 blocks of 128 B per t
template <bool BLOCKED>
__global__ __launch_bounds__(256) void sparse_kernel(const float2* __restrict__ emb, int T, int X, int Y, float* __restrict__ sink, int n) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= n) return;
    const int t = min((int)(rnd01(px, 3u) * T), T - 1), x = min((int)(rnd01(px, 4u) * X), X - 1), y = min((int)(rnd01(px, 5u) * Y), Y - 1);
    float acc = 0.f;
    const int nbx = (X + 3) >> 2, nby = (Y + 3) >> 2;
#pragma unroll
    for (int i = -1; i <= 1; ++i)
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
            const int a = min(max(x + i, 0), X - 1), b = min(max(y + j, 0), Y - 1);
            long long cell;
            if (BLOCKED) cell = (long long)t * nbx * nby * 16 + ((long long)(a >> 2) * nby + (b >> 2)) * 16 + (a & 3) * 4 + (b & 3);
            else cell = ((long long)t * X + a) * Y + b;
            const float2 v = emb[cell];
            acc += v.x + v.y;
        }
    if (acc == 123.456f) sink[0] = acc;
}

template <typename K, typename... A>
float time_kernel(const char* name, K kern, dim3 grid, A... args) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, args...);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, args...);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us per launch\n", name, ms / reps * 1e3f);
    return ms / reps;
}

int main() {
    const int n = 1245184;
    Lv lv;
    long long cells = 0;
    for (int l = 0; l < kLevels; ++l) {
        const double s = 16.0 * pow(1.35, 8 + l) - 1.0;
        lv.scale[l] = (float)s;
        lv.res[l] = (int)ceil(s) + 1;
        lv.off[l] = cells;
        const int nb = (lv.res[l] + 3) / 4;
        cells += (long long)nb * nb * 16;                       // room for the blocked layout (row-major uses res * res of it)
    }
    float2* tab; float* sink;
    CK(hipMalloc(&tab, cells * sizeof(float2)));
    CK(hipMemset(tab, 0, cells * sizeof(float2)));
    CK(hipMalloc(&sink, 4));
    const int T = 600, X = 300, Y = 300;
    const long long scells = (long long)T * ((X + 3) / 4) * ((Y + 3) / 4) * 16;
    float2* emb;
    CK(hipMalloc(&emb, scells * sizeof(float2)));
    CK(hipMemset(emb, 0, scells * sizeof(float2)));
    dim3 grid((n + 255) / 256);
    printf("N = %d pixels; xt-plane levels 8..15: %.1f MB; sparse grid %.0f MB\n", n, cells * 8 / 1e6, scells * 8 / 1e6);
    printf("ideal bytes: dense 8 levels x 4 corners x 8 B = %.3f GB; sparse 9 cells x 8 B = %.3f GB\n", n * 8.0 * 4 * 8 / 1e9, n * 72.0 / 1e9);
    time_kernel("dense xt levels 8-15, row-major (state_dict)", dense_kernel<false>, grid, (const float2*)tab, lv, sink, n);
    time_kernel("dense xt levels 8-15, 4x4 blocks (128 B)", dense_kernel<true>, grid, (const float2*)tab, lv, sink, n);
    time_kernel("sparse 3x3, [T][X][Y][F] (state_dict)", sparse_kernel<false>, grid, (const float2*)emb, T, X, Y, sink, n);
    time_kernel("sparse 3x3, 4x4 (x,y) blocks (128 B)", sparse_kernel<true>, grid, (const float2*)emb, T, X, Y, sink, n);
    return 0;
}
