// Dense AdamW over a list of fp32 tensors in ONE launch (reference training.py:13-14,73-76:
// torch.optim.AdamW(lr, weight_decay=1e-3), stepped every iteration after backward).
//
// Update rule = torch.optim.AdamW (decoupled weight decay, no amsgrad), in its single-tensor order:
//   p *= 1 - lr*wd;  m += (g - m)(1 - b1);  v = v*b2 + (1 - b2) g g;
//   p += -(lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with every product/sum rounded separately (this file is built with -ffp-contract=off).  `grad_scale`
// multiplies g first (data parallel: 1/world after a SUM all-reduce, so no separate scaling pass).
//
// Bound: HBM.  16 B read (p,g,m,v) + 12 B written (p,m,v) per element = 3.8 GB per step for nvp_s; one
// pass, 16-B accesses, nothing cached (the state is 2.2 GB, far beyond L2/MALL).
#include "nvp_common.h"
#include "adamw.h"

namespace {

constexpr int kMaxSegs = 24;
constexpr int kBlockFloats = 256 * 4 * 4;        // 256 threads x 4 float4

struct AdamSegs {
    float* p[kMaxSegs];
    const float* g[kMaxSegs];
    float* m[kMaxSegs];
    float* v[kMaxSegs];
    int64_t n[kMaxSegs];
    int32_t blk0[kMaxSegs + 1];
    int32_t aligned[kMaxSegs];
    int32_t n_segs;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamSegs A, AdamScalars S) {
    int s = 0;
    while (s + 1 < A.n_segs && (int)blockIdx.x >= A.blk0[s + 1]) ++s;        // wave-uniform scan over <= 24 entries
    const int64_t base = (int64_t)((int)blockIdx.x - A.blk0[s]) * kBlockFloats;
    const int64_t n = A.n[s];
    float* __restrict__ P = A.p[s];
    const float* __restrict__ G = A.g[s];
    float* __restrict__ M = A.m[s];
    float* __restrict__ V = A.v[s];
    if (A.aligned[s]) {
        float4 p4[4], g4[4], m4[4], v4[4];
        int64_t at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            at[u] = base + ((int64_t)u * 256 + threadIdx.x) * 4;
            if (at[u] + 3 < n) {
                p4[u] = *reinterpret_cast<const float4*>(P + at[u]);
                g4[u] = *reinterpret_cast<const float4*>(G + at[u]);
                m4[u] = *reinterpret_cast<const float4*>(M + at[u]);
                v4[u] = *reinterpret_cast<const float4*>(V + at[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (at[u] + 3 < n) {
                adam1(p4[u].x, g4[u].x, m4[u].x, v4[u].x, S);
                adam1(p4[u].y, g4[u].y, m4[u].y, v4[u].y, S);
                adam1(p4[u].z, g4[u].z, m4[u].z, v4[u].z, S);
                adam1(p4[u].w, g4[u].w, m4[u].w, v4[u].w, S);
                *reinterpret_cast<float4*>(P + at[u]) = p4[u];
                *reinterpret_cast<float4*>(M + at[u]) = m4[u];
                *reinterpret_cast<float4*>(V + at[u]) = v4[u];
            } else {
                for (int64_t i = at[u]; i < n && i < at[u] + 4; ++i) {        // ragged tail of the tensor
                    float p = P[i], m = M[i], v = V[i];
                    adam1(p, G[i], m, v, S);
                    P[i] = p; M[i] = m; V[i] = v;
                }
            }
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < n && i < base + kBlockFloats; i += 256) {
            float p = P[i], m = M[i], v = V[i];
            adam1(p, G[i], m, v, S);
            P[i] = p; M[i] = m; V[i] = v;
        }
    }
}

}  // namespace

extern "C" int nvp_adamw_step(const nvp_adamw_seg* segs, int32_t n_segs, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int64_t step, double grad_scale, void* stream) {
    if (n_segs < 0 || (n_segs > 0 && !segs) || step < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1)) return NVP_ERR_BADARG;
    const AdamScalars S = adam_scalars(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    int s = 0;
    while (s < n_segs) {
        AdamSegs A;
        int cnt = 0;
        int64_t blocks = 0;
        for (; s < n_segs && cnt < kMaxSegs; ++s) {
            const nvp_adamw_seg& q = segs[s];
            if (q.n < 0 || (q.n > 0 && (!q.param || !q.grad || !q.exp_avg || !q.exp_avg_sq))) return NVP_ERR_BADARG;
            if (q.n == 0) continue;
            A.p[cnt] = q.param; A.g[cnt] = q.grad; A.m[cnt] = q.exp_avg; A.v[cnt] = q.exp_avg_sq; A.n[cnt] = q.n;
            A.aligned[cnt] = ((((uintptr_t)q.param | (uintptr_t)q.grad | (uintptr_t)q.exp_avg | (uintptr_t)q.exp_avg_sq) & 15) == 0) ? 1 : 0;
            A.blk0[cnt] = (int32_t)blocks;
            blocks += (q.n + kBlockFloats - 1) / kBlockFloats;
            if (blocks > 0x7fffffff) return NVP_ERR_UNSUPPORTED;
            ++cnt;
        }
        if (cnt == 0) break;
        A.blk0[cnt] = (int32_t)blocks;
        A.n_segs = cnt;
        hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, S);
        NVP_LAUNCH_CHECK();
    }
    return 0;
}
