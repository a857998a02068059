// The AdamW update of ONE element, shared by the optimizer kernel (optim.hip) and by the scatter kernels that apply it in their
// flush (encode_bwd.hip: the gradient of a grid element goes from the LDS table straight into the update - it is never written
// to or read from HBM).  Both translation units are built with -ffp-contract=off: every product and sum is rounded separately,
// in torch.optim.AdamW's single-tensor order (training.py:13-14):
//   p *= 1 - lr*wd;  m += (g - m)(1 - b1);  v = v*b2 + (1 - b2) g g;  p += -(lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>

struct AdamScalars {
    float decay;          // 1 - lr*wd
    float one_m_b1, b2, one_m_b2;
    float neg_step;       // -lr / (1 - b1^t)
    float bc2_sqrt;       // sqrt(1 - b2^t)
    float eps, grad_scale;
    int scale_grad;
};

// scalar pre-computation in double, like the Python side of torch.optim.AdamW
inline AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, double grad_scale) {
    AdamScalars S;
    S.decay = (float)(1.0 - lr * weight_decay);
    S.one_m_b1 = (float)(1.0 - beta1);
    S.b2 = (float)beta2;
    S.one_m_b2 = (float)(1.0 - beta2);
    S.neg_step = (float)(-(lr / (1.0 - pow(beta1, (double)step))));
    S.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    S.eps = (float)eps;
    S.grad_scale = (float)grad_scale;
    S.scale_grad = grad_scale != 1.0;
    return S;
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamScalars& S) {
    if (S.scale_grad) g = g * S.grad_scale;
    p = p * S.decay;
    m = m + (g - m) * S.one_m_b1;
    v = v * S.b2 + S.one_m_b2 * g * g;
    const float denom = sqrtf(v) / S.bc2_sqrt + S.eps;
    p = p + S.neg_step * (m / denom);
}
