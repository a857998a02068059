// Weight gradients of the modulator MLP + modulated SIREN (the dW half of R12):
//   dW_k[out][in] = sum_px dY_k[out][px] * X_k[in][px]        db_k[out] = sum_px dY_k[out][px]
// i.e. GEMMs whose CONTRACTION axis is the pixel axis (1.2 M long) and whose output is tiny
// (128 x {114,242,128}).  They are run split-K over pixel chunks on fp32 MFMA:
//
//  * both operands come from pixel-tile-major (PTM) streams [tile][feature][32 px] written
//    by mlp_fwd/mlp_bwd, so an MFMA fragment for feature row i is "16 consecutive pixels of
//    row i" = one 64-B run per lane, a fully coalesced 4-KiB read per wave - no transposes;
//    lane (i, h) feeds pixel 16h + k at k-step k on BOTH operands, so the pairing is exact.
//  * a job = one (layer, group of 4 column tiles) = a 128 x 128 block of one dW; grid = (pixel
//    chunk, job).  Each of the 4 waves owns a 64 x 64 sub-block = 2 x 2 MFMA tiles (64
//    accumulator registers resident across the whole pixel chunk; 2 A + 2 B fragments feed
//    64 MFMAs per 32-pixel tile); the next tile's fragments are prefetched into a second
//    register set while the current tile's MFMAs run.
//  * per-chunk partial results are written in the parameters' natural [out][in] layout
//    (a D fragment row is 32 consecutive `in` columns = one 128-B line), then summed over
//    chunks by a second kernel in a fixed order -> deterministic gradients.
//
// Bound: fp32 MFMA (219 648 FLOP/px for nvp_s) with ~7 KB/px of HBM reads riding along.
#include "mlp_layout.h"

namespace {

struct DwJob {
    const float* a;       // dY stream (PTM, 128 rows) or nullptr for the "small" job
    const float* b;       // X stream (PTM, b_rows rows)
    int b_rows;           // rows per tile in the B stream (its PTM stride)
    int b_row0;           // first B row of this job's 4 column tiles
    int n_cols;           // valid columns (features) from b_row0 on, <= 128
    int64_t w_off;        // offset of W[0][col0] inside a partial
    int ld;               // leading dimension of W
    int64_t bias_off;     // >= 0: this job also produces the bias gradient
};

struct DwArgs {
    DwJob job[12];
    int n_jobs;           // the last job is the "small" one (last layer + SIREN layer 0)
    const float* drgb;
    const float* steps;
    const float* x2;
    const float* dq0s;
    int64_t last_w, last_b, sir0_w, sir0_b;
    int64_t total;        // floats per partial
};

__device__ __forceinline__ void load_frag(float (&f)[16], const float* __restrict__ row_ptr) {
    const float4* p = reinterpret_cast<const float4*>(row_ptr);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
}

// fragment set of one 32-pixel tile for a wave that owns a 64-row x 64-column block (2 x 2 MFMA tiles)
struct Frags {
    float a[2][16];
    float b[2][16];
};

__device__ __forceinline__ void load_frags(Frags& f, const DwJob& J, int64_t t, int wr, int wc, int i, int h, const bool (&bval)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) load_frag(f.a[r], J.a + ((t * NVP_H + 64 * wr + 32 * r + i) * 32 + 16 * h));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (bval[c]) load_frag(f.b[c], J.b + ((t * J.b_rows + J.b_row0 + 64 * wc + 32 * c + i) * 32 + 16 * h));
        else {
#pragma unroll
            for (int k = 0; k < 16; ++k) f.b[c][k] = 0.f;
        }
    }
}

__device__ __forceinline__ void mma_frags(f32x16 (&acc)[2][2], const Frags& f, float& bsum, bool want_bias) {
    if (want_bias) {
#pragma unroll
        for (int k = 0; k < 16; ++k) bsum += f.a[0][k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        acc[0][0] = nvp_mfma(f.a[0][k], f.b[0][k], acc[0][0]);
        acc[0][1] = nvp_mfma(f.a[0][k], f.b[1][k], acc[0][1]);
        acc[1][0] = nvp_mfma(f.a[1][k], f.b[0][k], acc[1][0]);
        acc[1][1] = nvp_mfma(f.a[1][k], f.b[1][k], acc[1][1]);
    }
}

__global__ __launch_bounds__(256, 2) void mlp_dw_kernel(DwArgs A, float* __restrict__ partials, int64_t n, int64_t ntiles, int tiles_per_chunk) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // provably wave-uniform
    const int i = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x;
    const int64_t t0 = (int64_t)chunk * tiles_per_chunk;
    const int64_t t1 = min(ntiles, t0 + tiles_per_chunk);
    float* part = partials + (int64_t)chunk * A.total;

    if ((int)blockIdx.y < A.n_jobs - 1) {
        const DwJob J = A.job[blockIdx.y];
        const int wr = w >> 1, wc = w & 1;            // wave owns rows 64wr.., columns 64wc..
        f32x16 acc[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[r][c] = nvp_zero16();
        float bsum = 0.f, bsum1 = 0.f;
        bool bval[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) bval[c] = (64 * wc + 32 * c + i) < J.n_cols;
        const bool want_bias = (J.bias_off >= 0) && (wc == 0);

        // software pipeline over pixel tiles: the next tile's 16 x 16-B loads are in flight while
        // the 64 MFMAs of the current tile run (two statically named fragment sets)
        Frags f0, f1;
        int64_t t = t0;
        if (t < t1) load_frags(f0, J, t, wr, wc, i, h, bval);
        while (t < t1) {
            if (t + 1 < t1) load_frags(f1, J, t + 1, wr, wc, i, h, bval);
            asm volatile("" ::: "memory");
            mma_frags(acc, f0, bsum, want_bias);
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bsum1 += f0.a[1][k];
            }
            if (++t >= t1) break;
            if (t + 1 < t1) load_frags(f0, J, t + 1, wr, wc, i, h, bval);
            asm volatile("" ::: "memory");
            mma_frags(acc, f1, bsum, want_bias);
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bsum1 += f1.a[1][k];
            }
            ++t;
        }
        // D[row = out][col = in]: lane holds column i of each tile, rows 8g+4h+e
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = 64 * wc + 32 * c + i;
            if (col < J.n_cols) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 64 * wr + 32 * r2 + nvp_frag_row(r, h);
                        part[J.w_off + (int64_t)row * J.ld + col] = acc[r2][c][r];
                    }
            }
        }
        if (want_bias) {
            bsum += __shfl_xor(bsum, 32);
            bsum1 += __shfl_xor(bsum1, 32);
            if (h == 0) {
                part[J.bias_off + 64 * wr + i] = bsum;
                part[J.bias_off + 64 * wr + 32 + i] = bsum1;
            }
        }
        return;
    }

    // ---- small job: last layer (3 x 128, A = drgb^T) and SIREN layer 0 (128 x 1) ----------
    {
        f32x16 acc = nvp_zero16();
        float bsum = 0.f;          // d last_b (rows 0..2, wave 0)
        float w0sum = 0.f, c0sum = 0.f;
        for (int64_t t = t0; t < t1; ++t) {
            const int64_t px0 = t * 32 + 16 * h;
            float a[16], b[16], q[16], s[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t px = px0 + k;
                a[k] = (i < 3 && px < n) ? A.drgb[px * 3 + i] : 0.f;
                s[k] = px < n ? A.steps[px] : 0.f;
            }
            load_frag(b, A.x2 + ((t * NVP_H + 32 * w + i) * 32 + 16 * h));
            load_frag(q, A.dq0s + ((t * NVP_H + 32 * w + i) * 32 + 16 * h));
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc = nvp_mfma(a[k], b[k], acc);
                bsum += a[k];
                w0sum = __fmaf_rn(q[k], s[k], w0sum);
                c0sum += q[k];
            }
        }
        // d last_w[c][32w + i]: D rows 0..2 live in lane half 0, registers 0..2
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) part[A.last_w + (int64_t)r * NVP_H + 32 * w + i] = acc[r];
        }
        bsum += __shfl_xor(bsum, 32);
        w0sum += __shfl_xor(w0sum, 32);
        c0sum += __shfl_xor(c0sum, 32);
        if (h == 0) {
            if (w == 0 && i < 3) part[A.last_b + i] = bsum;
            part[A.sir0_w + 32 * w + i] = w0sum;
            part[A.sir0_b + 32 * w + i] = c0sum;
        }
    }
}

struct ReduceArgs {
    float* dst[14];
    int64_t off[15];
};

__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ partials, ReduceArgs R, int n_chunks, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float s = 0.f;
    for (int c = 0; c < n_chunks; ++c) s += partials[(int64_t)c * total + idx];
    int t = 0;
    while (idx >= R.off[t + 1]) ++t;
    R.dst[t][idx - R.off[t]] = s;
}

}  // namespace

extern "C" int nvp_mlp_bwd_dw(const float* drgb, const float* steps, const float* zt, const float* saved,
                              const float* dy, const float* xs, float* partials, int32_t n_chunks,
                              const nvp_mlp_grads* g, int64_t n, int32_t d, void* stream) {
    if (!drgb || !steps || !zt || !saved || !dy || !xs || !partials || !g || n < 0 || d < 1 || n_chunks < 1) return NVP_ERR_BADARG;
    const NvpParamLayout P = nvp_param_layout(d);
    const int64_t ntiles = nvp_ntiles(n);
    const int rows = nvp_rows_even(d);
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    if (rows > 256) return NVP_ERR_UNSUPPORTED;

    DwArgs A;
    int nj = 0;
    // modulator layers: A = dp_k; B = [h_{k-1} ; z]
    for (int k = 0; k < 3; ++k) {
        const int ld = (k == 0) ? d : NVP_H + d;
        bool bias_done = false;
        if (k > 0) {
            DwJob& J = A.job[nj++];
            J.a = dy + (int64_t)k * act; J.b = saved + (int64_t)(k - 1) * act; J.b_rows = NVP_H; J.b_row0 = 0;
            J.n_cols = NVP_H; J.w_off = P.mod_w[k]; J.ld = ld; J.bias_off = P.mod_b[k];
            bias_done = true;
        }
        for (int c0 = 0; c0 < d; c0 += 128) {
            DwJob& J = A.job[nj++];
            J.a = dy + (int64_t)k * act; J.b = zt; J.b_rows = rows; J.b_row0 = c0;
            J.n_cols = (d - c0 < 128) ? d - c0 : 128;
            J.w_off = P.mod_w[k] + (k == 0 ? 0 : NVP_H) + c0; J.ld = ld;
            J.bias_off = bias_done ? -1 : P.mod_b[k];
            bias_done = true;
        }
    }
    // SIREN layers 1, 2: A = dq_k; B = x_{k-1}
    for (int k = 1; k <= 2; ++k) {
        DwJob& J = A.job[nj++];
        J.a = dy + (int64_t)(3 + k) * act; J.b = xs + (int64_t)(k - 1) * act; J.b_rows = NVP_H; J.b_row0 = 0;
        J.n_cols = NVP_H; J.w_off = P.sir_w[k]; J.ld = NVP_H; J.bias_off = P.sir_b[k];
    }
    A.n_jobs = nj + 1;
    A.drgb = drgb; A.steps = steps; A.x2 = xs + 2 * act; A.dq0s = dy + 3 * act;
    A.last_w = P.last_w; A.last_b = P.last_b; A.sir0_w = P.sir_w[0]; A.sir0_b = P.sir_b[0];
    A.total = P.total;

    const int tiles_per_chunk = (int)((ntiles + n_chunks - 1) / n_chunks);
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(n_chunks, A.n_jobs), dim3(256), 0, (hipStream_t)stream, A, partials, n, ntiles, tiles_per_chunk);
    NVP_LAUNCH_CHECK();

    ReduceArgs R;
    int t = 0;
    for (int k = 0; k < 3; ++k) { R.dst[t] = g->mod_w[k]; R.off[t++] = P.mod_w[k]; R.dst[t] = g->mod_b[k]; R.off[t++] = P.mod_b[k]; }
    for (int k = 0; k < 3; ++k) { R.dst[t] = g->sir_w[k]; R.off[t++] = P.sir_w[k]; R.dst[t] = g->sir_b[k]; R.off[t++] = P.sir_b[k]; }
    R.dst[t] = g->last_w; R.off[t++] = P.last_w;
    R.dst[t] = g->last_b; R.off[t++] = P.last_b;
    R.off[t] = P.total;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, R, n_chunks, P.total);
    NVP_LAUNCH_CHECK();
    return 0;
}
