// Weight gradients of the modulator MLP + modulated SIREN (the dW half of R12):
//   dW_k[out][in] = sum_px dY_k[out][px] * X_k[in][px]        db_k[out] = sum_px dY_k[out][px]
// i.e. GEMMs whose CONTRACTION axis is the pixel axis (1.2 M long) and whose output is tiny
// (128 x {114,242,128}).  They are run split-K over pixel chunks on the matrix cores - split-operand 16-bit MFMA by
// default (NVP_DW_B3; fragments split after the LDS read, see mlp_b3.h: fp16 x 2 with a running block scale per operand,
// or bf16 x 3), fp32 MFMA otherwise:
//
//  * both operands come from the PTM4 streams written by mlp_fwd/mlp_bwd.  A 128-row x 32-px
//    tile is 16 KiB contiguous: the block stages it through LDS with full-line 16-B loads
//    (the ablation/PMC runs showed fragment-shaped global loads - 32 lines touched per wave
//    instruction - starve the MFMA pipe), double-buffered, one barrier per pixel tile;
//    lane (i, h) feeds pixel 16h + k at k-step k on BOTH operands, so the pairing is exact.
//  * a job = one (layer, group of 4 column tiles) = a 128 x 128 block of one dW; grid = (pixel
//    chunk, job).  Each of the 4 waves owns a 64 x 64 sub-block = 2 x 2 MFMA tiles (64
//    accumulator registers resident across the whole pixel chunk; 2 A + 2 B fragments feed
//    64 fp32 / 48 bf16 MFMAs per 32-pixel tile); the next tile's global data is prefetched into
//    registers while the current tile's MFMAs run.
//  * per-chunk partial results are written in the parameters' natural [out][in] layout
//    (a D fragment row is 32 consecutive `in` columns = one 128-B line), then summed over
//    chunks by a second kernel in a fixed order -> deterministic gradients.
//
//  * the two tiny layers (last layer 3 x 128, SIREN layer 0: 643 values) are not GEMM jobs: the backward
//    chain kernel already reduced them over each tile's 32 pixels (mlp_bwd.hip, kRec* records); here
//    the records are summed per pixel chunk into the same partials.
//
// Bound: matrix pipe + the operand-split VALU work + ~6 KB/px of HBM reads (219 648 FLOP/px for nvp_s); they add.
#include <cstdlib>
#include <initializer_list>
#include "mlp_b3.h"

#ifndef NVP_DW_B3
#define NVP_DW_B3 1           // dW GEMMs on bf16 x 3 split MFMA (fragments split after the LDS read): 2.72 -> 2.21 ms
#endif

// experiment: wave priority of the compute phase (LDS fragment reads, operand splits, MFMAs) against the staging phase of the other
// workgroups' waves on the SIMD (1: compute outranks staging, 2: staging outranks compute)
#ifndef NVP_DW_PRIO
#define NVP_DW_PRIO 2            // measured (profiles/r04_ab_wave_priority.txt): 1.66 -> 1.57 ms (1: 1.59); bit-identical
#endif
#if NVP_DW_PRIO == 1
#define NVP_DW_COMPUTE_ENTER() __builtin_amdgcn_s_setprio(2)
#define NVP_DW_COMPUTE_LEAVE() __builtin_amdgcn_s_setprio(0)
#elif NVP_DW_PRIO == 2
#define NVP_DW_COMPUTE_ENTER() __builtin_amdgcn_s_setprio(0)
#define NVP_DW_COMPUTE_LEAVE() __builtin_amdgcn_s_setprio(2)
#else
#define NVP_DW_COMPUTE_ENTER()
#define NVP_DW_COMPUTE_LEAVE()
#endif
#ifndef NVP_DW_PAIR_DEFAULT
#define NVP_DW_PAIR_DEFAULT 0
#endif

namespace {

inline bool getenv_on(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }

struct DwJob {
    const float* a;       // dY stream (PTM4, 128 rows)
    const float* b;       // X stream (PTM4, b_rows rows): z, h_k, or the h factor of x_k = sin(q_k) h_k
    const float* b2;      // mode 1: the q_k stream; otherwise == b
    int mode;             // 0: B = b;  1: B = sin(b2) * b;  2: B = sin(30 (w0[row] s[px] + c0[row])) * b
    int b_rows;           // rows per tile in the B stream (its PTM stride)
    int b_row0;           // first B row of this job's 4 column tiles
    int n_cols;           // valid columns (features) from b_row0 on, <= 128
    int64_t w_off;        // offset of W[0][col0] inside a partial
    int ld;               // leading dimension of W
    int64_t bias_off;     // >= 0: this job also produces the bias gradient
    // second B tile of a MERGED job (NB == 2 kernels: the two 128-column halves [h_{k-1} ; z] of one modulator layer share
    // the dp_k tile in LDS, so dp_k is staged once instead of twice - the dW stage is HBM-bound on its operand streams)
    const float* bb;
    int bb_rows, bb_row0, nn_cols;
    int64_t ww_off;
};

struct DwArgs {
    DwJob job[12];
    int n_jobs;
    const float* steps;
    const float* sir0_wp;   // SIREN layer 0 weight / bias (mode 2)
    const float* sir0_bp;
    int64_t total;        // floats per partial
};

// ---- LDS staging ------------------------------------------------------------------------
// A 128-row x 32-pixel tile of a PTM4 stream is 1024 contiguous float4 (16 KiB): the 256 threads
// fetch it with four fully coalesced 16-B loads each.  It is written to LDS row-major with a
// 36-float row stride: rows stay 16-B aligned, so a lane reads its fragment (row i, 16 consecutive
// pixels) with four ds_read_b128 - 8 consecutive lanes cover all 32 banks - and the 4-B transposing
// writes (32 lanes = 32 consecutive pixels of one row) are conflict-free too.  The MFMA fragment for
// feature row i at k-step k is pixel 16h + k on BOTH operands.
#ifndef NVP_DW_STRIDE
#define NVP_DW_STRIDE 36
#endif
constexpr int kRowStride = NVP_DW_STRIDE;
constexpr int kTileFloats = 128 * kRowStride;          // 4608 floats = 18 KiB

#ifdef NVP_ABL_DW_NOMFMA          // ablation builds only: one VALU op instead of an MFMA
__device__ __forceinline__ f32x16 nvp_abl_fake_mfma(float a, float b, f32x16 c) { c[0] = __fmaf_rn(a, b, c[0]); return c; }
#endif

template <int NB>
struct Stage {
    float4 a[4 / NB];
    float4 b[4];
    float4 b2[4];
};

// 256 * NB threads stage one pixel tile: the A tile is spread over all of them (4 / NB float4 each), thread group g = tid >> 8
// stages B tile g (4 float4 per thread)
template <bool XF, int NB>
__device__ __forceinline__ void load_stage(Stage<NB>& s, const DwJob& J, int64_t t, int tid) {
    const int g = NB == 2 ? (tid >> 8) : 0, ft = tid & 255;
    const float* bsrc = g ? J.bb : J.b;
    const int brows = g ? J.bb_rows : J.b_rows, brow0 = g ? J.bb_row0 : J.b_row0;
    const float4* A4 = reinterpret_cast<const float4*>(J.a) + t * 1024;
    const float4* B4 = reinterpret_cast<const float4*>(bsrc) + (t * (brows >> 2) + (brow0 >> 2)) * 32;
    const float4* Q4 = reinterpret_cast<const float4*>(J.b2) + (t * (J.b_rows >> 2) + (J.b_row0 >> 2)) * 32;
    // rows past the end of the B stream (latent: 116 of 128) must read as zero: clamp the address here and
    // select on the DATA in write_stage - a conditional load would become a branch with a vmcnt(0) wait
    // per element, and a select right here would wait for the prefetch immediately
    const int nvalid = min(1024, ((brows - brow0) >> 2) * 32);
#ifdef NVP_ABL_DW_NOLOAD          // ablation builds only
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.b[k] = make_float4(1e-3f, 2e-3f, 3e-3f, (float)tid); s.b2[k] = s.b[k]; }
#pragma unroll
    for (int k = 0; k < 4 / NB; ++k) s.a[k] = make_float4(1e-3f, 2e-3f, 3e-3f, (float)t);
    return;
#endif
#pragma unroll
    for (int k = 0; k < 4 / NB; ++k) s.a[k] = A4[k * (256 * NB) + tid];
#pragma unroll
    for (int k = 0; k < 4; ++k) s.b[k] = B4[min(k * 256 + ft, nvalid - 1)];          // zeroed (if past the end) when it is written to LDS
    if (XF && J.mode == 1) {                      // wave-uniform (kernarg) branch: only x_k = sin(q_k) h_k jobs read q_k
#pragma unroll
        for (int k = 0; k < 4; ++k) s.b2[k] = Q4[min(k * 256 + ft, nvalid - 1)];
    }
}

// `t` is the tile the stage holds (mode 2 needs its temporal steps); lb = the B tile of this thread's group
// ma / mb: largest magnitude (bit pattern) among the A / B values this thread wrote (the fp16 x 2 split's block scale)
template <bool XF, int NB>
__device__ __forceinline__ void write_stage(float* __restrict__ la, float* __restrict__ lb0, const Stage<NB>& s, const DwJob& J,
                                            const DwArgs& A, const float* __restrict__ tab, int64_t t, int64_t n, int tid,
                                            unsigned& ma, unsigned& mb) {
    ma = 0u; mb = 0u;
#ifdef NVP_ABL_DW_NOWRITE         // ablation builds only
    if (t != 0x7fffffff) return;
#endif
    const int g = NB == 2 ? (tid >> 8) : 0, ft = tid & 255;
    float* __restrict__ lb = lb0 + g * kTileFloats;
    const int brows = g ? J.bb_rows : J.b_rows, brow0 = g ? J.bb_row0 : J.b_row0;
    const int nvalid = min(1024, ((brows - brow0) >> 2) * 32);
    float sp = 0.f;
    if (XF && J.mode == 2) sp = A.steps[min(t * 32 + (tid & 31), n - 1)];       // wave-uniform branch; px = f & 31 = tid & 31
#pragma unroll
    for (int k = 0; k < 4 / NB; ++k) {
        const int f = k * (256 * NB) + tid;
        const int o = (4 * (f >> 5)) * kRowStride + (f & 31);
        la[o] = s.a[k].x; la[o + kRowStride] = s.a[k].y; la[o + 2 * kRowStride] = s.a[k].z; la[o + 3 * kRowStride] = s.a[k].w;
        if (NVP_SPLIT_H2) ma = max(ma, __float_as_uint(absmax_f4(0.f, s.a[k])));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = k * 256 + ft;
        const int o = (4 * (f >> 5)) * kRowStride + (f & 31);
        const bool ok = f < nvalid;
        float bv[4] = {s.b[k].x, s.b[k].y, s.b[k].z, s.b[k].w};
        if (XF && J.mode == 1) {                 // x_k = sin(q_k) * h_k            (modulation.py:88-90)
            const float qv[4] = {s.b2[k].x, s.b2[k].y, s.b2[k].z, s.b2[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = nvp_sin(qv[e]) * bv[e];
        } else if (XF && J.mode == 2) {          // x_0 = sin(30 (w s + c)) * h_0
            const int row = 4 * (f >> 5);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = nvp_sin(30.0f * __fmaf_rn(sp, tab[row + e], tab[NVP_H + row + e])) * bv[e];
        }
        lb[o] = ok ? bv[0] : 0.f; lb[o + kRowStride] = ok ? bv[1] : 0.f;
        lb[o + 2 * kRowStride] = ok ? bv[2] : 0.f; lb[o + 3 * kRowStride] = ok ? bv[3] : 0.f;
        if (NVP_SPLIT_H2 && ok) mb = max(mb, __float_as_uint(absmax_f4(0.f, make_float4(bv[0], bv[1], bv[2], bv[3]))));
        __builtin_amdgcn_sched_barrier(0);       // one float4 at a time: keeps the sincos temporaries of 16 values from piling up
    }
}

__device__ __forceinline__ void read_frag(float (&f)[16], const float* __restrict__ tile, int row, int h) {
    const float* p = tile + row * kRowStride + 16 * h;
    if (kRowStride % 4 == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 t = p4[k];
            f[4 * k] = t.x; f[4 * k + 1] = t.y; f[4 * k + 2] = t.z; f[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = p[k];
    }
}

// ---- running block scale of the fp16 x 2 split (mlp_b3.h) ----------------------------------------------------------------------
// The contraction axis is the pixel axis, so a scale must be constant along it: one power of two per operand and workgroup,
// taken from the largest magnitude seen SO FAR in the chunk (the staging threads reduce each tile's maximum while they write
// it to LDS; the maxima travel through LDS with the tile).  When a tile raises the maximum, the accumulators are multiplied by
// the (exact, <= 1) ratio of the new to the old scale - a wave-uniform branch taken a handful of times per chunk.  Elements
// more than 2^-16 below the running maximum lose relative precision (absolute error <= 2^-38 of the maximum), which is far
// below the fp32 rounding of the sums they are added to.
constexpr int kMxW = 8;                               // wave slots per operand (two 4-wave groups for the merged jobs)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));      // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));      // row_mirror: every lane holds its row's max
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
// The staging wave `wall` publishes its share of a tile's maxima next to the tile (slot set `par`) - but only when one of its
// values exceeds the running maximum (run_a / run_b: what every wave of the workgroup has derived from the tiles so far): a
// tile that raises nothing needs no reduction, and whatever older tile's maxima its slots still hold are <= the running
// maximum already, so they change nothing.  After the first few tiles of a chunk the wave-uniform branch is not taken.
__device__ __forceinline__ void publish_max(unsigned* __restrict__ mx, int par, int wall, unsigned ma, unsigned mb, unsigned run_a, unsigned run_b, int lane) {
#if NVP_SPLIT_H2
    if (!__any(ma > run_a || mb > run_b)) return;
    ma = wave_umax(ma); mb = wave_umax(mb);
    if (lane == 0) { mx[(par * 2 + 0) * kMxW + wall] = ma; mx[(par * 2 + 1) * kMxW + wall] = mb; }
#endif
}

// KIND 0: the modulator jobs (plain operands); KIND 1: SIREN layers 1-2, whose B operand x_k is rebuilt
// on the fly.  Separate instantiations keep each variant's register budget tight (the union spilled).
#ifndef NVP_DW_BUFS
#define NVP_DW_BUFS 1        // LDS tile buffers of the PLAIN jobs: 1 = single (two barriers per tile, 36 KB: three workgroups per CU at 168 VGPRs;
                             // dW 1.735 -> 1.69 ms with the fp16 x 2 split), 2 = double-buffered (one barrier, two workgroups per CU).  The
                             // transform-capable instantiation (214 VGPRs) is always double-buffered.
#endif
template <int KIND, int NB>
__global__ __launch_bounds__(256 * NB, (KIND == 0 && NVP_DW_BUFS == 1 && NB == 1) ? 3 : 2) void mlp_dw_kernel(DwArgs A, float* __restrict__ partials, int64_t n, int64_t ntiles, int tiles_per_chunk, int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [2 buffers][A tile | B tile (| second B tile)]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);           // provably wave-uniform
    const int w = wall & 3, wb = wall >> 2;                              // wave within its 128 x 128 block, B tile of the block
    const int i = lane & 31, h = lane >> 5;
    // XCD-aware block mapping.  Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8,
    // observed; only speed depends on it).  The jobs of one pixel chunk share streams (z is read by three
    // jobs, h0/h1/dp1/dp2 by two), so they are given consecutive slots on the SAME XCD: they start together
    // and the second reader of a stream finds it in that XCD's L2 instead of HBM.
    int chunk, job;
    {
        const int L = blockIdx.x, nj = A.n_jobs;
#ifndef NVP_DW_XCD
#define NVP_DW_XCD 1
#endif
        if (NVP_DW_XCD && (n_chunks & 7) == 0) {
            const int xcd = L & 7, slot = L >> 3;
            chunk = (slot / nj) * 8 + xcd;
            job = slot - (slot / nj) * nj;
        } else {
            chunk = L / nj;
            job = L - chunk * nj;
        }
    }
    const int64_t t0 = (int64_t)chunk * tiles_per_chunk;
    const int64_t t1 = min(ntiles, t0 + tiles_per_chunk);
    float* part = partials + (int64_t)chunk * A.total;
    constexpr int BUFS = (KIND == 0 && NVP_DW_BUFS == 1 && NB == 1) ? 1 : 2;     // only the plain variant fits 3 waves/SIMD
    constexpr int kBufFloats = (1 + NB) * kTileFloats;                // one buffer: A tile + NB B tiles
    constexpr bool XF = KIND != 0;
    const DwJob J = A.job[job];
    const int wr = w >> 1, wc = w & 1;                // regular jobs: wave owns rows 64wr.., columns 64wc..

    // SIREN layer 0's weight and bias (mode 2 rebuilds x_0 from them) live in LDS behind the tile buffers:
    // a global load at the point of use would put a vmcnt(0) wait into the pipelined loop
    unsigned* mx = reinterpret_cast<unsigned*>(lds + BUFS * kBufFloats);    // [2 slot sets][A | B][kMxW] tile maxima (fp16 x 2 block scale)
    float* tab = lds + BUFS * kBufFloats + 4 * kMxW;
    if (XF && tid < NVP_H) { tab[tid] = A.sir0_wp[tid]; tab[NVP_H + tid] = A.sir0_bp[tid]; }
    if (XF) __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = nvp_zero16();
    float bsum0 = 0.f, bsum1 = 0.f;                   // bias sums
    const bool want_bias = (J.bias_off >= 0) && (wc == 0) && (wb == 0);

    // Global loads run TWO tiles ahead of the MFMAs (one tile in LDS, the next two in registers st/st2):
    // under load the HBM round trip exceeds one tile's 4096 MFMA cycles.
#ifndef NVP_DW_DEPTH
#define NVP_DW_DEPTH 1
#endif
    Stage<NB> st, st2;
    unsigned wma, wmb;
    if (t0 < t1) {
        load_stage<XF, NB>(st, J, t0, tid);
        write_stage<XF, NB>(lds, lds + kTileFloats, st, J, A, tab, t0, n, tid, wma, wmb);
        if (tid < 4 * kMxW) mx[tid] = 0u;            // a skipped publish leaves its slots untouched: start them below every running maximum
        __syncthreads();
        publish_max(mx, 0, wall, wma, wmb, 0u, 0u, lane);
    }
    // block-scale state: running maxima (bit patterns) per operand, the scale S = sA sB the accumulators are in, and 1 / S
    unsigned runA = __float_as_uint(kTinyMax), runB = __float_as_uint(kTinyMax);
    PxScale qa = px_scale(kTinyMax), qb = qa;
    float curS = qa.s * qb.s, curU = qa.u * qb.u;
    int par = 0;
#if NVP_DW_DEPTH == 2
    if (t0 + 1 < t1) load_stage<XF, NB>(st, J, t0 + 1, tid);
#endif
    __syncthreads();
    int cur = 0;
    for (int64_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
#if NVP_DW_DEPTH == 2
        if (t + 2 < t1) load_stage<XF, NB>(st2, J, t + 2, tid);
#else
        if (more) load_stage<XF, NB>(st, J, t + 1, tid);
#endif
        const float* la = lds + (BUFS == 2 ? cur : 0) * kBufFloats;
        const float* lb = la + (1 + wb) * kTileFloats;
#if NVP_DW_B3
        NVP_DW_COMPUTE_ENTER();
        {
            // split-operand MFMA (mlp_b3.h): a lane's 16 pixels of a row are two k-steps of 8 (the SAME pixels on both
            // operands); fragments are split after the LDS read, the part products per (row tile, column tile, k-step)
            // accumulate in fp32
            if (NVP_SPLIT_H2) {
                const unsigned* mr = mx + par * 2 * kMxW;
                unsigned mA = 0u, mB = 0u;
#pragma unroll
                for (int u = 0; u < 4 * NB; ++u) mA = max(mA, mr[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) mB = max(mB, mr[kMxW + 4 * wb + u]);
                runA = max(runA, (unsigned)__builtin_amdgcn_readfirstlane((int)mA));
                runB = max(runB, (unsigned)__builtin_amdgcn_readfirstlane((int)mB));
                qa = px_scale(__uint_as_float(runA)); qb = px_scale(__uint_as_float(runB));
                const float S = qa.s * qb.s;
                if (S != curS) {                     // wave-uniform: a tile raised a running maximum
                    const float ratio = S * curU;   // <= 1, a power of two
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[r][c] *= ratio;
                    curS = S; curU = qa.u * qb.u;
                }
            }
            float fa[16], fb[2][16];
            read_frag(fa, la, 64 * wr + i, h);
#pragma unroll
            for (int c = 0; c < 2; ++c) read_frag(fb[c], lb, 64 * wc + 32 * c + i, h);
            BOp pb[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float x[8] = {fb[c][8 * s2], fb[c][8 * s2 + 1], fb[c][8 * s2 + 2], fb[c][8 * s2 + 3],
                                        fb[c][8 * s2 + 4], fb[c][8 * s2 + 5], fb[c][8 * s2 + 6], fb[c][8 * s2 + 7]};
                    split8(x, qb.s, pb[c][s2]);
                }
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                if (r2 == 1) read_frag(fa, la, 64 * wr + 32 + i, h);
                if (want_bias) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { if (r2 == 0) bsum0 += fa[k]; else bsum1 += fa[k]; }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float x[8] = {fa[8 * s2], fa[8 * s2 + 1], fa[8 * s2 + 2], fa[8 * s2 + 3], fa[8 * s2 + 4], fa[8 * s2 + 5], fa[8 * s2 + 6], fa[8 * s2 + 7]};
                    BOp pa;
                    split8(x, qa.s, pa);
#pragma unroll
                    for (int c = 0; c < 2; ++c) mac_parts(acc[r2][c], pa.p, pb[c][s2]);
                }
            }
        }
        NVP_DW_COMPUTE_LEAVE();
#else
        {
            float fa[16], fb[2][16];
            read_frag(fa, la, 64 * wr + i, h);
#pragma unroll
            for (int c = 0; c < 2; ++c) read_frag(fb[c], lb, 64 * wc + 32 * c + i, h);
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bsum0 += fa[k];
            }
#ifdef NVP_ABL_DW_NOMFMA
#define nvp_mfma(a, b, c) nvp_abl_fake_mfma(a, b, c)
#endif
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[0][0] = nvp_mfma(fa[k], fb[0][k], acc[0][0]);
                acc[0][1] = nvp_mfma(fa[k], fb[1][k], acc[0][1]);
            }
            read_frag(fa, la, 64 * wr + 32 + i, h);          // second row tile reuses the A registers
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bsum1 += fa[k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[1][0] = nvp_mfma(fa[k], fb[0][k], acc[1][0]);
                acc[1][1] = nvp_mfma(fa[k], fb[1][k], acc[1][1]);
            }
        }
#endif
        if (BUFS == 1) {
            __syncthreads();                        // everyone finished reading the single buffer
            if (more) write_stage<XF, NB>(lds, lds + kTileFloats, st, J, A, tab, t + 1, n, tid, wma, wmb);
        } else {
            if (more) write_stage<XF, NB>(lds + (cur ^ 1) * kBufFloats, lds + (cur ^ 1) * kBufFloats + kTileFloats, st, J, A, tab, t + 1, n, tid, wma, wmb);
        }
        if (more) publish_max(mx, par ^ 1, wall, wma, wmb, runA, runB, lane);
        par ^= 1;
#if NVP_DW_DEPTH == 2
        st = st2;
#endif
        __syncthreads();
        cur ^= 1;
    }

    {
        // D[row = out][col = in]: lane holds column i of each tile, rows 8g+4h+e
        const int ncols = wb ? J.nn_cols : J.n_cols;
        const int64_t woff = wb ? J.ww_off : J.w_off;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = 64 * wc + 32 * c + i;
            if (col < ncols) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 64 * wr + 32 * r2 + nvp_frag_row(r, h);
                        part[woff + (int64_t)row * J.ld + col] = (NVP_DW_B3 && NVP_SPLIT_H2) ? acc[r2][c][r] * curU : acc[r2][c][r];
                    }
            }
        }
        if (want_bias) {
            bsum0 += __shfl_xor(bsum0, 32);
            bsum1 += __shfl_xor(bsum1, 32);
            if (h == 0) {
                part[J.bias_off + 64 * wr + i] = bsum0;
                part[J.bias_off + 64 * wr + 32 + i] = bsum1;
            }
        }
    }
}

#if NVP_EXPERIMENTS
// ---- paired jobs: dp_k x [B | BB] in ONE 256-thread workgroup ---------------------------------------------------------------------
// Two jobs that share their A operand (dp_k x h_{k-1} and dp_k x z; for wide latents also dp_0 x z[0:128] and dp_0 x z[128:]) as
// one workgroup that stages the dp_k tile ONCE: wave (wr, wc) owns the 64 x 64 sub-block (wr, wc) of BOTH outputs (128
// accumulator registers; two workgroups per CU instead of three), the A fragments are split once for both.  Unlike the merged
// jobs above (512 threads, 108 KiB, one workgroup per CU) the workgroup keeps the per-job kernel's shape: single-buffered tiles
// (54 KiB), three tiles = 48 KiB of loads in flight per workgroup, 96 KiB per CU - the same as three per-job workgroups.  Same
// fragments, same MFMA order per accumulator, same running block scales: BIT-identical to the two separate jobs.
struct PairStage { float4 a[4], b[4], bb[4]; };

__device__ __forceinline__ void pair_load(PairStage& s, const DwJob& J, int64_t t, int tid) {
    const float4* A4 = reinterpret_cast<const float4*>(J.a) + t * 1024;
    const float4* B4 = reinterpret_cast<const float4*>(J.b) + (t * (J.b_rows >> 2) + (J.b_row0 >> 2)) * 32;
    const float4* C4 = reinterpret_cast<const float4*>(J.bb) + (t * (J.bb_rows >> 2) + (J.bb_row0 >> 2)) * 32;
    const int nv = min(1024, ((J.b_rows - J.b_row0) >> 2) * 32), nvv = min(1024, ((J.bb_rows - J.bb_row0) >> 2) * 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) s.a[k] = A4[k * 256 + tid];
#pragma unroll
    for (int k = 0; k < 4; ++k) s.b[k] = B4[min(k * 256 + tid, nv - 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k) s.bb[k] = C4[min(k * 256 + tid, nvv - 1)];
}

__device__ __forceinline__ unsigned pair_write_tile(float* __restrict__ l, const float4 (&v)[4], int nvalid, int tid) {
    unsigned m = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = k * 256 + tid;
        const int o = (4 * (f >> 5)) * kRowStride + (f & 31);
        const bool ok = f < nvalid;
        l[o] = ok ? v[k].x : 0.f; l[o + kRowStride] = ok ? v[k].y : 0.f; l[o + 2 * kRowStride] = ok ? v[k].z : 0.f; l[o + 3 * kRowStride] = ok ? v[k].w : 0.f;
        if (NVP_SPLIT_H2 && ok) m = max(m, __float_as_uint(absmax_f4(0.f, v[k])));
    }
    return m;
}

__global__ __launch_bounds__(256, 2) void mlp_dw_pair_kernel(DwArgs A, float* __restrict__ partials, int64_t n, int64_t ntiles, int tiles_per_chunk, int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [A tile | B tile | BB tile] | maxima
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    int chunk, job;
    {
        const int L = blockIdx.x, nj = A.n_jobs;
        if (NVP_DW_XCD && (n_chunks & 7) == 0) {
            const int xcd = L & 7, slot = L >> 3;
            chunk = (slot / nj) * 8 + xcd;
            job = slot - (slot / nj) * nj;
        } else {
            chunk = L / nj;
            job = L - chunk * nj;
        }
    }
    const int64_t t0 = (int64_t)chunk * tiles_per_chunk;
    const int64_t t1 = min(ntiles, t0 + tiles_per_chunk);
    float* part = partials + (int64_t)chunk * A.total;
    const DwJob J = A.job[job];
    const int wr = w >> 1, wc = w & 1;
    const int nv = min(1024, ((J.b_rows - J.b_row0) >> 2) * 32), nvv = min(1024, ((J.bb_rows - J.bb_row0) >> 2) * 32);
    float* la = lds;
    float* lb[2] = {lds + kTileFloats, lds + 2 * kTileFloats};
    unsigned* mx = reinterpret_cast<unsigned*>(lds + 3 * kTileFloats);      // [2 slot sets][A | B | BB][4 waves]

    f32x16 acc[2][2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[g][r][c] = nvp_zero16();
    float bsum0 = 0.f, bsum1 = 0.f;
    const bool want_bias = (J.bias_off >= 0) && (wc == 0);

    // a staging wave publishes its share of a tile's maxima only when one of them exceeds the running maximum (publish_max)
    auto publish = [&](int par, unsigned ma, unsigned mb, unsigned mbb, unsigned ra, unsigned rb, unsigned rbb) {
#if NVP_SPLIT_H2
        if (!__any(ma > ra || mb > rb || mbb > rbb)) return;
        ma = wave_umax(ma); mb = wave_umax(mb); mbb = wave_umax(mbb);
        if (lane == 0) { mx[(par * 3 + 0) * 4 + w] = ma; mx[(par * 3 + 1) * 4 + w] = mb; mx[(par * 3 + 2) * 4 + w] = mbb; }
#endif
    };

    PairStage st;
    unsigned wma = 0u, wmb = 0u, wmbb = 0u;
    if (t0 < t1) {
        pair_load(st, J, t0, tid);
        wma = pair_write_tile(la, st.a, 1024, tid);
        wmb = pair_write_tile(lb[0], st.b, nv, tid);
        wmbb = pair_write_tile(lb[1], st.bb, nvv, tid);
        if (tid < 24) mx[tid] = 0u;
        __syncthreads();
        publish(0, wma, wmb, wmbb, 0u, 0u, 0u);
    }
    unsigned runA = __float_as_uint(kTinyMax), runB[2] = {runA, runA};
    PxScale qa = px_scale(kTinyMax), qb[2] = {qa, qa};
    float curS[2] = {qa.s * qa.s, qa.s * qa.s}, curU[2] = {qa.u * qa.u, qa.u * qa.u};
    int par = 0;
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) pair_load(st, J, t + 1, tid);
        if (NVP_SPLIT_H2) {
            const unsigned* mr = mx + par * 12;
            unsigned mA = 0u, mB = 0u, mC = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) { mA = max(mA, mr[u]); mB = max(mB, mr[4 + u]); mC = max(mC, mr[8 + u]); }
            runA = max(runA, (unsigned)__builtin_amdgcn_readfirstlane((int)mA));
            runB[0] = max(runB[0], (unsigned)__builtin_amdgcn_readfirstlane((int)mB));
            runB[1] = max(runB[1], (unsigned)__builtin_amdgcn_readfirstlane((int)mC));
            qa = px_scale(__uint_as_float(runA));
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                qb[g] = px_scale(__uint_as_float(runB[g]));
                const float S = qa.s * qb[g].s;
                if (S != curS[g]) {                  // wave-uniform: a tile raised a running maximum
                    const float ratio = S * curU[g];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[g][r][c] *= ratio;
                    curS[g] = S; curU[g] = qa.u * qb[g].u;
                }
            }
        }
        {
            // the A fragments of both row tiles, split once for both outputs
            BOp pa[2][2];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                float fa[16];
                read_frag(fa, la, 64 * wr + 32 * r2 + i, h);
                if (want_bias) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { if (r2 == 0) bsum0 += fa[k]; else bsum1 += fa[k]; }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float x[8] = {fa[8 * s2], fa[8 * s2 + 1], fa[8 * s2 + 2], fa[8 * s2 + 3], fa[8 * s2 + 4], fa[8 * s2 + 5], fa[8 * s2 + 6], fa[8 * s2 + 7]};
                    split8(x, qa.s, pa[r2][s2]);
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float fb[16];
                    read_frag(fb, lb[g], 64 * wc + 32 * c + i, h);
                    BOp pb[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const float x[8] = {fb[8 * s2], fb[8 * s2 + 1], fb[8 * s2 + 2], fb[8 * s2 + 3], fb[8 * s2 + 4], fb[8 * s2 + 5], fb[8 * s2 + 6], fb[8 * s2 + 7]};
                        split8(x, qb[g].s, pb[s2]);
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) mac_parts(acc[g][r2][c], pa[r2][s2].p, pb[s2]);
                }
        }
        __syncthreads();                            // everyone finished reading the single buffer
        if (more) {
            wma = pair_write_tile(la, st.a, 1024, tid);
            wmb = pair_write_tile(lb[0], st.b, nv, tid);
            wmbb = pair_write_tile(lb[1], st.bb, nvv, tid);
            publish(par ^ 1, wma, wmb, wmbb, runA, runB[0], runB[1]);
        }
        par ^= 1;
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int ncols = g ? J.nn_cols : J.n_cols;
        const int64_t woff = g ? J.ww_off : J.w_off;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = 64 * wc + 32 * c + i;
            if (col < ncols) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 64 * wr + 32 * r2 + nvp_frag_row(r, h);
                        part[woff + (int64_t)row * J.ld + col] = NVP_SPLIT_H2 ? acc[g][r2][c][r] * curU[g] : acc[g][r2][c][r];
                    }
            }
        }
    }
    if (want_bias) {
        bsum0 += __shfl_xor(bsum0, 32);
        bsum1 += __shfl_xor(bsum1, 32);
        if (h == 0) {
            part[J.bias_off + 64 * wr + i] = bsum0;
            part[J.bias_off + 64 * wr + 32 + i] = bsum1;
        }
    }
}

// ---- grouped jobs: every operand stream of a (modulator layer k, SIREN layer k) pair is staged ONCE per workgroup -----------------
// The seven GEMM jobs share operands: dp_k feeds dp_k x h_{k-1} and dp_k x z, h_{k-1} feeds dp_k x h_{k-1} and (as x_{k-1} = sin(.) h_{k-1})
// dq_k x x_{k-1}.  As separate 256-thread jobs every stream is fetched once per job that uses it (7.5 KB per pixel requested
// against 4.56 KB of distinct streams; PMC: 8.8 GB per step), and the stage is bound by exactly those bytes.  Here ONE 768-thread
// workgroup per pixel chunk owns the three jobs of layer k (k = 1, 2):
//     waves 0-3: dp_k x h_{k-1}      waves 4-7: dp_k x z      waves 8-11: dq_k x x_{k-1}
// and stages the four streams they need (dp_k, dq_k, h_{k-1} [+ q_{k-1}, steps], z) once into five LDS tiles (the x tile is built
// from the h float4 a thread holds anyway): 2.0 / 2.5 KB per pixel for k = 1 / 2.  Each wave owns a 64 x 64 sub-block exactly as
// in mlp_dw_kernel - same fragments, same MFMA order, same running block scales (the scale of an operand depends on the tiles
// seen so far, not on who staged them) - so the results are BIT-identical to the per-job kernel's.  dp_0 x z stays a plain job.
// Twelve waves share one 90 KiB single-buffered tile set (one workgroup per CU, three waves per SIMD like the plain jobs).
struct DwGroupArgs {
    const float* a1;          // dp_k   (PTM4, 128 rows)
    const float* a2;          // dq_k
    const float* h;           // h_{k-1}
    const float* q;           // q_{k-1} (K == 2)
    const float* z;           // latent (PTM4, z_rows rows, d valid)
    const float* steps;
    const float* sir0_wp;     // SIREN layer 0 weight / bias (K == 1 rebuilds x_0 from them)
    const float* sir0_bp;
    int z_rows, d;
    int ld_mod;               // 128 + d
    int64_t w_h, w_z, w_x;    // offsets inside a partial: W_mod[k][:, 0], W_mod[k][:, 128], W_sir[k]
    int64_t b_mod, b_sir;
    int64_t total;
};

constexpr int kGroupThreads = 768;
constexpr int kGroupItems = 6;                 // float4 items per thread and tile: 4 streams x 1024 / 768, rounded up

template <int K>
struct GroupStage {
    float4 v[kGroupItems];
    float4 q[K == 2 ? 2 : 1];                  // q_{k-1} of this thread's (at most two) h items
};

// item e = 768 it + tid of a tile: stream e >> 10 (wave-uniform: the boundaries are multiples of 256), float4 e & 1023 of that stream's tile
template <int K>
__device__ __forceinline__ void group_load(GroupStage<K>& s, const DwGroupArgs& A, int64_t t, int tid, int nvalid) {
#pragma unroll
    for (int it = 0; it < kGroupItems; ++it) {
        const int e = kGroupThreads * it + tid;
        const int sid = e >> 10, f = e & 1023;
        // the streams an iteration can meet are known at compile time (boundaries at multiples of 256): it 0: dp; 1: dp, dq;
        // 2: dq, h; 3: h; 4: z; 5: z (threads 0-255 only)
        if (it <= 1 && sid == 0) s.v[it] = reinterpret_cast<const float4*>(A.a1)[t * 1024 + f];
        if ((it == 1 || it == 2) && sid == 1) s.v[it] = reinterpret_cast<const float4*>(A.a2)[t * 1024 + f];
        if ((it == 2 || it == 3) && sid == 2) {
            s.v[it] = reinterpret_cast<const float4*>(A.h)[t * 1024 + f];
            if (K == 2) s.q[it == 3 ? 1 : 0] = reinterpret_cast<const float4*>(A.q)[t * 1024 + f];
        }
        if (it >= 4 && sid == 3) s.v[it] = reinterpret_cast<const float4*>(A.z)[t * (A.z_rows >> 2) * 32 + min(f, nvalid - 1)];     // zeroed past the end when written
    }
}

// tiles in LDS: [dp | dq | h | x | z]; mx[5]: running maxima (bit patterns) of the five tiles, raised with ds_max_u32
template <int K>
__device__ __forceinline__ void group_write(float* __restrict__ lds, unsigned* __restrict__ mx, const unsigned (&run)[5], const GroupStage<K>& s,
                                            const DwGroupArgs& A, const float* __restrict__ tab, int64_t t, int64_t n, int tid, int lane, int nvalid) {
    float sp = 0.f;
    if (K == 1) sp = A.steps[min(t * 32 + (tid & 31), n - 1)];       // px = f & 31 = tid & 31 (768 and 1024 are multiples of 32)
#pragma unroll
    for (int it = 0; it < kGroupItems; ++it) {
        const int e = kGroupThreads * it + tid;
        const int sid = e >> 10, f = e & 1023;
        if (sid > 3) continue;                      // it == 5, threads 256..767: nothing left
        const int o = (4 * (f >> 5)) * kRowStride + (f & 31);
        const float4 v = s.v[it];
        unsigned m = 0u, m2 = 0u;
        if ((it == 2 || it == 3) && sid == 2) {
            float* lh = lds + 2 * kTileFloats;
            float* lx = lds + 3 * kTileFloats;
            lh[o] = v.x; lh[o + kRowStride] = v.y; lh[o + 2 * kRowStride] = v.z; lh[o + 3 * kRowStride] = v.w;
            float xv[4];
            const float hv[4] = {v.x, v.y, v.z, v.w};
            if (K == 2) {                            // x_1 = sin(q_1) * h_1            (modulation.py:88-90)
                const float4 qq = s.q[it == 3 ? 1 : 0];
                const float qv[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[c] = nvp_sin(qv[c]) * hv[c];
            } else {                                 // x_0 = sin(30 (w s + c)) * h_0
                const int row = 4 * (f >> 5);
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[c] = nvp_sin(30.0f * __fmaf_rn(sp, tab[row + c], tab[NVP_H + row + c])) * hv[c];
            }
            lx[o] = xv[0]; lx[o + kRowStride] = xv[1]; lx[o + 2 * kRowStride] = xv[2]; lx[o + 3 * kRowStride] = xv[3];
            if (NVP_SPLIT_H2) { m = __float_as_uint(absmax_f4(0.f, v)); m2 = __float_as_uint(absmax_f4(0.f, make_float4(xv[0], xv[1], xv[2], xv[3]))); }
        } else {
            float* l = lds + (sid == 3 ? 4 : sid) * kTileFloats;
            const bool ok = sid != 3 || f < nvalid;
            l[o] = ok ? v.x : 0.f; l[o + kRowStride] = ok ? v.y : 0.f; l[o + 2 * kRowStride] = ok ? v.z : 0.f; l[o + 3 * kRowStride] = ok ? v.w : 0.f;
            if (NVP_SPLIT_H2 && ok) m = __float_as_uint(absmax_f4(0.f, v));
        }
#if NVP_SPLIT_H2
        // the tile maxima only matter when they raise a running maximum: a wave-uniform test, then one LDS atomic per wave
        const int slot = sid == 3 ? 4 : sid;
        if (__any(m > run[slot])) { const unsigned w = wave_umax(m); if (lane == 0) atomicMax(&mx[slot], w); }
        if (sid == 2 && __any(m2 > run[3])) { const unsigned w = wave_umax(m2); if (lane == 0) atomicMax(&mx[3], w); }
#endif
        __builtin_amdgcn_sched_barrier(0);           // one item at a time: keeps the sine temporaries from piling up
    }
}

template <int K>
__global__ __launch_bounds__(kGroupThreads, 1) void mlp_dw_group_kernel(DwGroupArgs A, float* __restrict__ partials, int64_t n, int64_t ntiles, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [dp | dq | h | x | z] tiles, 5 running maxima, (K == 1) SIREN-0 table
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);           // provably wave-uniform
    const int job = wall >> 2, w = wall & 3;                             // 0: dp x h, 1: dp x z, 2: dq x x
    const int i = lane & 31, h = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const int chunk = blockIdx.x;
    const int64_t t0 = (int64_t)chunk * tiles_per_chunk;
    const int64_t t1 = min(ntiles, t0 + tiles_per_chunk);
    float* part = partials + (int64_t)chunk * A.total;
    unsigned* mx = reinterpret_cast<unsigned*>(lds + 5 * kTileFloats);
    float* tab = lds + 5 * kTileFloats + 8;
    const int nvalid = min(1024, (A.z_rows >> 2) * 32);
    if (tid < 8) mx[tid] = __float_as_uint(kTinyMax);
    if (K == 1 && tid < NVP_H) { tab[tid] = A.sir0_wp[tid]; tab[NVP_H + tid] = A.sir0_bp[tid]; }
    __syncthreads();

    const int ta = job == 2 ? 1 : 0;                                     // A tile of this wave's job
    const int tb = job == 0 ? 2 : (job == 1 ? 4 : 3);                    // B tile
    const float* la = lds + ta * kTileFloats;
    const float* lb = lds + tb * kTileFloats;

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = nvp_zero16();
    float bsum0 = 0.f, bsum1 = 0.f;
    const bool want_bias = job != 1 && wc == 0;

    GroupStage<K> st;
    unsigned run[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) run[u] = __float_as_uint(kTinyMax);
    if (t0 < t1) {
        group_load<K>(st, A, t0, tid, nvalid);
        group_write<K>(lds, mx, run, st, A, tab, t0, n, tid, lane, nvalid);
    }
    PxScale qa = px_scale(kTinyMax), qb = qa;
    float curS = qa.s * qb.s, curU = qa.u * qb.u;
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) group_load<K>(st, A, t + 1, tid, nvalid);
#if NVP_DW_B3
        {
            if (NVP_SPLIT_H2) {
#pragma unroll
                for (int u = 0; u < 5; ++u) run[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)mx[u]);
                qa = px_scale(__uint_as_float(run[ta])); qb = px_scale(__uint_as_float(run[tb]));
                const float S = qa.s * qb.s;
                if (S != curS) {                     // wave-uniform: a tile raised a running maximum
                    const float ratio = S * curU;   // <= 1, a power of two
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[r][c] *= ratio;
                    curS = S; curU = qa.u * qb.u;
                }
            }
            float fa[16], fb[2][16];
            read_frag(fa, la, 64 * wr + i, h);
#pragma unroll
            for (int c = 0; c < 2; ++c) read_frag(fb[c], lb, 64 * wc + 32 * c + i, h);
            BOp pb[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float x[8] = {fb[c][8 * s2], fb[c][8 * s2 + 1], fb[c][8 * s2 + 2], fb[c][8 * s2 + 3],
                                        fb[c][8 * s2 + 4], fb[c][8 * s2 + 5], fb[c][8 * s2 + 6], fb[c][8 * s2 + 7]};
                    split8(x, qb.s, pb[c][s2]);
                }
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                if (r2 == 1) read_frag(fa, la, 64 * wr + 32 + i, h);
                if (want_bias) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { if (r2 == 0) bsum0 += fa[k]; else bsum1 += fa[k]; }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float x[8] = {fa[8 * s2], fa[8 * s2 + 1], fa[8 * s2 + 2], fa[8 * s2 + 3], fa[8 * s2 + 4], fa[8 * s2 + 5], fa[8 * s2 + 6], fa[8 * s2 + 7]};
                    BOp pa;
                    split8(x, qa.s, pa);
#pragma unroll
                    for (int c = 0; c < 2; ++c) mac_parts(acc[r2][c], pa.p, pb[c][s2]);
                }
            }
        }
#else
        {
            float fa[16], fb[2][16];
            read_frag(fa, la, 64 * wr + i, h);
#pragma unroll
            for (int c = 0; c < 2; ++c) read_frag(fb[c], lb, 64 * wc + 32 * c + i, h);
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bsum0 += fa[k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[0][0] = nvp_mfma(fa[k], fb[0][k], acc[0][0]);
                acc[0][1] = nvp_mfma(fa[k], fb[1][k], acc[0][1]);
            }
            read_frag(fa, la, 64 * wr + 32 + i, h);
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bsum1 += fa[k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[1][0] = nvp_mfma(fa[k], fb[0][k], acc[1][0]);
                acc[1][1] = nvp_mfma(fa[k], fb[1][k], acc[1][1]);
            }
        }
#endif
        __syncthreads();                            // everyone finished reading the tiles
        if (more) group_write<K>(lds, mx, run, st, A, tab, t + 1, n, tid, lane, nvalid);
        __syncthreads();
    }

    {
        const int ncols = job == 1 ? A.d : NVP_H;
        const int64_t woff = job == 0 ? A.w_h : (job == 1 ? A.w_z : A.w_x);
        const int ld = job == 2 ? NVP_H : A.ld_mod;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = 64 * wc + 32 * c + i;
            if (col < ncols) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 64 * wr + 32 * r2 + nvp_frag_row(r, h);
                        part[woff + (int64_t)row * ld + col] = (NVP_DW_B3 && NVP_SPLIT_H2) ? acc[r2][c][r] * curU : acc[r2][c][r];
                    }
            }
        }
        if (want_bias) {
            bsum0 += __shfl_xor(bsum0, 32);
            bsum1 += __shfl_xor(bsum1, 32);
            if (h == 0) {
                const int64_t boff = job == 0 ? A.b_mod : A.b_sir;
                part[boff + 64 * wr + i] = bsum0;
                part[boff + 64 * wr + 32 + i] = bsum1;
            }
        }
    }
}

#endif  // NVP_EXPERIMENTS

// Sum the per-tile small-gradient records of one pixel chunk into that chunk's partial (the chain kernel
// wrote one kRecFloats record per 32-pixel tile into stream 3 of `dy`).  Thread = one record element:
// consecutive threads read consecutive floats of a record, tiles are visited in order -> deterministic.
__global__ __launch_bounds__(256) void dw_records_kernel(const float* __restrict__ records, float* __restrict__ partials,
                                                          int64_t ntiles, int tiles_per_chunk, int64_t total,
                                                          int64_t p_last_w, int64_t p_last_b, int64_t p_sir0_w, int64_t p_sir0_b) {
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= kRecFloats || e == kRecLastB + 3) return;
    const int chunk = blockIdx.x;
    const int64_t t0 = (int64_t)chunk * tiles_per_chunk;
    const int64_t t1 = min(ntiles, t0 + tiles_per_chunk);
    const float* src = records + e;
    float s = 0.f;
    int64_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(t + u) * (int64_t)(NVP_H * 32)];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; t < t1; ++t) s += src[t * (int64_t)(NVP_H * 32)];
    int64_t idx;
    if (e < kRecLastB) idx = p_last_w + e;
    else if (e < kRecSir0W) idx = p_last_b + (e - kRecLastB);
    else if (e < kRecSir0B) idx = p_sir0_w + (e - kRecSir0W);
    else idx = p_sir0_b + (e - kRecSir0B);
    partials[(int64_t)chunk * total + idx] = s;
}

struct ReduceArgs {
    float* dst[14];
    int64_t off[15];
    int nch[14];          // pixel chunks whose partials hold this tensor (the plain and the transform jobs may be chunked differently)
};

// Element idx of every chunk's partial, summed in chunk order (fixed order -> deterministic).
// 16 independent loads are kept in flight per thread; the additions stay sequential.
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ partials, ReduceArgs R, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int tt = 0;
    while (idx >= R.off[tt + 1]) ++tt;
    const int n_chunks = R.nch[tt];
    const float* src = partials + idx;
    float s = 0.f;
    int c = 0;
    for (; c + 16 <= n_chunks; c += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(int64_t)(c + u) * total];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; c < n_chunks; ++c) s += src[(int64_t)c * total];
    R.dst[tt][idx - R.off[tt]] = s;
}

}  // namespace

#if NVP_EXPERIMENTS
int nvp_mlp_dw_glds_launch(const float* steps, const float* zt, const float* saved, const float* dy, const nvp_mlp_params* p,
                           float* partials, int32_t n_chunks, int64_t n, int32_t d, void* stream);        // mlp_dw_glds.hip
#endif

extern "C" int nvp_mlp_bwd_dw(const float* drgb, const float* steps, const float* zt, const float* saved,
                              const float* dy, const nvp_mlp_params* p, float* partials, int32_t n_chunks,
                              const nvp_mlp_grads* g, int64_t n, int32_t d, void* stream) {
    if (!drgb || !steps || !zt || !saved || !dy || !p || !partials || !g || n < 0 || d < 1 || n_chunks < 1) return NVP_ERR_BADARG;
    const NvpParamLayout P = nvp_param_layout(d);
    const int64_t ntiles = nvp_ntiles(n);
    const int rows = nvp_rows4(d);
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    if (rows > 256) return NVP_ERR_UNSUPPORTED;

    // NVP_DW_MERGE=1 (environment, read once): the two 128-column blocks [h_{k-1} | z] of modulator layers 1 and 2 become ONE
    // 512-thread job each, so dp_1 / dp_2 are staged once instead of twice (-14 % operand bytes).  Bit-identical; measured
    // SLOWER in the step (2.225 vs 2.152 ms): one 8-wave workgroup with 108 KiB of LDS per CU loses more than the saved reads
    // (which the XCD-local L2 partly served anyway) give back.  OFF by default; kept as the A/B evidence.
#if NVP_EXPERIMENTS
    static const bool merge_on = [] { const char* e = getenv("NVP_DW_MERGE"); return e && e[0] == '1'; }();
#else
    constexpr bool merge_on = false;              // (the switches below exist in the experiments build only)
#endif
    const bool merge = merge_on && d <= 128;          // one latent column block: [h ; z] = exactly two B tiles
    // NVP_DW_PAIR (environment, read once): dp_k x h_{k-1} and dp_k x z (wide latents: also the two column blocks of dp_0 x z) as ONE
    // 256-thread workgroup that stages dp_k once (mlp_dw_pair_kernel); bit-identical to the separate jobs
#if NVP_EXPERIMENTS
    static const bool pair_on = [] { const char* e = getenv("NVP_DW_PAIR"); return e ? e[0] == '1' : (NVP_DW_PAIR_DEFAULT != 0); }();
    const bool pair = pair_on && NVP_DW_B3 && !merge && !getenv_on("NVP_DW_GLDS") && !getenv_on("NVP_DW_GROUP") && !getenv_on("NVP_DW_ONE_LAUNCH");
#else
    constexpr bool pair = false;
#endif
    DwArgs P0, P2, P1, PP;                            // plain single-tile jobs, merged two-tile jobs, transform jobs, paired jobs
    int n0 = 0, n2 = 0, n1 = 0, np = 0;
    auto plain = [&](DwJob& J, int k, const float* b, int b_rows, int b_row0, int n_cols, int64_t w_off, int ld, int64_t bias_off) {
        J.a = dy + (int64_t)k * act; J.b = b; J.b2 = b; J.mode = 0; J.b_rows = b_rows; J.b_row0 = b_row0;
        J.n_cols = n_cols; J.w_off = w_off; J.ld = ld; J.bias_off = bias_off;
        J.bb = b; J.bb_rows = b_rows; J.bb_row0 = b_row0; J.nn_cols = 0; J.ww_off = w_off;
    };
    // modulator layers: A = dp_k; B = [h_{k-1} ; z]
    for (int k = 0; k < 3; ++k) {
        const int ld = (k == 0) ? d : NVP_H + d;
        if (k > 0 && merge) {
            // one job: dp_k x [h_{k-1} | z]: the dp_k tile is staged ONCE for both 128-column halves
            DwJob& J = P2.job[n2++];
            plain(J, k, saved + (int64_t)(k - 1) * act, NVP_H, 0, NVP_H, P.mod_w[k], ld, P.mod_b[k]);
            J.bb = zt; J.bb_rows = rows; J.bb_row0 = 0; J.nn_cols = d; J.ww_off = P.mod_w[k] + NVP_H;
            continue;
        }
        bool bias_done = false;
        int c_first = 0;
        if (pair && (k > 0 || d > 128)) {
            // A = dp_k staged once; B = h_{k-1} (k > 0) or the first column block of z (k = 0), BB = the next column block of z
            DwJob& J = PP.job[np++];
            if (k > 0) {
                plain(J, k, saved + (int64_t)(k - 1) * act, NVP_H, 0, NVP_H, P.mod_w[k], ld, P.mod_b[k]);
                J.bb = zt; J.bb_rows = rows; J.bb_row0 = 0; J.nn_cols = d < 128 ? d : 128; J.ww_off = P.mod_w[k] + NVP_H;
                c_first = 128;
            } else {
                plain(J, k, zt, rows, 0, 128, P.mod_w[k], ld, P.mod_b[k]);
                J.bb = zt; J.bb_rows = rows; J.bb_row0 = 128; J.nn_cols = (d - 128 < 128) ? d - 128 : 128; J.ww_off = P.mod_w[k] + 128;
                c_first = 256;
            }
            bias_done = true;
        } else if (k > 0) {
            plain(P0.job[n0++], k, saved + (int64_t)(k - 1) * act, NVP_H, 0, NVP_H, P.mod_w[k], ld, P.mod_b[k]);
            bias_done = true;
        }
        for (int c0 = c_first; c0 < d; c0 += 128) {
            plain(P0.job[n0++], k, zt, rows, c0, (d - c0 < 128) ? d - c0 : 128, P.mod_w[k] + (k == 0 ? 0 : NVP_H) + c0, ld,
                  bias_done ? -1 : P.mod_b[k]);
            bias_done = true;
        }
    }
    // SIREN layers 1, 2: A = dq_k; B = x_{k-1} = sin(q_{k-1}) h_{k-1}, rebuilt from the saved h (and q) streams
    for (int k = 1; k <= 2; ++k) {
        DwJob& J = P1.job[n1++];
        plain(J, 3 + k, saved + (int64_t)(k - 1) * act, NVP_H, 0, NVP_H, P.sir_w[k], NVP_H, P.sir_b[k]);
        if (k == 1) { J.mode = 2; } else { J.b2 = saved + 3 * act; J.mode = 1; }
    }
    // NVP_DW_ONE_LAUNCH=1 (environment, read once): all jobs through the transform-capable kernel in ONE launch, so that the jobs
    // of a pixel chunk sit on one XCD together and h_0 / h_1 (read by a modulator job AND by a SIREN job) could be fetched from
    // HBM once instead of once per launch (PMC: 8.0 GB per step against 5.7 GB of distinct operand streams).  Bit-identical;
    // measured SLOWER (1.87 vs 1.71 ms): the plain jobs run on the heavier instantiation.  OFF by default.
#if NVP_EXPERIMENTS
    static const bool one_launch = [] { const char* e = getenv("NVP_DW_ONE_LAUNCH"); return e && e[0] == '1'; }();
#else
    constexpr bool one_launch = false;
#endif
    // NVP_DW_GROUP=1 (environment, read once; default 0): layers 1 and 2 run as register-staged GROUPED workgroups that stage every operand stream once
    // (mlp_dw_group_kernel); only dp_0 x z stays a plain job.  0: the seven per-job workgroups (bit-identical results).
#if NVP_EXPERIMENTS
    static const bool group_on = [] { const char* e = getenv("NVP_DW_GROUP"); return e && e[0] == '1'; }();      // register-staged grouping: measured slower, off
#else
    constexpr bool group_on = false;
#endif
    // NVP_DW_GLDS=1 (environment, read once; default 0): layers 1 and 2 as grouped workgroups fed by LDS DMA, every operand stream
    // read once and split once (mlp_dw_glds.hip).  Correct (same tolerances; not bit-identical: other summation order) and
    // MEASURED SLOWER on MI355X - 0.99 + 0.92 ms for the two launches against 1.30 ms for the six jobs they replace: the DMA
    // skeleton alone streams at 6 TB/s (0.5 ms per launch) and the MFMA phase hides under it, but the staging pass between them
    // (three barriers and an LDS latency chain per 16-pixel step) adds 0.4 ms per launch.  Kept as the measured alternative.
#if NVP_EXPERIMENTS
    static const bool glds_on = [] { const char* e = getenv("NVP_DW_GLDS"); return e && e[0] == '1'; }();
#else
    constexpr bool glds_on = false;
#endif
    const bool glds = glds_on && NVP_DW_B3 && NVP_SPLIT_H2 && d <= 128 && (n & 3) == 0 && !merge && !one_launch;
    const bool group = !glds && group_on && d <= 128 && !merge && !one_launch;
    if (one_launch && n0 + n1 <= 12) {
        for (int j = 0; j < n0; ++j) P1.job[n1++] = P0.job[j];
        n0 = 0;
    }
    for (DwArgs* Q : {&P0, &P2, &P1, &PP}) { Q->steps = steps; Q->sir0_wp = p->sir_w[0]; Q->sir0_bp = p->sir_b[0]; Q->total = P.total; }
    P0.n_jobs = n0; P2.n_jobs = n2; P1.n_jobs = n1; PP.n_jobs = np;

    const int tiles_per_chunk = (int)((ntiles + n_chunks - 1) / n_chunks);
    // The transform jobs (SIREN layers 1, 2: two workgroups per CU, two jobs) fill the chip with 256 chunks exactly; the five plain
    // jobs (three workgroups per CU) want 5 n_chunks close below a multiple of 768 (304: 1520 of 1536 slots in two rounds, instead
    // of 1280 = one round and two thirds).  NVP_DW_CHUNKS_XF (environment, read once) caps the transform jobs' chunk count.
#if NVP_EXPERIMENTS
    static const int xf_cap = [] { const char* e = getenv("NVP_DW_CHUNKS_XF"); return e ? atoi(e) : 256; }();
#else
    constexpr int xf_cap = 256;
#endif
    const int nch1 = (xf_cap > 0 && xf_cap < n_chunks) ? xf_cap : n_chunks;
    const int tiles_per_chunk1 = (int)((ntiles + nch1 - 1) / nch1);
#if NVP_EXPERIMENTS
    if (glds) {
        // dp_0 x z as the one remaining plain job, then the two DMA-fed grouped launches
        DwArgs Q0 = P0;
        Q0.n_jobs = 1;                              // job 0 of P0 is (k = 0, c0 = 0): dp_0 x z with the bias
        const size_t lds0 = ((NVP_DW_BUFS == 1 ? 1 : 2) * 2 * kTileFloats + 4 * kMxW) * sizeof(float);
        hipLaunchKernelGGL((mlp_dw_kernel<0, 1>), dim3(n_chunks), dim3(256), lds0, (hipStream_t)stream, Q0, partials, n, ntiles, tiles_per_chunk, n_chunks);
        NVP_LAUNCH_CHECK();
        const int rc = nvp_mlp_dw_glds_launch(steps, zt, saved, dy, p, partials, n_chunks, n, d, stream);
        if (rc) return rc;
        n0 = n2 = n1 = 0;
    }
    if (group) {
        // dp_0 x z as the one remaining plain job, then the two grouped launches
        DwArgs Q0 = P0;
        Q0.n_jobs = 1;                              // job 0 of P0 is (k = 0, c0 = 0): dp_0 x z with the bias
        const size_t lds0 = ((NVP_DW_BUFS == 1 ? 1 : 2) * 2 * kTileFloats + 4 * kMxW) * sizeof(float);
        hipLaunchKernelGGL((mlp_dw_kernel<0, 1>), dim3(n_chunks), dim3(256), lds0, (hipStream_t)stream, Q0, partials, n, ntiles, tiles_per_chunk, n_chunks);
        NVP_LAUNCH_CHECK();
        for (int k = 1; k <= 2; ++k) {
            DwGroupArgs G;
            G.a1 = dy + (int64_t)k * act; G.a2 = dy + (int64_t)(3 + k) * act;
            G.h = saved + (int64_t)(k - 1) * act; G.q = saved + 3 * act;
            G.z = zt; G.z_rows = rows; G.d = d; G.steps = steps; G.sir0_wp = p->sir_w[0]; G.sir0_bp = p->sir_b[0];
            G.ld_mod = NVP_H + d; G.w_h = P.mod_w[k]; G.w_z = P.mod_w[k] + NVP_H; G.w_x = P.sir_w[k];
            G.b_mod = P.mod_b[k]; G.b_sir = P.sir_b[k]; G.total = P.total;
            const size_t ldsg = (5 * kTileFloats + 8 + 2 * NVP_H) * sizeof(float);
            if (k == 1) hipLaunchKernelGGL((mlp_dw_group_kernel<1>), dim3(n_chunks), dim3(kGroupThreads), ldsg, (hipStream_t)stream, G, partials, n, ntiles, tiles_per_chunk);
            else hipLaunchKernelGGL((mlp_dw_group_kernel<2>), dim3(n_chunks), dim3(kGroupThreads), ldsg, (hipStream_t)stream, G, partials, n, ntiles, tiles_per_chunk);
            NVP_LAUNCH_CHECK();
        }
        n0 = n2 = n1 = 0;                           // nothing left for the per-job launches
    }
#else
    (void)glds; (void)group;
#endif
    // GEMM launches: plain jobs, merged jobs (512 threads, three LDS tiles per buffer), transform jobs; plus the record sums
    // tile buffers (the kernel places the tile maxima and the table behind them), 4 kMxW maxima
    const size_t lds_bytes = (2 * 2 * kTileFloats + 4 * kMxW + 2 * NVP_H) * sizeof(float);
    const size_t lds_bytes0 = ((NVP_DW_BUFS == 1 ? 1 : 2) * 2 * kTileFloats + 4 * kMxW) * sizeof(float);
    const size_t lds_bytes2 = (2 * 3 * kTileFloats + 4 * kMxW) * sizeof(float);          // 108 KiB: one 8-wave workgroup per CU
#if NVP_EXPERIMENTS
    if (np) {
        const size_t lds_pair = (3 * kTileFloats + 24) * sizeof(float);                     // 54 KiB: two workgroups per CU
        hipLaunchKernelGGL(mlp_dw_pair_kernel, dim3(n_chunks * np), dim3(256), lds_pair, (hipStream_t)stream, PP, partials, n, ntiles, tiles_per_chunk, n_chunks);
        NVP_LAUNCH_CHECK();
    }
#endif
    if (n0) hipLaunchKernelGGL((mlp_dw_kernel<0, 1>), dim3(n_chunks * n0), dim3(256), lds_bytes0, (hipStream_t)stream, P0, partials, n, ntiles, tiles_per_chunk, n_chunks);
    NVP_LAUNCH_CHECK();
#if NVP_EXPERIMENTS
    if (n2) hipLaunchKernelGGL((mlp_dw_kernel<0, 2>), dim3(n_chunks * n2), dim3(512), lds_bytes2, (hipStream_t)stream, P2, partials, n, ntiles, tiles_per_chunk, n_chunks);
    NVP_LAUNCH_CHECK();
#else
    (void)lds_bytes2; (void)np; (void)n2;
#endif
    const bool xf_own = n1 > 0 && !one_launch;       // (one launch: the plain jobs ride in P1 and share its chunking)
    if (n1) hipLaunchKernelGGL((mlp_dw_kernel<1, 1>), dim3((xf_own ? nch1 : n_chunks) * n1), dim3(256), lds_bytes, (hipStream_t)stream, P1, partials, n, ntiles,
                               xf_own ? tiles_per_chunk1 : tiles_per_chunk, xf_own ? nch1 : n_chunks);
    NVP_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_records_kernel, dim3(n_chunks, (kRecFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy + 3 * act, partials,
                       ntiles, tiles_per_chunk, P.total, P.last_w, P.last_b, P.sir_w[0], P.sir_b[0]);
    NVP_LAUNCH_CHECK();

    ReduceArgs R;
    int t = 0;
    for (int k = 0; k < 3; ++k) { R.dst[t] = g->mod_w[k]; R.off[t++] = P.mod_w[k]; R.dst[t] = g->mod_b[k]; R.off[t++] = P.mod_b[k]; }
    for (int k = 0; k < 3; ++k) { R.dst[t] = g->sir_w[k]; R.off[t++] = P.sir_w[k]; R.dst[t] = g->sir_b[k]; R.off[t++] = P.sir_b[k]; }
    R.dst[t] = g->last_w; R.off[t++] = P.last_w;
    R.dst[t] = g->last_b; R.off[t++] = P.last_b;
    R.off[t] = P.total;
    for (int u = 0; u < 14; ++u) R.nch[u] = n_chunks;
    if (xf_own) { R.nch[8] = R.nch[9] = R.nch[10] = R.nch[11] = nch1; }      // sir_w[1], sir_b[1], sir_w[2], sir_b[2]: the transform jobs' tensors
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, R, P.total);
    NVP_LAUNCH_CHECK();
    return 0;
}
