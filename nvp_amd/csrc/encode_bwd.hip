// Gradient scatter-add of the NVP encoders (R3 + R6) for gfx950, atomic-free on the dense
// planes.
//
// Measured on MI355X (tools/probes): global fp32 atomics sustain only ~21 Gop/s whatever the
// footprint or scope, LDS ds_add_f32 ~0.2 Top/s, but LDS *integer* atomics 3-8 Top/s.  The
// reference-shaped scatter (one fp32 atomic per corner x feature = 478 M atomics per step for
// nvp_s) therefore costs ~25 ms.  This file replaces it by a deterministic scheme:
//
//  1. sort the batch's pixels by the row coordinate of each plane (y for the xy and yt planes,
//     x for the xt plane) - two 1.2 M-key radix sorts (rocPRIM);
//  2. permute: coordinates and the per-level latent gradients are rewritten in sorted order,
//     level-major, so every later access is a coalesced stream; the same pass finds
//     max|dz|, which fixes the fixed-point scale;
//  3. row tables: for every level, the first sorted pixel of each grid row (binary searches);
//  4. band kernel: a workgroup owns rows [r0, r1) of one level of one plane, keeps them as an
//     int64 fixed-point table in LDS (36 KB by default) and visits exactly the sorted pixel range that
//     can touch those rows.  Bilinear corner weights x gradient are added with ds_add_u64:
//     integer addition is exact and order-independent, so the result is bit-reproducible.
//     Bands with few rows-per-pixel are additionally split over several workgroups
//     ("splits"); split tables go to slabs that step 5 sums.  Fine levels are exclusive and
//     are stored straight into the gradient.  Every gradient element is written exactly once
//     with plain coalesced stores - no atomics, no zero-fill pass needed.
//  5. slab reduction for the split levels.
//
// The sparse 3x3 grid uses the same scheme with key = t_idx * X + x_idx (see below).
//
// Quantisation: contributions are scaled by 2^k with k chosen from max|dz| so that the sum of
// N contributions cannot overflow 2^62; each contribution keeps >= 40 significant bits below
// max|dz| (fp32 carries 24), so the scatter is more accurate than an fp32 atomic chain.
#include <cstring>
#include "grid_math.h"
#include <rocprim/rocprim.hpp>

#pragma clang fp contract(off)

namespace {

#ifndef NVP_BAND_THREADS
#define NVP_BAND_THREADS 512
#endif
#ifndef NVP_BAND_ENTRIES
#define NVP_BAND_ENTRIES 4500
#endif
constexpr int kBandThreads = NVP_BAND_THREADS;
constexpr int kLdsEntries = NVP_BAND_ENTRIES;   // target int64 entries per table (36 KB: four workgroups per CU overlap their
                                                // zero / accumulate / flush phases; measured best of 4.5k..18k, tools/ablate_scatter.*);
                                                // grows to one full grid row when a row is wider (nvp_l: 1443 x 4)
constexpr int kMaxLdsEntries = 20000;           // 160 KB
constexpr int kTargetVisits = 16384;          // pixel visits per band workgroup
constexpr int kMaxSlots = 256;                // dzmax slots

struct LevelPlan {
    int first_block;     // first blockIdx of this (plane, level)
    int bands, splits;
    int rows;            // rows per band
    int rs_off;          // offset of this level's row-start table
    int pad;
    long long slab_off;  // float offset of the split slabs, -1 when exclusive (splits == 1)
};

struct Plan {
    LevelPlan lp[3][NVP_MAX_LEVELS];
    int total_blocks;
    int rs_total[3];         // row-start entries per plane
    long long slab_floats;   // total slab floats
    int reduce_items;        // number of (plane, level) pairs with splits > 1
    int entries;             // int64 entries of the LDS table (>= the widest grid row)
};

void make_plan(Plan& P, const nvp_levels* lv[3], int64_t n) {
    P.entries = kLdsEntries;
    for (int p = 0; p < 3; ++p)
        for (int l = 0; l < lv[p]->n_levels; ++l)
            if (lv[p]->res[l] * lv[p]->n_features > P.entries) P.entries = lv[p]->res[l] * lv[p]->n_features;
    int blocks = 0;
    long long slab = 0;
    P.reduce_items = 0;
    for (int p = 0; p < 3; ++p) {
        int rs = 0;
        const int F = lv[p]->n_features;
        for (int l = 0; l < lv[p]->n_levels; ++l) {
            LevelPlan& L = P.lp[p][l];
            const int res = lv[p]->res[l];
            int rows = P.entries / (res * F);
            if (rows < 1) rows = 1;
            if (rows > res) rows = res;
            L.rows = rows;
            L.bands = (res + rows - 1) / rows;
            // pixels visited by one band ~ n * (rows + 2) / res  (whole level: n)
            double visits = (L.bands == 1) ? (double)n : (double)n * (rows + 2) / res;
            int splits = (int)((visits + kTargetVisits - 1) / kTargetVisits);
            if (splits < 1) splits = 1;
            if (splits > 256) splits = 256;
            L.splits = splits;
            L.first_block = blocks;
            blocks += L.bands * splits;
            L.rs_off = rs;
            rs += res + 1;
            L.pad = 0;
            if (splits > 1) {
                L.slab_off = slab;
                slab += (long long)splits * res * res * F;
                ++P.reduce_items;
            } else {
                L.slab_off = -1;
            }
        }
        P.rs_total[p] = rs;
    }
    P.total_blocks = blocks;
    P.slab_floats = slab;
}

// ------------------------------------------------------------------------------------------
struct Ws {                      // workspace carve (byte offsets)
    size_t keys_in[2], keys_out[2], iota, order[2], cs[3], dzs[3], rowstart[3], dzmax, slabs, sort_tmp, total;
    size_t skey_in, skey_out, sorder, srowstart, sdzmax;      // sparse grid
    size_t sort_tmp_bytes;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int carve(Ws& W, const Plan& P, const nvp_levels* lv[3], const nvp_sparse_shape* sh, int64_t n) {
    size_t o = 0;
    for (int k = 0; k < 2; ++k) { W.keys_in[k] = o; o = align_up(o + n * 4); }
    for (int k = 0; k < 2; ++k) { W.keys_out[k] = o; o = align_up(o + n * 4); }
    W.iota = o; o = align_up(o + n * 4);
    for (int k = 0; k < 2; ++k) { W.order[k] = o; o = align_up(o + n * 4); }
    for (int p = 0; p < 3; ++p) { W.cs[p] = o; o = align_up(o + n * 8); }
    for (int p = 0; p < 3; ++p) { W.dzs[p] = o; o = align_up(o + (size_t)n * lv[p]->n_levels * lv[p]->n_features * 4); }
    for (int p = 0; p < 3; ++p) { W.rowstart[p] = o; o = align_up(o + (size_t)P.rs_total[p] * 4); }
    W.dzmax = o; o += kMaxSlots * 4;
    W.sdzmax = o; o = align_up(o + kMaxSlots * 4);             // adjacent to dzmax: nvp_encode_bwd_prepare zeroes both with one memset
    W.slabs = o; o = align_up(o + (size_t)P.slab_floats * 4);
    W.skey_in = o; o = align_up(o + n * 4);
    W.skey_out = o; o = align_up(o + n * 4);
    W.sorder = o; o = align_up(o + n * 4);
    W.srowstart = o; o = align_up(o + ((size_t)sh->t_res * sh->x_res + 1) * 4);
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    size_t tmp2 = 0;
    e = rocprim::radix_sort_pairs(nullptr, tmp2, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                  (int*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    if (tmp2 > tmp) tmp = tmp2;
    W.sort_tmp_bytes = tmp;
    W.sort_tmp = o; o = align_up(o + tmp);
    W.total = o;
    return 0;
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int row_of(float c1, float scale, int flags) { return (int)floorf(nvp_grid_pos(c1, scale, flags)); }

// cs_xy / cs_yt (optional): the (dim0, dim1) coordinate pairs of the xy and yt planes in batch order - what the permute pass
// would write for them when their sorted order is the identity (NVP_DZ_PLANES_READY).
// kx: the sort key of the xt plane.  The scatter needs the pixels ordered so that the grid ROW index is non-decreasing at EVERY
// level; ordering by x itself does that, but costs four 8-bit radix passes over the float's 32 bits.  key(x) = sum over the levels
// of row_l(x) is a non-decreasing step function that steps exactly where some level's row index steps, so equal keys mean equal
// rows at every level (ties may sit in any order: the fixed-point accumulation does not depend on it) - and it has ~14 bits: two
// passes.  Rows are clamped to [0, res + 1]; coordinates outside [0, 1] keep a valid (if arbitrary) order, as before.
__global__ __launch_bounds__(256) void keys_kernel(const float* __restrict__ coords, float* __restrict__ ky, unsigned* __restrict__ kx,
                                                   int* __restrict__ iota, float2* __restrict__ cs_xy, float2* __restrict__ cs_yt,
                                                   nvp_levels lvx, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = coords[i * 3], x = coords[i * 3 + 1], y = coords[i * 3 + 2];
    unsigned key = 0u;
    for (int l = 0; l < lvx.n_levels; ++l) key += (unsigned)min(max(row_of(x, lvx.scale[l], lvx.flags), 0), lvx.res[l] + 1);
    kx[i] = key;
    ky[i] = y;
    iota[i] = (int)i;
    if (cs_xy) { cs_xy[i] = make_float2(x, y); cs_yt[i] = make_float2(t, y); }
}

struct PermArgs {
    const int* order[3];      // sorted->pixel id, per plane
    float2* cs[3];
    float* dzs[3];
    int col0[3], nlev[3];
    int c0[3], c1[3];         // coordinate columns feeding (dim0, dim1) of each plane
    int scol0, scols;         // the sparse grid's latent-gradient columns (only their max|.| is taken here)
    unsigned* sdzmax;
    int plane0;               // first plane this launch handles (2 when the y-sorted planes arrive level-major already)
};

// A workgroup rewrites kPermPix<F> consecutive sorted positions of one plane.  A pixel's segment of this plane in the
// row-major latent gradient (16 F floats = 128 B for F = 2) is read by NV = 4 F consecutive lanes as ONE coalesced
// run (a thread-per-pixel loop strides 464 B between lanes and re-fetches every line several times: 1.6 GB of
// fetch for 0.5 GB of data), transposed through LDS (row stride 16 F + 1: conflict-free both ways), and written
// level-major with consecutive threads on consecutive pixels (2 KB runs per level).  The same pass takes max|dz|
// for the plane and, from the last plane's workgroups, over the sparse grid's columns.
template <int F> struct PermCfg {
    static constexpr int NV = 16 * F / 4;              // float4 per pixel segment = lanes per pixel
    static constexpr int PIX = F <= 2 ? 256 : (F == 4 ? 128 : 64);      // pixels per workgroup: LDS tile <= 34 KB
    static constexpr int STRIDE = 16 * F + 1;
};

template <int F>
__global__ __launch_bounds__(256) void permute_kernel(const float* __restrict__ coords, const float* __restrict__ dz, int dz_stride,
                                                      PermArgs A, unsigned* __restrict__ dzmax, int64_t n) {
    constexpr int NV = PermCfg<F>::NV, PIX = PermCfg<F>::PIX, STRIDE = PermCfg<F>::STRIDE;
    constexpr int PPP = 256 / NV;                      // pixels per pass
    extern __shared__ float tile[];                    // [PIX][STRIDE]
    const int plane = blockIdx.y + A.plane0;
    const int64_t p0 = (int64_t)blockIdx.x * PIX;
    const int t = threadIdx.x;
    const int nl = A.nlev[plane];
    const int qmax = (nl * F - 1) >> 2;
    const int* order = A.order[plane];
    const int q = t % NV, pp = t / NV;
    // max|dz| is taken on the BIT PATTERNS of |v| (unsigned compare): finite values order like floats, Inf and NaN
    // sort above every finite value, so a non-finite latent gradient reaches the band kernels, which then poison
    // their output with NaN (fmaxf would silently drop a NaN and to_fixed would turn it into a finite integer).
    unsigned ms = 0u;
    // ---- phase 1: coalesced segment reads -> LDS
#pragma unroll
    for (int i = 0; i < PIX / PPP; ++i) {
        const int pix = i * PPP + pp;
        const int64_t p = p0 + pix;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < n) {
            const int id = order[p];
            if (q <= qmax) v = reinterpret_cast<const float4*>(dz + (int64_t)id * dz_stride + A.col0[plane])[q];
            if (plane == 2 && 4 * q < A.scols) {       // the sparse columns follow this plane's segment in the same row
                const float4 sv = reinterpret_cast<const float4*>(dz + (int64_t)id * dz_stride + A.scol0)[q];
                ms = max(ms, max(max(__float_as_uint(fabsf(sv.x)), __float_as_uint(fabsf(sv.y))),
                                 max(__float_as_uint(fabsf(sv.z)), __float_as_uint(fabsf(sv.w)))));
            }
        }
        float* d = tile + pix * STRIDE + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // ---- phase 2: thread = pixel, level-major stores
    unsigned m = 0u;
    for (int pix = t; pix < PIX; pix += 256) {
        const int64_t p = p0 + pix;
        if (p < n) {
            const int id = order[p];
            const float* c = coords + (int64_t)id * 3;
            A.cs[plane][p] = make_float2(c[A.c0[plane]], c[A.c1[plane]]);
            float* dst = A.dzs[plane];
            const float* srow = tile + pix * STRIDE;
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                if (l >= nl) break;
                float v[F];
#pragma unroll
                for (int f = 0; f < F; ++f) { v[f] = srow[l * F + f]; m = max(m, __float_as_uint(fabsf(v[f]))); }
#pragma unroll
                for (int f = 0; f < F; ++f) dst[((int64_t)l * n + p) * F + f] = v[f];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = max(m, (unsigned)__shfl_xor((int)m, o)); ms = max(ms, (unsigned)__shfl_xor((int)ms, o)); }
    if ((t & 63) == 0) {
        const int slot = (blockIdx.x * 4 + (t >> 6)) & (kMaxSlots - 1);
        if (m > 0u) atomicMax(dzmax + slot, m);
        if (plane == 2 && ms > 0u) atomicMax(A.sdzmax + slot, ms);
    }
}

struct RowArgs {
    const float2* cs[3];
    int* rowstart[3];
    nvp_levels lv[3];
    int rs_off[3][NVP_MAX_LEVELS];
    int rs_total[3];
};

// rowstart[level][r] = first sorted position whose row index at that level is >= r  (r in [0, res])
__global__ __launch_bounds__(256) void rowstart_kernel(RowArgs A, int64_t n) {
    const int plane = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= A.rs_total[plane]) return;
    int l = 0;
    while (l + 1 < A.lv[plane].n_levels && idx >= A.rs_off[plane][l + 1]) ++l;
    const int r = idx - A.rs_off[plane][l];
    const float scale = A.lv[plane].scale[l];
    const int flags = A.lv[plane].flags;
    const float2* cs = A.cs[plane];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (row_of(cs[mid].y, scale, flags) >= r) hi = mid; else lo = mid + 1;
    }
    A.rowstart[plane][idx] = (int)lo;
}

// value * 2^k as a signed 64-bit integer (|value * 2^k| < 2^41 by the choice of k: max|dz| * 2^k < 2^(62 - headroom), headroom >= 21).
// The band kernels are VALU-bound on this conversion (478 M of them per step), so it is built from what the hardware converts
// natively: t = v 2^k (exact, v_ldexp_f32) is cut into hi = trunc(t 2^-20) (|hi| < 2^21: v_cvt_i32_f32) and lo = t - hi 2^20
// (one exact fma, |lo| < 2^20, rounded to nearest even) - 11 instructions instead of the 23 of a hand-rolled mantissa shift.
// NVP_TO_FIXED_SHIFT=1 selects that earlier formulation (round-half-up in magnitude; differs from this one only on exact ties).
#ifndef NVP_TO_FIXED_SHIFT
#define NVP_TO_FIXED_SHIFT 0
#endif
__device__ __forceinline__ long long to_fixed(float v, int k) {
#if NVP_TO_FIXED_SHIFT
    const unsigned b = __float_as_uint(v);
    const int e = (b >> 23) & 0xff;
    const unsigned m = (b & 0x7fffffu) | (e ? 0x800000u : 0u);
    const int sh = (e ? e : 1) - 150 + k;
    long long q;
    if (sh >= 0) q = (long long)m << (sh > 38 ? 38 : sh);
    else {
        const int rs = -sh;
        q = rs > 24 ? 0 : (long long)((m + (1u << (rs - 1))) >> rs);
    }
    return (b >> 31) ? -q : q;
#else
    const float th = truncf(ldexpf(v, k - 20));
    const float lo = __builtin_fmaf(th, -1048576.0f, ldexpf(v, k));
    return ((long long)(int)th << 20) + (long long)__float2int_rn(lo);
#endif
}

struct BandArgs {
    Plan plan;
    nvp_levels lv[3];
    const float2* cs[3];
    const float* dzs[3];
    const int* rowstart[3];
    float* grad[3];
    float* slabs;
    const unsigned* dzmax;
    int headroom_bits;
};

template <int F>
__global__ __launch_bounds__(kBandThreads) void band_kernel(BandArgs A, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ int s_k;
    __shared__ bool s_poison;
    // ---- decode the work item
    int plane = 0, level = 0;
    {
        const int b = blockIdx.x;
        // (plane, level) pairs are laid out in increasing first_block order
        for (int p = 0; p < 3; ++p)
            for (int l = 0; l < A.lv[p].n_levels; ++l)
                if (b >= A.plan.lp[p][l].first_block) { plane = p; level = l; }
    }
    const LevelPlan L = A.plan.lp[plane][level];
    const int local = blockIdx.x - L.first_block;
    const int band = local / L.splits, split = local - band * L.splits;
    const int res = A.lv[plane].res[level];
    const float scale = A.lv[plane].scale[level];
    const int r0 = band * L.rows;
    const int r1 = min(res, r0 + L.rows);
    const int entries = (r1 - r0) * res * F;

    for (int i = threadIdx.x; i < entries; i += kBandThreads) tab[i] = 0ull;
    if (threadIdx.x < 64) {
        // fixed-point scale from max|dz| (kMaxSlots candidates, 4 per lane)
        unsigned m = 0;
        for (int i = threadIdx.x; i < kMaxSlots; i += 64) m = max(m, A.dzmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) {
            const int e = (int)((m >> 23) & 0xff) - 127;            // floor(log2 max|dz|); m == 0 -> -127
            s_k = 62 - A.headroom_bits - (e + 1);
            s_poison = m >= 0x7f800000u;                            // an Inf / NaN latent gradient somewhere in the batch
        }
    }
    __syncthreads();
    const int k = s_k;
    const int lflags = A.lv[plane].flags;

    // ---- sorted pixel ranges that can touch rows [r0, r1): rows iy in [r0-2, r1-1], plus the
    //      wrap-around of the last two rows into rows 0/1 (cell index is taken mod res^2)
    const int* rs = A.rowstart[plane] + L.rs_off;
    const int loA = rs[max(r0 - 2, 0)], hiA = rs[r1];
    int loW = 0, hiW = 0;
    if (r0 < 2) { loW = max(rs[max(res - 2, 0)], hiA); hiW = (int)n; if (loW > hiW) loW = hiW; }   // rows 0 and 1 receive the wrap
    const int lenA = hiA - loA, lenT = lenA + (hiW - loW);
    const int kb = (int)(((long long)lenT * split) / L.splits);
    const int ke = (int)(((long long)lenT * (split + 1)) / L.splits);

    const float2* cs = A.cs[plane];
    const float* dz = A.dzs[plane] + (int64_t)level * n * F;
    for (int kk = kb + threadIdx.x; kk < ke; kk += kBandThreads) {
        const int p = kk < lenA ? loA + kk : loW + (kk - lenA);
        const float2 c = cs[p];
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { g[f] = dz[(int64_t)p * F + f]; any |= (g[f] != 0.f); }
        if (!any) continue;
        const NvpBilerp bl = nvp_bilerp_setup(c.x, c.y, scale, res, lflags);
#pragma unroll
        for (int cnr = 0; cnr < 4; ++cnr) {
            const int off = bl.cell[cnr] - r0 * res;      // row test: cell in [r0*res, r1*res)
            if (off < 0 || off >= (r1 - r0) * res) continue;
            const float w = bl.w[cnr];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const long long q = to_fixed(w * g[f], k);
                atomicAdd(&tab[off * F + f], (unsigned long long)q);
            }
        }
    }
    __syncthreads();

    // ---- flush: exclusive bands go straight to the gradient, split bands to their slab
    const double inv = ldexp(1.0, -k);
    const int64_t lvl_off = (int64_t)A.lv[plane].offset[level] * F;
    float* dst = (L.slab_off < 0) ? A.grad[plane] + lvl_off + (int64_t)r0 * res * F
                                  : A.slabs + L.slab_off + (int64_t)split * res * res * F + (int64_t)r0 * res * F;
    if (s_poison) {          // non-finite dz: the reference's index_put / atomics would carry NaN / Inf; fixed point cannot, so say so loudly
        for (int i = threadIdx.x; i < entries; i += kBandThreads) dst[i] = __uint_as_float(0x7fc00000u);
        return;
    }
    for (int i = threadIdx.x; i < entries; i += kBandThreads) dst[i] = (float)((double)(long long)tab[i] * inv);
}

struct ReduceArgs {
    Plan plan;
    nvp_levels lv[3];
    float* grad[3];
    const float* slabs;
};

// grid.y enumerates the (plane, level) pairs that were split
__global__ __launch_bounds__(256) void slab_reduce_kernel(ReduceArgs A) {
    int want = blockIdx.y, plane = -1, level = -1;
    for (int p = 0; p < 3 && plane < 0; ++p)
        for (int l = 0; l < A.lv[p].n_levels; ++l)
            if (A.plan.lp[p][l].splits > 1 && want-- == 0) { plane = p; level = l; break; }
    if (plane < 0) return;
    const LevelPlan L = A.plan.lp[plane][level];
    const int F = A.lv[plane].n_features;
    const int64_t cells = (int64_t)A.lv[plane].res[level] * A.lv[plane].res[level] * F;
    float* dst = A.grad[plane] + (int64_t)A.lv[plane].offset[level] * F;
    const float* src = A.slabs + L.slab_off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int sp = 0; sp < L.splits; ++sp) s += src[(int64_t)sp * cells + i];
        dst[i] = s;
    }
}

// ---- sparse 3x3 grid (R6), same scheme --------------------------------------------------
// Key = t_idx * X + x_idx (nearest indices, reference sparsegrid.py:43-51).  A workgroup owns the
// cells (t, x in [r0, r1), all y) as an int64 fixed-point LDS table and visits the sorted pixels
// with x_idx in [r0-1, r1]: a pixel's clamped 3x3 patch only touches x rows x_idx-1..x_idx+1.
// Clamped border duplicates accumulate exactly like the reference's index_put_(accumulate=True).
// Every cell of the gradient is written exactly once (zeros included): no memset, no atomics.
#ifndef NVP_SPARSE_ENTRIES
#define NVP_SPARSE_ENTRIES 3000
#endif
constexpr int kSparseThreads = 256;
constexpr int kSparseEntries = NVP_SPARSE_ENTRIES;   // target int64 entries per table (24 KB); grows to one x-row if wider

__global__ __launch_bounds__(256) void sparse_keys_kernel(const float* __restrict__ coords, unsigned* __restrict__ keys, int64_t n, nvp_sparse_shape sh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* c = coords + i * 3;
    keys[i] = (unsigned)(nvp_nearest_idx(c[0], sh.t_res) * sh.x_res + nvp_nearest_idx(c[1], sh.x_res));
}

// srowstart[k] = first sorted position whose key is >= k, k in [0, T*X]
__global__ __launch_bounds__(256) void sparse_rowstart_kernel(const unsigned* __restrict__ keys_sorted, int* __restrict__ rowstart, int nkeys, int64_t n) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > nkeys) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys_sorted[mid] >= (unsigned)k) hi = mid; else lo = mid + 1;
    }
    rowstart[k] = (int)lo;
}

__global__ __launch_bounds__(kSparseThreads) void sparse_band_kernel(const float* __restrict__ coords, const float* __restrict__ dz, int dz_stride, int col0,
                                                                     const int* __restrict__ order, const int* __restrict__ rowstart,
                                                                     const unsigned* __restrict__ dzmax, float* __restrict__ demb,
                                                                     nvp_sparse_shape sh, int rows_per_band, int bands, int headroom_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ int s_k;
    __shared__ bool s_poison;
    const int F = sh.n_features;
    const int t = blockIdx.x / bands, band = blockIdx.x - t * bands;
    const int r0 = band * rows_per_band, r1 = min(sh.x_res, r0 + rows_per_band);
    const int entries = (r1 - r0) * sh.y_res * F;
    for (int i = threadIdx.x; i < entries; i += kSparseThreads) tab[i] = 0ull;
    if (threadIdx.x < 64) {
        unsigned m = 0;
        for (int i = threadIdx.x; i < kMaxSlots; i += 64) m = max(m, dzmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) { s_k = 62 - headroom_bits - ((int)((m >> 23) & 0xff) - 127 + 1); s_poison = m >= 0x7f800000u; }
    }
    __syncthreads();
    const int k = s_k;
    const int lo = rowstart[t * sh.x_res + max(r0 - 1, 0)];
    const int hi = rowstart[t * sh.x_res + min(r1, sh.x_res - 1) + 1];
    // one work unit = (pixel, patch cell): 9 units per pixel
    const int units = (hi - lo) * 9;
    for (int u = threadIdx.x; u < units; u += kSparseThreads) {
        const int p = lo + u / 9, cell9 = u - (u / 9) * 9;
        const int id = order[p];
        const float* c = coords + (int64_t)id * 3;
        const int xi = nvp_nearest_idx(c[1], sh.x_res), yi = nvp_nearest_idx(c[2], sh.y_res);
        const int i = cell9 / 3, j = cell9 - i * 3;
        const int vx = min(max(xi + i - 1, 0), sh.x_res - 1);
        if (vx < r0 || vx >= r1) continue;
        const int vy = min(max(yi + j - 1, 0), sh.y_res - 1);
        const float* g = dz + (int64_t)id * dz_stride + col0 + cell9 * F;
        unsigned long long* dst = tab + ((vx - r0) * sh.y_res + vy) * F;
        for (int f = 0; f < F; ++f) {
            const float v = g[f];
            if (v != 0.f) atomicAdd(dst + f, (unsigned long long)to_fixed(v, k));
        }
    }
    __syncthreads();
    const double inv = ldexp(1.0, -k);
    float* out = demb + (((int64_t)t * sh.x_res + r0) * sh.y_res) * F;
    if (s_poison) {
        for (int i = threadIdx.x; i < entries; i += kSparseThreads) out[i] = __uint_as_float(0x7fc00000u);
        return;
    }
    for (int i = threadIdx.x; i < entries; i += kSparseThreads) out[i] = (float)((double)(long long)tab[i] * inv);
}

bool levels_ok(const nvp_levels* lv) {
    if (!lv || lv->n_levels < 1 || lv->n_levels > NVP_MAX_LEVELS) return false;
    const int f = lv->n_features;
    if (!(f == 1 || f == 2 || f == 4 || f == 8)) return false;
    for (int l = 0; l < lv->n_levels; ++l)
        if (lv->res[l] < 2 || lv->res[l] * f > kMaxLdsEntries) return false;
    return true;
}

// Everything the scatter derives from the COORDINATES alone: sort keys, the xt plane's order (and the y order for unsorted
// batches), the sparse grid's (t, x) order and row table.  A dozen small latency-bound kernels (~0.26 ms back to back) that a host
// can run early on a side stream - underneath the backward chain kernel - through nvp_encode_bwd_presort.
int presort(const float* coords, int64_t n, const nvp_levels* lv[3], const nvp_sparse_shape* sh, char* ws, const Ws& W, int flags, hipStream_t s) {
    float* ky = (float*)(ws + W.keys_in[0]);
    unsigned* kx = (unsigned*)(ws + W.keys_in[1]);
    int* iota = (int*)(ws + W.iota);
    const bool y_sorted = (flags & NVP_COORDS_SORTED_BY_Y) != 0;
    const bool planes_ready = y_sorted && (flags & NVP_DZ_PLANES_READY) != 0;
    hipLaunchKernelGGL(keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, coords, ky, kx, iota,
                       planes_ready ? (float2*)(ws + W.cs[0]) : (float2*)nullptr, planes_ready ? (float2*)(ws + W.cs[1]) : (float2*)nullptr, *lv[2], n);
    int kx_bits = 1;
    {
        int64_t kmax = 0;
        for (int l = 0; l < lv[2]->n_levels; ++l) kmax += lv[2]->res[l] + 1;
        while (((int64_t)1 << kx_bits) <= kmax && kx_bits < 32) ++kx_bits;
    }
    size_t tmp = W.sort_tmp_bytes;
    if (!y_sorted) {                               // otherwise the batch already arrives in ascending y: identity order
        hipError_t e = rocprim::radix_sort_pairs((void*)(ws + W.sort_tmp), tmp, (const float*)(ws + W.keys_in[0]), (float*)(ws + W.keys_out[0]),
                                                 (const int*)iota, (int*)(ws + W.order[0]), (size_t)n, 0, 32, s);
        if (e != hipSuccess) return (int)e;
    }
    {
        hipError_t e = rocprim::radix_sort_pairs((void*)(ws + W.sort_tmp), tmp, (const unsigned*)kx, (unsigned*)(ws + W.keys_out[1]),
                                                 (const int*)iota, (int*)(ws + W.order[1]), (size_t)n, 0, kx_bits, s);
        if (e != hipSuccess) return (int)e;
    }
    unsigned* sk_in = (unsigned*)(ws + W.skey_in);
    unsigned* sk_out = (unsigned*)(ws + W.skey_out);
    hipLaunchKernelGGL(sparse_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, coords, sk_in, n, *sh);
    const int nkeys = sh->t_res * sh->x_res;
    int end_bit = 1;
    while (((int64_t)1 << end_bit) < nkeys && end_bit < 32) ++end_bit;
    size_t tmp2 = W.sort_tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs((void*)(ws + W.sort_tmp), tmp2, (const unsigned*)sk_in, sk_out, (const int*)iota, (int*)(ws + W.sorder), (size_t)n, 0, end_bit, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sparse_rowstart_kernel, dim3((nkeys + 1 + 255) / 256), dim3(256), 0, s, (const unsigned*)sk_out, (int*)(ws + W.srowstart), nkeys, n);
    return 0;
}

template <int F>
int launch_all(const float* coords, const float* dz, int dz_stride, float* g0, float* g1, float* g2, float* demb, int64_t n,
               const nvp_levels* lv[3], const nvp_sparse_shape* sh, char* ws, const Ws& W, const Plan& P, int flags, hipStream_t s) {
    int* iota = (int*)(ws + W.iota);
    const bool y_sorted = (flags & NVP_COORDS_SORTED_BY_Y) != 0;
    // xy / yt latent gradients already level-major in ws AND the sparse columns' max|dz| already in its slots (chain kernel)
    const bool planes_ready = y_sorted && (flags & NVP_DZ_PLANES_READY) != 0;
    const bool do_sparse = !(flags & NVP_SCATTER_DENSE_ONLY), do_dense = !(flags & NVP_SCATTER_SPARSE_ONLY);
    if ((!do_sparse || !do_dense) && !planes_ready) return NVP_ERR_BADARG;      // the split calls need the sparse max from the chain kernel
    int hb = 1;
    while (((int64_t)1 << hb) < n) ++hb;
    // n contributions of magnitude <= max|dz| must fit 2^62: log2(n) + 1 bits of headroom - and never fewer than 22, so that a
    // single contribution stays below 2^40 (to_fixed converts it in two 32-bit halves) whatever the batch size
    const int headroom_bits = hb + 1 > 22 ? hb + 1 : 22;
    int col = 0;
    for (int p = 0; p < 3; ++p) col += lv[p]->n_levels * lv[p]->n_features;
    const int scol0 = col, scols = 9 * sh->n_features;
    if ((scol0 & 3) != 0 || scol0 + ((scols + 3) & ~3) > dz_stride) return NVP_ERR_UNSUPPORTED;
    if (scols > 16 * F) return NVP_ERR_UNSUPPORTED;            // the sparse columns are scanned by the xt plane's own lanes
    const bool presorted = (flags & NVP_SCATTER_PRESORTED) != 0;      // nvp_encode_bwd_presort already ran on this workspace
    if (!presorted) {
        int rc0 = presort(coords, n, lv, sh, ws, W, flags, s);
        if (rc0) return rc0;
    }

    // ---- the three dense planes
    auto dense = [&]() -> int {
        if (!planes_ready) {          // otherwise nvp_encode_bwd_prepare zeroed both slot arrays before the chain kernel fed them
            hipError_t me = hipMemsetAsync(ws + W.dzmax, 0, 2 * kMaxSlots * 4, s);
            if (me != hipSuccess) return (int)me;
        }
        // planes in latent order: xy <- (x, y) sorted by y ; yt <- (t, y) sorted by y ; xt <- (t, x) sorted by x
        PermArgs PA;
        int c = 0;
        const int c0[3] = {1, 0, 0}, c1[3] = {2, 2, 1}, ord[3] = {0, 0, 1};
        for (int p = 0; p < 3; ++p) {
            PA.order[p] = (ord[p] == 0 && y_sorted) ? (const int*)iota : (const int*)(ws + W.order[ord[p]]);
            PA.cs[p] = (float2*)(ws + W.cs[p]);
            PA.dzs[p] = (float*)(ws + W.dzs[p]);
            PA.col0[p] = c; PA.nlev[p] = lv[p]->n_levels; PA.c0[p] = c0[p]; PA.c1[p] = c1[p];
            c += lv[p]->n_levels * lv[p]->n_features;
        }
        PA.scol0 = scol0; PA.scols = planes_ready ? 0 : scols;      // 0: the sparse max came from the chain kernel
        PA.sdzmax = (unsigned*)(ws + W.sdzmax);
        PA.plane0 = planes_ready ? 2 : 0;
        hipLaunchKernelGGL((permute_kernel<F>), dim3((unsigned)((n + PermCfg<F>::PIX - 1) / PermCfg<F>::PIX), 3 - PA.plane0), dim3(256),
                           (size_t)PermCfg<F>::PIX * PermCfg<F>::STRIDE * sizeof(float), s, coords, dz, dz_stride, PA, (unsigned*)(ws + W.dzmax), n);

        RowArgs RA;
        int rs_max = 0;
        for (int p = 0; p < 3; ++p) {
            RA.cs[p] = (const float2*)(ws + W.cs[p]);
            RA.rowstart[p] = (int*)(ws + W.rowstart[p]);
            RA.lv[p] = *lv[p];
            for (int l = 0; l < lv[p]->n_levels; ++l) RA.rs_off[p][l] = P.lp[p][l].rs_off;
            RA.rs_total[p] = P.rs_total[p];
            if (P.rs_total[p] > rs_max) rs_max = P.rs_total[p];
        }
        hipLaunchKernelGGL(rowstart_kernel, dim3((rs_max + 255) / 256, 3), dim3(256), 0, s, RA, n);

        BandArgs BA;
        BA.plan = P;
        float* grads[3] = {g0, g1, g2};
        for (int p = 0; p < 3; ++p) {
            BA.lv[p] = *lv[p];
            BA.cs[p] = (const float2*)(ws + W.cs[p]);
            BA.dzs[p] = (const float*)(ws + W.dzs[p]);
            BA.rowstart[p] = (const int*)(ws + W.rowstart[p]);
            BA.grad[p] = grads[p];
        }
        BA.slabs = (float*)(ws + W.slabs);
        BA.dzmax = (const unsigned*)(ws + W.dzmax);
        BA.headroom_bits = headroom_bits;
        hipLaunchKernelGGL((band_kernel<F>), dim3(P.total_blocks), dim3(kBandThreads), (size_t)P.entries * 8, s, BA, n);

        if (P.reduce_items > 0) {
            ReduceArgs R;
            R.plan = P;
            for (int p = 0; p < 3; ++p) { R.lv[p] = *lv[p]; R.grad[p] = grads[p]; }
            R.slabs = (const float*)(ws + W.slabs);
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(64, P.reduce_items), dim3(256), 0, s, R);
        }
        return 0;
    };

    // ---- sparse grid (its max|dz| slots were filled by the permute pass above, or by the chain kernel when planes_ready)
    auto sparse = [&]() -> int {
        int* sorder = (int*)(ws + W.sorder);
        int* srs = (int*)(ws + W.srowstart);
        unsigned* sdzmax = (unsigned*)(ws + W.sdzmax);
        int sentries = kSparseEntries;
        if (sh->y_res * sh->n_features > sentries) sentries = sh->y_res * sh->n_features;
        if (sentries > kMaxLdsEntries) return NVP_ERR_UNSUPPORTED;          // one x-row does not fit the LDS
        int rows = sentries / (sh->y_res * sh->n_features);
        if (rows > sh->x_res) rows = sh->x_res;
        const int bands = (sh->x_res + rows - 1) / rows;
        hipLaunchKernelGGL(sparse_band_kernel, dim3((unsigned)(sh->t_res * bands)), dim3(kSparseThreads), (size_t)sentries * 8, s,
                           coords, dz, dz_stride, scol0, (const int*)sorder, (const int*)srs, (const unsigned*)sdzmax, demb, *sh, rows, bands, headroom_bits + 2);
        return 0;
    };

    // With the sparse max already known the sparse grid goes FIRST: it is 80 % of the gradient bytes, and a data-parallel host
    // that splits the call (NVP_SCATTER_SPARSE_ONLY, then NVP_SCATTER_DENSE_ONLY) can start exchanging it while the planes scatter.
    int rc = 0;
    if (planes_ready) {
        if (do_sparse) rc = sparse();
        if (rc == 0 && do_dense) rc = dense();
    } else {
        rc = dense();
        if (rc == 0) rc = sparse();
    }
    return rc;
}

}  // namespace

extern "C" {

int64_t nvp_encode_bwd_workspace_bytes(int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                                       const nvp_sparse_shape* sh) {
    if (n < 0 || !levels_ok(lv_xy) || !levels_ok(lv_yt) || !levels_ok(lv_xt) || !sh) return NVP_ERR_BADARG;
    if (n == 0) return 256;
    const nvp_levels* lv[3] = {lv_xy, lv_yt, lv_xt};
    Plan P;
    make_plan(P, lv, n);
    Ws W;
    if (carve(W, P, lv, sh, n)) return NVP_ERR_BADARG;
    return (int64_t)W.total;
}

int nvp_encode_bwd_prepare(int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                           const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, nvp_scatter_lm* out, void* stream) {
    if (n < 1 || !levels_ok(lv_xy) || !levels_ok(lv_yt) || !levels_ok(lv_xt) || !sh || !workspace || !out) return NVP_ERR_BADARG;
    const nvp_levels* lv[3] = {lv_xy, lv_yt, lv_xt};
    Plan P;
    make_plan(P, lv, n);
    Ws W;
    int rc = carve(W, P, lv, sh, n);
    if (rc) return rc;
    if ((int64_t)W.total > workspace_bytes) return NVP_ERR_BADARG;
    char* ws = (char*)workspace;
    if (W.sdzmax != W.dzmax + kMaxSlots * 4) return NVP_ERR_UNSUPPORTED;       // carve() keeps the two slot arrays adjacent: one memset
    hipError_t e = hipMemsetAsync(ws + W.dzmax, 0, 2 * kMaxSlots * 4, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    out->dzs[0] = (float*)(ws + W.dzs[0]);
    out->dzs[1] = (float*)(ws + W.dzs[1]);
    out->dzmax = (uint32_t*)(ws + W.dzmax);
    out->sdzmax = (uint32_t*)(ws + W.sdzmax);
    out->scol0 = 0;
    for (int p = 0; p < 3; ++p) out->scol0 += lv[p]->n_levels * lv[p]->n_features;
    out->scols = 9 * sh->n_features;
    return 0;
}

// The coordinate-only part of the scatter (sort keys, orders, the sparse row table) on `stream`; nvp_encode_bwd then takes
// NVP_SCATTER_PRESORTED with the SAME flags otherwise.  The caller orders the two streams (an event between them).
int nvp_encode_bwd_presort(const float* coords, int64_t n, const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                           const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags, void* stream) {
    if (n < 1 || !coords || !levels_ok(lv_xy) || !levels_ok(lv_yt) || !levels_ok(lv_xt) || !sh || !workspace) return NVP_ERR_BADARG;
    if (n >= ((int64_t)1 << 31)) return NVP_ERR_UNSUPPORTED;
    const nvp_levels* lv[3] = {lv_xy, lv_yt, lv_xt};
    Plan P;
    make_plan(P, lv, n);
    Ws W;
    int rc = carve(W, P, lv, sh, n);
    if (rc) return rc;
    if ((int64_t)W.total > workspace_bytes) return NVP_ERR_BADARG;
    return presort(coords, n, lv, sh, (char*)workspace, W, flags, (hipStream_t)stream);
}

// dz: row-major latent gradient [>= n][dz_stride] (columns xy | yt | xt | sparse).
// d_kf_* and d_emb: every element is OVERWRITTEN (no zero-fill needed).
int nvp_encode_bwd(const float* coords, const float* dz, int32_t dz_stride,
                   float* d_kf_xy, float* d_kf_yt, float* d_kf_xt, float* d_emb, int64_t n,
                   const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                   const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags, void* stream) {
    if (!levels_ok(lv_xy) || !levels_ok(lv_yt) || !levels_ok(lv_xt) || !sh || n < 0) return NVP_ERR_BADARG;
    if (lv_xy->n_features != lv_yt->n_features || lv_xy->n_features != lv_xt->n_features) return NVP_ERR_UNSUPPORTED;
    if (n >= ((int64_t)1 << 31)) return NVP_ERR_UNSUPPORTED;
    const nvp_levels* lv[3] = {lv_xy, lv_yt, lv_xt};
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        // an empty batch still owes zero gradients for the planes it "overwrites"
        for (int p = 0; p < 3; ++p) {
            float* g = p == 0 ? d_kf_xy : (p == 1 ? d_kf_yt : d_kf_xt);
            hipError_t e = hipMemsetAsync(g, 0, (size_t)lv[p]->offset[lv[p]->n_levels] * lv[p]->n_features * 4, s);
            if (e != hipSuccess) return (int)e;
        }
        hipError_t e = hipMemsetAsync(d_emb, 0, (size_t)sh->t_res * sh->x_res * sh->y_res * sh->n_features * 4, s);
        return (int)e;
    }
    if (!coords || !dz || !d_kf_xy || !d_kf_yt || !d_kf_xt || !d_emb || !workspace) return NVP_ERR_BADARG;
    Plan P;
    make_plan(P, lv, n);
    Ws W;
    int rc = carve(W, P, lv, sh, n);
    if (rc) return rc;
    if ((int64_t)W.total > workspace_bytes) return NVP_ERR_BADARG;
    const int need = lv_xy->n_levels * lv_xy->n_features + lv_yt->n_levels * lv_yt->n_features + lv_xt->n_levels * lv_xt->n_features +
                     9 * sh->n_features;
    if (dz_stride < need || (dz_stride & 3)) return NVP_ERR_BADARG;
    if ((lv_xy->n_levels * lv_xy->n_features) & 3 || (lv_yt->n_levels * lv_yt->n_features) & 3 || (lv_xt->n_levels * lv_xt->n_features) & 3)
        return NVP_ERR_UNSUPPORTED;             // 16-B aligned per-plane row segments (true for every 4-level-multiple config)
    switch (lv_xy->n_features) {
        case 1: rc = launch_all<1>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s); break;
        case 2: rc = launch_all<2>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s); break;
        case 4: rc = launch_all<4>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s); break;
        case 8: rc = launch_all<8>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s); break;
        default: return NVP_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    NVP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
