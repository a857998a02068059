// Gradient scatter-add of the NVP encoders (R3 + R6) for gfx950, atomic-free on the dense
// planes.
//
// Measured on MI355X (tools/probes): global fp32 atomics sustain only ~21 Gop/s whatever the
// footprint or scope, LDS ds_add_f32 ~0.2 Top/s, but LDS *integer* atomics 3-8 Top/s.  The
// reference-shaped scatter (one fp32 atomic per corner x feature = 478 M atomics per step for
// nvp_s) therefore costs ~25 ms.  This file replaces it by a deterministic scheme:
//
//  1. sort the batch's pixels by the row coordinate of each plane (y for the xy and yt planes,
//     x for the xt plane) - hand-written counting sorts on small integer keys (the sum over the levels of the
//     row index: 13-14 bits, monotone for every level at once; csort_* below);
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
#include <cstdlib>
#include <cstring>
#include "adamw.h"
#include "grid_math.h"

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
constexpr int kScanTile = 4096;               // counting-sort scan: bins per workgroup (256 threads x 16 consecutive bins)

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

// Target table size per feature width.  A band's visits scale with (rows + 2) / rows; with F = 4 the four finest levels (587 .. 1443 cells
// x 4 floats per row) get ONE row per band from the F = 2 target - three visits per pixel and level.  NVP_BAND_ENTRIES_F4 (build) /
// the experiments build's environment variable of the same name tune it.
#ifndef NVP_BAND_ENTRIES_F4
#define NVP_BAND_ENTRIES_F4 NVP_BAND_ENTRIES
#endif
#ifndef NVP_SPARSE_ENTRIES_F4
#define NVP_SPARSE_ENTRIES_F4 NVP_SPARSE_ENTRIES
#endif
inline int env_int_or(const char* name, int dflt) {
#if NVP_EXPERIMENTS
    const char* e = getenv(name);
    if (e && atoi(e) > 0) return atoi(e);
#endif
    (void)name;
    return dflt;
}
inline int band_entries_target(int F) {
    static const int f4 = env_int_or("NVP_BAND_ENTRIES_F4", NVP_BAND_ENTRIES_F4);
    return F >= 4 ? f4 : kLdsEntries;
}

void make_plan(Plan& P, const nvp_levels* lv[3], int64_t n) {
    P.entries = band_entries_target(lv[0]->n_features);
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
    size_t keys[2], order[2], cs[3], dzs[3], rowstart[3], dzmax, slabs, total;
    size_t kstart[2], kcursor[2];                             // counting sorts of the two dense keys: exclusive starts, scatter cursors
    size_t skey, sorder, srowstart, scursor, sdzmax;         // sparse grid
    size_t tsum;                                              // per-tile sums of the counting sorts' scans (shared: the sorts run one after the other)
    int nkeys[2];                                             // key ranges of the two dense sorts (y key, x key)
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// key(c) = sum over the levels of clamp(row_l(c), 0, res_l + 1): its range
int key_range(const nvp_levels* a, const nvp_levels* b) {
    int r = 1;
    for (int l = 0; l < a->n_levels; ++l) r += a->res[l] + 1;
    if (b) for (int l = 0; l < b->n_levels; ++l) r += b->res[l] + 1;
    return r;
}

int carve(Ws& W, const Plan& P, const nvp_levels* lv[3], const nvp_sparse_shape* sh, int64_t n) {
    size_t o = 0;
    W.nkeys[0] = key_range(lv[0], lv[1]);          // y: rows of the xy AND the yt plane
    W.nkeys[1] = key_range(lv[2], nullptr);        // x: rows of the xt plane
    for (int k = 0; k < 2; ++k) { W.keys[k] = o; o = align_up(o + n * 4); }
    for (int k = 0; k < 2; ++k) { W.order[k] = o; o = align_up(o + n * 4); }
    for (int k = 0; k < 2; ++k) { W.kstart[k] = o; o = align_up(o + ((size_t)W.nkeys[k] + 1) * 4); }
    for (int k = 0; k < 2; ++k) { W.kcursor[k] = o; o = align_up(o + ((size_t)W.nkeys[k] + 1) * 4); }
    for (int p = 0; p < 3; ++p) { W.cs[p] = o; o = align_up(o + n * 8); }
    for (int p = 0; p < 3; ++p) { W.dzs[p] = o; o = align_up(o + (size_t)n * lv[p]->n_levels * lv[p]->n_features * 4); }
    for (int p = 0; p < 3; ++p) { W.rowstart[p] = o; o = align_up(o + (size_t)P.rs_total[p] * 4); }
    W.dzmax = o; o += kMaxSlots * 4;
    W.sdzmax = o; o = align_up(o + kMaxSlots * 4);             // adjacent to dzmax: nvp_encode_bwd_prepare zeroes both with one memset
    W.slabs = o; o = align_up(o + (size_t)P.slab_floats * 8);  // int64 fixed-point partial tables
    W.skey = o; o = align_up(o + n * 4);
    W.sorder = o; o = align_up(o + n * 4);
    const size_t nsk = (size_t)sh->t_res * sh->x_res + 1;
    W.srowstart = o; o = align_up(o + nsk * 4);
    W.scursor = o; o = align_up(o + nsk * 4);
    W.tsum = o; o = align_up(o + (nsk / kScanTile + 2) * 4 + ((size_t)(W.nkeys[0] > W.nkeys[1] ? W.nkeys[0] : W.nkeys[1]) / kScanTile + 2) * 4);
    W.total = o;
    return 0;
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int row_of(float c1, float scale, int flags) { return (int)floorf(nvp_grid_pos(c1, scale, flags)); }

// Sort keys.  The scatter needs the pixels of a plane ordered so that the grid ROW index is non-decreasing at EVERY level;
// ordering by the row coordinate itself does that, but a float key costs four 8-bit radix passes.  key(c) = sum over the levels
// of row_l(c) is a non-decreasing step function of c that steps exactly where some level's row index steps, so equal keys
// mean equal rows at every level (ties may sit in any order: the fixed-point accumulation is exact, hence order-independent) -
// and it has 13-14 bits: ONE counting-sort pass.  Rows are clamped to [0, res + 1]; coordinates outside [0, 1] keep a valid (if
// arbitrary) order.  kx: the xt plane's key (rows indexed by x); ky (only for batches that do not arrive y-sorted): rows of the
// xy and of the yt plane, both indexed by y.
// cs_xy / cs_yt (optional): the (dim0, dim1) coordinate pairs of the xy and yt planes in batch order - what the permute pass
// would write for them when their sorted order is the identity (NVP_DZ_PLANES_READY).
__global__ __launch_bounds__(256) void keys_kernel(const float* __restrict__ coords, unsigned* __restrict__ ky, unsigned* __restrict__ kx,
                                                   float2* __restrict__ cs_xy, float2* __restrict__ cs_yt,
                                                   nvp_levels lv0, nvp_levels lv1, nvp_levels lvx, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = coords[i * 3], x = coords[i * 3 + 1], y = coords[i * 3 + 2];
    unsigned key = 0u;
    for (int l = 0; l < lvx.n_levels; ++l) key += (unsigned)min(max(row_of(x, lvx.scale[l], lvx.flags), 0), lvx.res[l] + 1);
    kx[i] = key;
    if (ky) {
        key = 0u;
        for (int l = 0; l < lv0.n_levels; ++l) key += (unsigned)min(max(row_of(y, lv0.scale[l], lv0.flags), 0), lv0.res[l] + 1);
        for (int l = 0; l < lv1.n_levels; ++l) key += (unsigned)min(max(row_of(y, lv1.scale[l], lv1.flags), 0), lv1.res[l] + 1);
        ky[i] = key;
    }
    if (cs_xy) { cs_xy[i] = make_float2(x, y); cs_yt[i] = make_float2(t, y); }
}

// ---- counting sort on small integer keys (replaces three library radix sorts) --------------------------------------------------
// keys in [0, nkeys).  Three launches: histogram (cnt[] zeroed by the caller), exclusive scan (start[k] = first sorted position
// whose key is >= k, k in [0, nkeys] - which IS the row table the sparse grid's band kernel needs - plus a copy as scatter cursors),
// scatter (order[pos] = pixel).  The order among equal keys is whatever the atomics make it: every consumer accumulates in exact
// integer arithmetic, so the gradients do not depend on it (the slabs of split levels are int64 for that reason).
constexpr int kCsortLdsBins = 16384;             // histograms up to this many bins are privatised in LDS (64 KB)
constexpr int kCsortChunk = 8192;                // keys per workgroup

__global__ __launch_bounds__(256) void csort_hist_kernel(const unsigned* __restrict__ keys, unsigned* __restrict__ cnt, int64_t n, int nkeys) {
    extern __shared__ unsigned h[];
    const int64_t i0 = (int64_t)blockIdx.x * kCsortChunk;
    const int64_t i1 = min(n, i0 + kCsortChunk);
    if (nkeys <= kCsortLdsBins) {
        for (int k = threadIdx.x; k < nkeys; k += 256) h[k] = 0u;
        __syncthreads();
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&h[min(keys[i], (unsigned)(nkeys - 1))], 1u);
        __syncthreads();
        for (int k = threadIdx.x; k < nkeys; k += 256) { const unsigned v = h[k]; if (v) atomicAdd(&cnt[k], v); }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&cnt[min(keys[i], (unsigned)(nkeys - 1))], 1u);
    }
}

// exclusive scan of the counts in two launches: per-tile sums (tiles of 4096 bins), then every tile adds the sums of the tiles
// before it (<= a few dozen values, summed redundantly) to its own scan.  start[k] = sum of cnt[0..k), k in [0, nkeys];
// cursor = the same values (cnt aliases cursor: a thread overwrites only elements it has read itself).

__global__ __launch_bounds__(256) void csort_tilesum_kernel(const unsigned* __restrict__ cnt, unsigned* __restrict__ tsum, int nkeys) {
    __shared__ unsigned part[4];
    const int k0 = blockIdx.x * kScanTile + threadIdx.x * 16;
    unsigned s = 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (k0 + u < nkeys) ? cnt[k0 + u] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += (unsigned)__shfl_xor((int)s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tsum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void csort_scan_kernel(const unsigned* cnt, const unsigned* __restrict__ tsum, int* __restrict__ start, unsigned* cursor, int nkeys) {
    __shared__ unsigned wsum[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned base = 0u;
    for (int b = 0; b < (int)blockIdx.x; ++b) base += tsum[b];        // <= nkeys / 4096 values
    const int k0 = blockIdx.x * kScanTile + t * 16;
    unsigned c[16], s = 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) { c[u] = (k0 + u < nkeys) ? cnt[k0 + u] : 0u; s += c[u]; }
    unsigned incl = s;                                                // inclusive scan of the per-thread sums within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = (unsigned)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned run = base + incl - s;
    for (int w2 = 0; w2 < wv; ++w2) run += wsum[w2];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (k0 + u < nkeys) { start[k0 + u] = (int)run; cursor[k0 + u] = run; }
        run += c[u];
    }
    if (k0 <= nkeys - 1 && nkeys - 1 < k0 + 16) { start[nkeys] = (int)run; cursor[nkeys] = run; }      // the thread that owns the last bin: run == n
}

__global__ __launch_bounds__(256) void csort_scatter_kernel(const unsigned* __restrict__ keys, unsigned* __restrict__ cursor, int* __restrict__ order, int64_t n, int nkeys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned pos = atomicAdd(&cursor[min(keys[i], (unsigned)(nkeys - 1))], 1u);
    order[pos] = (int)i;
}

// cnt aliases `cursor` for the histogram phase (the scan reads cnt and overwrites it in place, element by element, after reading it)
int csort(const unsigned* keys, int64_t n, int nkeys, int* start, unsigned* cursor, unsigned* tsum, int* order, hipStream_t s) {
    hipError_t e = hipMemsetAsync(cursor, 0, ((size_t)nkeys + 1) * 4, s);
    if (e != hipSuccess) return (int)e;
    const unsigned nb = (unsigned)((n + kCsortChunk - 1) / kCsortChunk);
    hipLaunchKernelGGL(csort_hist_kernel, dim3(nb), dim3(256), nkeys <= kCsortLdsBins ? (size_t)nkeys * 4 : 0, s, keys, cursor, n, nkeys);
    const unsigned nt = (unsigned)((nkeys + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(csort_tilesum_kernel, dim3(nt), dim3(256), 0, s, (const unsigned*)cursor, tsum, nkeys);
    hipLaunchKernelGGL(csort_scan_kernel, dim3(nt), dim3(256), 0, s, (const unsigned*)cursor, (const unsigned*)tsum, start, cursor, nkeys);
    hipLaunchKernelGGL(csort_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, cursor, order, n, nkeys);
    return 0;
}

struct PermArgs {
    const int* order[3];      // sorted->pixel id, per plane (nullptr: identity - the batch arrives in that plane's order)
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
            const int id = order ? order[p] : (int)p;
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
            const int id = order ? order[p] : (int)p;
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
    const int* order[3];      // sorted->pixel id per plane (nullptr: identity)
    int col[3];               // coordinate column that indexes the plane's grid ROW (dim 1 of the plane): y, y, x
    int* rowstart[3];
    nvp_levels lv[3];
    int rs_off[3][NVP_MAX_LEVELS];
    int rs_total[3];
};

// rowstart[level][r] = first sorted position whose row index at that level is >= r  (r in [0, res]).  Coordinate-only: it
// searches the batch's coordinates through the plane's order, so it runs with the sorts (nvp_encode_bwd_presort), not behind the
// permute pass.
__global__ __launch_bounds__(256) void rowstart_kernel(const float* __restrict__ coords, RowArgs A, int64_t n) {
    const int plane = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= A.rs_total[plane]) return;
    int l = 0;
    while (l + 1 < A.lv[plane].n_levels && idx >= A.rs_off[plane][l + 1]) ++l;
    const int r = idx - A.rs_off[plane][l];
    const float scale = A.lv[plane].scale[l];
    const int flags = A.lv[plane].flags;
    const int* order = A.order[plane];
    const float* c1 = coords + A.col[plane];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t id = order ? order[mid] : mid;
        if (row_of(c1[id * 3], scale, flags) >= r) hi = mid; else lo = mid + 1;
    }
    A.rowstart[plane][idx] = (int)lo;
}

// value * 2^k as a signed 64-bit integer, rounded to nearest even (|value * 2^k| < 2^41 by the choice of k: max|dz| * 2^k <
// 2^(62 - headroom), headroom >= 21).  The band kernels are VALU-bound on this conversion (478 M of them per step), so it is three
// instructions: d = fma(double(v), 2^k, 1.5 * 2^52) lands in the binade [2^52, 2^53), whose 52 mantissa bits ARE the integer
// (two's complement around the bias); the bias 0x4338000000000000 has a zero low word, so removing it is one 32-bit subtraction
// on the high word.  Bit-identical to the earlier formulations (NVP_TO_FIXED=1: ldexp / trunc / fma / two 32-bit conversions,
// 11 instructions; =2: the hand-rolled mantissa shift, 23, round-half-up in magnitude: differs on exact ties only) - checked
// exhaustively on the host for 4e7 random (v, k) pairs and by the gradient tests.
#ifndef NVP_TO_FIXED
#define NVP_TO_FIXED 0
#endif
struct FixedScale { double s; int k; };
__device__ __forceinline__ FixedScale fixed_scale(int k) { FixedScale f; f.k = k; f.s = __longlong_as_double((long long)(1023 + k) << 52); return f; }
__device__ __forceinline__ long long to_fixed(float v, const FixedScale& fs) {
#if NVP_TO_FIXED == 2
    const int k = fs.k;
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
#elif NVP_TO_FIXED == 1
    const int k = fs.k;
    const float th = truncf(ldexpf(v, k - 20));
    const float lo = __builtin_fmaf(th, -1048576.0f, ldexpf(v, k));
    return ((long long)(int)th << 20) + (long long)__float2int_rn(lo);
#else
    const double d = __builtin_fma((double)v, fs.s, 6755399441055744.0);
    return __double_as_longlong(d) - 0x4338000000000000LL;
#endif
}

// The three dense planes' optimizer step applied in the flushes (band_kernel: exclusive bands; slab_reduce_kernel: split levels): the
// gradient of a cell goes from the int64 table / slab sum straight into torch.optim.AdamW's update (adamw.h, same arithmetic as
// nvp_adamw_step) - it is never written to HBM, and no optimizer launch is left for the planes.  on == 0: plain gradient output.
struct DenseAdam { float* p[3]; float* m[3]; float* v[3]; AdamScalars S[3]; int on; };

struct BandArgs {
    DenseAdam ad;
    Plan plan;
    nvp_levels lv[3];
    const float2* cs[3];
    const float* dzs[3];
    const int* rowstart[3];
    float* grad[3];
    long long* slabs;
    const unsigned* dzmax;
    int headroom_bits;
};

// fixed-point exponent from the max|dz| slots: k such that n contributions of magnitude <= max|dz| fit 2^62
__device__ __forceinline__ void scale_from_slots(const unsigned* __restrict__ slots, int headroom_bits, int* s_k, bool* s_poison) {
    if (threadIdx.x < 64) {
        unsigned m = 0;
        for (int i = threadIdx.x; i < kMaxSlots; i += 64) m = max(m, slots[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) {
            const int e = (int)((m >> 23) & 0xff) - 127;            // floor(log2 max|dz|); m == 0 -> -127
            *s_k = 62 - headroom_bits - (e + 1);
            *s_poison = m >= 0x7f800000u;                           // an Inf / NaN latent gradient somewhere in the batch
        }
    }
}

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
    scale_from_slots(A.dzmax, A.headroom_bits, &s_k, &s_poison);
    __syncthreads();
    const FixedScale fs = fixed_scale(s_k);
    const int lflags = A.lv[plane].flags;

    // ---- sorted pixel ranges that can touch rows [r0, r1): rows iy in [r0-1, r1-1] (a pixel touches iy and iy + 1), the
    //      row r0-2 for the few pixels whose dim-0 corner i0 + 1 == res wraps into the next row (checked on the coordinates
    //      alone, before anything else is loaded), plus the wrap-around of the last two rows into rows 0/1 (the cell index
    //      is taken mod res^2)
    const int* rs = A.rowstart[plane] + L.rs_off;
    const int loX = rs[max(r0 - 2, 0)], loA = rs[max(r0 - 1, 0)], hiA = rs[r1];
    int loW = 0, hiW = 0;
    if (r0 < 2) { loW = max(rs[max(res - 2, 0)], hiA); hiW = (int)n; if (loW > hiW) loW = hiW; }   // rows 0 and 1 receive the wrap
    const int lenX = loA - loX, lenA = hiA - loA, lenT = lenX + lenA + (hiW - loW);
    const int kb = (int)(((long long)lenT * split) / L.splits);
    const int ke = (int)(((long long)lenT * (split + 1)) / L.splits);

    const float2* cs = A.cs[plane];
    const float* dz = A.dzs[plane] + (int64_t)level * n * F;
    const int span = (r1 - r0) * res;
    const int base = r0 * res;
    // The visit of pixel kk + kBandThreads is REQUESTED (coordinates and latent gradient: 8 + 4 F bytes) before pixel kk is worked
    // on: the loop body is ~40 VALU operations and a few LDS atomics behind two loads, and left to itself every trip waited out
    // its own round trip (NVP_BAND_PREFETCH=0: the plain loop).
#ifndef NVP_BAND_PREFETCH
#define NVP_BAND_PREFETCH 1
#endif
    auto visit_index = [&](int kk) { return kk < lenX ? loX + kk : (kk < lenX + lenA ? loA + (kk - lenX) : loW + (kk - lenX - lenA)); };
    float2 c_nx = make_float2(0.f, 0.f);
    float g_nx[F];
#pragma unroll
    for (int f = 0; f < F; ++f) g_nx[f] = 0.f;
    // Visits of the EXTRA range (row r0 - 2: only a pixel whose dim-0 corner wraps reaches this band from there - decided on the coordinates
    // alone) request the coordinates only; the latent gradient of the rare pixel that passes the test is fetched on demand
    // (NVP_BAND_EXTRA_LEAN=0: the gradient is requested for every visit).  One-row bands (the finest levels) make three visits per pixel
    // and level, one of them extra: 8 instead of 8 + 4 F bytes for that third.
#ifndef NVP_BAND_EXTRA_LEAN
#define NVP_BAND_EXTRA_LEAN 1
#endif
    if (NVP_BAND_PREFETCH && kb + (int)threadIdx.x < ke) {
        const int p = visit_index(kb + threadIdx.x);
        c_nx = cs[p];
        if (!NVP_BAND_EXTRA_LEAN || kb + (int)threadIdx.x >= lenX) {
#pragma unroll
            for (int f = 0; f < F; ++f) g_nx[f] = dz[(int64_t)p * F + f];
        }
    }
#ifndef NVP_BAND_PRIO
#define NVP_BAND_PRIO 0          // experiment: wave priority of the visit loop (1) or of the flush (2) against the other workgroups of the CU
#endif
    if (NVP_BAND_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    for (int kk = kb + threadIdx.x; kk < ke; kk += kBandThreads) {
        const bool extra = kk < lenX;
        const int p = visit_index(kk);
        float2 c;
        float g[F];
        if (NVP_BAND_PREFETCH) {
            c = c_nx;
#pragma unroll
            for (int f = 0; f < F; ++f) g[f] = g_nx[f];
            if (kk + kBandThreads < ke) {
                const int pn = visit_index(kk + kBandThreads);
                c_nx = cs[pn];
                if (!NVP_BAND_EXTRA_LEAN || kk + kBandThreads >= lenX) {
#pragma unroll
                    for (int f = 0; f < F; ++f) g_nx[f] = dz[(int64_t)pn * F + f];
                }
            }
        } else {
            c = cs[p];
        }
        const float p0 = nvp_grid_pos(c.x, scale, lflags), p1 = nvp_grid_pos(c.y, scale, lflags);
        const float f0 = floorf(p0), f1 = floorf(p1);
        const int i0 = (int)f0, i1 = (int)f1;
        if (extra && ((lflags & NVP_GRID_CLAMP) || i0 + 1 < res)) continue;      // row r0-2 only reaches row r0 through the wrap
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { if (!NVP_BAND_PREFETCH || (NVP_BAND_EXTRA_LEAN && extra)) g[f] = dz[(int64_t)p * F + f]; any |= (g[f] != 0.f); }
        // cells first: a visit that touches no row of this band ends here, before any weight or conversion is computed
        int off[4];
        bool hit = false;
#pragma unroll
        for (int cnr = 0; cnr < 4; ++cnr) {
            off[cnr] = nvp_grid_cell(i0, i1, cnr & 1, cnr >> 1, res, lflags) - base;      // row test: cell in [r0*res, r1*res)
            hit |= (unsigned)off[cnr] < (unsigned)span;
        }
        if (!any || !hit) continue;
        const float w0 = nvp_sub_rn(p0, f0), w1 = nvp_sub_rn(p1, f1);
        const float u0 = nvp_sub_rn(1.0f, w0), u1 = nvp_sub_rn(1.0f, w1);
        const float w[4] = {nvp_mul_rn(u0, u1), nvp_mul_rn(w0, u1), nvp_mul_rn(u0, w1), nvp_mul_rn(w0, w1)};      // == nvp_bilerp_setup
#pragma unroll
        for (int cnr = 0; cnr < 4; ++cnr) {
            if ((unsigned)off[cnr] >= (unsigned)span) continue;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const long long q = to_fixed(nvp_mul_rn(w[cnr], g[f]), fs);
                atomicAdd(&tab[off[cnr] * F + f], (unsigned long long)q);
            }
        }
    }
    if (NVP_BAND_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (NVP_BAND_PRIO == 2) __builtin_amdgcn_s_setprio(2);
    __syncthreads();

    // ---- flush: exclusive bands go straight to the gradient; split bands keep their exact int64 sums in a slab (the order of
    //      the pixels among equal sort keys - hence which split sees which pixel - must not show in the result)
    if (L.slab_off >= 0) {
        long long* dst = A.slabs + L.slab_off + (int64_t)split * res * res * F + (int64_t)r0 * res * F;
        for (int i = threadIdx.x; i < entries; i += kBandThreads) dst[i] = (long long)tab[i];
        return;
    }
    const double inv = ldexp(1.0, -s_k);
    const int64_t lvl_off = (int64_t)A.lv[plane].offset[level] * F;
    float* dst = A.ad.on ? nullptr : A.grad[plane] + lvl_off + (int64_t)r0 * res * F;
    if (s_poison && !A.ad.on) {          // non-finite dz: the reference's index_put / atomics would carry NaN / Inf; fixed point cannot, so say so loudly
        for (int i = threadIdx.x; i < entries; i += kBandThreads) dst[i] = __uint_as_float(0x7fc00000u);
        return;
    }
    if (A.ad.on) {
        float* pp = A.ad.p[plane] + lvl_off + (int64_t)r0 * res * F;
        float* pm = A.ad.m[plane] + lvl_off + (int64_t)r0 * res * F;
        float* pv = A.ad.v[plane] + lvl_off + (int64_t)r0 * res * F;
        for (int i = threadIdx.x; i < entries; i += kBandThreads) {
            float x = pp[i], m = pm[i], v = pv[i];
            adam1(x, s_poison ? __uint_as_float(0x7fc00000u) : (float)((double)(long long)tab[i] * inv), m, v, A.ad.S[plane]);       // (a NaN gradient poisons the
            pp[i] = x; pm[i] = m; pv[i] = v;                                                                                    // parameter, as the gradient route would)
        }
        return;
    }
    for (int i = threadIdx.x; i < entries; i += kBandThreads) dst[i] = (float)((double)(long long)tab[i] * inv);
}

struct ReduceArgs {
    DenseAdam ad;
    Plan plan;
    nvp_levels lv[3];
    float* grad[3];
    const long long* slabs;
    const unsigned* dzmax;
    int headroom_bits;
};

// grid.y enumerates the (plane, level) pairs that were split: exact int64 sum over the splits, one conversion
__global__ __launch_bounds__(256) void slab_reduce_kernel(ReduceArgs A) {
    __shared__ int s_k;
    __shared__ bool s_poison;
    scale_from_slots(A.dzmax, A.headroom_bits, &s_k, &s_poison);
    __syncthreads();
    int want = blockIdx.y, plane = -1, level = -1;
    for (int p = 0; p < 3 && plane < 0; ++p)
        for (int l = 0; l < A.lv[p].n_levels; ++l)
            if (A.plan.lp[p][l].splits > 1 && want-- == 0) { plane = p; level = l; break; }
    if (plane < 0) return;
    const LevelPlan L = A.plan.lp[plane][level];
    const int F = A.lv[plane].n_features;
    const int64_t cells = (int64_t)A.lv[plane].res[level] * A.lv[plane].res[level] * F;
    const int64_t lvl_off = (int64_t)A.lv[plane].offset[level] * F;
    float* dst = A.ad.on ? nullptr : A.grad[plane] + lvl_off;
    const long long* src = A.slabs + L.slab_off;
    const double inv = ldexp(1.0, -s_k);
    const bool poison = s_poison;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
        long long sum = 0;
        for (int sp = 0; sp < L.splits; ++sp) sum += src[(int64_t)sp * cells + i];
        const float g = poison ? __uint_as_float(0x7fc00000u) : (float)((double)sum * inv);
        if (A.ad.on) {
            float x = A.ad.p[plane][lvl_off + i], m = A.ad.m[plane][lvl_off + i], v = A.ad.v[plane][lvl_off + i];
            adam1(x, g, m, v, A.ad.S[plane]);
            A.ad.p[plane][lvl_off + i] = x; A.ad.m[plane][lvl_off + i] = m; A.ad.v[plane][lvl_off + i] = v;
        } else {
            dst[i] = g;
        }
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

// rowstart[k] = first sorted position whose key is >= k, k in [0, T*X]: the counting sort's exclusive starts (csort_scan_kernel).
// One work item = (t, band of x rows): ~50 pixels.  An item's life is a chain of four dependent loads (row table -> order ->
// coordinates -> latent gradient) in front of a few microseconds of LDS work, so the workgroups are PERSISTENT and software-
// pipelined: while an item's table is flushed to the gradient (and zeroed for the next item in the same pass), the next item's
// chain is already in flight into registers (two (pixel, patch x-row) units per thread; the rare item with more falls back to
// loading in place).
template <int F>
struct SparseUnit { int row, yi; bool on; float g[3 * F]; };

// ADAM: the flush applies the optimizer step instead of writing the gradient (adamw.h): the gradient of a sparse-grid element
// goes from the LDS table straight into the update of (p, m, v) and never touches HBM - 432 MB written and 432 MB read per step
// less for nvp_s.  The item's (p, m, v) are requested at the item's START (kAdamTrips float4 of each per thread, in registers)
// and arrive while the item's units are added into the table.  Needs y_res * F % 4 == 0 and 16-B aligned tensors (host check).
constexpr int kAdamTrips = 4;                     // float4 trips per thread and item: tables up to 4 096 entries
struct SparseAdam { float* p; float* m; float* v; AdamScalars S; };

template <int F, bool ADAM>
__global__ __launch_bounds__(kSparseThreads) void sparse_band_kernel(const float* __restrict__ coords, const float* __restrict__ dz, int dz_stride, int col0,
                                                                     const int* __restrict__ order, const int* __restrict__ rowstart,
                                                                     const unsigned* __restrict__ dzmax, float* __restrict__ demb,
                                                                     nvp_sparse_shape sh, int rows_per_band, int bands, int n_items, int headroom_bits,
                                                                     SparseAdam ad) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ int s_k;
    __shared__ bool s_poison;
    scale_from_slots(dzmax, headroom_bits, &s_k, &s_poison);
    const int max_entries = rows_per_band * sh.y_res * F;
    for (int i = threadIdx.x; i < max_entries; i += kSparseThreads) tab[i] = 0ull;
    __syncthreads();
    const FixedScale fs = fixed_scale(s_k);
    const double inv = ldexp(1.0, -s_k);
    const bool poison = s_poison;

    SparseUnit<F> un[2];
    int lo_n = 0, hi_n = 0;
    auto unit_load = [&](SparseUnit<F>& u, int p, int i, int r0, int r1) {
        const int id = order[p];
        const float* c = coords + (int64_t)id * 3;
        const int xi = nvp_nearest_idx(c[1], sh.x_res);
        const int vx = min(max(xi + i - 1, 0), sh.x_res - 1);
        u.on = vx >= r0 && vx < r1;
        u.row = vx - r0;
        u.yi = nvp_nearest_idx(c[2], sh.y_res);
        if (u.on) {
            const float* g = dz + (int64_t)id * dz_stride + col0 + 3 * i * F;
#pragma unroll
            for (int e = 0; e < 3 * F; ++e) u.g[e] = g[e];
        }
    };
    auto unit_apply = [&](const SparseUnit<F>& u) {
        if (!u.on) return;
        unsigned long long* row = tab + u.row * sh.y_res * F;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int vy = min(max(u.yi + j - 1, 0), sh.y_res - 1);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float v = u.g[j * F + f];
                if (v != 0.f) atomicAdd(row + vy * F + f, (unsigned long long)to_fixed(v, fs));
            }
        }
    };
    auto prefetch = [&](int item) {
        const int t = item / bands, band = item - t * bands;
        const int r0 = band * rows_per_band, r1 = min(sh.x_res, r0 + rows_per_band);
        lo_n = rowstart[t * sh.x_res + max(r0 - 1, 0)];
        hi_n = rowstart[t * sh.x_res + min(r1, sh.x_res - 1) + 1];
        const int units = (hi_n - lo_n) * 3;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int u = threadIdx.x + kSparseThreads * s2;
            un[s2].on = false;
            if (u < units) { const int q = u / 3; unit_load(un[s2], lo_n + q, u - q * 3, r0, r1); }
        }
    };

    int item = blockIdx.x;
    if (item < n_items) prefetch(item);
    for (; item < n_items; item += gridDim.x) {
        const int t = item / bands, band = item - t * bands;
        const int r0 = band * rows_per_band, r1 = min(sh.x_res, r0 + rows_per_band);
        const int entries = (r1 - r0) * sh.y_res * F;
        const int lo = lo_n, units = (hi_n - lo_n) * 3;
        const int64_t item0 = (((int64_t)t * sh.x_res + r0) * sh.y_res) * F;
        float4 pp[kAdamTrips], mm[kAdamTrips], vv[kAdamTrips];
        if (ADAM) {
#pragma unroll
            for (int k = 0; k < kAdamTrips; ++k) {
                const int q = threadIdx.x + kSparseThreads * k;
                if (4 * q < entries) {
                    pp[k] = reinterpret_cast<const float4*>(ad.p + item0)[q];
                    mm[k] = reinterpret_cast<const float4*>(ad.m + item0)[q];
                    vv[k] = reinterpret_cast<const float4*>(ad.v + item0)[q];
                }
            }
        }
#ifndef NVP_SBAND_PRIO
#define NVP_SBAND_PRIO 0         // experiment: wave priority of the accumulate phase (1) or of the flush (2) of the sparse band kernel
#endif
        if (NVP_SBAND_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        if (NVP_SBAND_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        unit_apply(un[0]);
        unit_apply(un[1]);
        for (int u = threadIdx.x + 2 * kSparseThreads; u < units; u += kSparseThreads) {      // rare: more than 512 units
            SparseUnit<F> x;
            const int q = u / 3;
            unit_load(x, lo + q, u - q * 3, r0, r1);
            unit_apply(x);
        }
        if (NVP_SBAND_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (NVP_SBAND_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        __syncthreads();
        if (item + (int)gridDim.x < n_items) prefetch(item + gridDim.x);          // in flight during the flush
        if (ADAM) {
#pragma unroll
            for (int k = 0; k < kAdamTrips; ++k) {
                const int q = threadIdx.x + kSparseThreads * k;
                if (4 * q < entries) {
                    ulonglong2* t2 = reinterpret_cast<ulonglong2*>(tab + 4 * q);
                    const ulonglong2 a = t2[0], b = t2[1];
                    t2[0] = make_ulonglong2(0ull, 0ull); t2[1] = make_ulonglong2(0ull, 0ull);          // ready for the next item
                    const float nanv = __uint_as_float(0x7fc00000u);
                    const float g0 = poison ? nanv : (float)((double)(long long)a.x * inv), g1 = poison ? nanv : (float)((double)(long long)a.y * inv);
                    const float g2 = poison ? nanv : (float)((double)(long long)b.x * inv), g3 = poison ? nanv : (float)((double)(long long)b.y * inv);
                    adam1(pp[k].x, g0, mm[k].x, vv[k].x, ad.S);
                    adam1(pp[k].y, g1, mm[k].y, vv[k].y, ad.S);
                    adam1(pp[k].z, g2, mm[k].z, vv[k].z, ad.S);
                    adam1(pp[k].w, g3, mm[k].w, vv[k].w, ad.S);
                    reinterpret_cast<float4*>(ad.p + item0)[q] = pp[k];
                    reinterpret_cast<float4*>(ad.m + item0)[q] = mm[k];
                    reinterpret_cast<float4*>(ad.v + item0)[q] = vv[k];
                }
            }
        } else {
            float* out = demb + item0;
            for (int i = threadIdx.x; i < entries; i += kSparseThreads) {
                const long long v = (long long)tab[i];
                tab[i] = 0ull;                                                          // ready for the next item
                out[i] = poison ? __uint_as_float(0x7fc00000u) : (float)((double)v * inv);
            }
        }
        __syncthreads();
    }
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
// batches), every plane's per-level row tables, the sparse grid's (t, x) order and row table.  A dozen small latency-bound kernels
// that a host can run early on a side stream - underneath the gather kernel - through nvp_encode_bwd_presort.
int presort(const float* coords, int64_t n, const nvp_levels* lv[3], const nvp_sparse_shape* sh, char* ws, const Ws& W, const Plan& P, int flags, hipStream_t s) {
    unsigned* ky = (unsigned*)(ws + W.keys[0]);
    unsigned* kx = (unsigned*)(ws + W.keys[1]);
    const bool y_sorted = (flags & NVP_COORDS_SORTED_BY_Y) != 0;
    const bool planes_ready = y_sorted && (flags & NVP_DZ_PLANES_READY) != 0;
    hipLaunchKernelGGL(keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, coords, y_sorted ? (unsigned*)nullptr : ky, kx,
                       planes_ready ? (float2*)(ws + W.cs[0]) : (float2*)nullptr, planes_ready ? (float2*)(ws + W.cs[1]) : (float2*)nullptr,
                       *lv[0], *lv[1], *lv[2], n);
    int rc = 0;
    if (!y_sorted)                                 // otherwise the batch already arrives in ascending y: identity order
        rc = csort(ky, n, W.nkeys[0], (int*)(ws + W.kstart[0]), (unsigned*)(ws + W.kcursor[0]), (unsigned*)(ws + W.tsum), (int*)(ws + W.order[0]), s);
    if (rc) return rc;
    rc = csort(kx, n, W.nkeys[1], (int*)(ws + W.kstart[1]), (unsigned*)(ws + W.kcursor[1]), (unsigned*)(ws + W.tsum), (int*)(ws + W.order[1]), s);
    if (rc) return rc;
    // per-level row tables of the three planes (binary searches over the coordinates in each plane's order)
    {
        RowArgs RA;
        int rs_max = 0;
        const int ord[3] = {0, 0, 1}, col[3] = {2, 2, 1};
        for (int p = 0; p < 3; ++p) {
            RA.order[p] = (ord[p] == 0 && y_sorted) ? (const int*)nullptr : (const int*)(ws + W.order[ord[p]]);
            RA.col[p] = col[p];
            RA.rowstart[p] = (int*)(ws + W.rowstart[p]);
            RA.lv[p] = *lv[p];
            for (int l = 0; l < lv[p]->n_levels; ++l) RA.rs_off[p][l] = P.lp[p][l].rs_off;
            RA.rs_total[p] = P.rs_total[p];
            if (P.rs_total[p] > rs_max) rs_max = P.rs_total[p];
        }
        hipLaunchKernelGGL(rowstart_kernel, dim3((rs_max + 255) / 256, 3), dim3(256), 0, s, coords, RA, n);
    }
    unsigned* sk = (unsigned*)(ws + W.skey);
    hipLaunchKernelGGL(sparse_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, coords, sk, n, *sh);
    // the counting sort's exclusive starts are the sparse row table
    return csort(sk, n, sh->t_res * sh->x_res, (int*)(ws + W.srowstart), (unsigned*)(ws + W.scursor), (unsigned*)(ws + W.tsum), (int*)(ws + W.sorder), s);
}

// Geometry of the sparse band kernel's LDS table and every "this layout is not supported" decision of the sparse half, in ONE place:
// launch_all runs it BEFORE anything is enqueued, and nvp_encode_bwd_sparse_adamw's "NVP_ERR_UNSUPPORTED = nothing enqueued" promise
// rests on the same function (same clamp of `rows` to x_res).
struct SparseGeom { int sentries, rows; };
int sparse_geom(const nvp_sparse_shape* sh, const SparseAdam* adam, SparseGeom& G) {
    const int row_floats = sh->y_res * sh->n_features;
    if (row_floats < 1 || sh->x_res < 1 || sh->t_res < 1) return NVP_ERR_BADARG;
    if (sh->n_features != 1 && sh->n_features != 2 && sh->n_features != 4 && sh->n_features != 8) return NVP_ERR_UNSUPPORTED;
    static const int s4 = env_int_or("NVP_SPARSE_ENTRIES_F4", NVP_SPARSE_ENTRIES_F4);
    const int starget = sh->n_features >= 4 ? s4 : kSparseEntries;
    G.sentries = starget > row_floats ? starget : row_floats;
    if (G.sentries > kMaxLdsEntries) return NVP_ERR_UNSUPPORTED;              // one x-row does not fit the LDS
    G.rows = G.sentries / row_floats;
    if (G.rows > sh->x_res) G.rows = sh->x_res;
    if (adam) {
        // whole float4 per table quad, 16-B aligned tensors, the item's table within kAdamTrips trips of the workgroup
        if ((row_floats & 3) || G.rows * row_floats > 4 * kSparseThreads * kAdamTrips ||
            (((uintptr_t)adam->p | (uintptr_t)adam->m | (uintptr_t)adam->v) & 15)) return NVP_ERR_UNSUPPORTED;
    }
    return 0;
}

template <int F>
int launch_all(const float* coords, const float* dz, int dz_stride, float* g0, float* g1, float* g2, float* demb, int64_t n,
               const nvp_levels* lv[3], const nvp_sparse_shape* sh, char* ws, const Ws& W, const Plan& P, int flags, hipStream_t s,
               const SparseAdam* adam = nullptr, const DenseAdam* dadam = nullptr) {
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
    SparseGeom SG{0, 0};
    if (do_sparse) {                                                  // every support decision of the sparse half BEFORE anything is enqueued
        const int rcg = sparse_geom(sh, adam, SG);
        if (rcg) return rcg;
    }
    const bool presorted = (flags & NVP_SCATTER_PRESORTED) != 0;      // nvp_encode_bwd_presort already ran on this workspace
    if (!presorted) {
        int rc0 = presort(coords, n, lv, sh, ws, W, P, flags, s);
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
            PA.order[p] = (ord[p] == 0 && y_sorted) ? (const int*)nullptr : (const int*)(ws + W.order[ord[p]]);
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

        BandArgs BA;
        memset(&BA.ad, 0, sizeof(BA.ad));
        if (dadam) BA.ad = *dadam;
        BA.plan = P;
        float* grads[3] = {g0, g1, g2};
        for (int p = 0; p < 3; ++p) {
            BA.lv[p] = *lv[p];
            BA.cs[p] = (const float2*)(ws + W.cs[p]);
            BA.dzs[p] = (const float*)(ws + W.dzs[p]);
            BA.rowstart[p] = (const int*)(ws + W.rowstart[p]);
            BA.grad[p] = grads[p];
        }
        BA.slabs = (long long*)(ws + W.slabs);
        BA.dzmax = (const unsigned*)(ws + W.dzmax);
        BA.headroom_bits = headroom_bits;
        hipLaunchKernelGGL((band_kernel<F>), dim3(P.total_blocks), dim3(kBandThreads), (size_t)P.entries * 8, s, BA, n);

        if (P.reduce_items > 0) {
            ReduceArgs R;
            R.ad = BA.ad;
            R.plan = P;
            for (int p = 0; p < 3; ++p) { R.lv[p] = *lv[p]; R.grad[p] = grads[p]; }
            R.slabs = (const long long*)(ws + W.slabs);
            R.dzmax = (const unsigned*)(ws + W.dzmax);
            R.headroom_bits = headroom_bits;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(64, P.reduce_items), dim3(256), 0, s, R);
        }
        return 0;
    };

    // ---- sparse grid (its max|dz| slots were filled by the permute pass above, or by the chain kernel when planes_ready)
    auto sparse = [&]() -> int {
        int* sorder = (int*)(ws + W.sorder);
        int* srs = (int*)(ws + W.srowstart);
        unsigned* sdzmax = (unsigned*)(ws + W.sdzmax);
        const int sentries = SG.sentries, rows = SG.rows;                  // sparse_geom(), checked before the presort
        const int bands = (sh->x_res + rows - 1) / rows;
        const int n_items = sh->t_res * bands;
        // persistent workgroups: as many as fit the chip at once (LDS: 160 KB / table, at most 8 per CU of 256 CUs)
        int per_cu = (int)((160 * 1024) / ((size_t)sentries * 8 + 64));
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        const int grid = n_items < 256 * per_cu ? n_items : 256 * per_cu;
        const size_t lds = (size_t)rows * sh->y_res * sh->n_features * 8;
        SparseAdam ad{};
        if (adam) ad = *adam;
        switch (sh->n_features) {
#define NVP_SPARSE_CASE(FF) case FF: \
            if (adam) hipLaunchKernelGGL((sparse_band_kernel<FF, true>), dim3((unsigned)grid), dim3(kSparseThreads), lds, s, coords, dz, dz_stride, scol0, \
                           (const int*)sorder, (const int*)srs, (const unsigned*)sdzmax, demb, *sh, rows, bands, n_items, headroom_bits + 2, ad); \
            else hipLaunchKernelGGL((sparse_band_kernel<FF, false>), dim3((unsigned)grid), dim3(kSparseThreads), lds, s, coords, dz, dz_stride, scol0, \
                           (const int*)sorder, (const int*)srs, (const unsigned*)sdzmax, demb, *sh, rows, bands, n_items, headroom_bits + 2, ad); \
            break;
            NVP_SPARSE_CASE(1) NVP_SPARSE_CASE(2) NVP_SPARSE_CASE(4) NVP_SPARSE_CASE(8)
#undef NVP_SPARSE_CASE
            default: return NVP_ERR_UNSUPPORTED;
        }
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
    return presort(coords, n, lv, sh, (char*)workspace, W, P, flags, (hipStream_t)stream);
}

// dz: row-major latent gradient [>= n][dz_stride] (columns xy | yt | xt | sparse).
// d_kf_* and d_emb: every element is OVERWRITTEN (no zero-fill needed).
static int encode_bwd_impl(const float* coords, const float* dz, int32_t dz_stride,
                   float* d_kf_xy, float* d_kf_yt, float* d_kf_xt, float* d_emb, int64_t n,
                   const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                   const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags, void* stream, const SparseAdam* adam,
                   const DenseAdam* dadam = nullptr) {
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
    // a split call only needs the outputs it produces; the fused-optimizer call produces no d_emb at all
    const bool want_dense = !(flags & NVP_SCATTER_SPARSE_ONLY) && !dadam, want_emb = !(flags & NVP_SCATTER_DENSE_ONLY) && !adam;
    if (!coords || !dz || !workspace || (want_dense && (!d_kf_xy || !d_kf_yt || !d_kf_xt)) || (want_emb && !d_emb)) return NVP_ERR_BADARG;
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
        case 1: rc = launch_all<1>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s, adam, dadam); break;
        case 2: rc = launch_all<2>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s, adam, dadam); break;
        case 4: rc = launch_all<4>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s, adam, dadam); break;
        case 8: rc = launch_all<8>(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv, sh, (char*)workspace, W, P, flags, s, adam, dadam); break;
        default: return NVP_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    NVP_LAUNCH_CHECK();
    return 0;
}

int nvp_encode_bwd(const float* coords, const float* dz, int32_t dz_stride,
                   float* d_kf_xy, float* d_kf_yt, float* d_kf_xt, float* d_emb, int64_t n,
                   const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                   const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags, void* stream) {
    return encode_bwd_impl(coords, dz, dz_stride, d_kf_xy, d_kf_yt, d_kf_xt, d_emb, n, lv_xy, lv_yt, lv_xt, sh, workspace, workspace_bytes, flags, stream, nullptr);
}

// The sparse grid's half of the split scatter (NVP_SCATTER_SPARSE_ONLY semantics: needs NVP_COORDS_SORTED_BY_Y |
// NVP_DZ_PLANES_READY) with the optimizer step applied in the flush: no d_emb is produced, `emb` / `exp_avg` / `exp_avg_sq` are
// updated in place exactly as nvp_adamw_step would update them from that gradient (same arithmetic: adamw.h).
// NVP_ERR_UNSUPPORTED (nothing enqueued) if y_res * F is not a multiple of 4, a tensor is not 16-B aligned or a table exceeds
// 4 096 entries: the caller then takes the two-kernel route.
int nvp_encode_bwd_sparse_adamw(const float* coords, const float* dz, int32_t dz_stride, int64_t n,
                                const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                                const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags,
                                float* emb, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                                double weight_decay, int64_t step, void* stream) {
    if (!emb || !exp_avg || !exp_avg_sq || step < 1 || n < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1)) return NVP_ERR_BADARG;
    if (!(flags & NVP_DZ_PLANES_READY) || !(flags & NVP_COORDS_SORTED_BY_Y) || (flags & NVP_SCATTER_DENSE_ONLY)) return NVP_ERR_BADARG;
    if (!sh) return NVP_ERR_BADARG;
    SparseAdam ad;
    ad.p = emb; ad.m = exp_avg; ad.v = exp_avg_sq;
    ad.S = adam_scalars(lr, beta1, beta2, eps, weight_decay, step, 1.0);
    // (launch_all repeats this check - the same function - before its first launch: UNSUPPORTED always means nothing was enqueued)
    SparseGeom G;
    const int rcg = sparse_geom(sh, &ad, G);
    if (rcg) return rcg;
    return encode_bwd_impl(coords, dz, dz_stride, nullptr, nullptr, nullptr, nullptr, n, lv_xy, lv_yt, lv_xt, sh, workspace, workspace_bytes,
                           flags | NVP_SCATTER_SPARSE_ONLY, stream, &ad);
}

// The dense planes' half of the split scatter (NVP_SCATTER_DENSE_ONLY semantics: needs NVP_COORDS_SORTED_BY_Y | NVP_DZ_PLANES_READY)
// with the optimizer step applied in the flushes: no d_kf_* is produced; params[q] / exp_avg[q] / exp_avg_sq[q] (q = xy, yt, xt) are
// updated in place exactly as nvp_adamw_step would update them from that gradient (adamw.h).  step[q]: the 1-based step count of
// plane q.  Together with nvp_encode_bwd_sparse_adamw no optimizer launch is left for the four grids (one GPU).
int nvp_encode_bwd_dense_adamw(const float* coords, const float* dz, int32_t dz_stride, int64_t n,
                               const nvp_levels* lv_xy, const nvp_levels* lv_yt, const nvp_levels* lv_xt,
                               const nvp_sparse_shape* sh, void* workspace, int64_t workspace_bytes, int32_t flags,
                               float* const* params, float* const* exp_avg, float* const* exp_avg_sq, double lr, double beta1, double beta2,
                               double eps, double weight_decay, const int64_t* step, void* stream) {
    if (!params || !exp_avg || !exp_avg_sq || !step || n < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1)) return NVP_ERR_BADARG;
    if (!(flags & NVP_DZ_PLANES_READY) || !(flags & NVP_COORDS_SORTED_BY_Y) || (flags & NVP_SCATTER_SPARSE_ONLY)) return NVP_ERR_BADARG;
    DenseAdam ad;
    for (int q = 0; q < 3; ++q) {
        if (!params[q] || !exp_avg[q] || !exp_avg_sq[q] || step[q] < 1) return NVP_ERR_BADARG;
        ad.p[q] = params[q]; ad.m[q] = exp_avg[q]; ad.v[q] = exp_avg_sq[q];
        ad.S[q] = adam_scalars(lr, beta1, beta2, eps, weight_decay, step[q], 1.0);
    }
    ad.on = 1;
    return encode_bwd_impl(coords, dz, dz_stride, nullptr, nullptr, nullptr, nullptr, n, lv_xy, lv_yt, lv_xt, sh, workspace, workspace_bytes,
                           flags | NVP_SCATTER_DENSE_ONLY, stream, nullptr, &ad);
}

}  // extern "C"
