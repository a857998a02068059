// LDS-staged gather for batches that arrive sorted by their y coordinate (NVP_COORDS_SORTED_BY_Y; nvp_amd's own
// sampler delivers them that way, and NVPFused sorts training-size batches that do not).
//
// R2 for the two planes whose ROW coordinate is y: xy <- (x, y) and yt <- (t, y) (modules.py:61,63; cell = i0 + i1 * res,
// i1 from y).  A workgroup owns a run of 256 consecutive sorted pixels.  Their y values span one or two image columns, so
// at every level all of the run's bilinear corners lie in at most three consecutive grid rows: that cell range is loaded
// ONCE with coalesced full-width loads (consecutive lanes, consecutive cells) into LDS, and the 256 x 4 corner fetches per
// level become LDS reads.  Compared with the per-lane global gather (encode.hip) this replaces ~1 KB per pixel and plane
// of 8-byte random L1/TA requests (each dragging a 64-byte sector) by ~0.4 KB per pixel and plane of streaming reads.
//
// Levels are processed in groups whose staged ranges fit the LDS budget together (one barrier pair per group).  A level
// whose range does not fit its slot (small or unsorted-in-practice batches: the run spans many columns), or whose range
// wraps around the end of the level (y == 1 rows), is gathered from global memory by the same threads - results are
// identical either way, bit for bit: both paths run nvp_bilerp_setup / nvp_blend4 (grid_math.h).
//
// The xt plane (both coordinates random within a run) and the sparse 3x3 grid stay on the global gather kernel
// (encode.hip, slot mask); the two launches write disjoint row ranges of the PTM4 latent.
#include "grid_math.h"

#pragma clang fp contract(off)

namespace {

constexpr int kRun = 256;                 // pixels per workgroup = threads
constexpr int kMaxGroups = 8;

struct LdsPlan {
    int n_groups;
    int first[kMaxGroups + 1];            // levels [first[g], first[g+1])
    int slot[NVP_MAX_LEVELS];             // float offset of the level's slot inside the LDS buffer
    int cap[NVP_MAX_LEVELS];              // capacity of the slot in cells (3 * res + 1)
};

struct LdsArgs {
    nvp_levels lv[2];                     // xy, yt
    LdsPlan plan[2];
    int col0[2];                          // first latent row of the plane
    int c0[2];                            // coordinate column feeding dim 0 of the plane (x for xy, t for yt); dim 1 is y
    int rows;                             // PTM4 rows of the latent
};

template <int F>
__device__ __forceinline__ void load_cell(float (&v)[F], const float* __restrict__ p) {
    if constexpr (F == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else if constexpr (F == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else {
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = p[f];
    }
}

template <int F>
__global__ __launch_bounds__(kRun) void encode_fwd_lds_kernel(const float* __restrict__ coords, const float* __restrict__ kf_xy,
                                                              const float* __restrict__ kf_yt, float* __restrict__ zt,
                                                              int64_t n, int64_t npad, LdsArgs A) {
    extern __shared__ __attribute__((aligned(16))) float stage[];
    const int64_t p0 = (int64_t)blockIdx.x * kRun;
    const int64_t px = p0 + threadIdx.x;
    const bool valid = px < n;
    float c[3] = {0.f, 0.f, 0.f};
    if (valid) { c[0] = coords[px * 3]; c[1] = coords[px * 3 + 1]; c[2] = coords[px * 3 + 2]; }
    // y of the run's first and last pixel (ascending order: they bound every y in between) - uniform across the workgroup
    const int64_t pf = min(p0, n - 1), pl = min(p0 + kRun, n) - 1;
    const float y_first = coords[pf * 3 + 2], y_last = coords[max(pl, pf) * 3 + 2];
    float4* __restrict__ zt4 = reinterpret_cast<float4*>(zt);
    const int64_t zbase = ((px >> 5) * (int64_t)(A.rows >> 2)) * 32 + (px & 31);

#pragma unroll 1
    for (int plane = 0; plane < 2; ++plane) {
        const nvp_levels& lv = A.lv[plane];
        const LdsPlan& P = A.plan[plane];
        const float* __restrict__ params = plane == 0 ? kf_xy : kf_yt;
        const float x0 = c[A.c0[plane]], x1 = c[2];
        const int flags = lv.flags;
        float stash[2] = {0.f, 0.f};            // F == 2: the even level of a pair waits for the odd one (one 16-B store per pair)
        // the staged cell range of level l: rows r_lo .. r_hi + 2 (+1 cell), see the header comment
        auto range = [&](int l, int& c_lo, int& len) -> bool {
            const int res = lv.res[l];
            const int r_lo = (int)floorf(nvp_grid_pos(y_first, lv.scale[l], flags));
            const int r_hi = (int)floorf(nvp_grid_pos(y_last, lv.scale[l], flags));
            c_lo = r_lo * res;
            len = (r_hi - r_lo + 2) * res + 1;
            return r_lo >= 0 && r_hi >= r_lo && len <= P.cap[l] && c_lo + len <= res * res;
        };
#pragma unroll 1
        for (int g = 0; g < P.n_groups; ++g) {
            const int l0 = P.first[g], l1 = P.first[g + 1];
            // ---- stage: coalesced loads, consecutive lanes on consecutive cells
            for (int l = l0; l < l1; ++l) {
                int c_lo, len;
                if (!range(l, c_lo, len)) continue;
                const float* __restrict__ src = params + ((int64_t)lv.offset[l] + c_lo) * F;
                float* __restrict__ dst = stage + P.slot[l];
                // eight loads in flight per thread before the first LDS write (a load -> write loop serialises on the memory
                // latency: 17 round trips for the widest level)
                if (len <= 2 * kRun) {                           // coarse levels: at most two cells per thread
                    float v[2][F];
                    const int i = threadIdx.x;
                    load_cell<F>(v[0], src + (int64_t)min(i, len - 1) * F);
                    load_cell<F>(v[1], src + (int64_t)min(i + kRun, len - 1) * F);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (i + u * kRun < len) {
#pragma unroll
                            for (int f = 0; f < F; ++f) dst[(i + u * kRun) * F + f] = v[u][f];
                        }
                    continue;
                }
                for (int i0 = threadIdx.x; i0 < len; i0 += 8 * kRun) {
                    float v[8][F];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * kRun;
                        load_cell<F>(v[u], src + (int64_t)min(i, len - 1) * F);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * kRun;
                        if (i < len) {
#pragma unroll
                            for (int f = 0; f < F; ++f) dst[i * F + f] = v[u][f];
                        }
                    }
                }
            }
            __syncthreads();
            // ---- interpolate: thread = pixel
            for (int l = l0; l < l1; ++l) {
                int c_lo, len;
                const bool staged = range(l, c_lo, len);
                float out[F];
#pragma unroll
                for (int f = 0; f < F; ++f) out[f] = 0.f;
                if (valid) {
                    const NvpBilerp b = nvp_bilerp_setup(x0, x1, lv.scale[l], lv.res[l], flags);
                    float v[4][F];
                    const float* __restrict__ gbase = params + (int64_t)lv.offset[l] * F;
                    const float* __restrict__ sbase = stage + P.slot[l];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rel = b.cell[k] - c_lo;
                        if (staged && rel >= 0 && rel < len) {
#pragma unroll
                            for (int f = 0; f < F; ++f) v[k][f] = sbase[rel * F + f];
                        } else {
                            load_cell<F>(v[k], gbase + (int64_t)b.cell[k] * F);
                        }
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) out[f] = nvp_blend4(b.w, v[0][f], v[1][f], v[2][f], v[3][f], flags);
                }
                // ---- PTM4 store of rows col .. col + F - 1 (pixels of the last tile beyond n are zero-filled)
                if (px < npad) {
                    const int col = A.col0[plane] + l * F;
                    if constexpr (F >= 4) {
#pragma unroll
                        for (int q = 0; q < F / 4; ++q)
                            zt4[zbase + (int64_t)((col >> 2) + q) * 32] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
                    } else if constexpr (F == 2) {
                        if ((col & 3) == 0 && l + 1 < lv.n_levels) { stash[0] = out[0]; stash[1] = out[1]; }
                        else if ((col & 3) == 2) zt4[zbase + (int64_t)(col >> 2) * 32] = make_float4(stash[0], stash[1], out[0], out[1]);
                        else { float* z = zt + (zbase + (int64_t)(col >> 2) * 32) * 4; z[0] = out[0]; z[1] = out[1]; }
                    } else {
                        zt[(zbase + (int64_t)(col >> 2) * 32) * 4 + (col & 3)] = out[0];
                    }
                }
            }
            __syncthreads();
        }
    }
}

// greedy grouping of a plane's levels under the LDS budget (floats); a level that cannot get a slot has cap 0 and is
// gathered from global memory
void make_lds_plan(LdsPlan& P, const nvp_levels& lv, int budget_floats) {
    const int F = lv.n_features;
    int ng = 0, used = 0;
    P.first[0] = 0;
    for (int l = 0; l < lv.n_levels; ++l) {
        const int cap = 3 * lv.res[l] + 1;
        const int need = (cap * F + 3) & ~3;
        P.slot[l] = 0;
        P.cap[l] = 0;
        if (need > budget_floats) continue;
        if (used + need > budget_floats && ng + 1 < kMaxGroups) { P.first[++ng] = l; used = 0; }
        if (used + need > budget_floats) continue;
        P.slot[l] = used;
        P.cap[l] = cap;
        used += need;
    }
    P.n_groups = ng + 1;
    P.first[ng + 1] = lv.n_levels;
}

}  // namespace

// xy and yt planes of nvp_encode_fwd for y-sorted batches (called from encode.hip).  Returns 0 or a hipError_t.
int nvp_encode_fwd_lds_launch(const float* coords, const float* kf_xy, const float* kf_yt, float* zt, int64_t n, int64_t npad,
                              const nvp_levels* lv_xy, const nvp_levels* lv_yt, int col0_xy, int col0_yt, int rows, hipStream_t stream) {
    LdsArgs A;
    A.lv[0] = *lv_xy; A.lv[1] = *lv_yt;
    A.col0[0] = col0_xy; A.col0[1] = col0_yt;
    A.c0[0] = 1; A.c0[1] = 0;              // xy <- (x, y) = coords[:, (1, 2)]; yt <- (t, y) = coords[:, (0, 2)]   (modules.py:61,63)
    A.rows = rows;
    const int F = lv_xy->n_features;
    // budget: 40 KB (four workgroups per CU) unless the widest level needs more on its own
    int budget = 40 * 1024 / 4;
    for (int p = 0; p < 2; ++p)
        for (int l = 0; l < A.lv[p].n_levels; ++l) {
            const int need = (((3 * A.lv[p].res[l] + 1) * A.lv[p].n_features) + 3) & ~3;
            if (need > budget && need * 4 <= 150 * 1024) budget = need;
        }
    make_lds_plan(A.plan[0], A.lv[0], budget);
    make_lds_plan(A.plan[1], A.lv[1], budget);
    const dim3 grid((unsigned)((npad + kRun - 1) / kRun));
    const size_t lds = (size_t)budget * 4;
    switch (F) {
        case 1: hipLaunchKernelGGL((encode_fwd_lds_kernel<1>), grid, dim3(kRun), lds, stream, coords, kf_xy, kf_yt, zt, n, npad, A); break;
        case 2: hipLaunchKernelGGL((encode_fwd_lds_kernel<2>), grid, dim3(kRun), lds, stream, coords, kf_xy, kf_yt, zt, n, npad, A); break;
        case 4: hipLaunchKernelGGL((encode_fwd_lds_kernel<4>), grid, dim3(kRun), lds, stream, coords, kf_xy, kf_yt, zt, n, npad, A); break;
        case 8: hipLaunchKernelGGL((encode_fwd_lds_kernel<8>), grid, dim3(kRun), lds, stream, coords, kf_xy, kf_yt, zt, n, npad, A); break;
        default: return NVP_ERR_UNSUPPORTED;
    }
    hipError_t e = hipGetLastError();
    return (int)e;
}
