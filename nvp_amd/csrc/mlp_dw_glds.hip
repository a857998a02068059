// Weight gradients of modulator layer k and SIREN layer k (k = 1, 2) with every operand stream read ONCE, fed to LDS by DMA and
// split into 16-bit MFMA operands ONCE.
//
// Why.  As seven independent 256-thread jobs (mlp_dw.hip) the dW stage requests 7.5 KB per pixel for 4.56 KB of distinct operand
// streams - jobs that share a stream each fetch it (PMC: 8.8 GB per step) - and every operand value is split into its fp16 hi / lo
// parts by each of the two (dp: four) waves that consume it: bytes and VALU instructions are what the stage costs.  Here ONE
// 768-thread workgroup per pixel chunk owns the three jobs of a layer
//     waves 0-3: dp_k x h_{k-1}      waves 4-7: dp_k x z      waves 8-11: dq_k x x_{k-1},   x = sin(.) h
// (2.0 / 2.5 KB per pixel for k = 1 / 2) and works in three phases per 16-pixel step:
//   1. DMA.  global_load_lds (16 B per lane, no registers) brings the raw PTM4 half tiles of dp, dq, h, (q,) z (+ the 16 temporal
//      steps) into a two-stage ring; one stage is always in flight across the barriers (raw s_barrier + counted vmcnt;
//      tools/probes/glds_stream_probe.hip: this access pattern streams at 6.0 TB/s from one workgroup per CU).
//   2. Stage.  Every thread takes the 16-byte units its own DMA instructions brought (4 rows x 1 pixel each), feeds the tiles'
//      running maxima (the power-of-two block scales of the fp16 x 2 split, mlp_b3.h), splits each value ONCE, builds x = sin(.) h
//      ONCE, and writes the hi / lo halves into fragment-shaped planes [row][16 px]: a lane's 8 consecutive k of one row are
//      16 contiguous bytes.
//   3. MFMA.  Each wave reads its two A and two B fragments as ds_read_b128 (hi, lo) and issues the 12 products of its 64 x 64
//      sub-block: no split, no sine, no address arithmetic in the wave that multiplies.
// Bias gradients (row sums of dp / dq) are accumulated by the staging threads, which see every value exactly once.
//
// Arithmetic: split-operand 16-bit MFMA under a running power-of-two block scale per operand tile, as mlp_dw_kernel; x_{k-1}
// shares h_{k-1}'s scale (|x| <= |h|).  Results equal the per-job kernels' to fp32 summation order, not bit for bit.
//
// All LDS traffic of the main loop is inline assembly: hipcc's waitcnt insertion treats a pending LDS DMA as a write to ALL of
// LDS and puts s_waitcnt vmcnt(0) in front of every LDS access it can see, which would drain the ring every step.
#include <cstdlib>
#include "mlp_b3.h"

#ifndef NVP_DW_B3
#define NVP_DW_B3 1
#endif
#ifndef NVP_GL_ABL
#define NVP_GL_ABL 0        // ablation builds only (timing, wrong results): 1 = no staging pass; 2 = also no fragment reads / MFMAs
#endif

namespace {

constexpr int kThreads = 768, kWaves = 12, kStages = 2;
constexpr int kPlaneTile = 8192;                 // one operand tile in the planes: hi [128 rows][16 px] halves (4 KB), then lo
constexpr int kPlanes = 5 * kPlaneTile;          // dp, dq, h, x, z

struct GArgs {
    const float* a1;          // dp_k   (PTM4, 128 rows)
    const float* a2;          // dq_k
    const float* h;           // h_{k-1}
    const float* q;           // q_{k-1} (K == 2)
    const float* z;           // latent (PTM4, z_rows rows, d valid)
    const float* steps;       // (K == 1)
    const float* sir0_wp;     // SIREN layer 0 weight / bias (K == 1 rebuilds x_0 from them)
    const float* sir0_bp;
    int z_rows, d, ld_mod;
    int64_t w_h, w_z, w_x, b_mod, b_sir, total;
};

// raw stage layout in 16-byte units, every stream on a wave-instruction (64-unit) boundary so that a DMA instruction has ONE
// source stream: [dp 512 | dq 512 | h 512 | (q 512) | z 512 (16 rgz used) | (steps 64, 4 used)]; unit = (row-group, pixel)
template <int K> struct Lay {
    static constexpr int oA1 = 0, oA2 = 512, oH = 1024, oQ = 1536;
    static constexpr int oZ = K == 2 ? 2048 : 1536;
    static constexpr int oS = oZ + 512;                       // K == 1: the 16 temporal steps of the half tile
    static constexpr int units = K == 2 ? oZ + 512 : oS + 64;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p); }
// global -> LDS DMA of 16 bytes per lane: LDS destination = M0 (wave-uniform byte address) + 16 x lane; source = uniform 64-bit base
// (scalar registers) + a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16(const void* ubase, unsigned lane_off, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(ubase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ float lds_rd(unsigned addr) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); return v; }
template <int O> __device__ __forceinline__ f32x4v lds_rd128(unsigned addr) { f32x4v v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(O)); return v; }
template <int O> __device__ __forceinline__ u32x4 lds_rdq(unsigned addr) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(O)); return v; }
template <int O> __device__ __forceinline__ void lds_w16(unsigned addr, unsigned v) { asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(O) : "memory"); }
template <int O> __device__ __forceinline__ void lds_w16hi(unsigned addr, unsigned v) { asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(O) : "memory"); }
__device__ __forceinline__ void lds_umax(unsigned addr, unsigned v) { asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {            // n in [0, 4], wave-uniform
    if (n >= 4) wait_vm<4>(); else if (n == 3) wait_vm<3>(); else if (n == 2) wait_vm<2>(); else if (n == 1) wait_vm<1>(); else wait_vm<0>();
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// the four values of a unit (rows 4 rg .. + 3 of one pixel), scaled and split, into the hi / lo planes of one tile:
// `pa` = LDS byte address of (row 4 rg, this pixel) in the tile's hi plane
__device__ __forceinline__ void split_store4(unsigned pa, const float (&v)[4], float s) {
    const f16x2 h01 = {(_Float16)(v[0] * s), (_Float16)(v[1] * s)}, h23 = {(_Float16)(v[2] * s), (_Float16)(v[3] * s)};
    const f16x2 l01 = {(_Float16)__builtin_fmaf(v[0], s, -(float)h01.x), (_Float16)__builtin_fmaf(v[1], s, -(float)h01.y)};
    const f16x2 l23 = {(_Float16)__builtin_fmaf(v[2], s, -(float)h23.x), (_Float16)__builtin_fmaf(v[3], s, -(float)h23.y)};
    const unsigned a = __builtin_bit_cast(unsigned, h01), b = __builtin_bit_cast(unsigned, h23);
    const unsigned c = __builtin_bit_cast(unsigned, l01), d = __builtin_bit_cast(unsigned, l23);
    lds_w16<0>(pa, a); lds_w16hi<32>(pa, a); lds_w16<64>(pa, b); lds_w16hi<96>(pa, b);                       // rows are 32 bytes apart
    lds_w16<4096>(pa, c); lds_w16hi<4096 + 32>(pa, c); lds_w16<4096 + 64>(pa, d); lds_w16hi<4096 + 96>(pa, d);
}

template <int K>
__global__ __launch_bounds__(kThreads, 1) void mlp_dw_glds_kernel(GArgs A, float* __restrict__ partials, int64_t n, int64_t ntiles, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = wall >> 2, w = wall & 3;                 // 0: dp x h, 1: dp x z, 2: dq x x
    const int i = lane & 31, hh = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const int rgz = A.z_rows >> 2;                           // row-groups of the latent
    constexpr int stage_bytes = Lay<K>::units * 16;
    constexpr int q_z0 = Lay<K>::oZ >> 6;                    // first DMA instruction of the latent
    const int n_z = (rgz * 16 + 63) >> 6;                    // ... and how many it takes
    constexpr int n_slots = Lay<K>::units >> 6;
    auto instr_live = [&](int qi) { return qi < n_slots && (qi < q_z0 + n_z || (K == 1 && qi == (Lay<K>::oS >> 6))); };
    int my_instr = 0;                                        // DMA instructions this wave issues per stage (wave-uniform, 2..4)
#pragma unroll
    for (int m = 0; m < 4; ++m) my_instr += instr_live(wall + kWaves * m) ? 1 : 0;

    const unsigned l0 = lds_addr(lds);
    const unsigned planes = l0 + kStages * stage_bytes;
    const unsigned mx_a = planes + kPlanes;                  // running maxima: dp, dq, h, z (bit patterns)
    unsigned* mx = reinterpret_cast<unsigned*>(lds + kStages * stage_bytes + kPlanes);
    float* tab = reinterpret_cast<float*>(lds + kStages * stage_bytes + kPlanes + 32);      // K == 1: SIREN-0 weight / bias

    const int64_t s0 = (int64_t)blockIdx.x * tiles_per_chunk * 2;                  // half tiles of this chunk
    const int64_t s1 = min(ntiles * 2, s0 + (int64_t)tiles_per_chunk * 2);
    const int nsteps = (int)(s1 - s0);
    float* part = partials + (int64_t)blockIdx.x * A.total;

    if (tid < 8) mx[tid] = __float_as_uint(kTinyMax);
    if (K == 1 && tid < NVP_H) { tab[tid] = A.sir0_wp[tid]; tab[NVP_H + tid] = A.sir0_bp[tid]; }
    // the latent's rows past its end (and whatever an absent row-group would have brought) must read as zero in its planes
    for (int e = tid; e < kPlaneTile / 4; e += kThreads) reinterpret_cast<unsigned*>(lds + kStages * stage_bytes + 4 * kPlaneTile)[e] = 0u;
    __syncthreads();                                         // plain LDS traffic: before any DMA is in flight

    // ---- this thread's units: unit (qi, lane) = row-group 4 (qi & 7) + (lane >> 4) of stream qi >> 3, pixel lane & 15
    const int rgl = lane >> 4, px = lane & 15;
    // K == 1: SIREN-0 weight / bias of the four rows of this thread's h unit (qi = wall + 12 is an h instruction for waves 4-11)
    float w0r[4] = {0.f, 0.f, 0.f, 0.f}, c0r[4] = {0.f, 0.f, 0.f, 0.f};
    if (K == 1 && wall >= 4) {
        const int row = 4 * (4 * ((wall + kWaves) & 7) + rgl);
#pragma unroll
        for (int e = 0; e < 4; ++e) { w0r[e] = tab[row + e]; c0r[e] = tab[NVP_H + row + e]; }
    }
    float bacc[2][4];                                        // row sums of this thread's dp / dq units (m = 0: all waves; m = 1: waves 0-3)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) bacc[m][e] = 0.f;

    // ---- DMA of half tile s0 + step into stage step % kStages
    auto issue = [&](int step) {
        const int64_t ht = s0 + step;
        const int64_t tile = ht >> 1;
        const int half = (int)(ht & 1);
        const unsigned stage = (unsigned)__builtin_amdgcn_readfirstlane((int)(l0 + (unsigned)(step % kStages) * stage_bytes));
        const unsigned lane_src = (unsigned)(rgl * 512 + px * 16);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int qi = wall + kWaves * m;                 // wave-uniform
            if (!instr_live(qi)) continue;
            if (K == 1 && qi == (Lay<K>::oS >> 6)) {          // the 16 temporal steps: 4 lanes
                const unsigned e = (unsigned)tile * 32u + 16u * (unsigned)half + 4u * (unsigned)lane;      // (n < 2^31) n % 4 == 0: a lane's four steps are inside the batch or all beyond it
                const unsigned lo = e + 4u <= (unsigned)n ? e * 4u : 0u;          // beyond the batch: any finite value will do (those pixels' dY are 0)
                if (lane < 4) glds16(A.steps, lo, stage + (unsigned)(qi * 1024));
                continue;
            }
            const int sid = qi >> 3;                          // 0 dp, 1 dq, 2 h, (K == 2: 3 q,) then the latent
            const bool lat = qi >= q_z0;
            const float* base = lat ? A.z : (sid == 0 ? A.a1 : (sid == 1 ? A.a2 : (sid == 2 ? A.h : A.q)));
            const int rg0 = 4 * (qi & 7);                     // first row-group of this instruction
            const int64_t tbytes = lat ? (int64_t)rgz * 512 : 16384;
            const char* ub = reinterpret_cast<const char*>(base) + tile * tbytes + rg0 * 512 + half * 256;      // uniform
            if (!lat || rg0 + rgl < rgz) glds16(ub, lane_src, stage + (unsigned)(qi * 1024));
        }
    };

    // ---- MFMA-phase addressing: row i of a 32-row block, the 8 pixels of lane half hh = 16 bytes of a plane row
    const int tA = job == 2 ? 1 : 0, tB = job == 0 ? 2 : (job == 1 ? 4 : 3);
    const int slotA = job == 2 ? 1 : 0, slotB = job == 1 ? 3 : 2;
    const unsigned adA = planes + (unsigned)(tA * kPlaneTile + (64 * wr + i) * 32 + 16 * hh);      // + 1024 for the second 32-row block, + 4096 for lo
    const unsigned adB = planes + (unsigned)(tB * kPlaneTile + (64 * wc + i) * 32 + 16 * hh);

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = nvp_zero16();
    unsigned run[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) run[u] = __float_as_uint(kTinyMax);
    float curS, curU;
    { const PxScale q0 = px_scale(kTinyMax); curS = q0.s * q0.s; curU = q0.u * q0.u; }

    if (nsteps > 0) issue(0);
    for (int s = 0; s < nsteps; ++s) {
        const unsigned stage = l0 + (unsigned)(s % kStages) * stage_bytes;
        if (s + 1 < nsteps) issue(s + 1);                    // into the other stage: its readers finished in step s - 1 (barrier C below)
        wait_vm_n(s + 1 < nsteps ? my_instr : 0);            // this wave's share of stage s has landed
        __builtin_amdgcn_s_barrier();                        // A: everyone's share landed; everyone is past the MFMAs of step s - 1
#if NVP_GL_ABL == 0
        // ---- stage: own units -> registers, tile maxima
        f32x4v val[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) val[m] = lds_rd128<0>(stage + (unsigned)((wall + kWaves * m) * 1024 + lane * 16));      // (dead slots read in-bounds garbage)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int qi = wall + kWaves * m;
            if (qi >= q_z0 + n_z) continue;                   // steps / dead slots
            const int slot = qi < 8 ? 0 : (qi < 16 ? 1 : (qi < 24 ? 2 : (qi < q_z0 ? -1 : 3)));     // q needs no scale
            if (slot < 0) continue;
            const bool live = qi < q_z0 || 4 * (qi & 7) + rgl < rgz;
            const unsigned mm = live ? __float_as_uint(fmaxf(fmaxf(fabsf(val[m][0]), fabsf(val[m][1])), fmaxf(fabsf(val[m][2]), fabsf(val[m][3])))) : 0u;
            if (__any(mm > run[slot])) { const unsigned wm = wave_umax(mm); if (lane == 0) lds_umax(mx_a + 4 * slot, wm); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // B: the running maxima include stage s
        {
            float r4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r4[u] = lds_rd(mx_a + 4 * u);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]));
#pragma unroll
            for (int u = 0; u < 4; ++u) run[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r4[u]));
        }
        // ---- stage: split own units into the planes (x from h; bias sums from dp / dq)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int qi = wall + kWaves * m;
            if (qi >= q_z0 + n_z) continue;
            const int sid = qi >> 3;
            const bool lat = qi >= q_z0;
            if (!lat && K == 2 && sid == 3) continue;         // q: consumed by the owners of the h units
            const int rg = 4 * (qi & 7) + rgl;
            const float v[4] = {val[m][0], val[m][1], val[m][2], val[m][3]};
            const int tile_i = lat ? 4 : sid;                 // dp 0, dq 1, h 2, z 4 (x 3 below)
            const unsigned pa = planes + (unsigned)(tile_i * kPlaneTile + rg * 128 + px * 2);
            const float sc = px_scale(__uint_as_float(run[lat ? 3 : sid])).s;
            if (!lat || rg < rgz) split_store4(pa, v, sc);
            if (!lat && sid < 2 && m < 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bacc[m][e] += v[e];
            }
            if (!lat && sid == 2) {                           // x_{k-1} = sin(.) * h_{k-1}            (modulation.py:88-90)
                float x[4];
                if (K == 2) {
                    f32x4v qv = lds_rd128<(Lay<K>::oQ - Lay<K>::oH) * 16>(stage + (unsigned)(qi * 1024 + lane * 16));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qv));
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = nvp_sin(qv[e]) * v[e];
                } else {
                    float sp = lds_rd(stage + (unsigned)(Lay<K>::oS * 16 + px * 4));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sp));
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = nvp_sin(30.0f * __fmaf_rn(sp, w0r[e], c0r[e])) * v[e];
                }
                split_store4(pa + kPlaneTile, x, sc);          // the x tile follows the h tile; same scale (|x| <= |h|)
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // C: the planes hold step s; raw stage s is free
#endif
#if NVP_GL_ABL < 2
        // ---- MFMA phase
        const PxScale qa = px_scale(__uint_as_float(run[slotA])), qb = px_scale(__uint_as_float(run[slotB]));
        {
            const float S = qa.s * qb.s;
            if (S != curS) {                                 // wave-uniform: a tile raised a running maximum
                const float ratio = S * curU;               // <= 1, a power of two
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[r][c] *= ratio;
                curS = S; curU = qa.u * qb.u;
            }
        }
        BOp pb[2], pa2[2];
        pb[0].p[0] = lds_rdq<0>(adB); pb[0].p[1] = lds_rdq<4096>(adB);
        pb[1].p[0] = lds_rdq<1024>(adB); pb[1].p[1] = lds_rdq<4096 + 1024>(adB);
        pa2[0].p[0] = lds_rdq<0>(adA); pa2[0].p[1] = lds_rdq<4096>(adA);
        pa2[1].p[0] = lds_rdq<1024>(adA); pa2[1].p[1] = lds_rdq<4096 + 1024>(adA);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb[0].p[0]), "+v"(pb[0].p[1]), "+v"(pb[1].p[0]), "+v"(pb[1].p[1]),
                     "+v"(pa2[0].p[0]), "+v"(pa2[0].p[1]), "+v"(pa2[1].p[0]), "+v"(pa2[1].p[1]));
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int c = 0; c < 2; ++c) mac_parts(acc[r2][c], pa2[r2].p, pb[c]);
#endif
    }

    // ---- store (plain global stores: no DMA is in flight any more)
    {
        const int ncols = job == 1 ? A.d : NVP_H;
        const int64_t woff = job == 0 ? A.w_h : (job == 1 ? A.w_z : A.w_x);
        const int ld = job == 2 ? NVP_H : A.ld_mod;
        const float un = curU;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = 64 * wc + 32 * c + i;
            if (col < ncols) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 64 * wr + 32 * r2 + nvp_frag_row(r, hh);
                        part[woff + (int64_t)row * ld + col] = acc[r2][c][r] * un;
                    }
            }
        }
        // bias gradients: a row's sum over the chunk = the sum over the 16 lanes (pixels) that staged its row-group
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int qi = wall + kWaves * m;
            if (qi >= 16) continue;                           // not a dp / dq instruction
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bacc[m][e];
                v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
                bacc[m][e] = v;
            }
            if (px == 0) {
                const int row = 4 * (4 * (qi & 7) + rgl);
                const int64_t boff = qi < 8 ? A.b_mod : A.b_sir;
#pragma unroll
                for (int e = 0; e < 4; ++e) part[boff + row + e] = bacc[m][e];
            }
        }
    }
}

}  // namespace

// called by nvp_mlp_bwd_dw (mlp_dw.hip): layers k = 1, 2 of a latent with <= 128 rows; dp_0 x z, the records and the reduction
// stay with the caller
int nvp_mlp_dw_glds_launch(const float* steps, const float* zt, const float* saved, const float* dy, const nvp_mlp_params* p,
                           float* partials, int32_t n_chunks, int64_t n, int32_t d, void* stream) {
    const NvpParamLayout P = nvp_param_layout(d);
    const int64_t ntiles = nvp_ntiles(n);
    const int rows = nvp_rows4(d);
    const int64_t act = ntiles * (int64_t)NVP_H * 32;
    const int tiles_per_chunk = (int)((ntiles + n_chunks - 1) / n_chunks);
    for (int k = 1; k <= 2; ++k) {
        GArgs G;
        G.a1 = dy + (int64_t)k * act; G.a2 = dy + (int64_t)(3 + k) * act;
        G.h = saved + (int64_t)(k - 1) * act; G.q = saved + 3 * act;
        G.z = zt; G.z_rows = rows; G.d = d; G.steps = steps; G.sir0_wp = p->sir_w[0]; G.sir0_bp = p->sir_b[0];
        G.ld_mod = NVP_H + d; G.w_h = P.mod_w[k]; G.w_z = P.mod_w[k] + NVP_H; G.w_x = P.sir_w[k];
        G.b_mod = P.mod_b[k]; G.b_sir = P.sir_b[k]; G.total = P.total;
        const size_t lds = (size_t)kStages * (k == 2 ? Lay<2>::units : Lay<1>::units) * 16 + kPlanes + 32 + 2 * NVP_H * sizeof(float);
        if (lds > 160 * 1024) return NVP_ERR_UNSUPPORTED;
        if (k == 1) hipLaunchKernelGGL((mlp_dw_glds_kernel<1>), dim3(n_chunks), dim3(kThreads), lds, (hipStream_t)stream, G, partials, n, ntiles, tiles_per_chunk);
        else hipLaunchKernelGGL((mlp_dw_glds_kernel<2>), dim3(n_chunks), dim3(kThreads), lds, (hipStream_t)stream, G, partials, n, ntiles, tiles_per_chunk);
        NVP_LAUNCH_CHECK();
    }
    return 0;
}
