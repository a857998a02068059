// Weight gradients of modulator layer k and SIREN layer k (k = 1, 2) with every operand stream read ONCE and fed to LDS by DMA.
//
// Why.  As seven independent 256-thread jobs (mlp_dw.hip) the dW stage requests 7.5 KB per pixel for 4.56 KB of distinct operand
// streams - jobs that share a stream each fetch it - and is bound by those bytes (PMC: 8.8 GB per step at 4.9 TB/s).  One 768-thread
// workgroup per pixel chunk that owns the three jobs of a layer
//     waves 0-3: dp_k x h_{k-1}      waves 4-7: dp_k x z      waves 8-11: dq_k x x_{k-1},   x = sin(.) h  rebuilt on the fly
// needs 2.0 / 2.5 KB per pixel (k = 1 / 2).  Staged through registers (mlp_dw_group_kernel) that workgroup is alone on its CU and
// keeps only ~70 KB in flight part of the time: 2.8 TB/s, slower than the jobs it replaces.  Here the tiles go global -> LDS with
// global_load_lds (16 B per lane, no registers, no write pass): a ring of three 16-pixel stages keeps two stages (65-80 KB)
// in flight all the time, across the barriers (raw s_barrier + counted vmcnt; tools/probes/glds_stream_probe.hip: this access
// pattern streams at 6.0 TB/s from one workgroup per CU).
//
// LDS image.  A DMA writes lane l of a wave instruction to (wave-uniform base) + 16 l: the image is the PTM4 half tile itself,
// [row-group][16 px] 16-byte units.  An MFMA fragment needs 8 consecutive pixels of ONE row = 8 dwords 16 bytes apart, and the
// 32 lanes of a half wave (8 row-groups x 4 rows) would hit 4 banks.  So the SOURCE pixel of unit (rg, slot) is rotated,
// slot = (px + 2 rg) & 15: the 8 row-groups of a fragment then start 8 banks apart and a ds_read_b32 is conflict free.
//
// Arithmetic: that of mlp_dw_kernel (split-operand 16-bit MFMA under a running power-of-two block scale per operand tile,
// mlp_b3.h), one 16-pixel k-step per stage.  The tile maxima the scales follow are taken from LDS after a stage has landed.
// x_{k-1} shares h_{k-1}'s scale (|x| <= |h|).  Results equal the per-job kernels' to fp32 summation order (the pixels of a
// 32-pixel tile enter the k-steps in a different order), not bit for bit.
#include <cstdlib>
#include "mlp_b3.h"

#ifndef NVP_DW_B3
#define NVP_DW_B3 1
#endif

namespace {

constexpr int kThreads = 768, kWaves = 12, kStages = 3;

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

// stage layout in 16-byte units, every stream on a wave-instruction (64-unit) boundary so that a DMA instruction has ONE source
// stream (uniform base in scalar registers + a 32-bit lane offset): [dp 512 | dq 512 | h 512 | (q 512) | z 512 (16 rgz used) | (steps 64, 4 used)]
template <int K> struct Lay {
    static constexpr int oA1 = 0, oA2 = 512, oH = 1024, oQ = 1536;
    static constexpr int oZ = K == 2 ? 2048 : 1536;
    static constexpr int oS = oZ + 512;                       // K == 1: the 16 temporal steps of the half tile
    static constexpr int units = K == 2 ? oZ + 512 : oS + 64;
};

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p); }

// global -> LDS DMA of 16 bytes per lane: LDS destination = M0 (wave-uniform byte address) + 16 x lane.  Inline assembly, not
// __builtin_amdgcn_global_load_lds: hipcc sinks the builtin below the MFMAs of the step (nothing it can see depends on it), which
// halves the time a stage spends in flight; asm volatile statements keep their program order among themselves (LDS reads, waits).
// Source = uniform 64-bit base (scalar registers) + a 32-bit per-lane byte offset.
__device__ __forceinline__ void glds16(const void* ubase, unsigned lane_off, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(ubase), "s"(lds_byte) : "memory");
}
// LDS accesses of the main loop are written as inline assembly: hipcc's waitcnt insertion treats a pending LDS DMA as a write to
// ALL of LDS and puts s_waitcnt vmcnt(0) in front of every ds_read it can see - which would drain the ring each step.  The waits
// (vmcnt counted by hand for the DMAs, lgkmcnt(0) for these reads) are placed explicitly.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lds_rd(unsigned addr) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); return v; }
template <int O> __device__ __forceinline__ float lds_rd_o(unsigned addr) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(O)); return v; }
__device__ __forceinline__ f32x4v lds_rd128(unsigned addr) { f32x4v v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); return v; }
__device__ __forceinline__ void lds_umax(unsigned addr, unsigned v) { asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// s_waitcnt lgkmcnt(0), tied to the registers the preceding reads fill (so that no use can be scheduled above it)
__device__ __forceinline__ void wait_lgkm8(float (&x)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void tie8(float (&x)[8]) {
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}

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

template <int K>
__global__ __launch_bounds__(kThreads, 1) void mlp_dw_glds_kernel(GArgs A, float* __restrict__ partials, int64_t n, int64_t ntiles, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = wall >> 2, w = wall & 3;                 // 0: dp x h, 1: dp x z, 2: dq x x
    const int i = lane & 31, hh = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const int rgz = A.z_rows >> 2;                           // row-groups of the latent
    const int zunits = rgz * 16;
    constexpr int stage_bytes = Lay<K>::units * 16;
    // DMA wave instructions per stage: the 128-row streams whole, the latent's (zunits + 63) / 64, (K == 1) one for the steps
    const int n_z = (zunits + 63) >> 6;
    constexpr int q_z0 = Lay<K>::oZ >> 6;                    // first instruction of the latent
    auto instr_live = [&](int qi) { return qi < q_z0 + n_z || (K == 1 && qi == (Lay<K>::oS >> 6)); };
    int my_instr = 0;                                        // ... issued by this wave (wave-uniform, 2..4)
#pragma unroll
    for (int m = 0; m < 4; ++m) my_instr += (wall + kWaves * m < (Lay<K>::units >> 6) && instr_live(wall + kWaves * m)) ? 1 : 0;
    unsigned* mx = reinterpret_cast<unsigned*>(lds + kStages * stage_bytes);      // running maxima: dp, dq, h, z (bit patterns)
    float* tab = reinterpret_cast<float*>(lds + kStages * stage_bytes + 32);      // K == 1: SIREN-0 weight / bias

    const int64_t s0 = (int64_t)blockIdx.x * tiles_per_chunk * 2;                  // half tiles of this chunk
    const int64_t s1 = min(ntiles * 2, s0 + (int64_t)tiles_per_chunk * 2);
    const int nsteps = (int)(s1 - s0);
    float* part = partials + (int64_t)blockIdx.x * A.total;

    if (tid < 8) mx[tid] = __float_as_uint(kTinyMax);
    if (K == 1 && tid < NVP_H) { tab[tid] = A.sir0_wp[tid]; tab[NVP_H + tid] = A.sir0_bp[tid]; }

    // ---- DMA of half tile s0 + step into stage step % kStages.  Lane l of instruction qi fills unit qi * 64 + l: row-group
    //      rg = 4 (qi & 7) + (l >> 4) of its stream, slot l & 15, i.e. source pixel (slot - 2 rg) & 15 (the bank rotation).
    //      A wave's instructions qi = wave + 12 m all have the wave's parity, so 2 rg = 8 (qi & 7) + 2 (l >> 4) gives every one
    //      of them the SAME per-lane source offset: one register; the stream / tile / row-group block is a scalar base.
    auto issue = [&](int step) {
        // (recomputed per step behind an opaque zero: as a loop invariant it only gets spilled, and a scratch reload waits on vmcnt)
        unsigned zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        const unsigned ln = (unsigned)lane + zero;
        const unsigned lane_src = (ln >> 4) * 512u + (((ln & 15u) - 8u * (unsigned)(wall & 1) - 2u * (ln >> 4)) & 15u) * 16u;
        const int64_t ht = s0 + step;
        const int64_t tile = ht >> 1;
        const int half = (int)(ht & 1);
        const unsigned stage = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_addr(lds) + (unsigned)(step % kStages) * stage_bytes));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int qi = wall + kWaves * m;                 // wave-uniform
            if (qi >= (Lay<K>::units >> 6) || !instr_live(qi)) continue;
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
            if (!lat || rg0 + (lane >> 4) < rgz) glds16(ub, lane_src, stage + (unsigned)(qi * 1024));
        }
    };

    // ---- per-lane fragment addressing (bytes inside a stage): row i of a 32-row tile, pixels 8 hh .. + 7.  ONE per-lane table:
    //      a tile's first row-group is a multiple of 8, so the rotation (8 hh + 2 rg) & 15 only depends on the lane; the tile
    //      (operand stream, 32-row block) is a wave-uniform byte offset.  The latent's region is a full 32 row-groups: the rows
    //      past its end hold stale LDS contents, which only reach output columns >= d (never stored).
    const int uA = job == 2 ? Lay<K>::oA2 : Lay<K>::oA1;
    const int uB = job == 1 ? Lay<K>::oZ : Lay<K>::oH;
    const int slotA = job == 2 ? 1 : 0, slotB = job == 1 ? 3 : 2;
    const unsigned baseA = (unsigned)(uA * 16 + wr * 4096), baseB = (unsigned)(uB * 16 + wc * 4096);      // + 2048 for the second 32-row block
    // adA[k] / adB[k]: LDS byte address of pixel 8 hh + k of row i of this wave's first A / B 32-row block IN THE CURRENT STAGE
    // (second block: + 2048; q behind h: + 8192 - immediates).  Advanced by one stage per step (16 adds) instead of rebuilt
    // for each of the 40 reads: the loop is VALU-bound.
    unsigned adA[8], adB[8];
    {
        const unsigned off_row = (unsigned)((i >> 2) * 256 + 4 * (i & 3));
        const unsigned off_c = (unsigned)((8 * hh + 2 * (i >> 2)) & 15);
        const unsigned l0 = lds_addr(lds);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned o = off_row + (((off_c + k) & 15u) << 4);
            adA[k] = l0 + baseA + o;
            adB[k] = l0 + baseB + o;
        }
    }
    float w0r[2] = {0.f, 0.f}, c0r[2] = {0.f, 0.f};          // K == 1, job 2: SIREN-0 weight / bias of this lane's two x rows

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = nvp_zero16();
    float bsum0 = 0.f, bsum1 = 0.f;
    const bool want_bias = job != 1 && wc == 0;
    unsigned run[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) run[u] = __float_as_uint(kTinyMax);
    PxScale qa = px_scale(kTinyMax), qb = qa;
    float curS = qa.s * qb.s, curU = qa.u * qb.u;

    __syncthreads();                                         // mx, tab (plain loads: before any DMA is in flight)
    if (K == 1 && job == 2) {
#pragma unroll
        for (int c = 0; c < 2; ++c) { w0r[c] = tab[64 * wc + 32 * c + i]; c0r[c] = tab[NVP_H + 64 * wc + 32 * c + i]; }
    }
    if (nsteps > 0) issue(0);
    if (nsteps > 1) issue(1);

    const unsigned lds0 = lds_addr(lds);
    const unsigned mx_a = lds0 + kStages * stage_bytes;
    for (int s = 0; s < nsteps; ++s) {
        const unsigned stage = lds0 + (unsigned)(s % kStages) * stage_bytes;
        // stage s has landed (this wave's share): the DMAs of stage s + 1 may stay in flight
        wait_vm_n(s + 1 < nsteps ? my_instr : 0);
        __builtin_amdgcn_s_barrier();                        // everyone's share landed; everyone is past the MFMAs of step s - 1
        if (s + 2 < nsteps) issue(s + 2);                    // into the stage step s - 1 used
#ifndef NVP_GL_ABL
#define NVP_GL_ABL 0        // ablation builds only (timing, wrong results): 1 = no tile maxima / second barrier; 2 = also no fragment reads, splits, MFMAs
#endif
#if NVP_SPLIT_H2 && NVP_GL_ABL == 0
        {   // tile maxima of stage s -> running maxima (ds_max_u32), only when something raises them
            f32x4v mv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int qi = wall + kWaves * m;                 // the units this wave's own DMA instructions brought
                const bool on = qi < q_z0 + n_z;                  // (not the steps)
                const int u = qi * 64 + lane;
                const bool live = on && (qi < q_z0 || u < Lay<K>::oZ + zunits);
                mv[m] = lds_rd128(stage + (live ? u : 0) * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mv[0]), "+v"(mv[1]), "+v"(mv[2]), "+v"(mv[3]));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int qi = wall + kWaves * m;
                if (qi >= q_z0 + n_z) continue;
                const int u = qi * 64 + lane;
                const bool live = qi < q_z0 || u < Lay<K>::oZ + zunits;
                const unsigned mm = live ? __float_as_uint(fmaxf(fmaxf(fabsf(mv[m][0]), fabsf(mv[m][1])), fmaxf(fabsf(mv[m][2]), fabsf(mv[m][3])))) : 0u;
                const int slot = qi < 8 ? 0 : (qi < 16 ? 1 : (qi < 24 ? 2 : (qi < q_z0 ? -1 : 3)));     // q needs no scale
                if (slot >= 0 && __any(mm > run[slot])) { const unsigned wm = wave_umax(mm); if (lane == 0) lds_umax(mx_a + 4 * slot, wm); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            float r4[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) r4[u] = lds_rd(mx_a + 4 * u);
#pragma unroll
            for (int u = 4; u < 8; ++u) r4[u] = 0.f;
            wait_lgkm8(r4);
#pragma unroll
            for (int u = 0; u < 4; ++u) run[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r4[u]));
        }
        qa = px_scale(__uint_as_float(run[slotA])); qb = px_scale(__uint_as_float(run[slotB]));
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
#endif
#if NVP_GL_ABL < 2
        // ---- B fragments (two 32-column tiles), one at a time (registers): read, wait, (rebuild x,) split
        BOp pb[2];
        {
            float sv[8];
            if (K == 1 && job == 2) {
                f32x4v sa = lds_rd128(stage + Lay<K>::oS * 16 + 32 * hh), sb = lds_rd128(stage + Lay<K>::oS * 16 + 32 * hh + 16);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sa), "+v"(sb));
#pragma unroll
                for (int k = 0; k < 4; ++k) { sv[k] = sa[k]; sv[4 + k] = sb[k]; }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float xb[8], xs[8];
                if (job == 2) {                              // the sine factor of x_{k-1} = sin(.) * h_{k-1} first   (modulation.py:88-90)
                    if (K == 2) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) xs[k] = c == 0 ? lds_rd_o<(Lay<K>::oQ - Lay<K>::oH) * 16>(adB[k]) : lds_rd_o<(Lay<K>::oQ - Lay<K>::oH) * 16 + 2048>(adB[k]);
                        wait_lgkm8(xs);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            xs[k] = nvp_sin(xs[k]);
                            if (k & 1) __builtin_amdgcn_sched_barrier(0);      // two sines at a time: their temporaries must not pile up on top of the accumulators
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            xs[k] = nvp_sin(30.0f * __fmaf_rn(sv[k], w0r[c], c0r[c]));
                            if (k & 1) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) xb[k] = c == 0 ? lds_rd_o<0>(adB[k]) : lds_rd_o<2048>(adB[k]);
                wait_lgkm8(xb);
                if (job == 2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) xb[k] = xs[k] * xb[k];
                }
                split8(xb, qb.s, pb[c]);
            }
        }
        // ---- A fragments (two 32-row tiles) and the products
        {
            float xa[2][8];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                for (int k = 0; k < 8; ++k) xa[r2][k] = r2 == 0 ? lds_rd_o<0>(adA[k]) : lds_rd_o<2048>(adA[k]);
            wait_lgkm8(xa[0]); tie8(xa[1]);
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                if (want_bias) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { if (r2 == 0) bsum0 += xa[r2][k]; else bsum1 += xa[r2][k]; }
                }
                BOp pa;
                split8(xa[r2], qa.s, pa);
#pragma unroll
                for (int c = 0; c < 2; ++c) mac_parts(acc[r2][c], pa.p, pb[c]);
            }
        }
#endif
        {   // next stage
            const unsigned d = (s % kStages) == kStages - 1 ? (unsigned)(-(kStages - 1) * stage_bytes) : (unsigned)stage_bytes;
#pragma unroll
            for (int k = 0; k < 8; ++k) { adA[k] += d; adB[k] += d; }
        }
    }

    // ---- store (plain global stores: no DMA is in flight any more)
    {
        const int ncols = job == 1 ? A.d : NVP_H;
        const int64_t woff = job == 0 ? A.w_h : (job == 1 ? A.w_z : A.w_x);
        const int ld = job == 2 ? NVP_H : A.ld_mod;
        const float un = (NVP_DW_B3 && NVP_SPLIT_H2) ? curU : 1.0f;
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
        if (want_bias) {
            bsum0 += __shfl_xor(bsum0, 32);
            bsum1 += __shfl_xor(bsum1, 32);
            if (hh == 0) {
                const int64_t boff = job == 0 ? A.b_mod : A.b_sir;
                part[boff + 64 * wr + i] = bsum0;
                part[boff + 64 * wr + 32 + i] = bsum1;
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
        const size_t lds = (size_t)kStages * (k == 2 ? Lay<2>::units : Lay<1>::units) * 16 + 32 + 2 * NVP_H * sizeof(float);
        if (lds > 160 * 1024) return NVP_ERR_UNSUPPORTED;
        if (k == 1) hipLaunchKernelGGL((mlp_dw_glds_kernel<1>), dim3(n_chunks), dim3(kThreads), lds, (hipStream_t)stream, G, partials, n, ntiles, tiles_per_chunk);
        else hipLaunchKernelGGL((mlp_dw_glds_kernel<2>), dim3(n_chunks), dim3(kThreads), lds, (hipStream_t)stream, G, partials, n, ntiles, tiles_per_chunk);
        NVP_LAUNCH_CHECK();
    }
    return 0;
}
