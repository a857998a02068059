// Workgroup-shared weight operands for the split-operand MFMA chains (mlp_fwd_b3r.hip, mlp_bwd_b3r.hip).  Sizes below are
// for the bf16 x 3 split (12 KiB per k-step); with the fp16 x 2 split (mlp_b3.h) a k-step is 8 KiB = 8 pieces of 1 KiB.
//
// Why.  In mlp_fwd_b3 / mlp_bwd_b3 every wave streams the complete packed weight set (12 KiB per 16-input k-step, 732 /
// 672 KiB per 32-pixel tile) through its own vector-memory path: 29 GB of L2->L1->VGPR traffic per launch.  A CU's vector
// L1 returns 64 B/clk; eight waves x 12 KiB per k-step is 1536 clk of that path per k-step - exactly the time the four
// SIMDs need for the k-step's 8 x 24 MFMAs.  The weight stream alone saturates the L1 return path, so the matrix pipe can
// never be much more than half busy (measured: 42 % forward, 28 % backward; 3.2e7 VMEM wave-instructions per launch).
//
// What.  The four waves of a workgroup walk the layers in lock step and share each k-step through LDS:
//   * every wave fetches ONE QUARTER of the step (kP x 1 KiB, kP 16-byte loads per lane) two steps ahead into registers,
//   * writes it to the free slot of a two-slot ring one step ahead (ds_write_b128, lane-contiguous),
//   * all four read their A operands from the current slot (ds_read_b128, lane-contiguous: conflict-free),
//   * one s_barrier per k-step both publishes slot (s+1) and retires slot (s-1).
// Vector-memory traffic for weights drops 4x (LDS returns 256 B/clk/CU, four times the L1 path), and the two workgroups
// that share a CU (one wave each per SIMD) stay out of phase with each other, so one's VALU phases overlap the other's
// MFMA phases as before.
//
// The packed stream must be laid out in CONSUMPTION order: k-step s of the kernel is the 768 u32x4 at stream + 768 s.
#pragma once
#include "mlp_b3.h"

constexpr int kRingQuads = kB3StepQuads;       // u32x4 per k-step (8 / 12 KiB)

struct WRing {
    u32x4* lds;                // two slots of kRingQuads
    const u32x4* g;            // packed stream, k-step 0
    int total;                 // k-steps in the stream
    int wv, lane;
    u32x4 sg[kP];              // this wave's quarter of the step that is two ahead of the one being consumed

    // pieces kP wv .. kP wv + kP - 1 of step s
    __device__ __forceinline__ void fetch(int s) {
        const u32x4* p = g + NVP_WSTRIDE((int64_t)s * kRingQuads) + (kP * wv) * 64;
#pragma unroll
        for (int q = 0; q < kP; ++q) sg[q] = (p + q * 64)[(unsigned)lane];
    }
    __device__ __forceinline__ void publish(int s) {
        u32x4* d = lds + (s & 1) * kRingQuads + (kP * wv) * 64;
#pragma unroll
        for (int q = 0; q < kP; ++q) (d + q * 64)[(unsigned)lane] = sg[q];
    }
    __device__ __forceinline__ void prologue() {
        fetch(0);
        publish(0);
        if (total > 1) fetch(1);
        __syncthreads();
    }
    // start of k-step s: hand step s+1 to the ring, fetch step s+2; returns the slot holding step s
    __device__ __forceinline__ const u32x4* begin(int s) {
        if (s + 1 < total) publish(s + 1);
        if (s + 2 < total) fetch(s + 2);
        return lds + (s & 1) * kRingQuads;
    }
    // end of k-step s: everyone has read slot s, slot s+1 is complete.  NOT __syncthreads(): its workgroup-scope release
    // fence makes hipcc wait for vmcnt(0) - i.e. for the weight fetch issued two steps ahead and for every stream store in
    // flight - at every k-step.  Only LDS traffic has to be settled here.
#ifdef NVP_ABL_NOBARRIER         // ablation builds only: results are wrong, the timing prices the lock step
    __device__ __forceinline__ void end(int) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
    __device__ __forceinline__ void end(int) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
};

// ---- flag-synchronised ring (NVP_RING_FLAGS) ---------------------------------------------------------------------------
// The barrier ring lock-steps the workgroup: every k-step costs the slowest of four waves, each of which shares its SIMD with a
// wave of the other workgroup (measured: removing the barriers - wrong results - takes the forward from 1.93 to 1.55 ms).
// Here the slots carry two LDS counters instead: `filled` (+1 by each of the four producers after its quarter is written) and
// `drained` (+1 by each consumer after its reads).  A consumer polls filled[slot] >= 4 (generation + 1) before reading, a
// producer polls drained[slot] >= 4 generation before overwriting; with R slots and a publish distance of D = R - 2 steps the
// waves may drift two k-steps apart before anyone waits.  LDS executes one wave's operations in order, so a counter update
// issued after the data accesses is seen after them; the signalling and polling are inline asm so that hipcc neither
// reorders memory operations across them nor attaches a vmcnt(0) to them.
template <int R>
struct WRingF {
    static constexpr int D = R - 2;
    u32x4* lds;                // R slots of kRingQuads, followed by 2 R counters
    const u32x4* g;
    int total;
    int wv, lane;
    u32x4 sg[kP];

    __device__ __forceinline__ unsigned cnt_addr(int which, int slot) const {      // LDS byte address of a counter
        return (unsigned)reinterpret_cast<uintptr_t>(lds + R * kRingQuads) + 4u * (unsigned)(which * R + slot);
    }
    __device__ __forceinline__ void wait_ge(unsigned addr, unsigned target) const {
        unsigned v;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        } while (__builtin_amdgcn_readfirstlane(v) < target);
    }
    __device__ __forceinline__ void signal(unsigned addr) const {
        if (lane == 0) { const unsigned one = 1u; asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(one) : "memory"); }
        else asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void fetch(int s) {
        const u32x4* p = g + NVP_WSTRIDE((int64_t)s * kRingQuads) + (kP * wv) * 64;
#pragma unroll
        for (int q = 0; q < kP; ++q) sg[q] = (p + q * 64)[(unsigned)lane];
    }
    __device__ __forceinline__ void publish(int p) {
        const int slot = p % R;
        wait_ge(cnt_addr(1, slot), 4u * (unsigned)(p / R));            // every consumer is done with the slot's previous step
        u32x4* d = lds + slot * kRingQuads + (kP * wv) * 64;
#pragma unroll
        for (int q = 0; q < kP; ++q) (d + q * 64)[(unsigned)lane] = sg[q];
        signal(cnt_addr(0, slot));
    }
    __device__ __forceinline__ void prologue() {
        unsigned* c = reinterpret_cast<unsigned*>(lds + R * kRingQuads);
        if (wv == 0 && lane < 2 * R) c[lane] = 0u;
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < D && i < total; ++i) { fetch(i); publish(i); }
        if (D < total) fetch(D);
    }
    __device__ __forceinline__ const u32x4* begin(int s) {
        if (s + D < total) publish(s + D);
        if (s + D + 1 < total) fetch(s + D + 1);
        wait_ge(cnt_addr(0, s % R), 4u * (unsigned)(s / R + 1));       // all four quarters of step s are in the slot
        return lds + (s % R) * kRingQuads;
    }
    __device__ __forceinline__ void end(int s) { signal(cnt_addr(1, s % R)); }
};

// one k-step into four output tiles, A operands from the ring slot `w` (LDS); the reads of tile T+1 are issued ahead of
// the MFMAs of tile T
__device__ __forceinline__ void step_b3_ring(f32x16 (&acc)[4], const u32x4* __restrict__ w, const BOp& b, int lane) {
    const unsigned ul = (unsigned)lane;
    u32x4 a[2][kP];
#pragma unroll
    for (int q = 0; q < kP; ++q) a[0][q] = (w + q * 64)[ul];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        if (T < 3) {
#pragma unroll
            for (int q = 0; q < kP; ++q) a[(T + 1) & 1][q] = (w + ((T + 1) * kP + q) * 64)[ul];
        }
        NVP_CHAIN_FENCE();
        mac_parts(acc[T], a[T & 1], b);
    }
}

// bias step: B = e_0 s (mlp_b3.h), A[.][0] = the parts of the bias
__device__ __forceinline__ void bias_b3_ring(f32x16 (&acc)[4], const u32x4* __restrict__ w, float s, int lane) {
    const unsigned ul = (unsigned)lane;
    const u32x4 e0 = bias_bop(s, lane);
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        u32x4 a[kP];
#pragma unroll
        for (int q = 0; q < kP; ++q) a[q] = (w + (T * kP + q) * 64)[ul];
        bias_mac(acc[T], a, e0);
    }
}

// 8 k-steps over the previous layer's D registers; `s` is the running k-step index of the kernel.  `pre` (optional) runs
// at the start of the LAST k-step: the caller prefetches what the following chain needs (its first latent rows) there,
// one k-step ahead, instead of keeping those registers alive through the whole chain.
template <typename Ring, typename Pre>
__device__ __forceinline__ void chain_h_b3_ring(f32x16 (&acc)[4], const f32x16 (&hin)[4], const float sc, Ring& R, int& s, int lane, Pre pre) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const u32x4* w = R.begin(s);
        if (c == 7) pre();
        float x[8];
        chain_in8(x, hin, c);
        BOp b;
        split8(x, sc, b);
        step_b3_ring(acc, w, b, lane);
        R.end(s);
        ++s;
    }
}

template <typename Ring>
__device__ __forceinline__ void chain_h_b3_ring(f32x16 (&acc)[4], const f32x16 (&hin)[4], const float sc, Ring& R, int& s, int lane) {
    chain_h_b3_ring(acc, hin, sc, R, s, lane, [] {});
}

// ---- half-step ring (mlp_bwd_b3r.hip) ------------------------------------------------------------------------------
// The backward chain keeps a 16.9 KiB transpose / parking tile per wave in LDS, which leaves 12 KiB per workgroup when two
// workgroups share a CU: the ring's slots then hold HALF a k-step (two output tiles x kP parts: 4 KiB fp16 x 2, 6 KiB
// bf16 x 3), one barrier per half-step.  fp16 x 2: four 1-KiB pieces per half-step, one per wave.  bf16 x 3: six pieces over
// four waves - wave w moves pieces w and 4 + (w & 1).
constexpr int kHalfPieces = 2 * kP;
constexpr int kHalfQuads = kHalfPieces * 64;

#ifndef NVP_HRING_DEEP
#define NVP_HRING_DEEP (NVP_SPLIT_H2 ? 2 : 0)     // fp16 x 2 only: extra half-steps of weights in flight per wave (0: 1.82, 1: 1.75, 2: 1.735, 3: 1.73 ms)
#endif
#if NVP_HRING_DEEP && !NVP_SPLIT_H2
#error "NVP_HRING_DEEP needs the fp16 x 2 split (one ring piece per wave)"
#endif
struct HRing {
    u32x4* lds;                // two slots of kHalfQuads
    const u32x4* g;            // packed stream in consumption order, half-step 0
    int chain_end;             // first half-step beyond the chain that is open
    int wv, lane;
    u32x4 sg[kHalfPieces > 4 ? 2 : 1];

    // bf16 x 3: pieces 4 and 5 are moved twice (identical data to identical addresses), which keeps fetch / publish free of
    // wave-dependent branches (a conditional second quad cost ~450 spill instructions: the branches split the chains'
    // scheduling regions)
    __device__ __forceinline__ void fetch(int hs) {
        const u32x4* p = g + (int64_t)hs * kHalfQuads;
        sg[0] = (p + wv * 64)[(unsigned)lane];
        if (kHalfPieces > 4) sg[kHalfPieces > 4 ? 1 : 0] = (p + (4 + (wv & 1)) * 64)[(unsigned)lane];
    }
    __device__ __forceinline__ void publish(int hs) {
        u32x4* d = lds + (hs & 1) * kHalfQuads;
        (d + wv * 64)[(unsigned)lane] = sg[0];
        if (kHalfPieces > 4) (d + (4 + (wv & 1)) * 64)[(unsigned)lane] = sg[kHalfPieces > 4 ? 1 : 0];
    }
    // Open a chain of `nh` half-steps starting at hs0: fill slot hs0, fetch hs0 + 1.  The ring is filled chain by chain (not
    // across chains) so that the staging registers are DEAD during the element-wise stages between the chains, where the
    // backward kernel has no register to spare (kept alive there they cost ~480 spill instructions); the price is one exposed
    // L2 round trip per chain (seven per tile, ~2-3 % of a tile's cycles, mostly covered by the partner workgroup's wave).
#if NVP_HRING_DEEP
    // fp16 x 2 (one piece per wave and half-step): 1 + NVP_HRING_DEEP half-steps in flight in registers - the piece published at
    // half-step hs was requested 2 + NVP_HRING_DEEP half-steps earlier instead of two (one more quad of registers per level; the L2
    // round trip under load is longer than two half-steps of MFMAs)
    u32x4 nx[NVP_HRING_DEEP];
    __device__ __forceinline__ u32x4 load_piece(int hs) const { return (g + (int64_t)hs * kHalfQuads + wv * 64)[(unsigned)lane]; }
    __device__ __forceinline__ void open(int hs0, int nh) {
        chain_end = hs0 + nh;
        fetch(hs0);
        publish(hs0);
        if (nh > 1) fetch(hs0 + 1);
#pragma unroll
        for (int k = 0; k < NVP_HRING_DEEP; ++k)
            if (nh > 2 + k) nx[k] = load_piece(hs0 + 2 + k);
        end();
    }
    __device__ __forceinline__ const u32x4* begin(int hs) {
        if (hs + 1 < chain_end) publish(hs + 1);
        sg[0] = nx[0];
#pragma unroll
        for (int k = 0; k + 1 < NVP_HRING_DEEP; ++k) nx[k] = nx[k + 1];
        if (hs + 2 + NVP_HRING_DEEP < chain_end) nx[NVP_HRING_DEEP - 1] = load_piece(hs + 2 + NVP_HRING_DEEP);
        return lds + (hs & 1) * kHalfQuads;
    }
#else
    __device__ __forceinline__ void open(int hs0, int nh) {
        chain_end = hs0 + nh;
        fetch(hs0);
        publish(hs0);
        if (nh > 1) fetch(hs0 + 1);
        end();
    }
    __device__ __forceinline__ const u32x4* begin(int hs) {
        if (hs + 1 < chain_end) publish(hs + 1);
        if (hs + 2 < chain_end) fetch(hs + 2);
        return lds + (hs & 1) * kHalfQuads;
    }
#endif
#ifdef NVP_ABL_NOBARRIER
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
};

// one k-step (two half-steps of the ring) into four output tiles.  LEAN: one tile's operand quads live at a time (what the
// bf16 x 3 split needed: two accumulator sets live in the shared pass, no registers to spare); otherwise both tiles' quads of a
// half-step are read ahead of the first MFMA - with the fp16 x 2 split's smaller operands that fits: 1.875 -> 1.80 ms.
#ifndef NVP_HRING_LEAN
#define NVP_HRING_LEAN 0
#endif
__device__ __forceinline__ void step_b3_hring(f32x16 (&acc)[4], HRing& R, int& hs, const BOp& b, int lane) {
    const unsigned ul = (unsigned)lane;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const u32x4* w = R.begin(hs);
#if NVP_HRING_LEAN
#pragma unroll
        for (int Tl = 0; Tl < 2; ++Tl) {
            const int T = 2 * half + Tl;
            u32x4 a[kP];
#pragma unroll
            for (int q = kP - 1; q >= 0; --q) a[q] = (w + (Tl * kP + q) * 64)[ul];
            NVP_CHAIN_FENCE();
            mac_parts(acc[T], a, b);
            NVP_CHAIN_FENCE();
        }
#else
        u32x4 a[2][kP];
#pragma unroll
        for (int Tl = 0; Tl < 2; ++Tl)
#pragma unroll
            for (int q = 0; q < kP; ++q) a[Tl][q] = (w + (Tl * kP + q) * 64)[ul];
#pragma unroll
        for (int Tl = 0; Tl < 2; ++Tl) {
            NVP_CHAIN_FENCE();
            mac_parts(acc[2 * half + Tl], a[Tl], b);
        }
#endif
        R.end();
        ++hs;
    }
}

// `open_nh`: half-steps to open the ring for (16 = this chain alone; more = the chains that follow are fed by the same fill,
// 0 = a previous chain already opened the ring for this one: one exposed L2 round trip less)
__device__ __forceinline__ void chain_h_b3_hring(f32x16 (&acc)[4], const f32x16 (&hin)[4], const float sc, HRing& R, int& hs, int lane, int open_nh = 16) {
    if (open_nh) R.open(hs, open_nh);
    NVP_CHAIN_ENTER();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
        chain_in8(x, hin, c);
        BOp b;
        split8(x, sc, b);
        step_b3_hring(acc, R, hs, b, lane);
    }
    NVP_CHAIN_LEAVE();
}

// two transposed GEMMs over the SAME input registers, one operand split per k-step; the packed stream interleaves the two
// weight streams k-step by k-step (a's step c, then b's step c)
__device__ __forceinline__ void chain_h2_b3_hring(f32x16 (&acc_a)[4], f32x16 (&acc_b)[4], const f32x16 (&hin)[4], const float sc, HRing& R, int& hs, int lane, int open_nh = 32) {
    if (open_nh) R.open(hs, open_nh);
    NVP_CHAIN_ENTER();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
        chain_in8(x, hin, c);
        BOp b;
        split8(x, sc, b);
        step_b3_hring(acc_a, R, hs, b, lane);
        step_b3_hring(acc_b, R, hs, b, lane);
    }
    NVP_CHAIN_LEAVE();
}
