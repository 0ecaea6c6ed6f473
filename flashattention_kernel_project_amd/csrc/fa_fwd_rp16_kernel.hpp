// fa_fwd_rp16_kernel.hpp -- the rolling half-tile pipeline of fa_fwd_rp.hip on v_mfma_f32_16x16x32 (d = 64).
//
// Why a second shape of the same stream: the d=64 forward runs at the package power cap, where wall time is joules per
// launch divided by the cap (fa_fwd_rp needs 15 % fewer cycles than fa_fwd_w64x and lands at the same
// 0.55 ms, the clock simply settles at 1.80 instead of 2.10 GHz).  Sustained at two waves per SIMD the slot model
// (tools/slot_energy.py, profiles/r02_slot_energy.txt) prices one slot -- two scores per lane -- at
//     32x32x16 + folded vector work      21.7 nJ per SIMD     2 x 16x16x32 + folded      19.6 nJ   (-9 %)
//     32x32x16 + exact vector work       25.7 nJ              2 x 16x16x32 + exact       23.4 nJ   (-9 %)
// although the 16x16x32 form needs a quarter more cycles per slot (it holds the issue port 8 of every 16 cycles).
// So: the same pipeline (QK^T one half tile ahead, PV one behind, the softmax of the half tile in between issued as
// slices between the matrix instructions, branch-free steady state, folded fast pass with the wave reference maximum as
// the accumulators' start value) with the lane roles and LDS images of fa_fwd_w64x.hip:
//   lane = 16 g + c; the accumulator of S^T = K.Q^T for (16-row query block x, 16-key block kb) holds query 16x + c on
//   the lane and keys 16kb + 4g + i in register i; the packed registers of key blocks 2s, 2s+1 are the B fragment of
//   k-step s of O^T += V^T.P^T; K row-major with the 16-B chunk index XORed by (row >> 1) & 7, V in 256-B blocks
//   [key/8][d/16] x [8 keys][16 cols] for ds_read_b64_tr_b16.
// A step = one half tile (32 keys) = 32 matrix instructions (16 QK^T + 16 PV, four K and four V^T fragments, each
// feeding the wave's four query blocks) around the vector work of 32 scores per lane.
#pragma once
#include "fa_tile.hpp"

#include <type_traits>
#include <utility>

namespace fa {

namespace rp16 {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
template <typename T> struct Mx;
// mfma_v*: the same instruction spelled out with its accumulator in ARCHITECTURAL registers.  A kernel that may use the
// accumulator half of the file (one wave per SIMD) gets every builtin matrix instruction in the form whose C/D live there:
// right for O and the row sums (only matrix instructions touch them), wrong for the scores (one v_accvgpr_read per score).
// The compiler does not know these statements are matrix instructions: the wait states between one of them and the first
// vector instruction that reads its result are the CALLER's (see kAsmQK; tools/mfma_hazard_lint.py checks the listing).
#define FA_MFMA_V(NAME, SFX)                                                                                              \
    static __device__ __forceinline__ void mfma_v_acc(f32x4& acc, u32x4 a, u32x4 b) {                                     \
        asm("v_mfma_f32_16x16x32_" SFX " %0, %1, %2, %0 ; fa_qk" : "+v"(acc) : "v"(a), "v"(b));                           \
    }                                                                                                                     \
    static __device__ __forceinline__ f32x4 mfma_v_init(u32x4 a, u32x4 b, f32x4 c) {                                      \
        f32x4 d;                                                                                                          \
        asm("v_mfma_f32_16x16x32_" SFX " %0, %1, %2, %3 ; fa_qk" : "=&v"(d) : "v"(a), "v"(b), "v"(c));                    \
        return d;                                                                                                         \
    }                                                                                                                     \
    static __device__ __forceinline__ f32x4 mfma_v_zero(u32x4 a, u32x4 b) {                                               \
        f32x4 d;                                                                                                          \
        asm("v_mfma_f32_16x16x32_" SFX " %0, %1, %2, 0 ; fa_qk" : "=&v"(d) : "v"(a), "v"(b));                             \
        return d;                                                                                                         \
    }
template <> struct Mx<F16> {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    FA_MFMA_V(F16, "f16")
};
template <> struct Mx<BF16> {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    FA_MFMA_V(BF16, "bf16")
};
#undef FA_MFMA_V
constexpr int kW = 8;
#ifndef FA_RP16_AHEAD
#define FA_RP16_AHEAD 2
#endif
constexpr int kAheadWide = FA_RP16_AHEAD;   // 64-row waves: fragments read ahead of their MFMAs (at most ring - 1), ring of 4 registers
constexpr float kHeadroom = 4.0f;      // exact optimistic pass: reference = the row's max over its first 32 keys + this
constexpr float kHeadroomFold = 1.0f;  // folded pass: the reference already is the maximum over the wave's 64 rows
constexpr float kFoldMax = 16.0f;  // folded pass: largest |reference| (log2 units) it accepts -- Q's fp16 rounding moves a logit by <= |logit| * 2^-11
constexpr float kFoldAim = 6.0f;       // folded pass: log2 of the row sum the reference is placed for
constexpr float kFoldShiftMin = 6.0f;  // ... and how far below the first scores' maximum it may go (weights stay below 2^16)
#ifndef FA_RP16_ABL
#define FA_RP16_ABL 0              // lab only (timing ablations, results are garbage): 1 no LDS fragment reads, 2 no softmax vector
#endif                             // work, 4 no matrix instructions, 8 no K/V staging, 16 no tile barrier, 32 no O stores, 64 no Q loads,
                                   // 128 no loads of an item's first two K/V tiles
#ifndef FA_RP16_GATES
#define FA_RP16_GATES 15           // lab only: which refusal gates of the folded pass are armed (1 sum overflow, 2 sum too small, 4 reference, 8 Q range)
#endif
#ifndef FA_RP16_SUMMFMA
#define FA_RP16_SUMMFMA 1          // 1: the optimistic passes take the row sums from the matrix pipe (one more PV block against a
#endif                             // fragment of ones: X matrix instructions per step instead of 32 v_add_f32 per lane)
#ifndef FA_RP16_PREFETCH
#define FA_RP16_PREFETCH 1         // 1: the next item's Q rows are requested under this item's epilogue, ahead of its stores
#endif
#ifndef FA_RP16_VALU_AT
#define FA_RP16_VALU_AT 0          // lab: a vector pair-step goes behind the last (0) or the first (1) matrix instruction of its group
#endif
#ifndef FA_RP16_ONES_POS
#define FA_RP16_ONES_POS 0         // lab: where in a step the X row-sum matrix instructions go: 0 one per kNF slots, 1 behind the first X
#endif                             // slots, 2 behind the last X
#ifndef FA_RP16_TOP_BARRIER
#define FA_RP16_TOP_BARRIER 0      // lab: 1 = a barrier at the top of every item but the first (what the votes make redundant)
#endif
#ifndef FA_RP16_STAGGER
#define FA_RP16_STAGGER 0          // lab: workgroups start in 8 phases, this many s_sleep units (~64 clocks each) apart, so that the
#endif                             // item boundaries (stores, next Q) of the CUs do not all hit the memory system together
#ifndef FA_RP16_PRIO
#define FA_RP16_PRIO 0             // lab only: 1 = waves 0-3 (one of the two on each SIMD) run at raised priority
#endif
#ifndef FA_RP16_RUNSUM
#define FA_RP16_RUNSUM 1           // 1: the optimistic passes keep their row-sum chains across steps (16 fewer v_add_f32 per tile)
#endif
#ifndef FA_RP16_VFIX
#define FA_RP16_VFIX 2             // 1: the V image's 32-B key rows are XORed with the head-dim block (db & 3) inside their 256-B block, so that the
#endif                             // eight lanes of a ds_write_b128 group (one key row, chunks 0..7) hit eight 16-B slots of the 128-B bank row
#ifndef FA_RP16_VSPLIT
#define FA_RP16_VSPLIT 1           // 1: a vector pair-step is spread over its matrix slots (one v_exp behind each of the first two, the v_cvt_pk
#endif                             // behind the second / third) instead of all behind the last
#ifndef FA_RP16_RAWBAR
#define FA_RP16_RAWBAR 1           // 1: the tile barrier waits for this wave's staging writes only (counted lgkmcnt), not for the fragment reads behind them
#endif
#ifndef FA_RP16_PAIR
#define FA_RP16_PAIR 1             // 1: narrow waves (X <= 2 at D = 64) run TWO tiles per barrier out of a ring of eight slots; 2: every D = 64 width (lab)
#endif
#ifndef FA_RP16_ASMQK
#define FA_RP16_ASMQK 1
#endif
#ifndef FA_RP16_OLDS
#define FA_RP16_OLDS 0             // lab: fp32 outputs at D = 64 go through a wave-private LDS region and leave as whole 256-B rows (four rows per
#endif                             // store instruction) instead of sixteen 64-B pieces of sixteen rows
#ifndef FA_RP16_STAGE_SLOT
#define FA_RP16_STAGE_SLOT 8       // matrix slot (of 32; scaled for narrower steps) of the second step in front of which tile j+2 is written to LDS
#endif
// Two tiles per loop iteration and barrier (ring of eight [K tile][V tile] slots, tiles landed three ahead instead of two): for
// the narrow waves a tile is a few hundred issue cycles per wave, and the barrier of eight waves plus the landing of the next
// tile cost as much again (stamps at B4 H8 N1024: 16 tiles took 13.8 us = 2000 cycles each).  D = 64 only (128 KB of LDS).
constexpr bool pair_tiles(int D, int X, bool dma) { return FA_RP16_PAIR != 0 && D == 64 && !dma && (X <= 2 || FA_RP16_PAIR == 2); }
}  // namespace rp16

#ifdef FA_EXPERIMENTS
// libfa_mi355_exp.so only (fa_lab_rp16_pass_ids): when set, every workgroup of a non-redo kernel records which pass produced its
// row block -- 0 folded fast pass, 1 exact optimistic pass, 2 running-max pass in the kernel, 3 left to the redo kernel -- at
// [head * blocks per head + block].  The product build carries neither the symbol nor the store.  (One copy per translation
// unit: device symbols do not link across them; fa_fwd_rp16.hip's rp16_set_pass_ids sets them all.)
static __device__ unsigned* g_rp16_pass_ids = nullptr;
static hipError_t rp16_set_pass_ids_tu(unsigned* dev_ptr) { return hipMemcpyToSymbol(HIP_SYMBOL(g_rp16_pass_ids), &dev_ptr, sizeof(dev_ptr)); }
#endif

// kDma: K/V tiles go HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, one 1-KB piece of the K image and one of the V image
// per wave and tile, the images' permutations applied on the SOURCE address) instead of through registers
// (buffer_load -> VGPR -> ds_write_b128).  This is the loader half of the reference's warp-specialised hand-off
// (flashattn_streaming_16x16_mw_v5_warp_specialize.cu:121-185, _v11.cu:189-258) as far as CDNA4 affords it: the
// register file is allocated per kernel, so a ninth (loader) wave would cut every wave to 170 registers, and a loader
// among the eight idles an eighth of the matrix capacity (fixed roles: 43 vs 37.7 cycles per slot in the slot model),
// so every wave issues the DMA for its own eighth of the tile and the hand-off is the counted wait + the tile barrier.
// D = head dim (64 or 128); X = 16-row query blocks per wave (4 at D = 64: 64 rows, 512-row workgroups; 2 at D = 128: 32 rows,
// 256-row workgroups).  A step always is 32 matrix instructions: 2*D/32 K fragments and D/16 V^T fragments, each feeding X blocks.
// kCausal: query row i attends to keys 0..i.  A workgroup runs the tiles up to its last row's diagonal; the tiles its row
// range crosses go through the masked copy of the step (key > row -> -inf), the ones before it through the branch-free
// loop.  Waves are not skipped individually (the pipeline is shared), which costs the upper rows' waves ~3.5 masked tiles
// per item; query blocks alternate direction from one round of the persistent grid to the next (last-to-first, then
// first-to-last), so that every CU's items add up to the same number of tiles.
// kScan: the REDO kernel of the full-width instantiations.  Their stream fills the 256 registers a wave gets at two waves per
// SIMD; the running-max bookkeeping on top of it spills the Q fragments (reloaded from scratch every step, and the allocator's
// choices for the fast passes suffer with it: +17 % on the bench shape).  So a full-width workgroup whose optimistic passes
// fail does not run the running-max pass itself: it leaves a marker word in the first output element of each half of its row
// block, and this kernel -- half-width waves, running-max pass only, launched right behind it on the same stream -- walks ITS
// row blocks, finds the marked ones and computes them.  A marker that happens to equal a genuine output word (a NaN pattern no
// kernel of ours produces) would only cause a block to be computed twice, with the same result; an unmarked failed block cannot
// occur (the marker store is the failing workgroup's only store to the block).
// kWv: waves per workgroup.  8 = two per SIMD, 256 registers each (every shape above).  4 = ONE wave per SIMD with the whole
// 512-register file (accumulators and Q in the upper half): at D = 128 that affords 64-row waves (X = 4), i.e. every LDS
// fragment feeds four matrix instructions instead of two and four waves instead of eight read each tile -- the structure the
// CDNA4 guide documents for d = 128 (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD"), built here on this stream.
// (kWv = 4 with 16-row waves at D = 64 -- 64-row workgroups, two per CU -- was measured for small grids and lost: 23.5 against
// 16.9 us at B4 H8 N1024, every workgroup stages every tile of its head.)
// kKeySplit = 2 (small grids; 16-row waves, D = 64, N a multiple of 128): the workgroup is TWO groups of eight waves over the
// same 128 query rows, group s running the whole algorithm -- its own LDS ring, its own reference, every pass -- over keys
// [s N/2, (s+1) N/2); the votes and barriers are workgroup-wide (both groups run the same number of tiles), and group 1 hands
// (O, l, m) to group 0 through LDS at the end, which merges with 2^(m_s - M) weights and stores.  A wave's chain of tiles
// halves and four waves share a SIMD instead of two -- for grids where a workgroup per CU runs a handful of tiles and waits
// on LDS latency and the barrier most of the time.
template <typename T, int D, int X, bool kOutF32, bool kFold, bool kDma = false, bool kCausal = false, bool kScan = false, int kWv = 8,
          int kKeySplit = 1>
__global__ __launch_bounds__(64 * kWv * kKeySplit, kWv * kKeySplit / 4)
void fa_fwd_rp16_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                        const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                        int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace rp16;
    constexpr int kW = kWv;   // (hides rp16::kW, the default)
    static_assert(kWv == 8 || (kWv == 4 && !kDma && !kScan), "waves per (key-split group of a) workgroup: 8 or 4");
    static_assert(kKeySplit == 1 || (kKeySplit == 2 && ((kWv == 8 && X == 1) || (kWv == 4 && X == 2)) && D == 64 && !kDma && !kCausal && !kScan),
                  "key split: 128-row workgroups at D = 64 (2 x 8 waves of 16 rows, or 2 x 4 waves of 32 rows)");
    using M = Mx<T>;
    using G = TileGeom<D>;
    // The folded pass multiplies Q'.K on the fp16 matrix instruction whatever the input type: Q' = fp16(Q * scale * log2 e)
    // (bf16's 8 bits would move a logit by |logit| * 2^-8), and bf16 K is converted to fp16 while it is staged -- exact for
    // every bf16 value up to 65504 in magnitude (larger ones raise the gate; smaller ones lose at most 2^-25 absolutely).
    // P and V stay in the input type for O^T += V^T.P^T.
    constexpr bool kCvtK = kFold && T::id == 1;
    static_assert(!kScan || (!kFold && !kDma), "the redo kernel runs the running-max pass only");
    // full-width waves (64 rows at D = 64, 32 at D = 128): the running-max pass lives in the redo kernel (see kScan)
    constexpr bool kSplitTrack = !kScan && !kDma && 16 * X * (D / 64) >= 64;
    constexpr unsigned kMarker = 0x7FA5C0DEu;
    constexpr bool kVSplit = FA_RP16_VSPLIT != 0 && (!kCausal || (kWv == 4 && kKeySplit == 1));   // (under the mask the pins cost the two-wave full-width kernels spills)
    static_assert(!(kCvtK && kDma), "the DMA path cannot convert K on the way");
    static_assert(!kDma || D == 64, "the DMA piece maps are written for 128-byte rows");
    constexpr int kRows = 16 * X * kW;
    constexpr int kKS = D / 32, kDB = D / 16;   // k-steps of QK^T, 16-row blocks of O^T
    constexpr int kNF = 2 * kKS + kDB;          // fragments per step (K and V^T alternate: 2 kKS == kDB)
    // fragment registers and read-ahead: a fragment feeds X matrix instructions, so the narrow waves (X < 4 at D = 64: small
    // grids) need more of them in flight to cover the LDS latency
#ifndef FA_RP16_D128_DEEP
#define FA_RP16_D128_DEEP 0        // lab: 32-row waves at D = 128 with the narrow waves' ring of eight fragment registers, read four ahead
#endif                             // (a fragment there feeds two matrix instructions only: 2 ahead = 64 cycles of cover)
    constexpr bool kWide = 16 * X * (D / 64) >= 64 && !(FA_RP16_D128_DEEP && D == 128);
    constexpr int kRing = kWide ? 4 : 8;
    constexpr int kAhead = kWide ? kAheadWide : (X == 2 ? 4 : 6);
    constexpr int kSlots = kNF * X;             // matrix instructions per step (32 for the 64-row waves: X = 4 at D = 64, 2 at D = 128)
    static_assert(2 * kKS == kDB && kNF % kRing == 0 && kSlots % (4 * X) == 0, "fragment ring / vector pair-steps divide a step");
    constexpr int kLoads = (kBlockN * G::kChunks) / (64 * kW);   // 16-B chunks of K (and of V) per thread and tile
    // LDS instructions a wave issues in the second step behind the landing of tile j+2 (fragment reads: one ds_read_b128 per K
    // fragment, two ds_read_b64_tr_b16 per V^T fragment)
    constexpr int kLandSlot = FA_RP16_STAGE_SLOT * kSlots / 32;
    // one wave per SIMD: no second wave issues while this one works through a bunch of loads or LDS writes, so tile j+2 is
    // requested one chunk per matrix slot (first step) and landed one chunk per slot (second step, from kLandSlot on)
#ifndef FA_RP16_SPREAD
#define FA_RP16_SPREAD 1
#endif
    constexpr bool kOneWave = kWv == 4 && kKeySplit == 1;   // one wave per SIMD
    constexpr bool kSpread = FA_RP16_SPREAD != 0 && kOneWave && !kDma;
    constexpr int kLandLast = kSpread ? kLandSlot + 2 * kLoads - 1 : kLandSlot;   // the slot of the last landing write
    // ... and the tile barrier is replaced by one flag word per wave behind the ring: a wave publishes "tile j+2 landed" (its
    // iteration count) right behind its last landing write and looks at all four flags only in front of its first read of that
    // tile, most of a step later -- LDS operations of a wave complete in order, so the flag follows the data and the reads
    // follow the look.  A lone wave per SIMD that waits at s_barrier for the slowest of four idles its matrix pipe; here the
    // waves may drift by most of a step.  (Ring reuse: a wave lands tile j+3 over tile j-1 only behind its look of iteration
    // j+1, i.e. when every wave has landed tile j+2 -- 16 slots into the second step of iteration j, past its last read of
    // tile j-1 in the first.)
#ifndef FA_RP16_FLAGBAR
#define FA_RP16_FLAGBAR 1
#endif
    constexpr bool kFlagBar = (FA_RP16_FLAGBAR == 2 ? (!kDma && !pair_tiles(D, X, kDma) && kKeySplit == 1 && !kScan && FA_RP16_OLDS == 0) : (kSpread && FA_RP16_FLAGBAR != 0)) &&
                              (FA_RP16_ABL & 24) == 0;
    constexpr int kFlagCheck = (kNF - kAhead) * X + X - 2, kFlagRead = kFlagCheck >= 8 ? kFlagCheck - 8 : 0;   // the slot in front of the first read-ahead into the next step
    static_assert(!kFlagBar || (kFlagRead >= 0 && kFlagRead < kFlagCheck && kLandLast + 1 < kSlots), "flag slots");
    static_assert(kLandLast < kSlots && 2 * kLoads <= kSlots, "the landing fits the step");
    constexpr int kLdsAfterLand = [] {
        int n = 0;
        for (int i = kLandLast; i < kSlots; ++i)
            if (i % X == X - 1) n += ((i / X + kAhead) & 1) ? 2 : 1;
        return n;
    }();
    constexpr unsigned kRowB = D * 2;
    constexpr unsigned kTile = kBlockN * D * 2;
    constexpr unsigned kSlotBytes = 2 * kTile;      // [K tile][V tile]
    constexpr bool kPair = pair_tiles(D, X, kDma) && kKeySplit == 1 && kWv == 8;   // (two rings of eight slots do not fit; one-wave kernels land per slot and use flags)
    // one wave per SIMD: QK^T spelled out with the scores in architectural registers (Mx::mfma_v*).  Every vector read of a score
    // lies at least X P.V matrix instructions behind the instruction that wrote it (the units alternate QK^T and P.V fragments and a
    // step ends with a P.V fragment); the prologue, whose reference maximum reads unit 0 at once, waits explicitly (settle).
    constexpr bool kAsmQK = FA_RP16_ASMQK != 0 && kWv == 4 && kKeySplit == 1;   // (the key-split kernel has four waves per GROUP: two per SIMD, builtins)
    constexpr unsigned kRingSlots = kPair ? 8u : 4u, kRingMask = kRingSlots - 1u;
    constexpr int kLook = kPair ? 3 : 2;            // a tile is landed this many tiles ahead of the iteration that starts with it
    extern __shared__ __attribute__((aligned(16))) char smem_all[];   // one ring of slots per key-split group
    const unsigned grp = kKeySplit == 1 ? 0u : (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x / (64u * kW));
    char* const smem = smem_all + grp * (kRingSlots * kSlotBytes);
    const unsigned tid  = kKeySplit == 1 ? threadIdx.x : threadIdx.x % (64u * kW);   // within the group
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned c16 = lane & 15u, g = lane >> 4;
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    const int Nkv = N / kKeySplit;                       // keys of this group (the host checked the divisibility)
    const size_t kv_first = (size_t)grp * Nkv * D;       // its first K / V element inside a head
    const unsigned kv_bytes = (unsigned)((size_t)Nkv * D * 2);
    const int ntiles = (Nkv + kBlockN - 1) / kBlockN;
    const bool partial = (Nkv % kBlockN) != 0;

    unsigned st_goff[kLoads], sv_goff[kLoads], k_lds[kLoads], v_lds[kLoads];
#pragma unroll
    for (int p = 0; p < kLoads; ++p) {
        const unsigned idx = tid + p * 64u * kW;
        unsigned srow = idx / G::kChunks, sch = idx % G::kChunks;
        st_goff[p] = srow * kRowB + sch * 16u;
        k_lds[p] = G::k_off(srow, sch);
        if constexpr (FA_RP16_VFIX == 2) {
            // V: the eight lanes of a ds_write_b128 group take 4 keys x 2 chunks of one head-dim block (128 contiguous bytes of
            // the image) instead of one key's 8 chunks (8 slots 256 B apart: 4-way on the 128-B bank row of a write)
            constexpr unsigned ndb = G::kChunks / 2, rpw = 64u / G::kChunks;   // head-dim blocks; key rows per wave-instruction
            const unsigned w = idx >> 6, l = idx & 63u, t = l >> 3;
            srow = w * rpw + 4u * (t / ndb) + ((l >> 1) & 3u);
            sch = 2u * (t % ndb) + (l & 1u);
        }
        sv_goff[p] = srow * kRowB + sch * 16u;
        v_lds[p] = kTile + ((srow >> 3) * (unsigned)kDB + (sch >> 1)) * 256u + (((srow & 7u) ^ (FA_RP16_VFIX == 1 ? ((sch >> 1) & 3u) : 0u)) << 5) + ((sch & 1u) << 4);
    }
    // LDS-DMA: this wave's 1-KB piece of an image is bytes [1024 wave, +1024), lane l lands at +16 l; where that comes from
    const unsigned dk_row = 8u * wave + (lane >> 3), dk_slot = lane & 7u;
    const unsigned k_src = dk_row * kRowB + ((dk_slot ^ G::k_swz(dk_row)) << 4);
    const unsigned dv_l = 1024u * wave + 16u * lane, dv_blk = dv_l >> 8;
    const unsigned dv_row = (dv_blk / (unsigned)kDB) * 8u + (((dv_l & 255u) >> 5) ^ (FA_RP16_VFIX == 1 ? ((dv_blk % (unsigned)kDB) & 3u) : 0u)), dv_ch = (dv_blk % (unsigned)kDB) * 2u + ((dv_l >> 4) & 1u);
    const unsigned v_src = dv_row * kRowB + dv_ch * 16u;
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_tile = [&](__amdgpu_buffer_rsrc_t rks, __amdgpu_buffer_rsrc_t rvs, unsigned tile_off, unsigned slot_off) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rks, (lds_void*)(smem + slot_off + 1024u * wave), 16, tile_off + k_src, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rvs, (lds_void*)(smem + slot_off + kTile + 1024u * wave), 16, tile_off + v_src, 0, 0, 0);
    };
    unsigned k_rd[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) k_rd[ks] = c16 * kRowB + (((4u * ks + g) ^ G::k_swz(c16)) << 4);
    // (FA_RP16_VFIX: one base per db & 3 -- the key row inside the 256-B block is XORed with it)
    unsigned v_rd4[4];
#pragma unroll
    for (unsigned dq = 0; dq < 4u; ++dq)
        v_rd4[dq] = kTile + (g >> 1) * (unsigned)kDB * 256u + ((4u * (g & 1u) + ((c16 >> 2) ^ (FA_RP16_VFIX == 1 ? dq : 0u))) << 5) + (c16 & 3u) * 8u;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4u = {0u, 0u, 0u, 0u};
    const std::true_type yes{};
    const std::false_type no{};
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    // v of the lane `sh` places further round the lane's row of 16 (DPP row_ror: no LDS round trip like ds_bpermute); four
    // doubling steps (1, 2, 4, 8) leave the row's maximum / sum in every lane
    auto row_ror = [&](float v, int sh) -> float {
        const int iv = __builtin_bit_cast(int, v);
        int r;
        switch (sh) {
            case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x121, 0xF, 0xF, true); break;
            case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0x122, 0xF, 0xF, true); break;
            case 4: r = __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xF, 0xF, true); break;
            default: r = __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, true); break;
        }
        return __builtin_bit_cast(float, r);
    };
    auto across_max = [&](float v) -> float {   // over the four lanes that share a query row
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        return fmaxf(v, __shfl_xor(v, 32, 64));
    };
    auto across_sum = [&](float v) -> float {
        v += __shfl_xor(v, 16, 64);
        return v + __shfl_xor(v, 32, 64);
    };

    if constexpr (FA_RP16_STAGGER > 0) {
        const unsigned phase = (blockIdx.x >> 3) & 7u;
        for (unsigned i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(FA_RP16_STAGGER);
    }
    if constexpr (FA_RP16_PRIO == 1) { if (wave < 4u) __builtin_amdgcn_s_setprio(2); }   // lab: one wave of each SIMD's pair ahead
    const unsigned nwg = total_wg;
    // work item -> (head, query block): XCD-aware remap of the persistent grid's item index
    auto locate = [&](unsigned bid_, unsigned& bh_, unsigned& qb_) __attribute__((always_inline)) {
        const unsigned xq = nwg >> 3, xr = nwg & 7u, xcd = bid_ & 7u;
        const unsigned wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid_ >> 3);
        bh_ = wgid / (unsigned)nqb;
        const unsigned qbi = wgid - bh_ * (unsigned)nqb;
        qb_ = qbi;
        if constexpr (kCausal) {
            // Alternate the direction of the query blocks from one round of the persistent grid to the next, so that a CU's
            // items add up to about the same number of tiles.  The direction must be a function of the HEAD alone (all its
            // query blocks flip together, else two items would compute the same block): take the round of the head's first
            // item, found through the inverse of the XCD remap above.
            const unsigned t0 = bh_ * (unsigned)nqb, big = xr * (xq + 1u);
            const unsigned x0 = t0 < big ? t0 / (xq + 1u) : xr + (t0 - big) / (xq ? xq : 1u);
            const unsigned start0 = x0 < xr ? x0 * (xq + 1u) : big + (x0 - xr) * xq;
            const unsigned bid0 = 8u * (t0 - start0) + x0;
            if (((bid0 / gridDim.x) & 1u) == 0u) qb_ = (unsigned)nqb - 1u - qbi;
        }
    };
    const size_t head_elems = (size_t)N * D;
    const unsigned head_bytes = (unsigned)(head_elems * 2);
    // Between two items of the persistent loop everything is a latency chain (stamps: Q 2.8-4.4 us, then K/V 2.2, reference
    // 2.2, gates 1.8, stores 1.6 of ~116 us per item at B8 H16 N4096).  kPrefetch: the NEXT item's Q rows are requested
    // (raw, into qf -- dead by then) as soon as the first pass' tile loop is over, i.e. ahead of this item's stores in the
    // in-order vector memory queue, and every item requests its first three K/V tiles before it waits for its Q.
    constexpr bool kPrefetch = FA_RP16_PREFETCH != 0 && !kDma;
    constexpr bool kCarryKV = FA_RP16_PREFETCH == 2;   // lab: the K/V tiles 0..2 carried in registers as well (the allocator spills them)
    u32x4 qf[X][kKS];   // B operand of QK^T: Q[row of block x][32 ks + 8 g .. +7]
    u32x4 kst[kLoads], vst[kLoads];
    u32x4 kst2[kLoads], vst2[kLoads];   // kPair: the second tile of an iteration
    u32x4 pfk[2][kLoads], pfv[2][kLoads];
    // hb: the head's Q; row_base: the wave's first row (wave-uniform, folded into the descriptor: the bounds check -- rows past
    // N read zeros -- covers the per-lane and the immediate offset only).  One per-lane address, recomputed here from the lane
    // id so that nothing of it lives across the tile loop.
    auto q_issue = [&](const uint16_t* hb, unsigned row_base) __attribute__((always_inline)) {
        unsigned l = lane;
        asm volatile("" : "+v"(l));
        const unsigned voff = (l & 15u) * kRowB + (l >> 4) * 16u;
        const unsigned rb = __builtin_amdgcn_readfirstlane(row_base);
        const unsigned skip = rb * kRowB;
        // (the pointer is wave-uniform by construction; saying so spares the descriptor a waterfall loop per load)
        const unsigned long long pa = (unsigned long long)(hb + (size_t)rb * D);
        const unsigned p_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa);   // (the builtin returns int: no sign
        const unsigned p_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));   // extension into the high half)
        const unsigned long long pu = (unsigned long long)p_lo | ((unsigned long long)p_hi << 32);
        const __amdgpu_buffer_rsrc_t rq_ = make_rsrc(reinterpret_cast<const uint16_t*>(pu),
                                                     __builtin_amdgcn_readfirstlane(skip < head_bytes ? head_bytes - skip : 0u));
#pragma unroll
        for (int x = 0; x < X; ++x)
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if constexpr ((FA_RP16_ABL & 64) != 0) qf[x][ks] = zero4u;
                else qf[x][ks] = buf_load16(rq_, voff + ((16u * x) * kRowB + 64u * ks));
            }
    };
    auto kv_issue = [&](__amdgpu_buffer_rsrc_t rk_, __amdgpu_buffer_rsrc_t rv_) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < kLoads; ++p) {   // all loads of the three tiles in flight together
            if constexpr ((FA_RP16_ABL & 128) != 0) {
                pfk[0][p] = pfv[0][p] = pfk[1][p] = pfv[1][p] = kst[p] = vst[p] = zero4u;
                continue;
            }
            pfk[0][p] = buf_load16(rk_, st_goff[p]);
            pfv[0][p] = buf_load16(rv_, sv_goff[p]);
            pfk[1][p] = buf_load16(rk_, kTile + st_goff[p]);
            pfv[1][p] = buf_load16(rv_, kTile + sv_goff[p]);
            kst[p] = buf_load16(rk_, 2u * kTile + st_goff[p]);
            vst[p] = buf_load16(rv_, 2u * kTile + sv_goff[p]);
        }
    };
    // kScan: the marked row blocks among those this workgroup owns (block b = blockIdx.x + i gridDim.x).  Thread t looks at the
    // t-th of them, all loads in flight together; the marked ones are collected in LDS behind the ring and computed in turn.
    constexpr unsigned kScanBatch = 64u * kW;
    unsigned scan_base = 0, scan_n = 0, scan_i = 0;
    auto scan_next = [&]() __attribute__((always_inline)) -> unsigned {
        unsigned* list = reinterpret_cast<unsigned*>(smem + kRingSlots * kSlotBytes);   // [0] = count, [1 ..] = block ids
        while (scan_i == scan_n) {
            if (blockIdx.x + scan_base * gridDim.x >= nwg) return nwg;
            const unsigned long long b = (unsigned long long)blockIdx.x + (unsigned long long)(scan_base + tid) * gridDim.x;
            bool marked = false;
            if (b < nwg) {
                unsigned bh_, qb_;
                locate((unsigned)b, bh_, qb_);
                constexpr unsigned es_ = kOutF32 ? 4u : 2u;
                const unsigned* w0 = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(Og) +
                                                                       ((size_t)bh_ * head_elems + (size_t)qb_ * kRows * D) * es_);
                marked = *w0 == kMarker;
            }
            scan_i = scan_n = 0;
            scan_base += kScanBatch;
            // (the vote is also the barrier behind which every wave is done with the previous batch's list and with the ring)
            if (!__syncthreads_or(marked ? 1 : 0)) continue;   // the common case: nothing to do in this batch
            if (tid == 0) list[0] = 0u;
            __syncthreads();
            if (marked) list[1u + atomicAdd(&list[0], 1u)] = (unsigned)b;
            __syncthreads();
            scan_n = __builtin_amdgcn_readfirstlane(list[0]);
        }
        const unsigned r = __builtin_amdgcn_readfirstlane(list[1u + scan_i]);
        ++scan_i;
        return r;
    };
    // (the block after the current one, if the list already holds it: its Q rows are requested ahead, like the fast kernels do)
    auto scan_peek = [&]() __attribute__((always_inline)) -> unsigned {
        const unsigned* list = reinterpret_cast<const unsigned*>(smem + kRingSlots * kSlotBytes);
        return scan_i < scan_n ? (unsigned)__builtin_amdgcn_readfirstlane(list[1u + scan_i]) : nwg;
    };
    const unsigned first_bid = kScan ? scan_next() : blockIdx.x;
    [[maybe_unused]] bool q_pending = false;   // kScan: the current block's Q rows were requested by the block before it
    constexpr unsigned kStores = (unsigned)(X * kDB);   // store instructions per item
    if constexpr (kPrefetch) {
        // The first item's inputs, requested the way every later item's are (at the end of the item before it, ahead of that
        // item's stores) -- including kStores stores, so that both ways into the loop look alike to the wait-count
        // bookkeeping (s_waitcnt vmcnt counts in order: with the same instructions behind the loads on both paths the waits
        // for Q and K/V can leave exactly the stores outstanding).  The stand-in stores put one zero chunk per wave on the first
        // row the wave will really store later (same wave, same address, program order: the real value wins).
        if (first_bid < nwg) {
            unsigned bh0, qb0;
            locate(first_bid, bh0, qb0);
            q_pending = true;
            bh0 = __builtin_amdgcn_readfirstlane(bh0);
            qb0 = __builtin_amdgcn_readfirstlane(qb0);
            const unsigned rb0 = qb0 * kRows + wave * (16u * X);
            q_issue(Qg + bh0 * head_elems, rb0);
            if constexpr (kCarryKV) kv_issue(make_rsrc(Kg + bh0 * head_elems, head_bytes), make_rsrc(Vg + bh0 * head_elems, head_bytes));
            constexpr unsigned es0 = kOutF32 ? 4u : 2u;
            const __amdgpu_buffer_rsrc_t ro0 =
                make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh0 * head_elems * es0, (unsigned)(head_elems * es0));
#pragma unroll
            for (unsigned i = 0; i < (grp == 0u ? kStores : 0u); ++i) {   // (key-split group 1 never stores to O)
                u32x4 z = zero4u;
                asm volatile("" : "+v"(z));
                if constexpr (kOutF32) buf_store16(ro0, rb0 * D * 4u, z);
                else buf_store8(ro0, rb0 * D * 2u, u32x2{z[0], z[1]});
            }
        }
    }
    [[maybe_unused]] unsigned epoch = 0u, flag_seen = 0u;   // kFlagBar: iterations this wave has published (uniform; the same in every wave)
    [[maybe_unused]] const unsigned flag_base = lds_addr(smem_all) + kRingSlots * kSlotBytes;
    if constexpr (kFlagBar) {
        if (tid < (unsigned)kW) lds_write4_at(flag_base + 4u * tid, 0u);
        __syncthreads();
    }
    for (unsigned bid = first_bid; bid < nwg; bid = kScan ? scan_next() : bid + gridDim.x) {
#ifdef FA_RP16_STAMPS   // lab: 100 MHz timestamps of the item's phases, written over O[first row of the item][0..7] (fp32 out only)
    unsigned long long ts[8] = {};
#define FA_STAMP(i) ts[i] = wall_clock64()
#else
#define FA_STAMP(i)
#endif
    FA_STAMP(0);
    // No barrier here: the ring is only written again after the item's prologue loads have arrived, and every wave's last
    // LDS read of the previous item (the epilogue's V fragments) lies before that item's vote (__syncthreads_or) -- or the
    // barrier behind the tracked pass, which has no vote after it.  A wave therefore requests its K/V tiles as soon as its
    // own stores are issued, not when the slowest wave's are.
    if constexpr (FA_RP16_TOP_BARRIER) { if (bid != blockIdx.x) __syncthreads(); }
    unsigned bh, qb;
    locate(bid, bh, qb);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems + kv_first, kv_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems + kv_first, kv_bytes);
    [[maybe_unused]] const char* const k_head = reinterpret_cast<const char*>(Kg + bh * head_elems + kv_first);
    [[maybe_unused]] const char* const v_head = reinterpret_cast<const char*>(Vg + bh * head_elems + kv_first);
    const unsigned q_row0 = qb * kRows + wave * (16u * X) + c16;   // row of block 0; block x is 16x rows further
    // causal: tiles [0, nt) with nt up to the diagonal of the workgroup's last (existing) row; tiles >= jc cross its row range
    const int nt = kCausal ? min(ntiles, (int)(min((unsigned)N - 1u, qb * kRows + kRows - 1u) / kBlockN) + 1) : ntiles;
    const int jc = kCausal ? (int)((qb * kRows) / kBlockN) : nt;

    int q_bad = 0;
    auto q_finish = [&](auto fold_c) __attribute__((always_inline)) {   // raw rows in qf -> the B operands of this pass
        constexpr bool fold = decltype(fold_c)::value;
#pragma unroll
        for (int x = 0; x < X; ++x) {
            float amax = 0.0f;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                u32x4 raw = qf[x][ks];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if constexpr (fold) {
                        const float lo = T::lo(raw[w]) * scale_log2e, hi = T::hi(raw[w]) * scale_log2e;
                        amax = max3(amax, fabsf(lo), fabsf(hi));
                        raw[w] = F16::pack2(lo, hi);
                    } else {
                        raw[w] ^= q_flip;
                    }
                }
                qf[x][ks] = raw;
            }
            if constexpr (fold) q_bad |= (int)!(amax <= 65504.0f) | ((int)(amax != 0.0f) & (int)(amax < 6.2e-5f));
        }
        // pin the flag HERE: left to itself the compiler evaluates it after the tile loop and keeps all 64 fp32
        // products alive (spilled) across it -- 33 MB of scratch written and read back per item
        if constexpr (fold) asm volatile("" : "+v"(q_bad));
    };
    auto load_q = [&](auto fold_c) __attribute__((always_inline)) {
        q_issue(Qg + bh * head_elems, q_row0 - c16);
        q_finish(fold_c);
    };

    // bf16 K chunk -> fp16 (folded pass of bf16 inputs); k_amax collects the largest magnitude this thread converted
    float k_amax = 0.0f;
    auto k_to_f16 = [&](u32x4 kb) -> u32x4 {
        u32x4 r;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float lo = BF16::lo(kb[w]), hi = BF16::hi(kb[w]);
            k_amax = max3(k_amax, fabsf(lo), fabsf(hi));
            r[w] = F16::pack2(lo, hi);
        }
        return r;
    };

    f32x4 o[X][kDB];
    float m_ref[X] = {}, l_part[X] = {};
    float ls[X][2];   // optimistic passes: two running row-sum chains per block, folded into l_part once per item
    u32x4 frag[kRing];
    f32x4 minit;   // folded pass: every score chain starts at -(wave reference maximum)
    // row sums on the matrix pipe: lacc[x][i] = sum over keys of the ROUNDED weights of row (lane & 15) of block x, the
    // same in every register and every lane group (all 16 "head-dim rows" of the ones fragment are equal)
    f32x4 lacc[X];
    u32x4 ones;   // written by an instruction the optimiser cannot hoist out of the item loop (and spill around the tile loop)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(ones[i]) : "s"(T::kOnes2));

    // K fragment (key block kbl of half h in slot offset so, k-step ks); V^T fragment (head-dim block db) of half h
    auto read_kf = [&](unsigned so, int h, int kbl, int ks) -> u32x4 {
        return lds_read16(smem, so + (unsigned)(2 * h + kbl) * 16u * kRowB + k_rd[ks]);
    };
    auto read_vf = [&](unsigned so, int h, int db) -> u32x4 {
        u32x4 vf;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const u32x2 half = lds_read_tr8(smem, so + v_rd4[FA_RP16_VFIX == 1 ? (db & 3) : 0] + (4u * h + 2u * jj) * (unsigned)kDB * 256u + db * 256u);
            vf[2 * jj] = half[0];
            vf[2 * jj + 1] = half[1];
        }
        return vf;
    };
    // kBases (one wave per SIMD): a step's fragment addresses from per-step lane bases (ring slot + lane offset, one v_add
    // each at the top of the step) plus immediates, instead of one s_add + v_add in front of every fragment read -- with a
    // lone wave per SIMD every such instruction is a cycle the matrix pipe waits for
    constexpr bool kBases = kSpread && FA_RP16_VFIX == 2 && (FA_RP16_ABL & 257) == 0;
    auto read_frag_b = [&](auto fc, const unsigned (&kb)[kKS], unsigned vb, int h_q, int h_v) {
        constexpr int f = decltype(fc)::value;
        if constexpr ((f & 1) == 0) {
            constexpr int kbl = (f >> 1) / kKS, ks = (f >> 1) % kKS;
            frag[f % kRing] = lds_read16_at(kb[ks] + (unsigned)(2 * h_q + kbl) * 16u * kRowB);
        } else {
            constexpr int db = f >> 1;
            u32x4 vf;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const u32x2 half = lds_read_tr8_at(vb + (4u * h_v + 2u * jj) * (unsigned)kDB * 256u + db * 256u);
                vf[2 * jj] = half[0];
                vf[2 * jj + 1] = half[1];
            }
            frag[f % kRing] = vf;
        }
    };
    // fragment f (0..kNF-1) of a step: even f -> K fragment (kbl = (f/2) / kKS, ks = (f/2) % kKS) of the QK^T unit,
    // odd f -> V^T fragment db = f/2 of the PV unit
    auto read_frag = [&](auto fc, unsigned so_q, int h_q, unsigned so_v, int h_v) {
        constexpr int f = decltype(fc)::value;
        if constexpr ((FA_RP16_ABL & 1) != 0) {   // "defined" without an instruction, so that no consumer is folded away
            asm volatile("" : "=v"(frag[f % kRing]));
            return;
        }
        if constexpr ((f & 1) == 0) frag[f % kRing] = read_kf(so_q, h_q, (f >> 1) / kKS, (f >> 1) % kKS);
        else frag[f % kRing] = read_vf(so_v, h_v, f >> 1);
        if constexpr ((FA_RP16_ABL & 256) != 0) {   // lab: every fragment read issued twice (the neighbouring fragment, discarded): what LDS operand traffic costs
            u32x4 dup;
            if constexpr ((f & 1) == 0) dup = read_kf(so_q, h_q, (f >> 1) / kKS, ((f >> 1) % kKS) ^ 1);
            else dup = read_vf(so_v, h_v, (f >> 1) ^ 1);
            asm volatile("" :: "v"(dup));
        }
    };
    auto mask_unit = [&](int tile, int h, f32x4 (&s)[X][2]) {   // keys >= N (causal: keys after the query) -> -inf (p = 0)
#pragma unroll
        for (int x = 0; x < X; ++x)
#pragma unroll
            for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = tile * kBlockN + 32 * h + 16 * kbl + 4 * (int)g + i;
                    if (key >= Nkv || (kCausal && (unsigned)key > q_row0 + 16u * x)) s[x][kbl][i] = -INFINITY;
                }
    };
    auto row_max = [&](const f32x4 (&s)[2]) -> float {   // this row's 32 keys of the unit, unscaled
        const float a = max3(s[0][0], s[0][1], s[0][2]), b = max3(s[1][0], s[1][1], s[1][2]);
        return across_max(max3(a, b, fmaxf(s[0][3], s[1][3])));
    };

    // One step of an optimistic pass.  h = half of tile `tile` being softmaxed (s_cur -> pk_cur); the QK^T unit is
    // (so_q, 1-h) -> s_nxt, the PV unit (so_v, 1-h) <- pk_prev.  so_nq / so_nv: slots of the NEXT step's units.
    auto step = [&](auto h_c, auto masked_c, auto fast_c, auto track_c, int tile, f32x4 (&s_cur)[X][2], f32x4 (&s_nxt)[X][2],
                    u32x4 (&pk_prev)[X], u32x4 (&pk_cur)[X], unsigned so_q, unsigned so_v, unsigned so_nq, unsigned so_nv,
                    unsigned so_land, auto set_c, auto req_c) __attribute__((always_inline)) {
        u32x4 (&k_land)[kLoads] = decltype(set_c)::value == 0 ? kst : kst2;   // the staging registers this step lands (h = 1)
        u32x4 (&v_land)[kLoads] = decltype(set_c)::value == 0 ? vst : vst2;
        constexpr int h = decltype(h_c)::value, ho = 1 - h;
        constexpr bool kFast = decltype(fast_c)::value;
        if constexpr (decltype(masked_c)::value) mask_unit(tile, h, s_cur);
        if constexpr (decltype(track_c)::value) {
            // The running maximum of the online softmax (flashattn_streaming_16x16_mw.cu:200-229, _v12f.cu:193-220), lazily: the
            // reference of a row moves only when this unit's scores exceed it by more than kThr log2 units (p <= 2^kThr fits the
            // 16-bit weights), decided by ONE wave vote over per-lane maxima -- no cross-lane step unless it fires.  When it does,
            // everything still in the old scale is multiplied by 2^(old - new) exactly once: O, the row sums and the packed P of
            // the previous unit, whose P.V is issued during this step (guide T13: the decision must not split a pending P.V).
            // (v_max3_f32 spelled out: fmaxf on matrix results gets a canonicalising v_max per operand in front of it)
            auto max3r = [](float a, float b, float d) -> float {
                float r;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(d));
                return r;
            };
            auto lane_max = [&](int x) -> float {   // this lane's 8 keys of block x's row, scaled
                const float a = max3r(s_cur[x][0][0], s_cur[x][0][1], s_cur[x][0][2]), b = max3r(s_cur[x][1][0], s_cur[x][1][1], s_cur[x][1][2]);
                return max3r(a, b, max3r(s_cur[x][0][3], s_cur[x][1][3], s_cur[x][1][3])) * c;
            };
            int need = 0;   // (no short-circuit: one compare per block, no branch)
#pragma unroll
            for (int x = 0; x < X; ++x) need |= (int)(lane_max(x) - m_ref[x] > kThr);
            if (__any(need != 0)) {
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    const float m_new = fmaxf(m_ref[x], across_max(lane_max(x)));   // (a row that stayed below simply moves to its own maximum)
                    const float alpha = fast_exp2(m_ref[x] - m_new);
                    m_ref[x] = m_new;
#pragma unroll
                    for (int db = 0; db < kDB; ++db)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[x][db][i] *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) lacc[x][i] *= alpha;
                    ls[x][0] *= alpha;
                    ls[x][1] *= alpha;
                    l_part[x] *= alpha;
#pragma unroll
                    for (int w = 0; w < 4; ++w) pk_prev[x][w] = T::pack2(T::lo(pk_prev[x][w]) * alpha, T::hi(pk_prev[x][w]) * alpha);
                }
            }
        }

        constexpr int kPairs = 4 * X;   // vector pair-steps: pair j = (block j/4, key block (j/2)&1, registers 2(j&1), 2(j&1)+1)
        if constexpr (!FA_RP16_RUNSUM) {
#pragma unroll
            for (int x = 0; x < X; ++x) ls[x][0] = ls[x][1] = 0.0f;
        }
        auto fma_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 2, kbl = (j >> 1) & 1, e = 2 * (j & 1);
            s_cur[x][kbl][e] = __builtin_fmaf(s_cur[x][kbl][e], c, -m_ref[x]);
            s_cur[x][kbl][e + 1] = __builtin_fmaf(s_cur[x][kbl][e + 1], c, -m_ref[x]);
        };
        auto exp_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 2, kbl = (j >> 1) & 1, e = 2 * (j & 1);
            s_cur[x][kbl][e] = fast_exp2(s_cur[x][kbl][e]);
            s_cur[x][kbl][e + 1] = fast_exp2(s_cur[x][kbl][e + 1]);
        };
        auto fma_one = [&](auto jc, auto ec) {
            constexpr int j = decltype(jc)::value, x = j >> 2, kbl = (j >> 1) & 1, e = 2 * (j & 1) + decltype(ec)::value;
            s_cur[x][kbl][e] = __builtin_fmaf(s_cur[x][kbl][e], c, -m_ref[x]);
        };
        auto exp_one = [&](auto jc, auto ec) {   // (pinned: the value exists at this point of the stream, not where its consumer is)
            constexpr int j = decltype(jc)::value, x = j >> 2, kbl = (j >> 1) & 1, e = 2 * (j & 1) + decltype(ec)::value;
            float p = fast_exp2(s_cur[x][kbl][e]);
            asm volatile("" : "+v"(p));
            s_cur[x][kbl][e] = p;
        };
        auto fin_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 2, kbl = (j >> 1) & 1, e = 2 * (j & 1);
            unsigned w = T::pack2(s_cur[x][kbl][e], s_cur[x][kbl][e + 1]);
            if constexpr (kVSplit) asm volatile("" : "+v"(w));
            pk_cur[x][2 * kbl + (j & 1)] = w;
            if constexpr (FA_RP16_SUMMFMA) {
            } else if constexpr (T::kSumRounded) {
                ls[x][j & 1] = T::sum2(w, ls[x][j & 1]);
            } else {
                ls[x][0] += s_cur[x][kbl][e];
                ls[x][1] += s_cur[x][kbl][e + 1];
            }
        };
        auto valu_step = [&](auto jc) {   // skewed: nothing waits on the instruction before it
            constexpr int j = decltype(jc)::value;
            if constexpr ((FA_RP16_ABL & 2) != 0) {   // keep the scores "used" without an instruction
                if constexpr (j == 0) {
#pragma unroll
                    for (int x = 0; x < X; ++x) asm volatile("" :: "v"(s_cur[x][0]), "v"(s_cur[x][1]));
                }
                return;
            }
            if constexpr (j + 2 < kPairs && !kFast) fma_pair(std::integral_constant<int, j + 2>{});
            if constexpr (j + 1 < kPairs) exp_pair(std::integral_constant<int, j + 1>{});
            fin_pair(jc);
        };
        auto issue_mfma = [&](auto ic) {
            constexpr int i = decltype(ic)::value, f = i / X, x = i % X;
            if constexpr ((FA_RP16_ABL & 4) != 0) {   // the fragment stays "used"
                if constexpr (x == 0) asm volatile("" :: "v"(frag[f % kRing]));
                return;
            }
            if constexpr ((f & 1) == 0) {
                constexpr int kbl = (f >> 1) / kKS, ks = (f >> 1) % kKS;
                using MQ = std::conditional_t<kCvtK && kFast, Mx<F16>, M>;
                if constexpr (kAsmQK) {
                    if constexpr (ks != 0) MQ::mfma_v_acc(s_nxt[x][kbl], frag[f % kRing], qf[x][ks]);
                    else if constexpr (kFast) s_nxt[x][kbl] = MQ::mfma_v_init(frag[f % kRing], qf[x][ks], minit);
                    else s_nxt[x][kbl] = MQ::mfma_v_zero(frag[f % kRing], qf[x][ks]);
                } else {
                    s_nxt[x][kbl] = MQ::mfma(frag[f % kRing], qf[x][ks], ks == 0 ? (kFast ? minit : zero4) : s_nxt[x][kbl]);
                }
            } else {
                constexpr int db = f >> 1;
                o[x][db] = M::mfma(frag[f % kRing], pk_prev[x], o[x][db]);
            }
        };

        if constexpr (!kFast && (FA_RP16_ABL & 2) == 0) {
            fma_pair(c0{});
            fma_pair(c1{});
        }
        if constexpr ((FA_RP16_ABL & 2) == 0) exp_pair(c0{});
        [[maybe_unused]] unsigned kb_q[kKS], kb_n[kKS], vb_v = 0u, vb_n = 0u, land_k = 0u, land_v = 0u;
        [[maybe_unused]] __amdgpu_buffer_rsrc_t rk_t = rk, rv_t = rv;
        // (each base is formed behind the matrix instruction in front of its first read, not in a bunch at the top of the step)
        static_assert(!kBases || kAhead <= 2 * kKS, "the fragments read ahead into the next step are of its first key block");
        auto form_base = [&](auto fc) {
            constexpr int f = decltype(fc)::value;   // fragment about to be read; f >= kNF: of the next step
            const unsigned smem_a = lds_addr(smem);
            if constexpr (f >= kNF) {
                constexpr int fp = f - kNF;
                if constexpr ((fp & 1) == 0) { kb_n[fp >> 1] = smem_a + so_nq + k_rd[fp >> 1]; asm volatile("" : "+v"(kb_n[fp >> 1])); }
                if constexpr (fp == 1) { vb_n = smem_a + so_nv + v_rd4[0]; asm volatile("" : "+v"(vb_n)); }
            } else if constexpr ((f & 1) == 0) {
                constexpr int ks = (f >> 1) % kKS;
                if constexpr (f - 2 * kKS < kAhead) {   // no earlier in-step fragment with this ks
                    kb_q[ks] = smem_a + so_q + k_rd[ks];
                    asm volatile("" : "+v"(kb_q[ks]));
                }
            } else if constexpr (f - 2 < kAhead) {
                vb_v = smem_a + so_v + v_rd4[0];
                asm volatile("" : "+v"(vb_v));
            }
        };
        sfor<kSlots>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kFlagBar) {
                if constexpr (h == 0 && i == kFlagRead) flag_seen = lds_read4_at(flag_base + 4u * (lane & (unsigned)(kW - 1)));
                if constexpr (h == 0 && i == kFlagCheck) {   // every wave has landed the tile the next fragment reads touch
                    while (!__all((int)(flag_seen - epoch) >= 0)) {
                        __builtin_amdgcn_s_sleep(1);
                        flag_seen = lds_read4_at(flag_base + 4u * (lane & (unsigned)(kW - 1)));
                    }
                }
                if constexpr (h == 1 && i == kLandLast + 1) {   // this wave's chunks of tile j+2 are in LDS (in order behind them)
                    ++epoch;
                    lds_write4_at(flag_base + 4u * wave, epoch);
                }
            }
            if constexpr (kSpread && (FA_RP16_ABL & 8) == 0) {
                if constexpr (h == 0 && i < 2 * kLoads && decltype(req_c)::value) {   // request chunk i of tile j+2
                    constexpr int p = i >> 1;
                    if constexpr (kBases) {
                        // the tile's offset goes into the DESCRIPTOR (base up, bytes down: scalar instructions, and the bounds still
                        // cut at the end of the head) instead of into eight per-lane offsets
                        if constexpr (i == 0) {
                            const unsigned t_off = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(tile + kLook) * kTile);
                            const unsigned left = (unsigned)__builtin_amdgcn_readfirstlane(t_off < kv_bytes ? kv_bytes - t_off : 0u);
                            auto uniform_ptr = [](const char* q) -> const char* {   // (uniform anyway: spares the descriptor a waterfall loop)
                                const unsigned long long a = (unsigned long long)q;
                                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
                                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
                                return reinterpret_cast<const char*>((unsigned long long)lo | ((unsigned long long)hi << 32));
                            };
                            rk_t = make_rsrc(uniform_ptr(k_head + t_off), left);
                            rv_t = make_rsrc(uniform_ptr(v_head + t_off), left);
                        }
                        if constexpr ((i & 1) == 0) kst[p] = buf_load16(rk_t, st_goff[p]);
                        else vst[p] = buf_load16(rv_t, sv_goff[p]);
                    } else {
                        if constexpr ((i & 1) == 0) kst[p] = buf_load16(rk, (unsigned)(tile + kLook) * kTile + st_goff[p]);
                        else vst[p] = buf_load16(rv, (unsigned)(tile + kLook) * kTile + sv_goff[p]);
                    }
                }
                if constexpr (h == 1 && i >= kLandSlot && i <= kLandLast) {   // land chunk i - kLandSlot
                    constexpr int p = (i - kLandSlot) >> 1;
                    if constexpr (kBases) {
                        // chunk p of a thread lies 64 kW / kChunks rows below chunk 0 in both images (the K swizzle and the V block
                        // map repeat every 16 rows at D = 128): one address each, the rest in the immediate
                        static_assert(!kBases || ((64 * kW) % G::kChunks == 0 && ((64 * kW) / G::kChunks) % 16 == 0), "chunk p = chunk 0 + p rows, a multiple of 16 (both image maps repeat)");
                        constexpr unsigned kStepK = (64u * kW / G::kChunks) * kRowB, kStepV = (64u * kW / G::kChunks / 8u) * (unsigned)kDB * 256u;
                        if constexpr (i == kLandSlot) {
                            land_k = lds_addr(smem) + so_land + k_lds[0];
                            asm volatile("" : "+v"(land_k));
                            land_v = lds_addr(smem) + so_land + v_lds[0];
                            asm volatile("" : "+v"(land_v));
                        }
                        if constexpr (((i - kLandSlot) & 1) == 0) lds_write16_at(land_k + p * kStepK, (kCvtK && kFast) ? k_to_f16(k_land[p]) : k_land[p]);
                        else lds_write16_at(land_v + p * kStepV, v_land[p]);
                    } else {
                        if constexpr (((i - kLandSlot) & 1) == 0) lds_write16(smem, so_land + k_lds[p], (kCvtK && kFast) ? k_to_f16(k_land[p]) : k_land[p]);
                        else lds_write16(smem, so_land + v_lds[p], v_land[p]);
                    }
                }
            } else if constexpr (h == 1 && i == kLandSlot && !kDma && (FA_RP16_ABL & 8) == 0) {   // land tile j+2 (requested at the top of the iteration)
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    lds_write16(smem, so_land + k_lds[p], (kCvtK && kFast) ? k_to_f16(k_land[p]) : k_land[p]);
                    lds_write16(smem, so_land + v_lds[p], v_land[p]);
                }
            }
            issue_mfma(ic);
            if constexpr (FA_RP16_SUMMFMA && (FA_RP16_ABL & 4) == 0) {   // the X row-sum instructions of the step
                if constexpr (FA_RP16_ONES_POS == 0 && i % kNF == kNF - 1) lacc[i / kNF] = M::mfma(ones, pk_prev[i / kNF], lacc[i / kNF]);
                if constexpr (FA_RP16_ONES_POS == 1 && i < X) lacc[i] = M::mfma(ones, pk_prev[i], lacc[i]);
                if constexpr (FA_RP16_ONES_POS == 2 && i >= kSlots - X) lacc[i - (kSlots - X)] = M::mfma(ones, pk_prev[i - (kSlots - X)], lacc[i - (kSlots - X)]);
            }
            if constexpr (i % X == X - 1) {   // the fragment just consumed X times is free: read kAhead ahead
                constexpr int f = i / X + kAhead;
                if constexpr (kBases) {
                    form_base(std::integral_constant<int, f>{});
                    if constexpr (f < kNF) read_frag_b(std::integral_constant<int, f>{}, kb_q, vb_v, ho, ho);
                    else read_frag_b(std::integral_constant<int, f - kNF>{}, kb_n, vb_n, h, h);
                } else {
                    if constexpr (f < kNF) read_frag(std::integral_constant<int, f>{}, so_q, ho, so_v, ho);
                    else read_frag(std::integral_constant<int, f - kNF>{}, so_nq, h, so_nv, h);
                }
            }
            constexpr int kPer = kSlots / kPairs;   // matrix slots per vector pair-step (2 at D = 64, 4 at D = 128)
            if constexpr (kVSplit && (FA_RP16_ABL & 2) == 0) {
                // the pair-step's instructions one by one behind consecutive matrix instructions: a v_exp (or v_cvt_pk) of ~8 issue
                // cycles fits in the shadow of the 16-cycle matrix instruction in front of it, three in a row do not
                constexpr int j = i / kPer, sub = i % kPer;
                if constexpr (sub < 2) {
                    if constexpr (j + 2 < kPairs && !kFast) fma_one(std::integral_constant<int, j + 2>{}, std::integral_constant<int, sub>{});
                    if constexpr (j + 1 < kPairs) exp_one(std::integral_constant<int, j + 1>{}, std::integral_constant<int, sub>{});
                }
                if constexpr (sub == (kPer == 2 ? 1 : 2)) fin_pair(std::integral_constant<int, j>{});
            } else {
                if constexpr (i % kPer == (FA_RP16_VALU_AT ? 0 : kPer - 1)) valu_step(std::integral_constant<int, i / kPer>{});
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FA_RP16_RUNSUM) {
#pragma unroll
            for (int x = 0; x < X; ++x) l_part[x] += ls[x][0] + ls[x][1];
        }
    };

    // a packed weight can only have overflowed if the fp32 row sum reached the 16-bit format's range; bf16 keeps a finite bound with
    // room for sum(p*v) in fp32
    const float lim = T::id == 1 ? 0x1p+96f : 60000.0f;
    constexpr int kCheckEvery = 8;   // exact optimistic pass, fp16: tiles between two looks at the row sums (an overflow ends the pass there)
    // mode 0: folded fast pass; 1: exact, reference max fixed after the first 32 keys; 2: exact, lazy running max (same pipeline)
    // returns true when the folded pass gave up right after its reference was known (nothing computed yet)
    // pre_c: the item's first three K/V tiles are already in flight (requested at the end of the previous item)
    auto run = [&](auto mode_c, auto pre_c) __attribute__((always_inline)) -> bool {
        constexpr int kMode = decltype(mode_c)::value;
        constexpr bool kPre = decltype(pre_c)::value;
        constexpr bool kTrack = kMode == 2, kFast = kMode == 0;
        const std::integral_constant<bool, kFast> fast_c{};
        const std::integral_constant<bool, kTrack> track_c{};
        f32x4 sA[X][2], sB[X][2];
        u32x4 pkA[X], pkB[X];
#pragma unroll
        for (int x = 0; x < X; ++x) {
#pragma unroll
            for (int db = 0; db < kDB; ++db) o[x][db] = zero4;
            l_part[x] = 0.0f;
            ls[x][0] = ls[x][1] = 0.0f;
            lacc[x] = zero4;
            pkB[x] = zero4u;   // "P(-1)" = 0 against the zeroed V of the ring's last slot
        }
        // ---- prologue: tiles 0 and 1 -> slots 0 and 1; V of slot 3 ("tile -1") zeroed ----
        if constexpr (kDma) {
#pragma unroll
            for (int p = 0; p < kLoads; ++p) lds_write16(smem, kRingMask * kSlotBytes + v_lds[p], zero4u);
            dma_tile(rk, rv, 0u, 0u);
            dma_tile(rk, rv, kTile, kSlotBytes);
        } else {   // tiles 0 and 1 -> LDS; tile 2 stays in the staging registers until iteration 0 lands it
            if constexpr (!kPre) kv_issue(rk, rv);
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                lds_write16(smem, kRingMask * kSlotBytes + v_lds[p], zero4u);
                lds_write16(smem, k_lds[p], (kCvtK && kFast) ? k_to_f16(pfk[0][p]) : pfk[0][p]);
                lds_write16(smem, v_lds[p], pfv[0][p]);
                lds_write16(smem, kSlotBytes + k_lds[p], (kCvtK && kFast) ? k_to_f16(pfk[1][p]) : pfk[1][p]);
                lds_write16(smem, kSlotBytes + v_lds[p], pfv[1][p]);
                if constexpr (kPair) {   // tile 2 as well: an iteration starts with the tiles up to two ahead of it in LDS
                    lds_write16(smem, 2u * kSlotBytes + k_lds[p], (kCvtK && kFast) ? k_to_f16(kst[p]) : kst[p]);
                    lds_write16(smem, 2u * kSlotBytes + v_lds[p], vst[p]);
                }
            }
        }
        __syncthreads();
        if constexpr (kMode == (kFold ? 0 : 1)) FA_STAMP(2);
        if constexpr (kAsmQK) {   // Q' may have been finished by vector instructions just above: wait states the compiler would count for a builtin
#pragma unroll
            for (int x = 0; x < X; ++x)
#pragma unroll
                for (int ks = 0; ks < kKS; ++ks) asm volatile("s_nop 4" : "+v"(qf[x][ks]));
        }
#pragma unroll
        for (int kbl = 0; kbl < 2; ++kbl)   // S(unit 0)
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                const u32x4 kf = read_kf(0u, 0, kbl, ks);
                using MQ = std::conditional_t<kCvtK && kFast, Mx<F16>, M>;
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    if constexpr (!kAsmQK) sA[x][kbl] = MQ::mfma(kf, qf[x][ks], ks == 0 ? zero4 : sA[x][kbl]);
                    else if (ks == 0) sA[x][kbl] = MQ::mfma_v_zero(kf, qf[x][ks]);
                    else MQ::mfma_v_acc(sA[x][kbl], kf, qf[x][ks]);
                }
            }
        if constexpr (kAsmQK) {   // the matrix results are read by vector instructions right below: 16 wait states behind each
#pragma unroll
            for (int x = 0; x < X; ++x) asm volatile("s_nop 15" : "+v"(sA[x][0]), "+v"(sA[x][1]));
        }
        {
            // reference max from the first 32 keys (masked copy when N < 32; the step masks again)
            f32x4 s0[X][2];
#pragma unroll
            for (int x = 0; x < X; ++x) { s0[x][0] = sA[x][0]; s0[x][1] = sA[x][1]; }
            if ((partial && ntiles == 1) || (kCausal && jc == 0)) mask_unit(0, 0, s0);
            if constexpr (kFast) {   // one reference for the wave; the folded scores already carry the scale
                float mw = -INFINITY;
#pragma unroll
                for (int x = 0; x < X; ++x) mw = fmaxf(mw, row_max(s0[x]));
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) mw = fmaxf(mw, row_ror(mw, sh));   // over the 16 rows of a lane group (DPP)
                if (kCausal || (partial && ntiles == 1)) {
                    mw += kHeadroomFold;
                } else {
                    // Place the reference so that a typical row sum lands mid-window (2^kFoldAim; the window is
                    // [N 2^-16, 60000) for fp16 weights): the mean weight of these 64 x 32 scores relative to their maximum
                    // predicts the row sum N * mean * 2^(max - reference).  With the maximum + 1 alone, rows of a wave whose
                    // first scores hold an outlier fell below the window once the logits spread a little (sigma ~ 3 log2 units).
                    float e = 0.0f;
#pragma unroll
                    for (int x = 0; x < X; ++x)
#pragma unroll
                        for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
                            for (int i = 0; i < 4; ++i) e += fast_exp2(s0[x][kbl][i] - mw);
                    e = across_sum(e);
#pragma unroll
                    for (int sh = 1; sh < 16; sh <<= 1) e += row_ror(e, sh);
                    // (N through an opaque copy: hoisted out of the item loop, the product would be spilled around the tile
                    // loop and its reload -- s_waitcnt vmcnt(0) -- would sit behind whatever memory traffic is in flight)
                    int n_here = Nkv;
                    asm volatile("" : "+s"(n_here));
                    const float shift = __builtin_amdgcn_logf((float)n_here * e * (1.0f / (16.0f * X * 32.0f))) - kFoldAim;
                    mw += fminf(fmaxf(shift, -kFoldShiftMin), kFoldMax);
                }
#pragma unroll
                for (int x = 0; x < X; ++x) m_ref[x] = mw;
                // the gates that are known now (reference beyond kFoldMax, folded Q out of range) end the pass before it costs
                // anything: one workgroup vote per item
                if (__syncthreads_or(((((FA_RP16_GATES & 4) != 0) && !(fabsf(mw) <= kFoldMax)) || (((FA_RP16_GATES & 8) != 0) && q_bad != 0)) ? 1 : 0)) return true;
#pragma unroll
                for (int i = 0; i < 4; ++i) minit[i] = -mw;
#pragma unroll
                for (int x = 0; x < X; ++x)
#pragma unroll
                    for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
                        for (int i = 0; i < 4; ++i) sA[x][kbl][i] -= mw;   // unit 0 was accumulated from zero
            } else {
#pragma unroll
                for (int x = 0; x < X; ++x) m_ref[x] = row_max(s0[x]) * c + (kTrack ? 0.0f : kHeadroom);
            }
        }
        // the first kAhead fragments of the first step: K(tile 0, half 1), V("tile -1")
        sfor<kAhead>([&](auto fc) { read_frag(fc, 0u, 1, kRingMask * kSlotBytes, 1); });

        // phase_c: j & 3 when the caller knows it at compile time (the unrolled steady state: ring slot offsets become
        // immediates of the LDS instructions instead of one v_add per fragment read), -1 otherwise
        // req_c: request tile j+2 at the top (not in iteration 0 of the non-DMA path: the prologue already has it in flight)
        auto tile_barrier = [&]() __attribute__((always_inline)) {
            if constexpr ((FA_RP16_ABL & 16) != 0 || kFlagBar) {
            } else if constexpr (FA_RP16_RAWBAR != 0 && !kDma && kLdsAfterLand <= 15) {
                // The barrier publishes this wave's ds_writes of the landed tile (first read at least one iteration later) and orders
                // the ring's reuse; it does not need the fragment reads issued since (LDS operations of a wave complete in order: once
                // at most kLdsAfterLand are outstanding, the writes are done).  __syncthreads() would wait for all of them
                // (s_waitcnt lgkmcnt(0)): the latency of the last read, exposed once per tile.
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(kLdsAfterLand) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            } else {
                __syncthreads();
            }
        };
        auto tile_iter = [&](int j, auto masked_c, auto phase_c, auto req_c) __attribute__((always_inline)) {
            constexpr int ph = decltype(phase_c)::value;
            const unsigned jj = ph >= 0 ? (unsigned)ph : (unsigned)j;
            const unsigned so_m1 = ((jj + kRingMask) & kRingMask) * kSlotBytes, so_0 = (jj & kRingMask) * kSlotBytes;
            const unsigned so_p1 = ((jj + 1u) & kRingMask) * kSlotBytes, so_ld = ((jj + (unsigned)kLook) & kRingMask) * kSlotBytes;
            // tile j + kLook: tiles past the end read zeros through the buffer bounds into a free slot
            if constexpr ((FA_RP16_ABL & 8) != 0 || !decltype(req_c)::value || kSpread) {   // (kSpread: inside the first step)
            } else if constexpr (kDma) {
                dma_tile(rk, rv, (unsigned)(j + 2) * kTile, so_ld);   // the barrier below waits for it (vmcnt) and publishes it
            } else {
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    kst[p] = buf_load16(rk, (unsigned)(j + kLook) * kTile + st_goff[p]);
                    vst[p] = buf_load16(rv, (unsigned)(j + kLook) * kTile + sv_goff[p]);
                }
            }
            //   h 0: softmax (j,0);  QK^T (j,1);    PV (j-1,1);  next step: QK^T (j+1,0), PV (j,0)
            //   h 1: softmax (j,1);  QK^T (j+1,0);  PV (j,0);    next step: QK^T (j+1,1), PV (j,1)
            step(c0{}, masked_c, fast_c, track_c, j, sA, sB, pkB, pkA, so_0, so_m1, so_p1, so_0, so_ld, c0{}, req_c);
            step(c1{}, masked_c, fast_c, track_c, j, sB, sA, pkA, pkB, so_p1, so_0, so_p1, so_0, so_ld, c0{}, req_c);
            tile_barrier();
        };
        // kPair: tiles j and j+1 in one iteration: tiles j+3 and j+4 requested at the top and landed in the second step of each tile,
        // one barrier behind both (the iteration starts with the tiles up to j+2 in LDS: the last step reads K of tile j+2)
        auto pair_iter = [&](int j) __attribute__((always_inline)) {
            const unsigned jj = (unsigned)j;
            const unsigned so_m1 = ((jj + kRingMask) & kRingMask) * kSlotBytes, so_0 = (jj & kRingMask) * kSlotBytes;
            const unsigned so_p1 = ((jj + 1u) & kRingMask) * kSlotBytes, so_p2 = ((jj + 2u) & kRingMask) * kSlotBytes;
            const unsigned so_p3 = ((jj + 3u) & kRingMask) * kSlotBytes, so_p4 = ((jj + 4u) & kRingMask) * kSlotBytes;
            if constexpr ((FA_RP16_ABL & 8) == 0) {
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    kst[p] = buf_load16(rk, (unsigned)(j + 3) * kTile + st_goff[p]);
                    vst[p] = buf_load16(rv, (unsigned)(j + 3) * kTile + sv_goff[p]);
                    kst2[p] = buf_load16(rk, (unsigned)(j + 4) * kTile + st_goff[p]);
                    vst2[p] = buf_load16(rv, (unsigned)(j + 4) * kTile + sv_goff[p]);
                }
            }
            step(c0{}, no, fast_c, track_c, j, sA, sB, pkB, pkA, so_0, so_m1, so_p1, so_0, so_p3, c0{}, no);
            step(c1{}, no, fast_c, track_c, j, sB, sA, pkA, pkB, so_p1, so_0, so_p1, so_0, so_p3, c0{}, no);
            step(c0{}, no, fast_c, track_c, j + 1, sA, sB, pkB, pkA, so_p1, so_0, so_p2, so_p1, so_p4, c1{}, no);
            step(c1{}, no, fast_c, track_c, j + 1, sB, sA, pkA, pkB, so_p2, so_p1, so_p2, so_p1, so_p4, c1{}, no);
            tile_barrier();
        };
        if constexpr (kMode == (kFold ? 0 : 1)) FA_STAMP(3);
        using dyn = std::integral_constant<int, -1>;
        // iteration 0 of the one-tile-per-iteration form requests nothing unless staging is by DMA: the prologue has tile 2 in flight
        const std::integral_constant<bool, kDma || kPair> req0{};
        // returns true when the pass was given up on a workgroup vote: fp16 weights of the exact optimistic pass overflowed (the
        // matrix-pipe row sums are complete in every lane, so the look costs a compare per block and a vote every kCheckEvery tiles)
        auto full_tiles = [&](int nfull) __attribute__((always_inline)) -> bool {
            constexpr bool kLook4Overflow = kMode == 1 && T::id == 0 && FA_RP16_SUMMFMA && FA_RP16_ABL == 0;
            auto overflowed = [&]() -> bool {
                bool over = false;
#pragma unroll
                for (int x = 0; x < X; ++x) over = over || !(lacc[x][0] < lim);
                return __syncthreads_or(over ? 1 : 0) != 0;
            };
            int j = 0;
            if constexpr (kPair) {
                for (; j + 1 < nfull; j += 2) {
                    pair_iter(j);
                    if constexpr (kLook4Overflow) { if (((j + 2) % kCheckEvery) == 0 && j + 2 < nfull && overflowed()) return true; }
                }
                if (j < nfull) tile_iter(j, no, dyn{}, yes);
                return false;
            }
            if (nfull > 0) { tile_iter(0, no, dyn{}, req0); j = 1; }
            if constexpr (kLook4Overflow) {
                while (j < nfull) {
                    const int je = min(nfull, j + kCheckEvery);
                    for (; j < je; ++j) tile_iter(j, no, dyn{}, yes);
                    if (j < nfull && overflowed()) return true;
                }
            } else {
                for (; j < nfull; ++j) tile_iter(j, no, dyn{}, yes);
            }
            return false;
        };
        // (a masked iteration 0 requests tile 2 once more: the same data into the same registers)
        if constexpr (kCausal) {
            if (full_tiles(jc)) return true;
            for (int j = jc; j < nt; ++j) tile_iter(j, yes, dyn{}, yes);
        } else {
            if (full_tiles(partial ? ntiles - 1 : ntiles)) return true;
            if (partial) tile_iter(ntiles - 1, yes, dyn{}, yes);
        }
        if constexpr (FA_RP16_RUNSUM && !FA_RP16_SUMMFMA) {
#pragma unroll
            for (int x = 0; x < X; ++x) l_part[x] = ls[x][0] + ls[x][1];
        }
        if constexpr (kMode == (kFold ? 0 : 1)) FA_STAMP(4);
        // ---- epilogue: O^T += V(last tile, half 1)^T.P^T ----
        {
            const unsigned so = ((unsigned)(nt - 1) & kRingMask) * kSlotBytes;
#pragma unroll
            for (int db = 0; db < kDB; ++db) {
                const u32x4 vf = read_vf(so, 1, db);
#pragma unroll
                for (int x = 0; x < X; ++x) o[x][db] = M::mfma(vf, pkB[x], o[x][db]);
            }
            if constexpr (FA_RP16_SUMMFMA) {
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    lacc[x] = M::mfma(ones, pkB[x], lacc[x]);
                }
            }
        }
        return false;
    };

    float l_row[X];
    bool bad = false, second_vote = false, direct = false;   // (second_vote, direct: workgroup-uniform)
    // an optimistic pass' row sum: complete in every lane when it came from the matrix pipe
    auto row_sum = [&](int x) -> float { return FA_RP16_SUMMFMA ? lacc[x][0] : across_sum(l_part[x]); };
    if constexpr (!kPrefetch) q_issue(Qg + bh * head_elems, q_row0 - c16);   // else: requested by the item before (next_in)
    else {
        if constexpr (kScan) { if (!q_pending) q_issue(Qg + bh * head_elems, q_row0 - c16); }   // (first block of a later list batch)
        if constexpr (!kCarryKV) kv_issue(rk, rv);   // tiles 0..2 on their way before Q is waited for
    }
    const std::integral_constant<bool, kPrefetch> pre_c{};
    // qf, pfk/pfv, kst/vst <- the next item's raw Q rows and K/V tiles 0..2 (called between the last pass and the stores)
    auto next_in = [&]() __attribute__((always_inline)) {
        if constexpr (kPrefetch) {
            const unsigned nbid = kScan ? scan_peek() : bid + gridDim.x;
            if constexpr (kScan) q_pending = nbid < nwg;
            if (nbid < nwg) {
                unsigned bh_n, qb_n;
                locate(nbid, bh_n, qb_n);
                bh_n = __builtin_amdgcn_readfirstlane(bh_n);   // (uniform anyway: spares the descriptors a waterfall loop)
                qb_n = __builtin_amdgcn_readfirstlane(qb_n);
                q_issue(Qg + bh_n * head_elems, qb_n * kRows + wave * (16u * X));
                if constexpr (kCarryKV) kv_issue(make_rsrc(Kg + bh_n * head_elems, head_bytes), make_rsrc(Vg + bh_n * head_elems, head_bytes));
            }
        }
    };
    bool redo = false;   // (workgroup-uniform) full-width waves: this block is left to the redo kernel
    [[maybe_unused]] unsigned pass_id = kFold ? 0u : 1u;
    if constexpr (kScan) {
        q_finish(no);
        direct = true;
    } else if constexpr (kFold) {
        k_amax = 0.0f;
        q_finish(yes);
        FA_STAMP(1);
        const bool gave_up = run(std::integral_constant<int, 0>{}, pre_c);
        // fp16 weights: each subnormal one is off by at most 2^-25, N of them by N * 2^-25 in the worst case (2^-13 * sqrt(N)
        // typically), which stays below 2^-9 of the row sum; bf16 weights only must not vanish in fp32 (a row more than ~100
        // log2 units below its wave's reference: p = 0, l = 0)
        int n_here = Nkv;
        asm volatile("" : "+s"(n_here));   // as above: no spilled constant behind the Q prefetch
        const float lo = T::id == 0 ? (float)n_here * 0x1p-16f : 0x1p-100f;
#pragma unroll
        for (int x = 0; x < X; ++x) {
            l_row[x] = row_sum(x);
            // causal: a row only has row+1 keys to add up
            const float lo_x = (kCausal && T::id == 0) ? (float)min((unsigned)n_here, q_row0 + 16u * x + 1u) * 0x1p-16f : lo;
            // upper gate: 60000 for BOTH input types.  For bf16 the weights themselves would hold far more (lim = 2^96), but a row
            // sum beyond 2^16 means logits more than 16 above a reference that is itself up to kFoldMax in magnitude, and the
            // rounding of Q' (|logit| 2^-11 per logit) then moves the ratio of two comparable dominant weights by up to ~0.5 %
            bad = bad || ((FA_RP16_GATES & 1) && !(l_row[x] < 60000.0f)) || ((FA_RP16_GATES & 2) && !(l_row[x] >= lo_x)) ||
                  ((FA_RP16_GATES & 4) && !(fabsf(m_ref[x]) <= kFoldMax));
        }
        bad = bad || gave_up || ((FA_RP16_GATES & 8) && q_bad != 0) || !(k_amax <= 65504.0f);
        if constexpr ((FA_RP16_ABL & 255) != 0) bad = false;
        // folded pass refused: the exact optimistic pass first (same pipeline, per-row reference, one v_fma per score --
        // it is what large logits need; bf16 weights cannot overflow in it), the tracked pass only if that overflows too
        if (__syncthreads_or(bad ? 1 : 0)) {
            if (T::id == 0 && gave_up) {
                // fp16, refused before anything was computed (reference beyond kFoldMax, Q' out of range): logits this large overflow
                // fixed-reference fp16 weights more often than not (profiles/r02_gate_cliff.txt) -- the running-max pass at once
                direct = true;
                if constexpr (!kSplitTrack) load_q(no);
            } else {
                load_q(no);
                pass_id = 1u;
                direct = run(std::integral_constant<int, 1>{}, no);   // (true: given up on an overflow vote)
                bad = false;
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    l_row[x] = row_sum(x);
                    bad = bad || !(l_row[x] < lim);
                }
                second_vote = !direct;
            }
        }
    } else {
        q_finish(no);
        direct = run(std::integral_constant<int, 1>{}, pre_c);
#pragma unroll
        for (int x = 0; x < X; ++x) {
            l_row[x] = row_sum(x);
            bad = bad || !(l_row[x] < lim);
        }
        second_vote = !direct;
    }
    if (direct || (second_vote && __syncthreads_or(bad ? 1 : 0))) {   // (both flags are uniform: workgroup votes decided them)
        pass_id = kSplitTrack ? 3u : 2u;
        if constexpr (kSplitTrack) {
            redo = true;
        } else {
            if constexpr (kScan) run(std::integral_constant<int, 2>{}, pre_c);   // (its K/V tiles 0..2 are on their way already)
            else run(std::integral_constant<int, 2>{}, no);
#pragma unroll
            for (int x = 0; x < X; ++x) l_row[x] = row_sum(x);
            __syncthreads();   // the next item's prologue writes the ring: every wave is past this pass' last LDS read
        }
    }
#ifdef FA_EXPERIMENTS
    if constexpr (!kScan) {
        unsigned* ids = g_rp16_pass_ids;
        if (ids != nullptr && tid == 0u) ids[bh * (unsigned)nqb + qb] = pass_id;
    }
#endif
    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
    if constexpr (kSplitTrack) {
        if (redo) {
            // the marker goes where the redo kernel's two half-width blocks of this block begin: the first rows of waves 0 and
            // kW/2, written by those waves (their stand-in stores to the same addresses precede it in program order)
            next_in();
            if (lane == 0u && (wave == 0u || wave == (unsigned)(kW / 2))) buf_store4(ro, (q_row0 - c16) * D * es, kMarker);
            continue;
        }
    }

    if constexpr (kKeySplit == 2) {
        // group 1 -> group 0: (O^T unnormalised, l, m) per lane through group 1's ring (both rings are idle: every wave is past
        // its last fragment read once it is past this barrier); weights 2^(m_s - M)
        float* const xch = reinterpret_cast<float*>(smem_all + kRingSlots * kSlotBytes) + tid;
        constexpr unsigned kStride = 64u * kW;
        __syncthreads();
        if (grp == 1u) {
#pragma unroll
            for (int x = 0; x < X; ++x) {
#pragma unroll
                for (int db = 0; db < kDB; ++db)
#pragma unroll
                    for (int i = 0; i < 4; ++i) xch[((x * kDB + db) * 4 + i) * kStride] = o[x][db][i];
                xch[(X * kDB * 4 + 2 * x) * kStride] = l_row[x];
                xch[(X * kDB * 4 + 2 * x + 1) * kStride] = m_ref[x];
            }
        }
        __syncthreads();
        if (grp == 0u) {
#pragma unroll
            for (int x = 0; x < X; ++x) {
                const float l1 = xch[(X * kDB * 4 + 2 * x) * kStride], m1 = xch[(X * kDB * 4 + 2 * x + 1) * kStride];
                const float mm = fmaxf(m_ref[x], m1);
                const float a0 = fast_exp2(m_ref[x] - mm), a1 = fast_exp2(m1 - mm);
#pragma unroll
                for (int db = 0; db < kDB; ++db)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[x][db][i] = o[x][db][i] * a0 + xch[((x * kDB + db) * 4 + i) * kStride] * a1;
                l_row[x] = l_row[x] * a0 + l1 * a1;
            }
        }
        __syncthreads();   // (group 1's next prologue writes the ring group 0 has just read)
        if (grp == 1u) {
            next_in();
            continue;
        }
    }
    // normalise in place FIRST (no temporaries alive when the prefetch takes its registers), then the next item's loads, then
    // the stores straight from the accumulators
#pragma unroll
    for (int x = 0; x < X; ++x) {
        const float inv = 1.0f / l_row[x];
#pragma unroll
        for (int db = 0; db < kDB; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[x][db][i] *= inv;
#ifdef FA_RP16_DIAG
        if (g == 0) { o[x][0][0] = l_row[x]; o[x][0][1] = m_ref[x]; }
#endif
    }
    FA_STAMP(5);
    if constexpr (kPrefetch) {
#pragma unroll
        for (int x = 0; x < X; ++x)
#pragma unroll
            for (int db = 0; db < kDB; ++db) asm volatile("" : "+v"(o[x][db]));   // the multiplies stay in front of the loads
    }
    next_in();
    // o[x][db][i] = O[q_row0 + 16x][16 db + 4 g + i]
    constexpr bool kOLds = FA_RP16_OLDS != 0 && kOutF32 && D == 64 && !kPair && !kScan && kKeySplit == 1 && (FA_RP16_ABL & 32) == 0;
    if constexpr (kOLds) {
        // One 16-row block at a time through this wave's 4 KB of LDS behind the ring: written as the accumulators hold it
        // (lane = row c16, 16-B chunk 4 db + g, chunk index XORed with row & 7: conflict-free both ways), read back with 16
        // lanes per 256-B row and stored four whole rows per instruction.
        char* const stg = smem_all + kRingSlots * kSlotBytes + wave * (16u * 256u);
        const unsigned wr = c16 * 256u, wx = c16 & 7u;
#pragma unroll
        for (int x = 0; x < X; ++x) {
#pragma unroll
            for (int db = 0; db < kDB; ++db)
                lds_write16(stg, wr + (((4u * db + g) ^ wx) << 4), __builtin_bit_cast(u32x4, o[x][db]));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned R = (lane >> 4) + 4u * j, cr = lane & 15u;
                const u32x4 v = lds_read16(stg, R * 256u + ((cr ^ (R & 7u)) << 4));
                buf_store16(ro, ((q_row0 - c16 + 16u * x + R) * D + cr * 4u) * 4u, v);
            }
        }
    } else
#pragma unroll
    for (int x = 0; x < X; ++x) {
        const unsigned row = q_row0 + 16u * x;
#pragma unroll
        for (int db = 0; db < kDB; ++db) {
            const unsigned col = 16u * db + 4u * g;
            if constexpr (kOutF32) {
                if constexpr ((FA_RP16_ABL & 32) != 0) asm volatile("" :: "v"(o[x][db]));
                else buf_store16(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, o[x][db]));
            } else {
                buf_store8(ro, (row * D + col) * 2u, u32x2{T::pack2(o[x][db][0], o[x][db][1]), T::pack2(o[x][db][2], o[x][db][3])});
            }
        }
    }
#ifdef FA_RP16_STAMPS
    FA_STAMP(6);
    if constexpr (kOutF32) {
        if (tid == 0) {
            float* orow = reinterpret_cast<float*>(Og) + ((size_t)bh * N + (size_t)qb * kRows) * D;
            orow[0] = (float)(ts[0] & 0xFFFFFFull);
            for (int i = 1; i < 7; ++i) orow[i] = (float)(long long)(ts[i] - ts[0]);
            orow[7] = (float)blockIdx.x;
        }
    }
#endif
#undef FA_STAMP
    }   // persistent loop over work items
}

template <typename T, int D, int X, bool kOutF32, bool kFold, bool kDma = false, bool kCausal = false, int kWv = 8, int kKeySplit = 1>
static hipError_t launch_rp16(const void* Q, const void* K, const void* V, void* O,
                              int BH, int N, float scale, hipStream_t stream)
{
    using namespace rp16;
    constexpr int kW = kWv;
    constexpr int lds_bytes = kKeySplit * ((pair_tiles(D, X, kDma) && kKeySplit == 1 && kWv == 8) ? 8 : 4) * 2 * kBlockN * D * 2;   // ring(s) of four (eight) [K tile][V tile] slots
    if (kKeySplit > 1 && N % (kBlockN * kKeySplit) != 0) return hipErrorInvalidValue;
    constexpr int lds_extra = ((FA_RP16_OLDS != 0 && kOutF32 && D == 64 && !pair_tiles(D, X, kDma) && kKeySplit == 1) ? kWv * 16 * 256 : 0)   // the output staging region
                              + 64;   // the waves' landing flags (kFlagBar)
    constexpr int kRows = 16 * X * kW;
    const int nqb = (N + kRows - 1) / kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const long long cap = device_cus();
    const unsigned grid = nwg > cap ? (unsigned)cap : (unsigned)nwg;
    auto kern = fa_fwd_rp16_kernel<T, D, X, kOutF32, kFold, kDma, kCausal, false, kWv, kKeySplit>;
    const hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_bytes + lds_extra);
    if (attr != hipSuccess) return attr;
    FA_LAUNCH(kern, dim3(grid), dim3(64 * kW * kKeySplit), lds_bytes + lds_extra, stream,
              static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K), static_cast<const uint16_t*>(V), O, N, nqb,
              scale * kLog2e, (unsigned)nwg);
    if (launch_status() != hipSuccess) return launch_status();
    if constexpr (!kDma && 16 * X * (D / 64) >= 64) {
        // full-width waves: the redo kernel for the row blocks whose optimistic passes failed (see kScan): half-width waves,
        // running-max pass only; with nothing marked it ends after one look at the marker words
        constexpr int kW2 = 8, X2 = X * kW / (2 * kW2), kRows2 = 16 * X2 * kW2;
        constexpr int lds2 = (pair_tiles(D, X2, false) ? 8 : 4) * 2 * kBlockN * D * 2 + 4 * (1 + 64 * kW2);   // (half this kernel's row block, on eight waves)
        const int nqb2 = (N + kRows2 - 1) / kRows2;
        const long long nwg2 = (long long)BH * nqb2;
        if (nwg2 > 0x7FFFFFFFll) return hipErrorInvalidValue;
        const unsigned grid2 = nwg2 > cap ? (unsigned)cap : (unsigned)nwg2;
        auto kern2 = fa_fwd_rp16_kernel<T, D, X2, kOutF32, false, false, kCausal, true>;
        const hipError_t attr2 = ensure_dyn_lds(reinterpret_cast<const void*>(kern2), lds2);
        if (attr2 != hipSuccess) return attr2;
        static_assert(2 * kRows2 == kRows, "the redo kernel's row block is half of this kernel's");
        FA_LAUNCH(kern2, dim3(grid2), dim3(64 * kW2), lds2, stream,
                  static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K), static_cast<const uint16_t*>(V), O, N, nqb2,
                  scale * kLog2e, (unsigned)nwg2);
    }
    return launch_status();
}

// One (D, X, staging, mask) family of the pipeline: its (input type, output type, folded-first) instantiations.  The families
// live in translation units of their own (fa_fwd_rp16_{d64,d64n,d128,c}.hip) so that they compile side by side.
template <int D, int X, bool kDma, bool kCausal, int kWv = 8, int kKeySplit = 1>
static hipError_t rp16_family(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale,
                              int in_dtype, int out_dtype, bool fold, hipStream_t stream)
{
#define RP16_L(T, OUT, FOLD) launch_rp16<T, D, X, OUT, FOLD, kDma, kCausal, kWv, kKeySplit>(Q, K, V, O, BH, N, scale, stream)
    if (in_dtype == 0) {
        if (fold) return out_dtype == 0 ? RP16_L(F16, true, true) : RP16_L(F16, false, true);
        return out_dtype == 0 ? RP16_L(F16, true, false) : RP16_L(F16, false, false);
    }
    if constexpr (!kDma) {   // (the LDS-DMA path cannot convert bf16 K on the way: exact passes only)
        if (fold) return out_dtype == 0 ? RP16_L(BF16, true, true) : RP16_L(BF16, false, true);
    }
    return out_dtype == 0 ? RP16_L(BF16, true, false) : RP16_L(BF16, false, false);
#undef RP16_L
}

}  // namespace fa
