// fa_fwd_rp.hip -- attention forward, d = 64 / 128: 64 (32) query rows per wave, 512-row (256-row) workgroups, a
// rolling half-tile pipeline with single-instruction fp32 vector work and a branch-free steady state.
//
// The unit of work is a HALF tile (32 keys x the wave's query rows); step u issues
//     MFMA:  S(u+1) = K(u+1).Q^T  (8)      O^T += V(u-1)^T.P(u-1)^T  (8)
//     VALU:  P(u) = 2^(c*S(u) - m), row sums, pack                       (16 X scores per lane)
// as 16 slots { MFMA ; LDS fragment read two fragments ahead ; one slice of the vector work }, fenced, so that
// inside ONE wave a matrix instruction is always followed by the vector work that fits under it.
// What round 2 measured about this shape (tools/microbench/slot_model.hip, profiles/r02_slot_model.txt; cycles
// per 32x32x16 MFMA slot = two scores per lane at d = 64, two waves per SIMD):
//     vector work of a slot beside its MFMA                        cycles
//     1 v_pk_fma_f32 + 2 v_exp + 1 v_pk_add_f32 + 1 v_cvt_pk        81    <- round 1 (packed fp32 stalls behind the matrix pipe)
//     2 v_fma_f32 + 2 v_exp + 2 v_add_f32 + 1 v_cvt_pk              45.5  <- here
//     the same on two v_mfma_f32_16x16x32 (8 of 16 issue cycles held) 55.7
//     1 slot : 1 slice 37.7 | 2 : 2 42.6 | 4 : 4 47 | 16 : 16 (phase-ordered) 47   <- interleave granularity
// and about the compiler: any branch inside the loop body (round 1 tested `last tile?` in every step to mask the
// keys past N) lets machine-sinking move a whole step's vector work below the branch, behind its MFMAs -- the
// fences order a scheduling region, not the CFG -- which silently turned every other step back into a
// matrix phase followed by a vector phase.  Here the steady-state loop has no branch: full tiles run unmasked, a
// ragged last tile runs through a second, masked copy of the two steps after the loop.
//
// Reference analogue: the warp-specialised kernels' hand-off of MMA, softmax and load roles per tile
// (flashattn_warp_spc/flashattn_streaming_16x16_mw_v10.cu:188-269, _v11.cu:189-258); with 64-lane waves and one
// matrix pipe per SIMD the roles are slots of one instruction stream instead of warps.
// Overflow safety, row sums, output, persistent XCD-aware grid: as fa_fwd_w64.hip (optimistic pass against a fixed
// reference max + exact detection + tracked re-run).
#include "fa_tile.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

namespace rp {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
constexpr int kW = 8;              // waves per workgroup
constexpr int kSlots = 4;          // LDS ring: tiles j-1 .. j+2
constexpr int kAhead = 2;          // fragments read ahead of their MFMAs
constexpr int kRing = 4;           // fragment registers (8 fragments per step: the phase repeats)
constexpr float kFoldMax = 24.0f;  // folded pass only while the reference maximum (log2 units) stays below this: the rounding of
                                   // Q*scale to fp16 moves a logit by at most |logit| * 2^-11
#ifndef FA_RP_SETPRIO
#define FA_RP_SETPRIO 0
#endif
#ifndef FA_RP_DOT2
#define FA_RP_DOT2 T::kSumRounded             // row sums of the ROUNDED p by v_dot2c: bf16 only (fa_common.hpp; +1.8 % wall when forced on for fp16)
#endif
#ifndef FA_RP_ABL
#define FA_RP_ABL 0   // timing ablations, wrong results (A/B builds only): 1 no LDS fragment reads in the loop, 8 no K/V staging, 16 no barrier
#endif
#ifndef FA_RP_SUMMFMA
#define FA_RP_SUMMFMA 1          // 1: row sums of the optimistic passes on the matrix pipe: one v_mfma_f32_16x16x32 per packed-P fragment against a
#endif                           // SELECTOR A operand (row 0 = ones over the k-groups of lanes 0-15 / 32-47, row 1 = ones over those of lanes 16-31 / 48-63):
                                 // 4 accumulator registers and 16 pipe cycles instead of 16 v_add_f32 (or a 32x32x16 against ones: 16 registers, 32 cycles)
#ifndef FA_RP_STAGE_SLOT
#define FA_RP_STAGE_SLOT 8       // MFMA slot of the second step in front of which tile j+2 is written to LDS
#endif
}  // namespace rp

// D = head dim (64 or 128), X = 32-row query blocks per wave (2 at D = 64, 1 at D = 128): a step is always
// 16 MFMAs (2*kKSteps fragments, each feeding X MFMAs) beside the softmax of 16*X scores per lane.
// kFold (fp16 only): fast first pass with scale*log2(e) folded into a rounded copy of Q and a WAVE-uniform reference
// maximum carried as the initial value of every score accumulator chain, so that p = 2^s' costs no arithmetic:
// 5 instead of 7 vector instructions per slot (37.7 vs 45.5 cycles in the slot model).  Accepted per workgroup only if
// every row sum stayed inside [N * 2^-14, 60000) (no fp16 overflow; the subnormal weights of a row add up to less than
// 2^-11 of it), the folded Q stayed in fp16's normal range and the reference is below kFoldMax; else the exact tracked
// pass re-runs the workgroup.
template <typename T, int D, int X, bool kOutF32, bool kFold = false>
__global__ __launch_bounds__(64 * rp::kW, 2)
void fa_fwd_rp_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                        const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                        int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace rp;
    using G = TileGeom<D>;
    constexpr int kRows = 32 * X * kW;                          // query rows per workgroup
    constexpr int kLoads = (kBlockN * G::kChunks) / (64 * kW);  // 16-B K (and V) chunks per thread and tile
    constexpr int kFrags = 2 * G::kKSteps;                      // fragments per step: K and V^T alternate
    static_assert(kFrags * X == 16 && kFrags % kRing == 0, "a step is 16 MFMAs; the fragment ring phase must repeat");
    constexpr unsigned kSlotBytes = G::kBufBytes;   // [K tile][V tile]
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned r = lane & 31u, h = lane >> 5;
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    // staging: kLoads 16-B chunks of K and of V per thread and tile
    unsigned st_goff[kLoads], k_lds[kLoads], v_lds[kLoads];
#pragma unroll
    for (int p = 0; p < kLoads; ++p) {
        const unsigned idx = tid + p * 64u * kW;
        const unsigned srow = idx / G::kChunks, sch = idx % G::kChunks;
        st_goff[p] = srow * G::kRowBytes + sch * 16u;
        k_lds[p] = G::k_off(srow, sch);
        v_lds[p] = G::kTileBytes + G::v_off(srow, sch);
    }

    const unsigned k_rd_row = r * G::kRowBytes;
    const unsigned k_rd_swz = G::k_swz(r);
    const unsigned i16 = lane & 15u, vq = i16 >> 2, vp = i16 & 3u, vg = (lane >> 4) & 1u;
    unsigned v_rd[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        v_rd[par] = G::kTileBytes + h * G::kDBlocks * 256u + ((vq ^ par) << 6) + vg * 32u + vp * 8u;

    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.0f;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    constexpr float kHeadroom = 4.0f;
    const std::true_type yes{};
    const std::false_type no{};
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;

    const unsigned nwg = total_wg;
    for (unsigned bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
    if (bid != blockIdx.x) __syncthreads();
    const unsigned xq = nwg >> 3, xr = nwg & 7u, xcd = bid & 7u;
    const unsigned wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const unsigned bh = wgid / (unsigned)nqb;
    const unsigned qb = wgid - bh * (unsigned)nqb;
    const size_t head_elems = (size_t)N * D;
    const unsigned head_bytes = (unsigned)(head_elems * 2);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, head_bytes);
    const unsigned q_row0 = qb * kRows + wave * (32u * X) + r;   // query block 0; block x is 32x rows further

    u32x4 qf[X][G::kKSteps];
    int q_bad = 0;   // folded Q left fp16's normal range
    auto load_q = [&](auto fold_c) __attribute__((always_inline)) {
        constexpr bool fold = decltype(fold_c)::value;
#pragma unroll
        for (int x = 0; x < X; ++x) {
            float amax = 0.0f;
#pragma unroll
            for (int s = 0; s < G::kKSteps; ++s) {
                u32x4 raw = buf_load16(rq, (q_row0 + 32u * x) * G::kRowBytes + (16u * s + 8u * h) * 2u);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if constexpr (fold) {
                        const float lo = T::lo(raw[w]) * scale_log2e, hi = T::hi(raw[w]) * scale_log2e;
                        amax = max3(amax, fabsf(lo), fabsf(hi));
                        raw[w] = T::pack2(lo, hi);
                    } else {
                        raw[w] ^= q_flip;
                    }
                }
                qf[x][s] = raw;
            }
            // overflow (or NaN), or a row whose elements ALL fell below fp16's normal range (branch-free, per lane)
            if constexpr (fold) q_bad |= (int)!(amax <= 65504.0f) | ((int)(amax != 0.0f) & (int)(amax < 6.2e-5f));
        }
        // pin the flag HERE: left to itself the compiler evaluates it after the tile loop and keeps all 64 fp32
        // products alive (spilled) across it -- 33 MB of scratch written and read back per item
        if constexpr (fold) asm volatile("" : "+v"(q_bad));
    };

    f32x16 o[X][G::kDBlocks];
    float m_ref[X] = {}, l_part[X] = {};
    u32x4 kst[kLoads], vst[kLoads];
    u32x4 frag[kRing];
    // Row sums on the matrix pipe.  The packed P of a 32x32 score block (query lane & 31 on the lane, 8 keys in the 4 registers,
    // the other 8 of the k-step in lane + 32) read as the B operand of a 16x16x32: column = lane & 15, k-group = lane >> 4.
    // A = selector: row 0 adds k-groups 0 and 2 (queries 0..15), row 1 adds k-groups 1 and 3 (queries 16..31).  The result's
    // rows 0 and 1 sit in lanes 0..15, registers 0 and 1: lsum[x][q >> 4] of lane q & 15 = row sum of query q of block x.
    f32x4 lsum[X];
    u32x4 sel;
    {
        const bool on = ((lane & 15u) == 0u && ((lane >> 4) & 1u) == 0u) || ((lane & 15u) == 1u && ((lane >> 4) & 1u) == 1u);
        const unsigned w = on ? T::kOnes2 : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sel[i] = w; asm volatile("" : "+v"(sel[i])); }
    }
    auto sum_mfma = [&](f32x4 acc, u32x4 pk) -> f32x4 {
        if constexpr (T::id == 0) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sel), __builtin_bit_cast(f16x8, pk), acc, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sel), __builtin_bit_cast(bf16x8, pk), acc, 0, 0, 0);
    };
    auto lsum_row = [&](int x) -> float {   // this lane's query: lsum[x][r >> 4] of lane r & 15
        const float a = __shfl(lsum[x][0], (int)(r & 15u), 64), b = __shfl(lsum[x][1], (int)(r & 15u), 64);
        return (r & 16u) ? b : a;
    };

    // LDS fragment addresses.  A K fragment of unit (tile slot offset `so`, key block kb), k-step ks;
    // a V^T fragment of unit (so, kb), head-dim block db, 16-key step ks2.
    auto read_kf = [&](unsigned so, int kb, int ks) -> u32x4 {
        return lds_read16(smem, so + kb * 32u * G::kRowBytes + k_rd_row + (((2u * ks + h) ^ k_rd_swz) << 4));
    };
    auto read_vf = [&](unsigned so, int kb, int db, int ks2) -> u32x4 {
        u32x4 vf;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const u32x2 half = lds_read_tr8(smem, so + v_rd[db & 1] + ((4u * (2 * kb + ks2) + 2u * jj) * G::kDBlocks + db) * 256u);
            vf[2 * jj] = half[0];
            vf[2 * jj + 1] = half[1];
        }
        return vf;
    };
    // fragment f (0..kFrags-1) of a step: even f -> K fragment ks = f/2 of the QK^T unit, odd f -> V^T
    // fragment (db = f/4, ks2 = (f/2)&1) of the PV unit
    auto read_frag = [&](auto fc, unsigned so_q, int kb_q, unsigned so_v, int kb_v) {
        constexpr int f = decltype(fc)::value;
        if constexpr (FA_RP_ABL & 1) return;
        if constexpr ((f & 1) == 0) frag[f % kRing] = read_kf(so_q, kb_q, f >> 1);
        else frag[f % kRing] = read_vf(so_v, kb_v, f >> 2, (f >> 1) & 1);
    };
    auto mask_unit = [&](int tile, int kb, f32x16 (&s)[X]) {   // keys >= N -> -inf (p = 0)
#pragma unroll
        for (int x = 0; x < X; ++x)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = tile * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                if (key >= N) s[x][i] = -INFINITY;
            }
    };
    auto row_max = [&](const f32x16& s) -> float {   // over this lane's 16 keys, both half-waves
        float a = max3(s[0], s[1], s[2]), b = max3(s[3], s[4], s[5]);
        a = max3(a, s[6], s[7]);
        b = max3(b, s[8], s[9]);
        a = max3(a, s[10], s[11]);
        b = max3(b, s[12], s[13]);
        a = max3(a, s[14], s[15]);
        a = fmaxf(a, b) * c;
        return fmaxf(a, swap_halves(a));
    };

    // One step of the optimistic pass.  kb = key block of the unit being softmaxed (s_cur = its raw
    // scores -> pk_cur); the QK^T unit is (so_q, 1-kb) -> s_nxt, the PV unit (so_v, 1-kb) <- pk_prev.
    // `so_nq/so_nv`: slot offsets of the NEXT step's QK^T / PV units, for the fragments read ahead.
    f32x16 minit;   // folded pass: every score chain starts at -(wave reference maximum)
    auto step = [&](auto kb_c, auto masked_c, auto fast_c, int tile, f32x16 (&s_cur)[X], f32x16 (&s_nxt)[X], u32x4 (&pk_prev)[X][2],
                    u32x4 (&pk_cur)[X][2], unsigned so_q, unsigned so_v, unsigned so_nq, unsigned so_nv,
                    unsigned so_land) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_c)::value, ko = 1 - kb;
        constexpr bool kFast = decltype(fast_c)::value;
        if constexpr (decltype(masked_c)::value) mask_unit(tile, kb, s_cur);

        constexpr int kSteps = 8 * X;   // VALU pair-steps: pairs 8x .. 8x+7 belong to query block x
        float ls[X][2];
#pragma unroll
        for (int x = 0; x < X; ++x) ls[x][0] = ls[x][1] = 0.0f;
        auto fma_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 3, e = 2 * (j & 7);
            s_cur[x][e] = __builtin_fmaf(s_cur[x][e], c, -m_ref[x]);
            s_cur[x][e + 1] = __builtin_fmaf(s_cur[x][e + 1], c, -m_ref[x]);
        };
        auto exp_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 3, e = 2 * (j & 7);
            s_cur[x][e] = fast_exp2(s_cur[x][e]);
            s_cur[x][e + 1] = fast_exp2(s_cur[x][e + 1]);
        };
        auto fin_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 3, e = 2 * (j & 7);
            const unsigned w = T::pack2(s_cur[x][e], s_cur[x][e + 1]);
            pk_cur[x][(j & 7) >> 2][j & 3] = w;
            if constexpr (FA_RP_SUMMFMA) {
            } else if constexpr (FA_RP_DOT2) {
                ls[x][j & 1] = T::sum2(w, ls[x][j & 1]);
            } else {
                ls[x][0] += s_cur[x][e];
                ls[x][1] += s_cur[x][e + 1];
            }
        };
        auto valu_step = [&](auto jc) {   // skewed: nothing waits on the instruction before it
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 2 < kSteps && !kFast) fma_pair(std::integral_constant<int, j + 2>{});
            if constexpr (j + 1 < kSteps) exp_pair(std::integral_constant<int, j + 1>{});
            fin_pair(jc);
        };
        auto issue_mfma = [&](auto ic) {
            constexpr int i = decltype(ic)::value, f = i / X, x = i % X;
            if constexpr ((f & 1) == 0) {
                constexpr int ks = f >> 1;
                s_nxt[x] = T::mfma32(frag[f % kRing], qf[x][ks], ks == 0 ? (kFast ? minit : zero16) : s_nxt[x]);
            } else {
                constexpr int db = f >> 2, ks2 = (f >> 1) & 1;
                o[x][db] = T::mfma32(frag[f % kRing], pk_prev[x][ks2], o[x][db]);
            }
        };

        if constexpr (!kFast) {
            fma_pair(c0{});
            fma_pair(c1{});
        }
        exp_pair(c0{});
        sfor<16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kb == 1 && i == FA_RP_STAGE_SLOT && !(FA_RP_ABL & 8)) {   // land tile j+2 (requested at the top of the iteration)
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    lds_write16(smem, so_land + k_lds[p], kst[p]);
                    lds_write16(smem, so_land + v_lds[p], vst[p]);
                }
            }
            if constexpr (FA_RP_SETPRIO) __builtin_amdgcn_s_setprio(1);
            issue_mfma(ic);
            if constexpr (FA_RP_SUMMFMA) {   // the 2 X row-sum instructions of the step (pk_prev[x][ks2]), spread over it
                constexpr int kEvery = 16 / (2 * X);
                if constexpr (i % kEvery == kEvery - 1) {
                    constexpr int n = i / kEvery, x = n >> 1, ks2 = n & 1;
                    lsum[x] = sum_mfma(lsum[x], pk_prev[x][ks2]);
                }
            }
            if constexpr (i % X == X - 1) {   // the fragment just consumed X times is free: read kAhead ahead
                constexpr int f = i / X + kAhead;
                if constexpr (f < kFrags) read_frag(std::integral_constant<int, f>{}, so_q, ko, so_v, ko);
                else read_frag(std::integral_constant<int, f - kFrags>{}, so_nq, kb, so_nv, kb);
            }
            if constexpr (FA_RP_SETPRIO) __builtin_amdgcn_s_setprio(0);
            // VALU pair-steps [i*kSteps/16, (i+1)*kSteps/16): one per slot at X = 2, one per two slots at X = 1
            constexpr int j0 = i * kSteps / 16, j1 = (i + 1) * kSteps / 16;
            sfor<j1 - j0>([&](auto dj) { valu_step(std::integral_constant<int, j0 + decltype(dj)::value>{}); });
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FA_RP_SUMMFMA) {
#pragma unroll
            for (int x = 0; x < X; ++x) l_part[x] += ls[x][0] + ls[x][1];
        }
    };

    // One step of the tracked (fallback) pass: same data flow in plain program order, with the lazy
    // running max per unit.  Speed is irrelevant here.
    auto step_tracked = [&](auto kb_c, int tile, f32x16 (&s_cur)[X], f32x16 (&s_nxt)[X], u32x4 (&pk_prev)[X][2],
                            u32x4 (&pk_cur)[X][2], unsigned so_q, unsigned so_v, unsigned so_land) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_c)::value, ko = 1 - kb;
        // O^T += V(u-1)^T.P(u-1)^T first: P(u-1) is in the scale of the current reference max
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const u32x4 vf = read_vf(so_v, ko, db, ks2);
#pragma unroll
                for (int x = 0; x < X; ++x) o[x][db] = T::mfma32(vf, pk_prev[x][ks2], o[x][db]);
            }
        if (partial && tile + 1 == ntiles) mask_unit(tile, kb, s_cur);
#pragma unroll
        for (int x = 0; x < X; ++x) {
            const float tmax = row_max(s_cur[x]);
            if (__any(tmax - m_ref[x] > kThr)) {
                const float m_new = fmaxf(tmax, m_ref[x]);
                const float alpha = fast_exp2(m_ref[x] - m_new);
                m_ref[x] = m_new;
#pragma unroll
                for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[x][db][i] *= alpha;
                l_part[x] *= alpha;
            }
            float ls0 = 0.0f, ls1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float p0 = fast_exp2(fmaf(s_cur[x][e], c, -m_ref[x]));
                const float p1 = fast_exp2(fmaf(s_cur[x][e + 1], c, -m_ref[x]));
                const unsigned w = T::pack2(p0, p1);
                pk_cur[x][e >> 3][(e >> 1) & 3] = w;
                if constexpr (FA_RP_DOT2) {
                    ls0 = T::sum2(w, ls0);
                } else {
                    ls0 += p0;
                    ls1 += p1;
                }
            }
            l_part[x] += ls0 + ls1;
        }
#pragma unroll
        for (int ks = 0; ks < G::kKSteps; ++ks) {
            const u32x4 kf = read_kf(so_q, ko, ks);
#pragma unroll
            for (int x = 0; x < X; ++x) s_nxt[x] = T::mfma32(kf, qf[x][ks], ks == 0 ? zero16 : s_nxt[x]);
        }
        if constexpr (kb == 1) {
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                lds_write16(smem, so_land + k_lds[p], kst[p]);
                lds_write16(smem, so_land + v_lds[p], vst[p]);
            }
        }
    };

    // mode 0: folded fast pass; 1: exact, reference max fixed after the first 32 keys; 2: exact, lazy running max
    auto run = [&](auto mode_c) __attribute__((always_inline)) {
        constexpr int kMode = decltype(mode_c)::value;
        constexpr bool kTrack = kMode == 2, kFast = kMode == 0;
        const std::integral_constant<bool, kFast> fast_c{};
        f32x16 sA[X], sB[X];
        u32x4 pkA[X][2], pkB[X][2];
#pragma unroll
        for (int x = 0; x < X; ++x) {
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db) o[x][db] = zero16;
            l_part[x] = 0.0f;
            lsum[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            pkB[x][0] = pkB[x][1] = zero4;   // "P(-1)" = 0 against the zeroed V of ring slot 3
        }
        // ---- prologue: tiles 0 and 1 -> slots 0 and 1; V of slot 3 ("tile -1") zeroed ------------
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                kst[p] = buf_load16(rk, (unsigned)pt * G::kTileBytes + st_goff[p]);
                vst[p] = buf_load16(rv, (unsigned)pt * G::kTileBytes + st_goff[p]);
            }
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                if (pt == 0) lds_write16(smem, 3u * kSlotBytes + v_lds[p], zero4);
                lds_write16(smem, (unsigned)pt * kSlotBytes + k_lds[p], kst[p]);
                lds_write16(smem, (unsigned)pt * kSlotBytes + v_lds[p], vst[p]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < G::kKSteps; ++ks) {   // S(unit 0)
            const u32x4 kf = read_kf(0u, 0, ks);
#pragma unroll
            for (int x = 0; x < X; ++x) sA[x] = T::mfma32(kf, qf[x][ks], ks == 0 ? zero16 : sA[x]);
        }
        {
            // reference max from the first 32 keys (masked copy when N < 32; the step masks again)
            f32x16 s0[X];
#pragma unroll
            for (int x = 0; x < X; ++x) s0[x] = sA[x];
            if (partial && ntiles == 1) mask_unit(0, 0, s0);
#pragma unroll
            for (int x = 0; x < X; ++x) m_ref[x] = row_max(s0[x]) + (kTrack ? 0.0f : kHeadroom);
            if constexpr (kFast) {
                // row_max() multiplied by c, but the folded scores already carry it: undo; then one reference for the wave
                float mw = -INFINITY;
#pragma unroll
                for (int x = 0; x < X; ++x) mw = fmaxf(mw, (m_ref[x] - kHeadroom) / c);
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) mw = fmaxf(mw, __shfl_xor(mw, off, 64));
                mw += kHeadroom;
#pragma unroll
                for (int x = 0; x < X; ++x) m_ref[x] = mw;
#pragma unroll
                for (int i = 0; i < 16; ++i) minit[i] = -mw;
#pragma unroll
                for (int x = 0; x < X; ++x)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sA[x][i] -= mw;   // unit 0 was accumulated from zero
            }
        }
        if constexpr (!kTrack) {   // fragments 0,1 of the first step: K(tile 0, kb 1), V("tile -1")
            read_frag(c0{}, 0u, 1, 3u * kSlotBytes, 1);
            read_frag(c1{}, 0u, 1, 3u * kSlotBytes, 1);
        }

        // one tile = two steps; `masked_c`: the ragged last tile (keys >= N -> -inf), outside the steady-state loop
        auto tile_iter = [&](int j, auto masked_c) __attribute__((always_inline)) {
            const unsigned so_m1 = ((unsigned)(j + 3) & 3u) * kSlotBytes, so_0 = ((unsigned)j & 3u) * kSlotBytes;
            const unsigned so_p1 = ((unsigned)(j + 1) & 3u) * kSlotBytes, so_p2 = ((unsigned)(j + 2) & 3u) * kSlotBytes;
            // tile j+2: tiles past the end read zeros through the buffer bounds into a free slot
            if constexpr (!(FA_RP_ABL & 8)) {
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    kst[p] = buf_load16(rk, (unsigned)(j + 2) * G::kTileBytes + st_goff[p]);
                    vst[p] = buf_load16(rv, (unsigned)(j + 2) * G::kTileBytes + st_goff[p]);
                }
            }
            if constexpr (kTrack) {
                step_tracked(c0{}, j, sA, sB, pkB, pkA, so_0, so_m1, so_p2);
                step_tracked(c1{}, j, sB, sA, pkA, pkB, so_p1, so_0, so_p2);
            } else {
                //   kb 0: softmax (j,0);  QK^T (j,1);    PV (j-1,1);  next step: QK^T (j+1,0), PV (j,0)
                //   kb 1: softmax (j,1);  QK^T (j+1,0);  PV (j,0);    next step: QK^T (j+1,1), PV (j,1)
                step(c0{}, masked_c, fast_c, j, sA, sB, pkB, pkA, so_0, so_m1, so_p1, so_0, so_p2);
                step(c1{}, masked_c, fast_c, j, sB, sA, pkA, pkB, so_p1, so_0, so_p1, so_0, so_p2);
            }
            if constexpr (!(FA_RP_ABL & 16)) __syncthreads();
        };
        if constexpr (kTrack) {
            for (int j = 0; j < ntiles; ++j) tile_iter(j, no);
        } else {
            const int nfull = partial ? ntiles - 1 : ntiles;
            for (int j = 0; j < nfull; ++j) tile_iter(j, no);
            if (partial) tile_iter(ntiles - 1, yes);
        }
        // ---- epilogue: O^T += V(last tile, kb 1)^T.P^T ---------------------------------------------
        {
            const unsigned so = ((unsigned)(ntiles - 1) & 3u) * kSlotBytes;
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    const u32x4 vf = read_vf(so, 1, db, ks2);
#pragma unroll
                    for (int x = 0; x < X; ++x) o[x][db] = T::mfma32(vf, pkB[x][ks2], o[x][db]);
                }
            if constexpr (FA_RP_SUMMFMA && !kTrack) {
#pragma unroll
                for (int x = 0; x < X; ++x)
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2) lsum[x] = sum_mfma(lsum[x], pkB[x][ks2]);
            }
        }
    };

    float l_row[X];
    // a packed p can only have overflowed if the fp32 row sum reached the 16-bit format's range; bf16 keeps a finite
    // bound with room for sum(p*v) in fp32 (2^96 * |V| * N stays finite)
    const float lim = T::id == 1 ? 0x1p+96f : 60000.0f;
    bool bad = false;
    if constexpr (kFold) {
        load_q(yes);
        run(std::integral_constant<int, 0>{});
        const float lo = (float)N * 0x1p-14f;
#pragma unroll
        for (int x = 0; x < X; ++x) {
            l_row[x] = FA_RP_SUMMFMA ? lsum_row(x) : l_part[x] + swap_halves(l_part[x]);
            bad = bad || !(l_row[x] < lim) || !(l_row[x] >= lo) || !(fabsf(m_ref[x]) <= kFoldMax);
        }
        bad = bad || q_bad != 0;
    } else {
        load_q(no);
        run(std::integral_constant<int, 1>{});
#pragma unroll
        for (int x = 0; x < X; ++x) {
            l_row[x] = FA_RP_SUMMFMA ? lsum_row(x) : l_part[x] + swap_halves(l_part[x]);
            bad = bad || !(l_row[x] < lim);
        }
    }
    if constexpr (FA_RP_ABL != 0) bad = false;   // timing builds: never the second pass
    if (__syncthreads_or(bad ? 1 : 0)) {
        if constexpr (kFold) load_q(no);
        run(std::integral_constant<int, 2>{});
#pragma unroll
        for (int x = 0; x < X; ++x) l_row[x] = l_part[x] + swap_halves(l_part[x]);
    }

    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
#pragma unroll
    for (int x = 0; x < X; ++x) {
        const float inv = 1.0f / l_row[x];
        const unsigned row = q_row0 + 32u * x;
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned col = db * 32u + 8u * g + 4u * h;
                const float a = o[x][db][4 * g] * inv, b = o[x][db][4 * g + 1] * inv;
                const float cc = o[x][db][4 * g + 2] * inv, d = o[x][db][4 * g + 3] * inv;
                if constexpr (kOutF32) {
                    const f32x4 v = {a, b, cc, d};
                    buf_store16(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
                } else {
                    const u32x2 v = {T::pack2(a, b), T::pack2(cc, d)};
                    buf_store8(ro, (row * D + col) * 2u, v);
                }
            }
    }
    }   // persistent loop over work items
}

template <typename T, int D, int X, bool kOutF32, bool kFold = false>
static hipError_t launch_rp(const void* Q, const void* K, const void* V, void* O,
                              int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    constexpr int kRows = 32 * X * rp::kW;
    const int nqb = (N + kRows - 1) / kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int grid_cap = device_cus();
    const unsigned grid = nwg > grid_cap ? (unsigned)grid_cap : (unsigned)nwg;
    const hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(&fa_fwd_rp_kernel<T, D, X, kOutF32, kFold>), rp::kSlots * G::kBufBytes);
    if (attr != hipSuccess) return attr;
    FA_LAUNCH((fa_fwd_rp_kernel<T, D, X, kOutF32, kFold>), dim3(grid), dim3(64 * rp::kW),
                       rp::kSlots * G::kBufBytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, (unsigned)nwg);
    return launch_status();
}

// fold: 1 = folded fast pass where it exists (fp16, d = 64), 0 = exact passes only
hipError_t rp_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype, int fold,
                         hipStream_t stream)
{
    if (!(scale == scale) || scale * kLog2e == 0.0f) fold = 0;   // NaN / zero scale: the exact pass defines the result
    if (D != 64 && D != 128) return hipErrorInvalidValue;
    if ((unsigned long long)(N + 64 * rp::kW + 3 * kBlockN) * (unsigned)D * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (D == 64) {
        if (in_dtype == 0 && fold)
            return out_dtype == 0 ? launch_rp<F16, 64, 2, true, true>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_rp<F16, 64, 2, false, true>(Q, K, V, O, BH, N, scale, stream);
        if (in_dtype == 0)
            return out_dtype == 0 ? launch_rp<F16, 64, 2, true>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_rp<F16, 64, 2, false>(Q, K, V, O, BH, N, scale, stream);
        return out_dtype == 0 ? launch_rp<BF16, 64, 2, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_rp<BF16, 64, 2, false>(Q, K, V, O, BH, N, scale, stream);
    }
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_rp<F16, 128, 1, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_rp<F16, 128, 1, false>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_rp<BF16, 128, 1, true>(Q, K, V, O, BH, N, scale, stream)
                          : launch_rp<BF16, 128, 1, false>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
