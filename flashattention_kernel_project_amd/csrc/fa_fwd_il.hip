// fa_fwd_il.hip -- interleaved (software-pipelined) tiled attention forward for gfx950.
//
// Same data layout, MFMA orientation and LDS images as fa_fwd_kernels.hip (see the header there);
// what changes is the instruction stream of a wave.  Measured on MI355X (tools/microbench/
// valu_rate.hip): the softmax VALU slice that belongs to one 32x32x16 MFMA at d=64 (2 fma, 2 exp,
// 1 cvt_pk, 2 add, 1 max3) costs ~44 issue cycles for a lone wave and ~32 per wave when both
// waves of a SIMD issue VALU together, and an MFMA placed between such slices adds only ~7 cycles.
// So the stream that keeps both pipes busy is NOT "matrix phase, then vector phase" but one MFMA
// followed by its slice of independent vector work, in every wave, all the time.
//
// That needs every MFMA of an iteration to be independent of that iteration's vector work:
//
//   iteration t:   MFMA:  S(t+1) = K(t+1).Q^T          and   O^T += V(t-1)^T.P(t-1)^T
//                  VALU:  P(t) = 2^(c*S(t) - m), pack, row sums;   row max of S(t+1) at the end
//
// i.e. QK^T runs one tile ahead of the softmax and PV one tile behind it (two score sets and two
// packed-P sets, named statically and swapped by unrolling the loop twice).  The iteration is
// written as a sequence of slots { MFMA c ; LDS operand reads for MFMA c+4 ; VALU slice c } with a
// scheduling fence between slots, so program order IS issue order.  The lazy running-max update
// (rare, wave-uniform branch at the end of the iteration) rescales O, l and the packed P(t) that
// has not been multiplied into O yet.
//
// The vector pipe is the bottleneck at d=64 (rocprofv3: VALU-active 64-72 % of SIMD time, MFMA 41 %), so
// vector instructions are traded for matrix ones where possible: the row sums l = sum_k P come out
// of four extra MFMAs per tile against an all-ones A fragment (no LDS traffic; every lane then holds
// the complete row sum of its query, no cross-half exchange), and c*S - m is a packed v_pk_fma_f32.
//
// K(t+2) and V(t) are fetched HBM/L2 -> registers at the top of iteration t and written to LDS at
// its end (2-deep rings for K and V, one barrier per iteration): the role of the reference's
// loader warp + cp.async ping-pong (flashattn_streaming_16x16_mw_v10.cu:156-195,
// flashattn_forward_wmma_v5_cp_async.cu:221-256).
#include "fa_tile.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#ifndef FA_IL_READ_AHEAD
#define FA_IL_READ_AHEAD 4
#endif
#ifndef FA_IL_SETPRIO
#define FA_IL_SETPRIO 1
#endif
#ifndef FA_IL_NO_MAX
#define FA_IL_NO_MAX 0   // experiment: skip the per-tile row max (UNSAFE: no overflow protection)
#endif
#ifndef FA_IL_SLOT_FENCE
#define FA_IL_SLOT_FENCE 1
#endif
#ifndef FA_IL_STAGE_NUM
#define FA_IL_STAGE_NUM 3
#endif
#ifndef FA_IL_FRONT_STEPS
#define FA_IL_FRONT_STEPS 0   // VALU pair-steps issued right after the barrier, under the first LDS reads' latency
#endif
#ifndef FA_IL_OCC
#define FA_IL_OCC 2   // waves per SIMD the register budget is held to (experiments: 3 forces spills)
#endif
#ifndef FA_IL_MFMA_SUM
#define FA_IL_MFMA_SUM 0   // 0: row sums by v_add on the fp32 p (default: the kernel is POWER-limited and 4 extra MFMAs per tile cost more energy than 34 v_add; -4.9 % wall); 1: by 4 MFMAs against an all-ones fragment
#endif
#ifndef FA_IL_MFMA_ORDER
#define FA_IL_MFMA_ORDER 0
#endif
constexpr int kReadAhead = FA_IL_READ_AHEAD;             // LDS operand reads run this many MFMAs ahead
constexpr int kFragRing  = kReadAhead + 1;

// kDiag: diagnostic build only (never the shipped path): per-wave s_memtime sums of the time spent
// computing, waiting for + writing the staged tiles, and at the barrier -> diag[wg][wave][4].
//
// W = waves per workgroup (32 query rows each).  W = 4 with two workgroups per CU is the shipped
// shape: the two waves that share a SIMD then belong to DIFFERENT workgroups.  The SIMD arbitrates
// between its waves by age, not fairly (measured: with W = 8 the older wave of each pair finishes a
// tile in ~1360 cycles, the younger in ~2180, and the older then idles at the workgroup barrier);
// across workgroups nobody waits for the slower wave, so the unfairness costs nothing.
template <typename T, int D, bool kOutF32, int W, bool kDiag = false, int kAblate = 0>
__global__ __launch_bounds__(64 * W, FA_IL_OCC)
void fa_fwd_il_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                      const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                      int N, int nqb, float scale_log2e, unsigned long long* __restrict__ diag = nullptr,
                      unsigned total_wg = 0)
{
    unsigned long long tm_c = 0, tm_w = 0, tm_b = 0, tm_last = 0, tm_entry = 0, tm_loop0 = 0, tm_loop1 = 0;
    if constexpr (kDiag) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm_entry)::"memory");
    auto stamp = [&](unsigned long long& acc) {
        if constexpr (kDiag) {
            unsigned long long now;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            acc += now - tm_last;
            tm_last = now;
        }
    };
    using G = TileGeom<D>;
    constexpr int kLoadsW = (kBlockN * G::kChunks) / (64 * W);   // 16-B staging loads per thread per tile
    constexpr int kRowsWG = 32 * W;                              // query rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0][K1][V0][V1]

    // ---- persistent workgroups: the launch fills the chip once (grid = CUs x workgroups per CU) and
    // each workgroup walks the work items bid, bid + grid, ...  This removes the per-workgroup
    // dispatch gap (8 back-to-back workgroups per CU at B=8,H=16,N=4096 otherwise) and lets the
    // output stores of one item drain under the next item's prologue.  Item order per CU is the
    // order the dispatcher would have used, so the XCD affinity of a head's K/V is unchanged.
    const unsigned nwg = total_wg ? total_wg : gridDim.x;
    for (unsigned bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
    if (bid != blockIdx.x) __syncthreads();   // the previous item's last LDS reads are done
    // ---- item -> (head, query block): items that share K/V sit on one XCD, consecutively ----
    const unsigned xq = nwg >> 3, xr = nwg & 7u, xcd = bid & 7u;
    const unsigned wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const unsigned bh = wgid / (unsigned)nqb;
    const unsigned qb = wgid - bh * (unsigned)nqb;

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned r = lane & 31u, h = lane >> 5;

    const size_t head_elems = (size_t)N * D;
    const unsigned head_bytes = (unsigned)(head_elems * 2);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, head_bytes);

    const unsigned q_row = qb * kRowsWG + wave * 32u + r;

    // ---- Q^T fragments (B operand of S^T = K.Q^T), resident for the whole kernel ---------------
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    u32x4 qf[G::kKSteps];
#pragma unroll
    for (int s = 0; s < G::kKSteps; ++s) {
        u32x4 raw = buf_load16(rq, q_row * G::kRowBytes + (16u * s + 8u * h) * 2u);
#pragma unroll
        for (int w = 0; w < 4; ++w) raw[w] ^= q_flip;
        qf[s] = raw;
    }

    // ---- staging ----------------------------------------------------------------------------
    unsigned g_off[kLoadsW], k_lds[kLoadsW], v_lds[kLoadsW];
#pragma unroll
    for (int p = 0; p < kLoadsW; ++p) {
        const unsigned idx = tid + p * 64u * W;
        const unsigned row = idx / G::kChunks, ch = idx % G::kChunks;
        g_off[p] = row * G::kRowBytes + ch * 16u;
        k_lds[p] = G::k_off(row, ch);
        v_lds[p] = 2u * G::kTileBytes + G::v_off(row, ch);
    }
    // Two staging register sets: the tiles written to LDS in iteration t were requested in
    // iteration t-1 (a loaded L2/HBM round trip is ~1250 cycles: the same order as an iteration).
    struct Stage { u32x4 k[kLoadsW], v[kLoadsW]; };
    Stage stA, stB;
    // Tiles past the end read zeros through the buffer bounds and land in ring slots nobody
    // reads any more: staging is unconditional and the loop body stays branch-free.
    auto load_k = [&](Stage& st, int tile) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) st.k[p] = buf_load16(rk, (unsigned)tile * kBlockN * G::kRowBytes + g_off[p]);
    };
    auto load_v = [&](Stage& st, int tile) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) st.v[p] = buf_load16(rv, (unsigned)tile * kBlockN * G::kRowBytes + g_off[p]);
    };
    auto write_k = [&](const Stage& st, unsigned buf) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) lds_write16(smem, buf * G::kTileBytes + k_lds[p], st.k[p]);
    };
    auto write_v = [&](const Stage& st, unsigned buf) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) lds_write16(smem, buf * G::kTileBytes + v_lds[p], st.v[p]);
    };

    // ---- per-lane LDS read addresses (see fa_fwd_kernels.hip) -----------------------------------
    const unsigned k_rd_row = r * G::kRowBytes;
    const unsigned k_rd_swz = G::k_swz(r);
    const unsigned i16 = lane & 15u, vq = i16 >> 2, vp = i16 & 3u, vg = (lane >> 4) & 1u;
    unsigned v_rd[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        v_rd[par] = 2u * G::kTileBytes + h * G::kDBlocks * 256u + ((vq ^ par) << 6) + vg * 32u + vp * 8u;

    // ---- running state -------------------------------------------------------------------------
    f32x16 o[G::kDBlocks];
#pragma unroll
    for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.0f;
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.0f;
    float m_ref = 0.0f;    // reference max of this lane's query row, log2 units (c*S)
    float l_part = 0.0f;   // FA_IL_MFMA_SUM == 0: this half-wave's share of the row sum (fp32 p, v_add)
    f32x16 o_l = zero16;   // row sums: accumulator of ones(32x16).P^T, every register = l of this lane's row
    const u32x4 ones = {T::kOnes2, T::kOnes2, T::kOnes2, T::kOnes2};

    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    constexpr int nQK = 2 * G::kKSteps;   // MFMAs of S(t+1) = K.Q^T
    constexpr int nPV = 4 * (G::kDBlocks + FA_IL_MFMA_SUM);  // MFMAs of O^T += V^T.P^T (plus the row-sum block, A = ones)

    auto keep_alive = [&](u32x4& v) { asm volatile("" : "+v"(v)); };
    auto mask_tail = [&](int tile, f32x16 (&s)[2]) {   // keys >= N -> -inf (p = 0)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = tile * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                if (key >= N) s[kb][i] = -INFINITY;
            }
    };
    auto rescale = [&](float tmax, u32x4 (&pk)[4]) {   // rare: raise the reference max
        const float mx = fmaxf(tmax, swap_halves(tmax));
        const float m_new = fmaxf(mx, m_ref);
        const float alpha = fast_exp2(m_ref - m_new);
        m_ref = m_new;
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) o_l[i] *= alpha;
        l_part *= alpha;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)   // P(t) is still waiting for its PV: bring it to the new scale too
#pragma unroll
            for (int w = 0; w < 4; ++w) pk[k4][w] = T::pack2(T::lo(pk[k4][w]) * alpha, T::hi(pk[k4][w]) * alpha);
    };

    // One iteration.  s_cur: raw S(t), overwritten by P(t); s_nxt: receives raw S(t+1);
    // pk_prev: packed P(t-1) (consumed by PV); pk_cur: receives packed P(t).
    auto iter = [&](auto track_c, auto has_prev_c, auto has_next_c, auto par_c, int t, bool mask_next,
                    f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], u32x4 (&pk_prev)[4], u32x4 (&pk_cur)[4],
                    Stage& st_land, Stage& st_fetch) {
        constexpr bool kTrack = decltype(track_c)::value;   // false: optimistic pass, no running max
        constexpr bool kHasPrev = decltype(has_prev_c)::value;
        constexpr bool kHasNext = decltype(has_next_c)::value;
        constexpr int nQ = kHasNext ? nQK : 0, nP = kHasPrev ? nPV : 0, nAll = nQ + nP;
        // slot -> MFMA index: QK^T first (order 0) or PV first (order 1); indices < nQ are QK^T MFMAs
        auto mfma_of = [](int slot) constexpr { return FA_IL_MFMA_ORDER == 1 ? (slot < nP ? nQ + slot : slot - nP) : slot; };
        constexpr int kSteps = 16;   // VALU pair-steps (2 scores each)
        constexpr int kStageSlot = nAll > 0 ? (FA_IL_STAGE_NUM * nAll) / 4 : -1;   // slot in front of which the staged tiles are written

        if constexpr (!(kAblate & 8)) {
            load_k(st_fetch, t + 3);   // requested now, landed in LDS during iteration t+1
            load_v(st_fetch, t + 1);
        }

        // ring slot of K(t+1) and of V(t-1): (t+1)&1, a compile-time constant in the unrolled steady loop
        constexpr int kPar = decltype(par_c)::value;
        const unsigned kbuf = kPar >= 0 ? (unsigned)kPar : ((unsigned)(t + 1) & 1u), vbuf = kbuf;
        u32x4 frag[kFragRing];
        auto issue_reads = [&](auto slot_c) {   // LDS operand reads of the MFMA in slot `slot`
            constexpr int slot = decltype(slot_c)::value;
            constexpr int i = slot < nAll ? mfma_of(slot) : nAll;
            if constexpr ((kAblate & 1) != 0) {   // timing ablation: no LDS operand reads
                if constexpr (i < nAll) frag[slot % kFragRing] = qf[i % G::kKSteps];
            } else if constexpr (i < nAll) {
                if constexpr (i < nQ) {
                    constexpr int kb = i / G::kKSteps, ks = i % G::kKSteps;
                    frag[slot % kFragRing] = lds_read16(smem, kbuf * G::kTileBytes + kb * 32u * G::kRowBytes + k_rd_row +
                                                               (((2u * ks + h) ^ k_rd_swz) << 4));
                } else if constexpr ((i - nQ) / 4 < G::kDBlocks) {
                    constexpr int j = i - nQ, db = j / 4, ks = j % 4;
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(
                            smem, vbuf * G::kTileBytes + v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    frag[slot % kFragRing] = vf;
                }
            }
        };
        auto issue_mfma = [&](auto slot_c) {
            constexpr int slot = decltype(slot_c)::value;
            constexpr int i = mfma_of(slot);
            if constexpr ((kAblate & 2) != 0) {   // timing ablation: no MFMA (operands kept alive)
                keep_alive(frag[slot % kFragRing]);
            } else if constexpr ((kAblate & 32) != 0) {   // timing ablation: reads issued, MFMA does not wait for them
                if constexpr (i < nQ) {
                    constexpr int kb = i / G::kKSteps, ks = i % G::kKSteps;
                    s_nxt[kb] = T::mfma32(qf[(ks + 1) % G::kKSteps], qf[ks], ks == 0 ? zero16 : s_nxt[kb]);
                } else if constexpr ((i - nQ) / 4 < G::kDBlocks) {
                    constexpr int j = i - nQ, db = j / 4, ks = j % 4;
                    o[db] = T::mfma32(qf[ks], pk_prev[ks], o[db]);
                } else {
                    o_l = T::mfma32(ones, pk_prev[(i - nQ) % 4], o_l);
                }
                if constexpr (i + 1 == nAll) {
#pragma unroll
                    for (int f = 0; f < kFragRing; ++f) keep_alive(frag[f]);
                }
            } else if constexpr (i < nQ) {
                constexpr int kb = i / G::kKSteps, ks = i % G::kKSteps;
                s_nxt[kb] = T::mfma32(frag[slot % kFragRing], qf[ks], ks == 0 ? zero16 : s_nxt[kb]);
            } else if constexpr ((i - nQ) / 4 < G::kDBlocks) {
                constexpr int j = i - nQ, db = j / 4, ks = j % 4;
                o[db] = T::mfma32(frag[slot % kFragRing], pk_prev[ks], o[db]);
            } else {
                o_l = T::mfma32(ones, pk_prev[(i - nQ) % 4], o_l);   // row sums: no LDS operand
            }
        };

        // VALU pair-steps, skewed so that nothing waits on the instruction before it:
        //   step j:  fma of pair j+2, exp of pair j+1, pack + row-sum of pair j,
        //            and (last six steps) three max chains over S(t+1).
        const float neg_m = -m_ref;
        float mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY, ls0 = 0.0f, ls1 = 0.0f;
#ifndef FA_IL_PKFMA
#define FA_IL_PKFMA 0   // 1: one v_pk_fma_f32 per pair (round 1).  Packed fp32 stalls behind the matrix pipe when it is issued
                        // between MFMAs (slot model: 81 vs 45.5 cycles per slot), so the interleaved stream uses two v_fma_f32
#endif
        auto fma_pair = [&](auto jc) {
            constexpr int e0 = 2 * decltype(jc)::value, e1 = e0 + 1;
            if constexpr (FA_IL_PKFMA) {
                const f32x2 c2 = {c, c}, neg_m2 = {neg_m, neg_m};
                f32x2 x = {s_cur[e0 >> 4][e0 & 15], s_cur[e1 >> 4][e1 & 15]};
                x = __builtin_elementwise_fma(x, c2, neg_m2);
                s_cur[e0 >> 4][e0 & 15] = x[0];
                s_cur[e1 >> 4][e1 & 15] = x[1];
            } else {
                s_cur[e0 >> 4][e0 & 15] = __builtin_fmaf(s_cur[e0 >> 4][e0 & 15], c, neg_m);
                s_cur[e1 >> 4][e1 & 15] = __builtin_fmaf(s_cur[e1 >> 4][e1 & 15], c, neg_m);
            }
        };
        auto exp_pair = [&](auto jc) {
            constexpr int e0 = 2 * decltype(jc)::value, e1 = e0 + 1;
            s_cur[e0 >> 4][e0 & 15] = fast_exp2(s_cur[e0 >> 4][e0 & 15]);
            s_cur[e1 >> 4][e1 & 15] = fast_exp2(s_cur[e1 >> 4][e1 & 15]);
        };
        auto fin_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, e0 = 2 * j, e1 = e0 + 1;
            pk_cur[j >> 2][j & 3] = T::pack2(s_cur[e0 >> 4][e0 & 15], s_cur[e1 >> 4][e1 & 15]);
            if constexpr (!FA_IL_MFMA_SUM) {
                if constexpr (T::kSumRounded) {
                    if constexpr ((j & 1) == 0) ls0 = T::sum2(pk_cur[j >> 2][j & 3], ls0);
                    else ls1 = T::sum2(pk_cur[j >> 2][j & 3], ls1);
                } else {
                    ls0 += s_cur[e0 >> 4][e0 & 15];
                    ls1 += s_cur[e1 >> 4][e1 & 15];
                }
            }
        };
        // The max chains read S(t+1) inside the slot sequence only when every QK^T MFMA has been
        // issued before the first of those steps (steady iterations); otherwise after the slots.
        constexpr bool kMaxInSlots = kTrack && FA_IL_MFMA_ORDER == 0 && kHasNext && kHasPrev && ((kSteps - 6) * nAll / kSteps >= nQ);
        auto max_step = [&](auto kc) {   // 6 scores of S(t+1), three independent chains
            constexpr int e0 = 6 * decltype(kc)::value;
            constexpr int a0 = e0 < 32 ? e0 : 31, a1 = e0 + 1 < 32 ? e0 + 1 : 31, a2 = e0 + 2 < 32 ? e0 + 2 : 31;
            constexpr int a3 = e0 + 3 < 32 ? e0 + 3 : 31, a4 = e0 + 4 < 32 ? e0 + 4 : 31, a5 = e0 + 5 < 32 ? e0 + 5 : 31;
            mx0 = max3(mx0, s_nxt[a0 >> 4][a0 & 15], s_nxt[a1 >> 4][a1 & 15]);
            mx1 = max3(mx1, s_nxt[a2 >> 4][a2 & 15], s_nxt[a3 >> 4][a3 & 15]);
            mx2 = max3(mx2, s_nxt[a4 >> 4][a4 & 15], s_nxt[a5 >> 4][a5 & 15]);
        };
        auto valu_step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr ((kAblate & 4) != 0) {   // timing ablation: no softmax VALU
                pk_cur[j >> 2][j & 3] = __builtin_bit_cast(unsigned, s_cur[(2 * j) >> 4][(2 * j) & 15]);
                return;
            }
            if constexpr (j + 2 < kSteps) fma_pair(std::integral_constant<int, j + 2>{});
            if constexpr (j + 1 < kSteps) exp_pair(std::integral_constant<int, j + 1>{});
            fin_pair(jc);
            if constexpr (kMaxInSlots && j >= kSteps - 6) max_step(std::integral_constant<int, j - (kSteps - 6)>{});
        };

        // ---- the slots -------------------------------------------------------------------------
        static_for<kReadAhead>([&](auto ic) { issue_reads(ic); });
        fma_pair(std::integral_constant<int, 0>{});
        fma_pair(std::integral_constant<int, 1>{});
        exp_pair(std::integral_constant<int, 0>{});
        constexpr int kFront = (nAll > 0 && !kMaxInSlots) ? FA_IL_FRONT_STEPS : 0;
        if constexpr (kFront > 0) {
            // Every LDS read of this iteration is of data that became visible at the barrier just
            // passed, so the first MFMA cannot issue for one LDS round trip: fill it with VALU work.
            __builtin_amdgcn_sched_barrier(0);
            static_for<kFront>([&](auto jc) { valu_step(jc); });
        }
        if constexpr (nAll == 0) {
            static_for<kSteps>([&](auto jc) { valu_step(jc); });
        } else {
            static_for<nAll>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (FA_IL_SLOT_FENCE) __builtin_amdgcn_sched_barrier(0);
                if constexpr (i == kStageSlot && !(kAblate & 8)) {
                    // land the staged tiles mid-iteration: the loads were issued at the top, and the
                    // LDS writes are long complete when the iteration reaches its barrier
                    write_k(st_land, (unsigned)t & 1u);   // K(t+2) -> slot t&1 (held K(t), last read in iteration t-1)
                    write_v(st_land, (unsigned)t & 1u);   // V(t)   -> slot t&1 (held V(t-2), last read in iteration t-1)
                }
                // The SIMD arbitrates between its two waves by age; raising the priority around the
                // short MFMA + LDS-read issue lets the younger wave feed the long-latency units as
                // soon as it gets there instead of queueing behind the older wave's VALU stream.
                if constexpr (FA_IL_SETPRIO) __builtin_amdgcn_s_setprio(1);
                issue_mfma(ic);
                issue_reads(std::integral_constant<int, i + kReadAhead>{});
                if constexpr (FA_IL_SETPRIO) __builtin_amdgcn_s_setprio(0);
                // VALU steps [i*kSteps/nAll, (i+1)*kSteps/nAll)
                constexpr int j0 = kFront + i * (kSteps - kFront) / nAll, j1 = kFront + (i + 1) * (kSteps - kFront) / nAll;
                static_for<j1 - j0>([&](auto dj) { valu_step(std::integral_constant<int, j0 + decltype(dj)::value>{}); });
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FA_IL_MFMA_SUM) l_part += ls0 + ls1;

        if constexpr (kHasNext && !kTrack) {
            if (mask_next) mask_tail(t + 1, s_nxt);   // only ever true in a peeled iteration
        }
        if constexpr (kHasNext && kTrack) {
            if constexpr (!kMaxInSlots) static_for<6>([&](auto kc) { max_step(kc); });
            if (mask_next) {   // only ever true in a peeled iteration
                mask_tail(t + 1, s_nxt);
                mx0 = -INFINITY;
#pragma unroll
                for (int e = 0; e < 32; ++e) mx0 = fmaxf(mx0, s_nxt[e >> 4][e & 15]);
                mx1 = mx2 = mx0;
            }
            const float tmax = max3(mx0, mx1, mx2) * c;
            if (__any(tmax - m_ref > kThr)) rescale(tmax, pk_cur);
        }
        stamp(tm_c);
        if constexpr (!(kAblate & 8) && nAll == 0) {
            write_k(st_land, (unsigned)t & 1u);
            write_v(st_land, (unsigned)t & 1u);
        }
        if constexpr (kDiag) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(tm_w);
        if constexpr (!(kAblate & 16)) __syncthreads();
        stamp(tm_b);
    };

    // Two passes at most.  The optimistic pass fixes the reference max after tile 0 (row max of
    // tile 0 plus kHeadroom) and never looks at a row max again: the 18 v_max3 + compare + branch
    // per tile are pure cost on a vector-issue-bound loop (-4.6 % measured).  It is exact unless a
    // later score exceeds the reference by more than the 16-bit format can hold (p = 2^(x-m) > 65504
    // for fp16, > 3e38 for bf16): then p becomes inf, the row sum -- accumulated by MFMA from the very
    // same packed P -- becomes inf/NaN, and that is tested once per row at the end.  If any row of the
    // workgroup overflowed, the whole workgroup (staging is cooperative) re-runs in the tracking mode
    // with the lazy running max.  Underflow is harmless: the reference is an actual score of the row,
    // so one term has weight 2^-kHeadroom and anything that underflows is < 2^-24 of it.
    constexpr float kHeadroom = 4.0f;
    f32x16 sA[2], sB[2];
    u32x4 pkA[4], pkB[4];
    const std::true_type yes{};
    const std::false_type no{};
    const std::integral_constant<int, -1> rt{};   // ring-slot parity known only at run time
    auto run = [&](auto track_c) {
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db) o[db] = zero16;
        o_l = zero16;
        l_part = 0.0f;
        // ---- prologue: K(0), K(1) into LDS; S(0) and the exact row max of tile 0 ---------------------
        load_k(stA, 0);
        load_k(stB, 1);
        write_k(stA, 0);
        write_k(stB, 1);
        load_k(stA, 2);   // what iteration 0 lands
        load_v(stA, 0);
        __syncthreads();
    #pragma unroll
        for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
            for (int ks = 0; ks < G::kKSteps; ++ks) {
                const u32x4 kf = lds_read16(smem, kb * 32u * G::kRowBytes + k_rd_row + (((2u * ks + h) ^ k_rd_swz) << 4));
                sA[kb] = T::mfma32(kf, qf[ks], ks == 0 ? zero16 : sA[kb]);
            }
        if (ntiles == 1 && partial) mask_tail(0, sA);
        {
            float tmax = -INFINITY;
    #pragma unroll
            for (int e = 0; e < 32; ++e) tmax = fmaxf(tmax, sA[e >> 4][e & 15]);
            tmax *= c;
            m_ref = fmaxf(tmax, swap_halves(tmax));
            if constexpr (!decltype(track_c)::value) m_ref += kHeadroom;
        }
        __syncthreads();   // all waves are done reading K(0) before iteration 0 overwrites its slot

        {
            unsigned long long dummy = 0;
            stamp(dummy);
            tm_loop0 = tm_last;
        }
        if (ntiles == 1) {
            iter(track_c, no, no, rt, 0, false, sA, sB, pkB, pkA, stA, stB);
        } else {
            iter(track_c, no, yes, rt, 0, partial && ntiles == 2, sA, sB, pkB, pkA, stA, stB);   // now: S in sB, P(0) in pkA
            // steady iterations t in [1, t_end): the next tile is full, no masking
            const int t_end = partial ? ntiles - 2 : ntiles - 1;
            int t = 1;
            for (; t + 1 < t_end; t += 2) {
                iter(track_c, yes, yes, std::integral_constant<int, 0>{}, t, false, sB, sA, pkA, pkB, stB, stA);       // t odd
                iter(track_c, yes, yes, std::integral_constant<int, 1>{}, t + 1, false, sA, sB, pkB, pkA, stA, stB);   // t+1 even
            }
            // leftovers in canonical naming (scores in sB, previous P in pkA), copying back each time
            for (; t + 1 < ntiles; ++t) {
                iter(track_c, yes, yes, rt, t, partial && (t + 2 == ntiles), sB, sA, pkA, pkB, stB, stA);
    #pragma unroll
                for (int kb = 0; kb < 2; ++kb) sB[kb] = sA[kb];
    #pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) pkA[k4] = pkB[k4];
                stB = stA;
            }
            iter(track_c, yes, no, rt, ntiles - 1, false, sB, sA, pkA, pkB, stB, stA);   // last tile: P in pkB
    #pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) pkA[k4] = pkB[k4];
        }
        if constexpr (kDiag) tm_loop1 = tm_last;
        // ---- drain: O^T += V(last)^T.P(last)^T  (P in pkA; V(last) landed at the end of the last iteration)
        {
            const unsigned vbuf = (unsigned)(ntiles - 1) & 1u;
    #pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db)
    #pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    u32x4 vf;
    #pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(
                            smem, vbuf * G::kTileBytes + v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    o[db] = T::mfma32(vf, pkA[ks], o[db]);
                }
    #pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if constexpr (FA_IL_MFMA_SUM) o_l = T::mfma32(ones, pkA[ks], o_l);
        }

    };
    run(no);
    {
        // without the MFMA row sum: a packed p can only have overflowed if the fp32 row sum reached the format's range
        const float l_chk = FA_IL_MFMA_SUM ? o_l[0] : l_part + swap_halves(l_part);
        const bool bad = !(__builtin_fabsf(l_chk) < (FA_IL_MFMA_SUM || T::id == 1 ? 0x1p+96f : 60000.0f));
        if (__syncthreads_or(bad ? 1 : 0)) {
            __syncthreads();   // everybody is out of the first pass's LDS reads
            run(yes);
        }
    }

    // ---- normalise and store: lane holds O[q_row][db*32 + 8g + 4h + 0..3] in o[db][4g..4g+3] ---
    const float inv = 1.0f / (FA_IL_MFMA_SUM ? o_l[0] : l_part + swap_halves(l_part));   // every register of o_l holds the full row sum
    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
#pragma unroll
    for (int db = 0; db < G::kDBlocks; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned col = db * 32u + 8u * g + 4u * h;
            const float a = o[db][4 * g] * inv, b = o[db][4 * g + 1] * inv;
            const float cc = o[db][4 * g + 2] * inv, d = o[db][4 * g + 3] * inv;
            if constexpr (kOutF32) {
                const f32x4 v = {a, b, cc, d};
                buf_store16(ro, (q_row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
            } else {
                const u32x2 v = {T::pack2(a, b), T::pack2(cc, d)};
                buf_store8(ro, (q_row * D + col) * 2u, v);
            }
        }
    }
    if constexpr (kDiag) {
        unsigned long long tm_exit;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm_exit)::"memory");
        if (lane == 0 && diag) {
            unsigned long long* dd = diag + ((size_t)bid * W + wave) * 8;
            dd[0] = tm_c;
            dd[1] = tm_w;
            dd[2] = tm_b;
            dd[3] = (unsigned long long)ntiles;
            dd[4] = tm_loop0 - tm_entry;   // prologue
            dd[5] = tm_loop1 - tm_loop0;   // main loop
            dd[6] = tm_exit - tm_loop1;    // drain + normalise + store (stores retired)
            dd[7] = tm_entry;
        }
    }
    }   // persistent loop over work items
}

template <typename T, int D, bool kOutF32, int W>
static hipError_t launch_il(const void* Q, const void* K, const void* V, void* O,
                            int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    auto kern = fa_fwd_il_kernel<T, D, kOutF32, W>;
    const int lds_bytes = G::kLdsBytes;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    const int nqb = (N + 32 * W - 1) / (32 * W);
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    // persistent grid: one resident generation of workgroups (8 waves per CU at <= 256 VGPRs)
    const int grid_cap = device_cus() * (8 / W);
    const unsigned grid = (grid_cap > 0 && nwg > grid_cap) ? (unsigned)grid_cap : (unsigned)nwg;
    FA_LAUNCH(kern, dim3(grid), dim3(64 * W), lds_bytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e,
                       static_cast<unsigned long long*>(nullptr), (unsigned)nwg);
    return launch_status();
}

#ifdef FA_EXPERIMENTS
// Diagnostic launch (fp16, d=64, fp32 out): diag[nwg][W][4].
hipError_t il_diag_dispatch(const void* Q, const void* K, const void* V, void* O,
                            int BH, int N, float scale, unsigned long long* diag, int waves, hipStream_t stream)
{
    using G = TileGeom<64>;
    if (waves >= 200) {   // ablations of the 4-wave kernel at ONE workgroup per CU (one wave per SIMD), no stamps
        const int nqb = (N + 127) / 128;
        const int lds = 96 * 1024;   // more than half of the CU's LDS: occupancy 1
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            FA_LAUNCH(kern, dim3((unsigned)(BH * nqb)), dim3(256), lds, stream,
                               static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                               static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, diag, 0u);
        };
        switch (waves - 200) {
            case 0: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 0>); break;
            case 1: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 1>); break;
            case 2: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 2>); break;
            case 4: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 4>); break;
            case 8: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 8>); break;
            case 9: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 9>); break;
            case 11: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 11>); break;
            case 13: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 13>); break;
            case 14: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 14>); break;
            case 12: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 12>); break;
            case 10: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 10>); break;
            case 31: go(fa_fwd_il_kernel<F16, 64, true, 4, false, 31>); break;
            default: return hipErrorInvalidValue;
        }
        return launch_status();
    }
    if (waves >= 100) {   // the same ablations WITHOUT the in-kernel stamps (time them with events)
        const int nqb = (N + 255) / 256;
        auto go = [&](auto kern) {
            FA_LAUNCH(kern, dim3((unsigned)(BH * nqb)), dim3(512), G::kLdsBytes, stream,
                               static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                               static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, diag, 0u);
        };
        switch (waves - 100) {
            case 0: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 0>); break;
            case 1: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 1>); break;
            case 2: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 2>); break;
            case 4: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 4>); break;
            case 5: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 5>); break;
            case 6: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 6>); break;
            case 7: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 7>); break;
            case 8: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 8>); break;
            case 15: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 15>); break;
            case 11: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 11>); break;
            case 13: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 13>); break;
            case 14: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 14>); break;
            case 9: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 9>); break;
            case 10: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 10>); break;
            case 12: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 12>); break;
            case 31: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 31>); break;
            case 3: go(fa_fwd_il_kernel<F16, 64, true, 8, false, 3>); break;
            default: return hipErrorInvalidValue;
        }
        return launch_status();
    }
    if (waves >= 10) {   // 8-wave workgroups with one piece of the iteration removed (timing only, wrong results)
        const int nqb = (N + 255) / 256;
        auto go = [&](auto kern) {
            FA_LAUNCH(kern, dim3((unsigned)(BH * nqb)), dim3(512), G::kLdsBytes, stream,
                               static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                               static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, diag, 0u);
        };
        switch (waves - 10) {   // bit mask: 1 no LDS operand reads, 2 no MFMA, 4 no softmax VALU, 8 no staging, 16 no barrier
            case 1: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 1>); break;
            case 2: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 2>); break;
            case 4: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 4>); break;
            case 5: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 5>); break;
            case 6: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 6>); break;
            case 7: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 7>); break;
            case 8: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 8>); break;
            case 15: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 15>); break;
            case 31: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 31>); break;
            case 3: go(fa_fwd_il_kernel<F16, 64, true, 8, true, 3>); break;
            default: return hipErrorInvalidValue;
        }
        return launch_status();
    }
    if (waves == 8) {
        const int nqb = (N + 255) / 256;
        FA_LAUNCH((fa_fwd_il_kernel<F16, 64, true, 8, true>), dim3((unsigned)(BH * nqb)), dim3(512), G::kLdsBytes, stream,
                           static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                           static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, diag, 0u);
    } else {
        const int nqb = (N + 127) / 128;
        FA_LAUNCH((fa_fwd_il_kernel<F16, 64, true, 4, true>), dim3((unsigned)(BH * nqb)), dim3(256), G::kLdsBytes, stream,
                           static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                           static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, diag, 0u);
    }
    return launch_status();
}
#endif  // FA_EXPERIMENTS

// waves: 8 = one 256-row workgroup per CU, 4 = two 128-row workgroups per CU
hipError_t il_dispatch(const void* Q, const void* K, const void* V, void* O,
                       int BH, int N, int D, float scale, int in_dtype, int out_dtype, int waves,
                       hipStream_t stream)
{
    if (D != 64) return hipErrorInvalidValue;
    if (waves == 8) {
        if (in_dtype == 0)
            return out_dtype == 0 ? launch_il<F16, 64, true, 8>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_il<F16, 64, false, 8>(Q, K, V, O, BH, N, scale, stream);
        return out_dtype == 0 ? launch_il<BF16, 64, true, 8>(Q, K, V, O, BH, N, scale, stream)
                              : launch_il<BF16, 64, false, 8>(Q, K, V, O, BH, N, scale, stream);
    }
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_il<F16, 64, true, 4>(Q, K, V, O, BH, N, scale, stream)
                              : launch_il<F16, 64, false, 4>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_il<BF16, 64, true, 4>(Q, K, V, O, BH, N, scale, stream)
                          : launch_il<BF16, 64, false, 4>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
