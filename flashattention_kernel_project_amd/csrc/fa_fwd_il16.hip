// fa_fwd_il16.hip -- interleaved attention forward with 16 query rows per wave (gfx950, d = 64).
//
// Same algorithm and pipeline as fa_fwd_il.hip (QK^T one tile ahead of the softmax, PV one tile
// behind it, optimistic pass + tracked fallback, row sums on the matrix pipe, persistent grid);
// what changes is the granularity.  Measured on MI355X (tools/microbench/valu_rate.hip): the cost of
// one { MFMA ; LDS operand reads ; VALU slice } slot grows linearly with the waves sharing a SIMD,
// cost = a + b*N, and the fixed part `a` (in-order issue latency a wave cannot hide from itself) is
// as large as b: a SIMD with four waves retires slots 1.3x faster than with two.  The 32-row kernel
// needs ~200 VGPRs and tops out at two waves per SIMD; with 16 rows per wave and
// v_mfma_f32_16x16x32 every per-wave array halves (~115 VGPRs), so sixteen waves -- four per SIMD --
// share one 256-row workgroup.  The price is LDS traffic: every wave still reads the whole K and V
// tile, now for 16 rows instead of 32 (256 KB per tile per CU, ~60 % of the LDS read rate).
//
// Lane roles (lane = 16*g + c): the accumulator of S^T = K.Q^T has the query c on the lane and keys
// 16*kb + 4*g + i in register i of block kb; four lanes (g = 0..3) share a query row.  Packed to 16
// bit, registers of key blocks 2s and 2s+1 are the B fragment of k-step s of O^T += V^T.P^T with
// k-slot 8g+j <-> key 32s + 16(j>>2) + 4g + (j&3); the transposed V reads fetch exactly those keys.
// V image: 256-B blocks [key/8][d/16] x [8 keys][16 cols] (a half-wave's transposed read covers one
// block: all 64 banks once; the once-per-tile ds_write_b128 of a key row is 4-way conflicted, accepted
// to keep one read address register).
#include "fa_tile.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

namespace il16 {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct F16x {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
struct BF16x {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <typename T> struct Mx;
template <> struct Mx<F16> : F16x {};
template <> struct Mx<BF16> : BF16x {};

constexpr int kW = 16;          // waves per workgroup
constexpr int kRows = 16 * kW;  // 256 query rows per workgroup
constexpr int kAhead = 2, kRing = kAhead + 1;
}  // namespace il16

template <typename T, bool kOutF32>
__global__ __launch_bounds__(64 * il16::kW, 4)
void fa_fwd_il16_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                        const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                        int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace il16;
    using M = Mx<T>;
    constexpr int D = 64;
    constexpr unsigned kTile = kBlockN * D * 2;   // 8 KB: one K or V tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0][K1][V0][V1]

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned c16 = lane & 15u, g = lane >> 4;
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    // staging role: waves 0-7 move K chunks, waves 8-15 move V chunks (one 16-B chunk per thread and tile)
    const bool v_role = wave >= 8;
    const unsigned sidx = tid & 511u, srow = sidx >> 3, sch = sidx & 7u;
    const unsigned st_goff = srow * 128u + sch * 16u;
    const unsigned st_lds = v_role
        ? 2u * kTile + ((srow >> 3) * 4u + (sch >> 1)) * 256u + ((srow & 7u) << 5) + ((sch & 1u) << 4)
        : srow * 128u + ((sch ^ ((srow >> 1) & 7u)) << 4);

    // K reads (A operand of QK^T): lane (c16,g) reads row 16*kb + c16, chunk 4*ks + g
    const unsigned k_swz = (c16 >> 1) & 7u;
    unsigned k_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) k_rd[ks] = c16 * 128u + (((4u * ks + g) ^ k_swz) << 4);
    // V^T reads (A operand of PV): 16-lane group g, lane 4*q4+p supplies key row q4, columns 4p..4p+3
    const unsigned q4 = c16 >> 2, p4 = c16 & 3u;
    const unsigned v_rd = 2u * kTile + (g >> 1) * 1024u + ((4u * (g & 1u) + q4) << 5) + p4 * 8u;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {T::kOnes2, T::kOnes2, T::kOnes2, T::kOnes2};
    constexpr float kHeadroom = 4.0f;
    const std::true_type yes{};
    const std::false_type no{};

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
    const __amdgpu_buffer_rsrc_t rkv = make_rsrc((v_role ? Vg : Kg) + bh * head_elems, head_bytes);
    const unsigned q_row = qb * kRows + wave * 16u + c16;

    u32x4 qf[2];   // B operand of QK^T: Q[q_row][32*ks + 8*g .. +7]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u32x4 raw = buf_load16(rq, q_row * 128u + (32u * ks + 8u * g) * 2u);
#pragma unroll
        for (int w = 0; w < 4; ++w) raw[w] ^= q_flip;
        qf[ks] = raw;
    }

    f32x4 o[4], o_l;
    f32x4 sA[4], sB[4];
    u32x4 pk[2];
    u32x4 st;
    float m_ref = 0.0f;

    // tile index per role: K waves run `kt`, V waves run `vt`
    auto stage_load = [&](u32x4& dst, int kt, int vt) {
        dst = buf_load16(rkv, (unsigned)(v_role ? vt : kt) * kTile + st_goff);
    };
    auto stage_write = [&](const u32x4& src, unsigned buf) { lds_write16(smem, buf * kTile + st_lds, src); };

    auto mask_tail = [&](int tile, f32x4 (&s)[4]) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tile * kBlockN + 16 * kb + 4 * (int)g + i >= N) s[kb][i] = -INFINITY;
    };
    auto row_max = [&](const f32x4 (&s)[4]) -> float {   // over this lane's 16 keys
        float a = max3(s[0][0], s[0][1], s[0][2]), b = max3(s[1][0], s[1][1], s[1][2]);
        float d = max3(s[2][0], s[2][1], s[2][2]), e = max3(s[3][0], s[3][1], s[3][2]);
        return fmaxf(max3(a, b, s[0][3]), max3(d, e, fmaxf(s[1][3], fmaxf(s[2][3], s[3][3]))));
    };
    auto across_groups = [&](float x) -> float {   // max over the four lanes that share a query row
        x = fmaxf(x, __shfl_xor(x, 16, 64));
        return fmaxf(x, __shfl_xor(x, 32, 64));
    };

    auto iter = [&](auto track_c, auto has_prev_c, auto has_next_c, int t, bool mask_next,
                    f32x4 (&s_cur)[4], f32x4 (&s_nxt)[4]) {
        constexpr bool kTrack = decltype(track_c)::value;
        constexpr bool kHasPrev = decltype(has_prev_c)::value, kHasNext = decltype(has_next_c)::value;
        constexpr int nP = kHasPrev ? 10 : 0, nQ = kHasNext ? 8 : 0, nAll = nP + nQ;   // PV (8) + row sums (2) first, then QK^T (8)

        stage_load(st, t + 2, t);   // landed in LDS at 3/4 of this iteration
        const unsigned buf = (unsigned)(t + 1) & 1u;   // ring slot of K(t+1) and of V(t-1)

        u32x4 frag[kRing];
        auto issue_reads = [&](auto slot_c) {
            constexpr int i = decltype(slot_c)::value;
            if constexpr (i < nP) {
                if constexpr (i < 8) {   // PV MFMA i: d-block db = i/2, k-step s = i%2
                    constexpr int db = i / 2, s = i % 2;
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(smem, buf * kTile + v_rd + (4u * s + 2u * jj) * 1024u + db * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    frag[i % kRing] = vf;
                }
            } else if constexpr (i < nAll) {   // QK^T MFMA: key block kb = j/2, k-step ks = j%2
                constexpr int j = i - nP, kb = j / 2, ks = j % 2;
                frag[i % kRing] = lds_read16(smem, buf * kTile + kb * 2048u + k_rd[ks]);
            }
        };
        auto issue_mfma = [&](auto slot_c) {
            constexpr int i = decltype(slot_c)::value;
            if constexpr (i < nP) {
                if constexpr (i < 8) o[i / 2] = M::mfma(frag[i % kRing], pk[i % 2], o[i / 2]);
                else o_l = M::mfma(ones, pk[i - 8], o_l);
            } else {
                constexpr int j = i - nP, kb = j / 2, ks = j % 2;
                s_nxt[kb] = M::mfma(frag[i % kRing], qf[ks], ks == 0 ? zero4 : s_nxt[kb]);
            }
        };

        // VALU pair-steps on S(t): 8 pairs.  fma + exp in place while PV (which still reads the previous
        // packed P) runs; pack into the single P register set while QK^T runs.
        const float neg_m = -m_ref;
        const f32x2 c2 = {c, c}, neg_m2 = {neg_m, neg_m};
        auto fma_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, kb = j >> 1, i0 = 2 * (j & 1);
            f32x2 x = {s_cur[kb][i0], s_cur[kb][i0 + 1]};
            x = __builtin_elementwise_fma(x, c2, neg_m2);
            s_cur[kb][i0] = x[0];
            s_cur[kb][i0 + 1] = x[1];
        };
        auto exp_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, kb = j >> 1, i0 = 2 * (j & 1);
            s_cur[kb][i0] = fast_exp2(s_cur[kb][i0]);
            s_cur[kb][i0 + 1] = fast_exp2(s_cur[kb][i0 + 1]);
        };
        auto cvt_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, kb = j >> 1, i0 = 2 * (j & 1);
            pk[j >> 2][j & 3] = T::pack2(s_cur[kb][i0], s_cur[kb][i0 + 1]);
        };
        // 24 VALU micro-steps in dependency order: fma(0), then {fma(j+1), exp(j)}, then cvt(0..7)
        auto micro = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m < 8) {
                if constexpr (m == 0) fma_pair(std::integral_constant<int, 0>{});
                if constexpr (m + 1 < 8) fma_pair(std::integral_constant<int, m + 1>{});
                exp_pair(mc);
            } else {
                cvt_pair(std::integral_constant<int, m - 8>{});
            }
        };
        constexpr int nExpSlots = kHasPrev ? nP : nAll;            // slots that may carry fma/exp steps
        constexpr int nCvtSlots = kHasPrev ? nQ : 0;               // slots that may carry cvt steps (after PV)

        il16::sfor<kAhead>([&](auto ic) { issue_reads(ic); });
        il16::sfor<nAll>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i == (3 * nAll) / 4) stage_write(st, (unsigned)t & 1u);   // K(t+2) / V(t) -> slot t&1
            issue_mfma(ic);
            issue_reads(std::integral_constant<int, i + kAhead>{});
            if constexpr (i < nExpSlots) {
                constexpr int m0 = i * 8 / nExpSlots, m1 = (i + 1) * 8 / nExpSlots;
                il16::sfor<m1 - m0>([&](auto dm) { micro(std::integral_constant<int, m0 + decltype(dm)::value>{}); });
            } else if constexpr (nCvtSlots > 0) {
                constexpr int k0 = (i - nExpSlots) * 8 / nCvtSlots, k1 = (i - nExpSlots + 1) * 8 / nCvtSlots;
                il16::sfor<k1 - k0>([&](auto dm) { micro(std::integral_constant<int, 8 + k0 + decltype(dm)::value>{}); });
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (nAll == 0) {
            il16::sfor<8>([&](auto mc) { micro(mc); });
            stage_write(st, (unsigned)t & 1u);
        }
        if constexpr (nCvtSlots == 0) il16::sfor<8>([&](auto mc) { micro(std::integral_constant<int, 8 + decltype(mc)::value>{}); });

        if constexpr (kHasNext) {
            if (mask_next) mask_tail(t + 1, s_nxt);
            if constexpr (kTrack) {
                const float tmax = row_max(s_nxt) * c;
                if (__any(tmax - m_ref > kThr)) {   // rare: raise the reference max; P(t) is still pending
                    const float m_new = fmaxf(across_groups(tmax), m_ref);
                    const float alpha = fast_exp2(m_ref - m_new);
                    m_ref = m_new;
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[db][i] *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o_l[i] *= alpha;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int w = 0; w < 4; ++w) pk[s2][w] = T::pack2(T::lo(pk[s2][w]) * alpha, T::hi(pk[s2][w]) * alpha);
                }
            }
        }
        __syncthreads();
    };

    auto run = [&](auto track_c) {
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = zero4;
        o_l = zero4;
        // prologue: K(0), K(1) (K waves) into LDS; first staged pair K(2) / V(0); S(0); reference max
        stage_load(st, 0, 0);
        if (!v_role) stage_write(st, 0);
        stage_load(st, 1, 1);
        if (!v_role) stage_write(st, 1);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                sA[kb] = M::mfma(lds_read16(smem, kb * 2048u + k_rd[ks]), qf[ks], ks == 0 ? zero4 : sA[kb]);
        if (ntiles == 1 && partial) mask_tail(0, sA);
        m_ref = across_groups(row_max(sA) * c) + (decltype(track_c)::value ? 0.0f : kHeadroom);
        __syncthreads();   // all waves are done reading K(0) before iteration 0 overwrites its slot

        if (ntiles == 1) {
            iter(track_c, no, no, 0, false, sA, sB);
        } else {
            iter(track_c, no, yes, 0, partial && ntiles == 2, sA, sB);   // S(1) in sB
            const int t_end = partial ? ntiles - 2 : ntiles - 1;
            int t = 1;
            for (; t + 1 < t_end; t += 2) {
                iter(track_c, yes, yes, t, false, sB, sA);
                iter(track_c, yes, yes, t + 1, false, sA, sB);
            }
            for (; t + 1 < ntiles; ++t) {   // leftovers in canonical naming (scores in sB, landing set stB)
                iter(track_c, yes, yes, t, partial && (t + 2 == ntiles), sB, sA);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) sB[kb] = sA[kb];
            }
            iter(track_c, yes, no, ntiles - 1, false, sB, sA);
        }
        // drain: O^T += V(last)^T.P(last)^T and its row sums
        {
            const unsigned buf = (unsigned)(ntiles - 1) & 1u;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(smem, buf * kTile + v_rd + (4u * s + 2u * jj) * 1024u + db * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    o[db] = M::mfma(vf, pk[s], o[db]);
                }
            o_l = M::mfma(ones, pk[0], o_l);
            o_l = M::mfma(ones, pk[1], o_l);
        }
    };

    run(no);
    {
        const bool bad = !(__builtin_fabsf(o_l[0]) < INFINITY);
        if (__syncthreads_or(bad ? 1 : 0)) {
            __syncthreads();
            run(yes);
        }
    }

    // o[db][i] = O[q_row][16*db + 4*g + i]
    const float inv = 1.0f / o_l[0];
    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const unsigned col = 16u * db + 4u * g;
        if constexpr (kOutF32) {
            const f32x4 v = {o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv};
            buf_store16(ro, (q_row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
        } else {
            const u32x2 v = {T::pack2(o[db][0] * inv, o[db][1] * inv), T::pack2(o[db][2] * inv, o[db][3] * inv)};
            buf_store8(ro, (q_row * D + col) * 2u, v);
        }
    }
    }   // persistent loop over work items
}

template <typename T, bool kOutF32>
static hipError_t launch_il16(const void* Q, const void* K, const void* V, void* O,
                              int BH, int N, float scale, hipStream_t stream)
{
    constexpr int lds_bytes = 4 * kBlockN * 64 * 2;   // 32 KB
    const int nqb = (N + il16::kRows - 1) / il16::kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    static const int grid_cap = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus;
    }();
    const unsigned grid = nwg > grid_cap ? (unsigned)grid_cap : (unsigned)nwg;
    hipLaunchKernelGGL((fa_fwd_il16_kernel<T, kOutF32>), dim3(grid), dim3(64 * il16::kW), lds_bytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, (unsigned)nwg);
    return hipGetLastError();
}

hipError_t il16_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                         hipStream_t stream)
{
    if (D != 64) return hipErrorInvalidValue;
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_il16<F16, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_il16<F16, false>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_il16<BF16, true>(Q, K, V, O, BH, N, scale, stream)
                          : launch_il16<BF16, false>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
