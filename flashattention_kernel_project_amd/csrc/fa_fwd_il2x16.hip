// fa_fwd_il2x16.hip -- interleaved attention forward on v_mfma_f32_16x16x32 (gfx950, d = 64).
//
// Same algorithm and pipeline as fa_fwd_il.hip (QK^T one tile ahead of the softmax, PV one tile
// behind it, optimistic pass + tracked fallback, persistent 256-row workgroups of 8 waves x 32 rows);
// what changes is the MFMA shape.  The kernel is power-limited, so joules per FLOP
// decide the wall time, and on this device the 16x16x32 form sustains 1.88 PF of fp16 matrix work at
// the package power cap against 1.62 PF for 32x32x16 (tools/microbench/mfma_power.hip, random
// operands): the same arithmetic for 14 % less energy.  Each wave therefore treats its 32 query rows
// as two 16-row blocks that share every K and V fragment read from LDS (LDS traffic is unchanged
// against the 32x32x16 kernel; only the number of MFMA instructions doubles).
// MEASURED OUTCOME: no gain in the full kernel -- 0.613 ms vs 0.612 ms (fp16), 0.590 vs 0.580 (bf16)
// at B8 H16 N4096: twice the MFMA issue slots and accumulator traffic eat the 14 %.  Kept selectable
// (FA_ALGO_IL2X16) and parity-tested; not the default.
//
// Lane roles (lane = 16*g + c): for query block qb the accumulator of S^T = K.Q^T has query
// 16*qb + c on the lane and keys 16*kb + 4*g + i in register i of key block kb; four lanes
// (g = 0..3) share a query row.  Packed to 16 bit, the registers of key blocks 2s and 2s+1 are the B
// fragment of k-step s of O^T += V^T.P^T with k-slot 8g+j <-> key 32s + 16(j>>2) + 4g + (j&3); the
// transposed V reads fetch exactly those keys.  V image: 256-B blocks [key/8][d/16] x [8 keys][16
// cols] -- a half-wave's ds_read_b64_tr_b16 covers one block, all 64 banks once.
#include "fa_tile.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

namespace il2 {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
template <typename T> struct Mx;
template <> struct Mx<F16> {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mx<BF16> {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
constexpr int kW = 8;           // waves per workgroup
constexpr int kRows = 32 * kW;  // 256 query rows per workgroup
constexpr int kAheadF = 2;      // LDS fragment reads run this many fragments (= 2x MFMAs) ahead
constexpr int kRing = kAheadF + 1;
}  // namespace il2

template <typename T, bool kOutF32>
__global__ __launch_bounds__(64 * il2::kW, 2)
void fa_fwd_il2x16_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                          const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                          int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace il2;
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

    // staging: thread -> one 16-B chunk of the K tile and one of the V tile
    const unsigned srow = tid >> 3, sch = tid & 7u;
    const unsigned st_goff = srow * 128u + sch * 16u;
    const unsigned k_lds = srow * 128u + ((sch ^ ((srow >> 1) & 7u)) << 4);
    const unsigned v_lds = 2u * kTile + ((srow >> 3) * 4u + (sch >> 1)) * 256u + ((srow & 7u) << 5) + ((sch & 1u) << 4);

    // K reads (A operand of QK^T): lane (c16,g) reads row 16*kb + c16, chunk 4*ks + g
    unsigned k_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) k_rd[ks] = c16 * 128u + (((4u * ks + g) ^ ((c16 >> 1) & 7u)) << 4);
    // V^T reads (A operand of PV): 16-lane group g, lane 4*q4+p supplies key row q4, columns 4p..4p+3
    const unsigned v_rd = 2u * kTile + (g >> 1) * 1024u + ((4u * (g & 1u) + (c16 >> 2)) << 5) + (c16 & 3u) * 8u;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
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
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, head_bytes);
    const unsigned q_row0 = qb * kRows + wave * 32u + c16;   // row of query block 0; block 1 is 16 rows further

    u32x4 qf[2][2];   // B operand of QK^T: Q[row of block x][32*ks + 8*g .. +7]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 raw = buf_load16(rq, (q_row0 + 16u * x) * 128u + (32u * ks + 8u * g) * 2u);
#pragma unroll
            for (int w = 0; w < 4; ++w) raw[w] ^= q_flip;
            qf[x][ks] = raw;
        }

    f32x4 o[2][4];
    f32x4 sA[2][4], sB[2][4];
    u32x4 pk[2][2];
    u32x4 kst, vst;
    float m_ref[2] = {0.0f, 0.0f}, l_part[2] = {0.0f, 0.0f};

    auto load_k = [&](int tile) { kst = buf_load16(rk, (unsigned)tile * kTile + st_goff); };
    auto load_v = [&](int tile) { vst = buf_load16(rv, (unsigned)tile * kTile + st_goff); };
    auto write_k = [&](unsigned buf) { lds_write16(smem, buf * kTile + k_lds, kst); };
    auto write_v = [&](unsigned buf) { lds_write16(smem, buf * kTile + v_lds, vst); };

    auto mask_tail = [&](int tile, f32x4 (&s)[2][4]) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (tile * kBlockN + 16 * kb + 4 * (int)g + i >= N) s[x][kb][i] = -INFINITY;
    };
    auto row_max = [&](const f32x4 (&s)[4]) -> float {   // over this lane's 16 keys of one query block
        float a = max3(s[0][0], s[0][1], s[0][2]), b = max3(s[1][0], s[1][1], s[1][2]);
        float d = max3(s[2][0], s[2][1], s[2][2]), e = max3(s[3][0], s[3][1], s[3][2]);
        return fmaxf(max3(a, b, s[0][3]), max3(d, e, fmaxf(s[1][3], fmaxf(s[2][3], s[3][3]))));
    };
    auto across_max = [&](float v) -> float {   // over the four lanes that share a query row
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        return fmaxf(v, __shfl_xor(v, 32, 64));
    };
    auto across_sum = [&](float v) -> float {
        v += __shfl_xor(v, 16, 64);
        return v + __shfl_xor(v, 32, 64);
    };

    // One iteration.  Fragment f of a phase feeds two MFMAs (query blocks 0 and 1).
    auto iter = [&](auto track_c, auto has_prev_c, auto has_next_c, int t, bool mask_next,
                    f32x4 (&s_cur)[2][4], f32x4 (&s_nxt)[2][4]) __attribute__((always_inline)) {
        constexpr bool kTrack = decltype(track_c)::value;
        constexpr bool kHasPrev = decltype(has_prev_c)::value, kHasNext = decltype(has_next_c)::value;
        constexpr int fP = kHasPrev ? 8 : 0, fQ = kHasNext ? 8 : 0, fAll = fP + fQ;   // fragments: PV first, then QK^T
        constexpr int nAll = 2 * fAll;                                                 // MFMA slots

        load_k(t + 2);   // landed in LDS at 3/4 of this iteration
        load_v(t);
        const unsigned buf = (unsigned)(t + 1) & 1u;   // ring slot of K(t+1) and of V(t-1)

        u32x4 frag[kRing];
        auto read_frag = [&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f < fP) {   // V^T fragment: d-block db = f/2, k-step s = f%2
                constexpr int db = f / 2, s = f % 2;
                u32x4 vf;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const u32x2 half = lds_read_tr8(smem, buf * kTile + v_rd + (4u * s + 2u * jj) * 1024u + db * 256u);
                    vf[2 * jj] = half[0];
                    vf[2 * jj + 1] = half[1];
                }
                frag[f % kRing] = vf;
            } else if constexpr (f < fAll) {   // K fragment: key block kb, k-step ks
                constexpr int j = f - fP, kb = j / 2, ks = j % 2;
                frag[f % kRing] = lds_read16(smem, buf * kTile + kb * 2048u + k_rd[ks]);
            }
        };
        auto issue_mfma = [&](auto slot_c) {
            constexpr int i = decltype(slot_c)::value, f = i / 2, x = i % 2;
            if constexpr (f < fP) {
                constexpr int db = f / 2, s = f % 2;
                o[x][db] = M::mfma(frag[f % kRing], pk[x][s], o[x][db]);
            } else {
                constexpr int j = f - fP, kb = j / 2, ks = j % 2;
                s_nxt[x][kb] = M::mfma(frag[f % kRing], qf[x][ks], ks == 0 ? zero4 : s_nxt[x][kb]);
            }
        };

        // VALU pair-steps on S(t): 16 pairs (query block x = j/8, pair jj = j%8: key block jj/2, regs 2(jj&1)..).
        // fma + exp in place while PV (still reading the previous packed P) runs; pack + row sums into the
        // single P register set while QK^T runs.
        const f32x2 c2 = {c, c};
        auto fma_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 3, kb = (j & 7) >> 1, i0 = 2 * (j & 1);
            const f32x2 nm = {-m_ref[x], -m_ref[x]};
            f32x2 v = {s_cur[x][kb][i0], s_cur[x][kb][i0 + 1]};
            v = __builtin_elementwise_fma(v, c2, nm);
            s_cur[x][kb][i0] = v[0];
            s_cur[x][kb][i0 + 1] = v[1];
        };
        auto exp_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 3, kb = (j & 7) >> 1, i0 = 2 * (j & 1);
            s_cur[x][kb][i0] = fast_exp2(s_cur[x][kb][i0]);
            s_cur[x][kb][i0 + 1] = fast_exp2(s_cur[x][kb][i0 + 1]);
        };
        auto cvt_pair = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = j >> 3, jj = j & 7, kb = jj >> 1, i0 = 2 * (jj & 1);
            pk[x][jj >> 2][jj & 3] = T::pack2(s_cur[x][kb][i0], s_cur[x][kb][i0 + 1]);
            l_part[x] += s_cur[x][kb][i0] + s_cur[x][kb][i0 + 1];
        };
        // 32 VALU micro-steps in dependency order: {fma(m+1), exp(m)} for m < 16, then cvt(m-16)
        auto micro = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m < 16) {
                if constexpr (m == 0) fma_pair(std::integral_constant<int, 0>{});
                if constexpr (m + 1 < 16) fma_pair(std::integral_constant<int, m + 1>{});
                exp_pair(mc);
            } else {
                cvt_pair(std::integral_constant<int, m - 16>{});
            }
        };
        constexpr int nExpSlots = kHasPrev ? 2 * fP : nAll;   // slots that may carry fma/exp steps
        constexpr int nCvtSlots = kHasPrev ? 2 * fQ : 0;      // slots that may carry pack steps (after PV)

        il2::sfor<kAheadF>([&](auto fc) { read_frag(fc); });
        il2::sfor<nAll>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i == (3 * nAll) / 4) {   // K(t+2) / V(t) -> slot t&1
                write_k((unsigned)t & 1u);
                write_v((unsigned)t & 1u);
            }
            issue_mfma(ic);
            if constexpr ((i & 1) == 1) read_frag(std::integral_constant<int, i / 2 + kAheadF>{});
            if constexpr (i < nExpSlots) {
                constexpr int m0 = i * 16 / nExpSlots, m1 = (i + 1) * 16 / nExpSlots;
                il2::sfor<m1 - m0>([&](auto dm) { micro(std::integral_constant<int, m0 + decltype(dm)::value>{}); });
            } else if constexpr (nCvtSlots > 0) {
                constexpr int k0 = (i - nExpSlots) * 16 / nCvtSlots, k1 = (i - nExpSlots + 1) * 16 / nCvtSlots;
                il2::sfor<k1 - k0>([&](auto dm) { micro(std::integral_constant<int, 16 + k0 + decltype(dm)::value>{}); });
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (nAll == 0) {
            il2::sfor<16>([&](auto mc) { micro(mc); });
            write_k((unsigned)t & 1u);
            write_v((unsigned)t & 1u);
        }
        if constexpr (nCvtSlots == 0) il2::sfor<16>([&](auto mc) { micro(std::integral_constant<int, 16 + decltype(mc)::value>{}); });

        if constexpr (kHasNext) {
            if (mask_next) mask_tail(t + 1, s_nxt);
            if constexpr (kTrack) {
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const float tmax = row_max(s_nxt[x]) * c;
                    if (__any(tmax - m_ref[x] > kThr)) {   // rare: raise the reference max; P(t) is still pending
                        const float m_new = fmaxf(across_max(tmax), m_ref[x]);
                        const float alpha = fast_exp2(m_ref[x] - m_new);
                        m_ref[x] = m_new;
#pragma unroll
                        for (int db = 0; db < 4; ++db)
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[x][db][i] *= alpha;
                        l_part[x] *= alpha;
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int w = 0; w < 4; ++w)
                                pk[x][s2][w] = T::pack2(T::lo(pk[x][s2][w]) * alpha, T::hi(pk[x][s2][w]) * alpha);
                    }
                }
            }
        }
        __syncthreads();
    };

    auto run = [&](auto track_c) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int db = 0; db < 4; ++db) o[x][db] = zero4;
            l_part[x] = 0.0f;
        }
        // prologue: K(0), K(1) into LDS; S(0); reference max
        load_k(0);
        write_k(0);
        load_k(1);
        write_k(1);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 kf = lds_read16(smem, kb * 2048u + k_rd[ks]);
#pragma unroll
                for (int x = 0; x < 2; ++x) sA[x][kb] = M::mfma(kf, qf[x][ks], ks == 0 ? zero4 : sA[x][kb]);
            }
        if (ntiles == 1 && partial) mask_tail(0, sA);
#pragma unroll
        for (int x = 0; x < 2; ++x)
            m_ref[x] = across_max(row_max(sA[x]) * c) + (decltype(track_c)::value ? 0.0f : kHeadroom);
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
            for (; t + 1 < ntiles; ++t) {   // leftovers in canonical naming (scores in sB)
                iter(track_c, yes, yes, t, partial && (t + 2 == ntiles), sB, sA);
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) sB[x][kb] = sA[x][kb];
            }
            iter(track_c, yes, no, ntiles - 1, false, sB, sA);
        }
        // drain: O^T += V(last)^T.P(last)^T
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
#pragma unroll
                    for (int x = 0; x < 2; ++x) o[x][db] = M::mfma(vf, pk[x][s], o[x][db]);
                }
        }
    };

    run(no);
    float l_row[2] = {across_sum(l_part[0]), across_sum(l_part[1])};
    {
        // a packed p can only have overflowed if the fp32 row sum reached the 16-bit format's range
        const float lim = T::id == 1 ? 0x1p+96f : 60000.0f;
        const bool bad = !(l_row[0] < lim) || !(l_row[1] < lim);
        if (__syncthreads_or(bad ? 1 : 0)) {
            __syncthreads();
            run(yes);
            l_row[0] = across_sum(l_part[0]);
            l_row[1] = across_sum(l_part[1]);
        }
    }

    // o[x][db][i] = O[q_row0 + 16x][16*db + 4*g + i]
    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const float inv = 1.0f / l_row[x];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const unsigned col = 16u * db + 4u * g;
            const unsigned row = q_row0 + 16u * x;
            if constexpr (kOutF32) {
                const f32x4 v = {o[x][db][0] * inv, o[x][db][1] * inv, o[x][db][2] * inv, o[x][db][3] * inv};
                buf_store16(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
            } else {
                const u32x2 v = {T::pack2(o[x][db][0] * inv, o[x][db][1] * inv), T::pack2(o[x][db][2] * inv, o[x][db][3] * inv)};
                buf_store8(ro, (row * D + col) * 2u, v);
            }
        }
    }
    }   // persistent loop over work items
}

template <typename T, bool kOutF32>
static hipError_t launch_il2x16(const void* Q, const void* K, const void* V, void* O,
                                int BH, int N, float scale, hipStream_t stream)
{
    constexpr int lds_bytes = 4 * kBlockN * 64 * 2;   // 32 KB
    const int nqb = (N + il2::kRows - 1) / il2::kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    static const int grid_cap = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus;
    }();
    const unsigned grid = nwg > grid_cap ? (unsigned)grid_cap : (unsigned)nwg;
    hipLaunchKernelGGL((fa_fwd_il2x16_kernel<T, kOutF32>), dim3(grid), dim3(64 * il2::kW), lds_bytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, (unsigned)nwg);
    return hipGetLastError();
}

hipError_t il2x16_dispatch(const void* Q, const void* K, const void* V, void* O,
                           int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                           hipStream_t stream)
{
    if (D != 64) return hipErrorInvalidValue;
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_il2x16<F16, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_il2x16<F16, false>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_il2x16<BF16, true>(Q, K, V, O, BH, N, scale, stream)
                          : launch_il2x16<BF16, false>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
