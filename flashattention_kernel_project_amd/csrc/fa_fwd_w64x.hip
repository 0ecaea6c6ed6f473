// fa_fwd_w64x.hip -- fa_fwd_w64 (64 query rows per wave, phase-ordered, d = 64) on v_mfma_f32_16x16x32.
//
// The wave's 64 rows are four 16-row blocks that share every K and V^T fragment read from LDS, so each
// fragment feeds four MFMAs (LDS bytes per wave-tile as fa_fwd_w64: 8 KB of K, 8 KB of V) and
// consecutive MFMAs always belong to different accumulator chains.  Why try the shape: on this device
// the 16x16x32 form sustains 1.88 PF of fp16 matrix work at the power cap against 1.62 PF for
// 32x32x16 (tools/microbench/mfma_power.hip) -- 14 % less energy for the same product -- and the
// kernel is power-limited.  fa_fwd_il2x16.hip tried it with 32 rows per wave and
// tied; here the MFMA issue slots per LDS byte double again.
//
// Lane roles (lane = 16*g + c): the accumulator of S^T = K.Q^T for (query block x, key block kb) holds
// query 16x + c on the lane and keys 16kb + 4g + i in register i; four lanes (g = 0..3) share a query
// row (row statistics meet across them once, at the end).  Packed to 16 bit, the registers of key
// blocks 2s and 2s+1 are the B fragment of k-step s of O^T += V^T.P^T with k-slot 8g+j <-> key
// 32s + 16(j>>2) + 4g + (j&3); the transposed V reads fetch exactly those keys.  LDS images as
// fa_fwd_il2x16.hip: K row-major with the 16-B chunk index XORed by (row>>1)&7; V in 256-B blocks
// [key/8][d/16] x [8 keys][16 cols].  Optimistic pass + tracked re-run, v_dot2c row sums, persistent grid
// as fa_fwd_w64.hip.
#include "fa_tile.hpp"

#include <type_traits>
#include <utility>

namespace fa {

namespace w64x {
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
#ifndef FA_W64X_WAVES
#define FA_W64X_WAVES 8
#endif
constexpr int kW = FA_W64X_WAVES;        // waves per workgroup (4: two independent workgroups per CU)
#ifndef FA_W64X_AHEAD
#define FA_W64X_AHEAD 2
#endif
#ifndef FA_W64X_STAGE_AT
#define FA_W64X_STAGE_AT 2   // quarter of the PV phase in front of which the staged tile is written to LDS (0..3)
#endif
#ifndef FA_W64X_MIDBAR
#define FA_W64X_MIDBAR 0   // 1: three K/V buffers, the one barrier per tile sits right behind the mid-PV staging write, and the
                           // next tile's first K fragments are read during the second half of PV (no LDS latency, no barrier, at the tile boundary)
#endif
#ifndef FA_W64X_PKADD
#define FA_W64X_PKADD 1   // fp16 row sums at d=64 by v_pk_add_f32, one instruction per pair (-1.8 % against two v_add; neutral at d=128: off there)
#endif
#ifndef FA_W64X_DOT2
#define FA_W64X_DOT2 0   // fp16 row sums: 0 = v_add of the fp32 p (-3.7 % at d=64, -2.5 % at d=128 against 1 = v_dot2c over the rounded weights); bf16 always v_dot2c (accuracy)
#endif
constexpr int kAhead = FA_W64X_AHEAD, kRing = kAhead + 1;
}  // namespace w64x

// D = head dim (64 or 128); X = 16-row query blocks per wave (4 at D = 64: 64 rows; 2 at D = 128: 32 rows).
template <typename T, int D, int X, bool kOutF32>
__global__ __launch_bounds__(64 * w64x::kW, 2)
void fa_fwd_w64x_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                        const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                        int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace w64x;
    using M = Mx<T>;
    using G = TileGeom<D>;
    constexpr int kRows = 16 * X * kW;              // query rows per workgroup
    constexpr int kKS = D / 32, kDB = D / 16;       // k-steps of QK^T, 16-column blocks of O^T
    constexpr int kLoads = (kBlockN * G::kChunks) / (64 * kW);
    constexpr unsigned kRowB = D * 2;
    constexpr unsigned kTile = kBlockN * D * 2;     // one K or V tile
    constexpr unsigned kBuf = 2 * kTile;            // [K tile][V tile]
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two buffers

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned c16 = lane & 15u, g = lane >> 4;
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    // staging: thread -> kLoads 16-B chunks of the K tile and of the V tile
    unsigned st_goff[kLoads], k_lds[kLoads], v_lds[kLoads];
#pragma unroll
    for (int p = 0; p < kLoads; ++p) {
        const unsigned idx = tid + p * 64u * kW;
        const unsigned srow = idx / G::kChunks, sch = idx % G::kChunks;
        st_goff[p] = srow * kRowB + sch * 16u;
        k_lds[p] = G::k_off(srow, sch);
        v_lds[p] = kTile + ((srow >> 3) * (unsigned)kDB + (sch >> 1)) * 256u + ((srow & 7u) << 5) + ((sch & 1u) << 4);
    }
    // K reads (A operand of QK^T): lane (c16,g) reads row 16*kb + c16, chunk 4*ks + g (the swizzle of row 16kb+c16 is that of c16)
    unsigned k_rd[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) k_rd[ks] = c16 * kRowB + (((4u * ks + g) ^ G::k_swz(c16)) << 4);
    // V^T reads (A operand of PV)
    const unsigned v_rd = kTile + (g >> 1) * (unsigned)kDB * 256u + ((4u * (g & 1u) + (c16 >> 2)) << 5) + (c16 & 3u) * 8u;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr float kHeadroom = 4.0f;
    const std::true_type yes{};
    const std::false_type no{};
    auto across_max = [&](float v) -> float {   // over the four lanes that share a query row
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        return fmaxf(v, __shfl_xor(v, 32, 64));
    };
    auto across_sum = [&](float v) -> float {
        v += __shfl_xor(v, 16, 64);
        return v + __shfl_xor(v, 32, 64);
    };

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
    const unsigned q_row0 = qb * kRows + wave * (16u * X) + c16;   // row of block 0; block x is 16x rows further

    u32x4 qf[X][kKS];   // B operand of QK^T: Q[row of block x][32*ks + 8*g .. +7]
#pragma unroll
    for (int x = 0; x < X; ++x)
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            u32x4 raw = buf_load16(rq, (q_row0 + 16u * x) * kRowB + (32u * ks + 8u * g) * 2u);
#pragma unroll
            for (int w = 0; w < 4; ++w) raw[w] ^= q_flip;
            qf[x][ks] = raw;
        }

    f32x4 o[X][kDB];
    float m_ref[X] = {}, l_part[X] = {};
    u32x4 kst[kLoads], vst[kLoads];

    auto run = [&](auto track_c) __attribute__((always_inline)) {
        constexpr bool kTrack = decltype(track_c)::value;
#pragma unroll
        for (int x = 0; x < X; ++x) {
#pragma unroll
            for (int db = 0; db < kDB; ++db) o[x][db] = zero4;
            l_part[x] = 0.0f;
        }
#pragma unroll
        for (int p = 0; p < kLoads; ++p) {
            kst[p] = buf_load16(rk, st_goff[p]);
            vst[p] = buf_load16(rv, st_goff[p]);
        }
#pragma unroll
        for (int p = 0; p < kLoads; ++p) {
            lds_write16(smem, k_lds[p], kst[p]);
            lds_write16(smem, v_lds[p], vst[p]);
        }
        __syncthreads();
        u32x4 kpre[kAhead];   // FA_W64X_MIDBAR: the first K fragments of the next tile, read ahead of the tile boundary
        if constexpr (FA_W64X_MIDBAR) {
#pragma unroll
            for (int i = 0; i < kAhead; ++i) kpre[i] = lds_read16(smem, (i % 4) * 16u * kRowB + k_rd[i / 4]);
        }
        unsigned cur = 0u;

        for (int t = 0; t < ntiles; ++t) {
            const unsigned nxt = FA_W64X_MIDBAR ? (cur == 2u * kBuf ? 0u : cur + kBuf) : kBuf - cur;
            // next tile: tiles past the end read zeros through the buffer bounds, into the free buffer
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                kst[p] = buf_load16(rk, (unsigned)(t + 1) * kTile + st_goff[p]);
                vst[p] = buf_load16(rv, (unsigned)(t + 1) * kTile + st_goff[p]);
            }

            // ---- S^T = K.Q^T: 4*kKS K fragments (key block kb, k-step ks), each feeding the X query blocks --------
            f32x4 s[X][4];
            u32x4 frag[kRing];
            auto read_k = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f < 4 * kKS) {
                    constexpr int kb = f % 4, ks = f / 4;
                    frag[f % kRing] = lds_read16(smem, cur + kb * 16u * kRowB + k_rd[ks]);
                }
            };
            if constexpr (FA_W64X_MIDBAR) {
                sfor<kAhead>([&](auto fc) { frag[decltype(fc)::value % kRing] = kpre[decltype(fc)::value]; });
            } else {
                sfor<kAhead>([&](auto fc) { read_k(fc); });
            }
            sfor<4 * kKS>([&](auto fc) {
                constexpr int f = decltype(fc)::value, kb = f % 4, ks = f / 4;
#pragma unroll
                for (int x = 0; x < X; ++x) s[x][kb] = M::mfma(frag[f % kRing], qf[x][ks], ks == 0 ? zero4 : s[x][kb]);
                read_k(std::integral_constant<int, f + kAhead>{});
            });

            if (partial && t + 1 == ntiles) {   // keys >= N -> -inf (p = 0)
#pragma unroll
                for (int x = 0; x < X; ++x)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (t * kBlockN + 16 * kb + 4 * (int)g + i >= N) s[x][kb][i] = -INFINITY;
            }

            // ---- reference max: tile 0 always; later tiles only in the tracked (fallback) pass ----
            if (kTrack || t == 0) {
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    float a = max3(s[x][0][0], s[x][0][1], s[x][0][2]), b = max3(s[x][1][0], s[x][1][1], s[x][1][2]);
                    float d = max3(s[x][2][0], s[x][2][1], s[x][2][2]), e = max3(s[x][3][0], s[x][3][1], s[x][3][2]);
                    float tmax = fmaxf(max3(a, b, s[x][0][3]), max3(d, e, fmaxf(s[x][1][3], fmaxf(s[x][2][3], s[x][3][3])))) * c;
                    if (t == 0) {
                        m_ref[x] = across_max(tmax) + (kTrack ? 0.0f : kHeadroom);
                    } else if (__any(tmax - m_ref[x] > kThr)) {
                        const float m_new = fmaxf(across_max(tmax), m_ref[x]);
                        const float alpha = fast_exp2(m_ref[x] - m_new);
                        m_ref[x] = m_new;
#pragma unroll
                        for (int db = 0; db < kDB; ++db)
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[x][db][i] *= alpha;
                        l_part[x] *= alpha;
                    }
                }
            }

            // ---- P = 2^(c*S - m), row sums over the rounded weights, packed to 16 bit ----------------------
            u32x4 pk[X][2];
            const f32x2 c2 = {c, c};
#pragma unroll
            for (int x = 0; x < X; ++x) {
                const f32x2 nm = {-m_ref[x], -m_ref[x]};
                float ls0 = 0.0f, ls1 = 0.0f;
                f32x2 lsv = {0.0f, 0.0f}, lsw = {0.0f, 0.0f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        f32x2 v = {s[x][kb][2 * pr], s[x][kb][2 * pr + 1]};
                        v = __builtin_elementwise_fma(v, c2, nm);
                        const float p0 = fast_exp2(v[0]), p1 = fast_exp2(v[1]);
                        const unsigned w = T::pack2(p0, p1);
                        pk[x][kb >> 1][(kb & 1) * 2 + pr] = w;
                        if constexpr (T::kSumRounded || FA_W64X_DOT2) {
                            if (pr) ls1 = T::sum2(w, ls1);
                            else ls0 = T::sum2(w, ls0);
                        } else if constexpr (FA_W64X_PKADD && D == 64) {   // one v_pk_add_f32 per pair, two chains
                            const f32x2 pv = {p0, p1};
                            if (pr) lsw = lsw + pv;
                            else lsv = lsv + pv;
                        } else {
                            ls0 += p0;
                            ls1 += p1;
                        }
                    }
                l_part[x] += (ls0 + ls1) + ((lsv[0] + lsv[1]) + (lsw[0] + lsw[1]));
            }

            // ---- O^T += V^T.P^T: 2*kDB V^T fragments (k-step sk, head-dim block db), each feeding the X blocks -------
            auto read_v = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f < 2 * kDB) {
                    constexpr int db = f % kDB, sk = f / kDB;
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(smem, cur + v_rd + (4u * sk + 2u * jj) * (unsigned)kDB * 256u + db * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    frag[f % kRing] = vf;
                }
            };
            sfor<kAhead>([&](auto fc) { read_v(fc); });
            sfor<2 * kDB>([&](auto fc) {
                constexpr int f = decltype(fc)::value, db = f % kDB, sk = f / kDB;
                if constexpr (f == (FA_W64X_STAGE_AT * 2 * kDB) / 4) {   // land the next tile in the other buffer (default: half way through PV)
#pragma unroll
                    for (int p = 0; p < kLoads; ++p) {
                        lds_write16(smem, nxt + k_lds[p], kst[p]);
                        lds_write16(smem, nxt + v_lds[p], vst[p]);
                    }
                    if constexpr (FA_W64X_MIDBAR) {
                        __syncthreads();   // tile t+1 visible; everybody is past tile t-1 (whose buffer tile t+2 will take)
#pragma unroll
                        for (int i = 0; i < kAhead; ++i) kpre[i] = lds_read16(smem, nxt + (i % 4) * 16u * kRowB + k_rd[i / 4]);
                    }
                }
#pragma unroll
                for (int x = 0; x < X; ++x) o[x][db] = M::mfma(frag[f % kRing], pk[x][sk], o[x][db]);
                read_v(std::integral_constant<int, f + kAhead>{});
            });
            if constexpr (!FA_W64X_MIDBAR) __syncthreads();
            cur = nxt;
        }
        if constexpr (FA_W64X_MIDBAR) __syncthreads();   // the buffers are rewritten by whatever runs next
    };

    run(no);
    float l_row[X];
    bool bad = false;
    const float lim = T::id == 1 ? 0x1p+96f : 60000.0f;   // a packed p can only have overflowed if the row sum got here
#pragma unroll
    for (int x = 0; x < X; ++x) {
        l_row[x] = across_sum(l_part[x]);
        bad = bad || !(l_row[x] < lim);
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        run(yes);
#pragma unroll
        for (int x = 0; x < X; ++x) l_row[x] = across_sum(l_part[x]);
    }

    // o[x][db][i] = O[q_row0 + 16x][16*db + 4*g + i]
    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
#pragma unroll
    for (int x = 0; x < X; ++x) {
        const float inv = 1.0f / l_row[x];
        const unsigned row = q_row0 + 16u * x;
#pragma unroll
        for (int db = 0; db < kDB; ++db) {
            const unsigned col = 16u * db + 4u * g;
            const float a = o[x][db][0] * inv, b = o[x][db][1] * inv, cc = o[x][db][2] * inv, d = o[x][db][3] * inv;
            if constexpr (kOutF32) {
                const f32x4 v = {a, b, cc, d};
                buf_store16(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
            } else {
                buf_store8(ro, (row * D + col) * 2u, u32x2{T::pack2(a, b), T::pack2(cc, d)});
            }
        }
    }
    }   // persistent loop over work items
}

template <typename T, int D, int X, bool kOutF32>
static hipError_t launch_w64x(const void* Q, const void* K, const void* V, void* O,
                              int BH, int N, float scale, hipStream_t stream)
{
    constexpr int lds_bytes = (FA_W64X_MIDBAR ? 6 : 4) * kBlockN * D * 2;   // two (three) [K tile][V tile] buffers
    constexpr int kRows = 16 * X * w64x::kW;
    const int nqb = (N + kRows - 1) / kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int grid_cap = device_cus();
    const long long cap = (long long)grid_cap * (8 / w64x::kW);
    const unsigned grid = nwg > cap ? (unsigned)cap : (unsigned)nwg;
    const hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(&fa_fwd_w64x_kernel<T, D, X, kOutF32>), lds_bytes);
    if (attr != hipSuccess) return attr;
    FA_LAUNCH((fa_fwd_w64x_kernel<T, D, X, kOutF32>), dim3(grid), dim3(64 * w64x::kW), lds_bytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, (unsigned)nwg);
    return launch_status();
}

#ifndef FA_W64X_X64
#define FA_W64X_X64 4    // 16-row blocks per wave at d = 64
#endif
#ifndef FA_W64X_X128
#define FA_W64X_X128 2   // 16-row blocks per wave at d = 128
#endif

hipError_t w64x_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                         hipStream_t stream)
{
    if (D != 64 && D != 128) return hipErrorInvalidValue;
    if ((unsigned long long)(N + 64 * w64x::kW) * (unsigned)D * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (D == 64) {
        if (in_dtype == 0)
            return out_dtype == 0 ? launch_w64x<F16, 64, FA_W64X_X64, true>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_w64x<F16, 64, FA_W64X_X64, false>(Q, K, V, O, BH, N, scale, stream);
        return out_dtype == 0 ? launch_w64x<BF16, 64, FA_W64X_X64, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_w64x<BF16, 64, FA_W64X_X64, false>(Q, K, V, O, BH, N, scale, stream);
    }
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_w64x<F16, 128, FA_W64X_X128, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_w64x<F16, 128, FA_W64X_X128, false>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_w64x<BF16, 128, FA_W64X_X128, true>(Q, K, V, O, BH, N, scale, stream)
                          : launch_w64x<BF16, 128, FA_W64X_X128, false>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
