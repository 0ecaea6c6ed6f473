// fa_fwd_tlp.hip -- thread-level-parallel variant of the tiled attention forward (gfx950, d = 64).
//
// Same data layout, MFMA orientation and LDS images as fa_fwd_kernels.hip.  Where fa_fwd_il.hip
// overlaps the matrix and vector pipes inside one wave's instruction stream (two score sets, two
// packed-P sets, ~220 VGPRs, two waves per SIMD), this kernel keeps the per-wave stream simple
// (QK^T -> softmax -> PV per tile, one score set) and small enough (<= 168 VGPRs) for THREE
// workgroups of four waves per CU: three independent waves per SIMD, never barrier-coupled to each
// other, cover each other's LDS / MFMA / barrier stalls.  It carries the same vector-work
// reductions: optimistic pass without per-tile row max (exact overflow detection through the MFMA
// row sum, tracked re-run as fallback), row sums on the matrix pipe, packed fma.
#include "fa_tile.hpp"

#include <type_traits>
#include <utility>

namespace fa {

namespace tlp {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
}  // namespace tlp

template <typename T, bool kOutF32, int kOcc>
__global__ __launch_bounds__(256, kOcc)
void fa_fwd_tlp_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                       const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                       int N, int nqb, float scale_log2e)
{
    constexpr int D = 64, W = 4;
    using G = TileGeom<D>;
    constexpr int kLoadsW = (kBlockN * G::kChunks) / (64 * W);   // 2
    constexpr int kAhead = 3, kRing = kAhead + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][K tile][V tile]

    const unsigned nwg = gridDim.x, bid = blockIdx.x;
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
    const unsigned q_row = qb * (32u * W) + wave * 32u + r;

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

    unsigned g_off[kLoadsW], k_lds[kLoadsW], v_lds[kLoadsW];
#pragma unroll
    for (int p = 0; p < kLoadsW; ++p) {
        const unsigned idx = tid + p * 64u * W;
        const unsigned row = idx / G::kChunks, ch = idx % G::kChunks;
        g_off[p] = row * G::kRowBytes + ch * 16u;
        k_lds[p] = G::k_off(row, ch);
        v_lds[p] = G::kTileBytes + G::v_off(row, ch);
    }
    u32x4 kst[kLoadsW], vst[kLoadsW];
    auto stage_load = [&](int tile) {   // tiles past the end read zeros through the buffer bounds
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
            kst[p] = buf_load16(rk, (unsigned)tile * kBlockN * G::kRowBytes + g_off[p]);
            vst[p] = buf_load16(rv, (unsigned)tile * kBlockN * G::kRowBytes + g_off[p]);
        }
    };
    auto stage_write = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
            lds_write16(smem, buf * G::kBufBytes + k_lds[p], kst[p]);
            lds_write16(smem, buf * G::kBufBytes + v_lds[p], vst[p]);
        }
    };

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
    f32x16 o[G::kDBlocks], o_l;
    const u32x4 ones = {T::kOnes2, T::kOnes2, T::kOnes2, T::kOnes2};
    float m_ref = 0.0f;

    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;
    constexpr float kHeadroom = 4.0f;
    constexpr int nQK = 2 * G::kKSteps, nPV = 4 * (G::kDBlocks + 1);

    auto run = [&](auto track_c) {
        constexpr bool kTrack = decltype(track_c)::value;
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db) o[db] = zero16;
        o_l = zero16;
        stage_load(0);
        stage_write(0);
        __syncthreads();

        for (int t = 0; t < ntiles; ++t) {
            const unsigned cur = (unsigned)t & 1u;
            stage_load(t + 1);

            // ---- S^T = K.Q^T, LDS operand reads kAhead MFMAs ahead ----------------------------
            f32x16 s[2];
            u32x4 frag[kRing];
            auto read_k = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < nQK) {
                    constexpr int kb = i / G::kKSteps, ks = i % G::kKSteps;
                    frag[i % kRing] = lds_read16(smem, cur * G::kBufBytes + kb * 32u * G::kRowBytes + k_rd_row +
                                                           (((2u * ks + h) ^ k_rd_swz) << 4));
                }
            };
            tlp::sfor<kAhead>([&](auto ic) { read_k(ic); });
            tlp::sfor<nQK>([&](auto ic) {
                constexpr int i = decltype(ic)::value, kb = i / G::kKSteps, ks = i % G::kKSteps;
                s[kb] = T::mfma32(frag[i % kRing], qf[ks], ks == 0 ? zero16 : s[kb]);
                read_k(std::integral_constant<int, i + kAhead>{});
            });

            if (partial && t + 1 == ntiles) {   // keys >= N -> -inf (p = 0)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = t * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                        if (key >= N) s[kb][i] = -INFINITY;
                    }
            }

            // ---- reference max: tile 0 always; later tiles only in the tracked (fallback) pass ----
            if (kTrack || t == 0) {
                float tmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 32; e += 2) tmax = max3(tmax, s[e >> 4][e & 15], s[(e + 1) >> 4][(e + 1) & 15]);
                tmax *= c;
                if (t == 0) {
                    m_ref = fmaxf(tmax, swap_halves(tmax)) + (kTrack ? 0.0f : kHeadroom);
                } else if (__any(tmax - m_ref > kThr)) {
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
                }
            }

            // ---- P = 2^(c*S - m) packed; O^T += V^T.P^T and l += 1.P^T, V reads kAhead ahead -------
            u32x4 pk[4];
            const float neg_m = -m_ref;
            const f32x2 c2 = {c, c}, neg_m2 = {neg_m, neg_m};
            auto softmax_quarter = [&](auto qc) {   // 8 scores -> pk[q]
                constexpr int q4 = decltype(qc)::value, kb = q4 >> 1, b8 = (q4 & 1) * 8;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    f32x2 x = {s[kb][b8 + 2 * w], s[kb][b8 + 2 * w + 1]};
                    x = __builtin_elementwise_fma(x, c2, neg_m2);
                    pk[q4][w] = T::pack2(fast_exp2(x[0]), fast_exp2(x[1]));
                }
            };
            auto read_v = [&](auto ic) {
                constexpr int i = decltype(ic)::value;   // PV MFMA index, ks-major: (ks, db)
                if constexpr (i < nPV) {
                    constexpr int ks = i / (G::kDBlocks + 1), db = i % (G::kDBlocks + 1);
                    if constexpr (db < G::kDBlocks) {
                        u32x4 vf;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const u32x2 half = lds_read_tr8(
                                smem, cur * G::kBufBytes + v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                            vf[2 * jj] = half[0];
                            vf[2 * jj + 1] = half[1];
                        }
                        frag[i % kRing] = vf;
                    }
                }
            };
            tlp::sfor<kAhead>([&](auto ic) { read_v(ic); });
            softmax_quarter(std::integral_constant<int, 0>{});
            tlp::sfor<nPV>([&](auto ic) {
                constexpr int i = decltype(ic)::value, ks = i / (G::kDBlocks + 1), db = i % (G::kDBlocks + 1);
                // the next quarter of P is exponentiated beside the MFMAs that consume this one
                if constexpr (db == 0 && ks + 1 < 4) softmax_quarter(std::integral_constant<int, ks + 1>{});
                if constexpr (i == nPV / 2) stage_write(cur ^ 1u);
                if constexpr (db < G::kDBlocks) o[db] = T::mfma32(frag[i % kRing], pk[ks], o[db]);
                else o_l = T::mfma32(ones, pk[ks], o_l);
                read_v(std::integral_constant<int, i + kAhead>{});
            });
            __syncthreads();
        }
    };

    run(std::false_type{});
    {
        const bool bad = !(__builtin_fabsf(o_l[0]) < INFINITY);
        if (__syncthreads_or(bad ? 1 : 0)) run(std::true_type{});
    }

    const float inv = 1.0f / o_l[0];
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
}

template <typename T, bool kOutF32, int kOcc>
static hipError_t launch_tlp(const void* Q, const void* K, const void* V, void* O,
                             int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<64>;
    const int nqb = (N + 127) / 128;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((fa_fwd_tlp_kernel<T, kOutF32, kOcc>), dim3((unsigned)nwg), dim3(256), G::kLdsBytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e);
    return hipGetLastError();
}

// occ: waves per SIMD the register budget is held to (3 or 4)
hipError_t tlp_dispatch(const void* Q, const void* K, const void* V, void* O,
                        int BH, int N, int D, float scale, int in_dtype, int out_dtype, int occ,
                        hipStream_t stream)
{
    if (D != 64) return hipErrorInvalidValue;
    if (occ == 4) {
        if (in_dtype == 0)
            return out_dtype == 0 ? launch_tlp<F16, true, 4>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_tlp<F16, false, 4>(Q, K, V, O, BH, N, scale, stream);
        return out_dtype == 0 ? launch_tlp<BF16, true, 4>(Q, K, V, O, BH, N, scale, stream)
                              : launch_tlp<BF16, false, 4>(Q, K, V, O, BH, N, scale, stream);
    }
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_tlp<F16, true, 3>(Q, K, V, O, BH, N, scale, stream)
                              : launch_tlp<F16, false, 3>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_tlp<BF16, true, 3>(Q, K, V, O, BH, N, scale, stream)
                          : launch_tlp<BF16, false, 3>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
