// fa_fwd_w64m.hip -- fa_fwd_w64 with ONE matrix phase per tile: QK^T one tile ahead, merged with PV.
//
// fa_fwd_w64 runs QK^T(t) -> softmax(t) -> PV(t) per tile: the softmax waits for the MFMA issued
// right before it and PV waits for the last pack, so every tile has two pipe-turnaround bubbles, and
// each matrix phase has only as many independent accumulator chains as one product offers (a dependent
// MFMA issued right behind its producer stalls; alternating accumulators was worth 1.6-2.1 %).  Here
//
//   iteration t:   matrix phase   S(t+1) = K(t+1).Q^T   interleaved with   O^T += V(t)^T.P(t)^T
//                  vector phase   P(t+1) = 2^(c*S(t+1) - m), row sums, pack
//
// so the matrix phase has both products' chains to alternate between (8 at d=64, 6 at d=128), the
// vector phase's input was produced at the START of the matrix phase, and the next matrix phase's
// QK^T does not depend on the vector phase at all.  Registers are those of fa_fwd_w64 (S and packed P
// are live together there too).  LDS: K and V rings of two tiles each, K one tile ahead of V; one
// barrier per tile.  Same optimistic pass + tracked re-run, row sums, persistent XCD-aware grid.
//
// MEASURED OUTCOME (bit-identical results to fa_fwd_w64): SLOWER -- B8 H16 N4096 d64 fp16 0.599 vs 0.578 ms,
// bf16 0.578 vs 0.546; N8192 d128 6.07 vs 4.16 ms.  Mixing K (ds_read_b128) and V^T (ds_read_b64_tr_b16)
// fragment reads in one stream and keeping S and packed P live across the matrix phase costs more than the two
// pipe turnarounds it removes.  Kept selectable (FA_ALGO 15) and parity-tested; not used by AUTO.
#include "fa_tile.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

namespace w64m {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#ifndef FA_W64M_AHEAD
#define FA_W64M_AHEAD 2
#endif
constexpr int kW = 8;
constexpr int kAhead = FA_W64M_AHEAD, kRing = kAhead + 1;
}  // namespace w64m

template <typename T, int D, int X, bool kOutF32>
__global__ __launch_bounds__(64 * w64m::kW, 2)
void fa_fwd_w64m_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                        const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                        int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace w64m;
    using G = TileGeom<D>;
    constexpr int kRows = 32 * X * kW;
    constexpr int kLoads = (kBlockN * G::kChunks) / (64 * kW);
    constexpr int nQK = 2 * G::kKSteps, nPV = 4 * G::kDBlocks;   // fragments of a matrix phase
    // LDS: [K0][K1][V0][V1]
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned r = lane & 31u, h = lane >> 5;
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    unsigned st_goff[kLoads], k_lds[kLoads], v_lds[kLoads];
#pragma unroll
    for (int p = 0; p < kLoads; ++p) {
        const unsigned idx = tid + p * 64u * kW;
        const unsigned srow = idx / G::kChunks, sch = idx % G::kChunks;
        st_goff[p] = srow * G::kRowBytes + sch * 16u;
        k_lds[p] = G::k_off(srow, sch);
        v_lds[p] = 2u * G::kTileBytes + G::v_off(srow, sch);
    }
    const unsigned k_rd_row = r * G::kRowBytes;
    const unsigned k_rd_swz = G::k_swz(r);
    const unsigned i16 = lane & 15u, vq = i16 >> 2, vp = i16 & 3u, vg = (lane >> 4) & 1u;
    unsigned v_rd[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        v_rd[par] = 2u * G::kTileBytes + h * G::kDBlocks * 256u + ((vq ^ par) << 6) + vg * 32u + vp * 8u;

    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.0f;
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
    const unsigned q_row0 = qb * kRows + wave * (32u * X) + r;

    u32x4 qf[X][G::kKSteps];
#pragma unroll
    for (int x = 0; x < X; ++x)
#pragma unroll
        for (int s = 0; s < G::kKSteps; ++s) {
            u32x4 raw = buf_load16(rq, (q_row0 + 32u * x) * G::kRowBytes + (16u * s + 8u * h) * 2u);
#pragma unroll
            for (int w = 0; w < 4; ++w) raw[w] ^= q_flip;
            qf[x][s] = raw;
        }

    f32x16 o[X][G::kDBlocks];
    float m_ref[X] = {}, l_part[X] = {};
    u32x4 kst[kLoads], vst[kLoads];
    u32x4 pk[X][4];        // packed P of the tile whose PV is next
    f32x16 s[X][2];        // raw scores of the tile being softmaxed

    auto mask_tile = [&](int tile) {   // keys >= N -> -inf (p = 0)
#pragma unroll
        for (int x = 0; x < X; ++x)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = tile * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                    if (key >= N) s[x][kb][i] = -INFINITY;
                }
    };
    // reference max (tile 0: set; later tiles, tracked pass only: raise lazily) then P = 2^(c*S - m) -> pk, row sums
    auto softmax = [&](auto track_c, int tile) __attribute__((always_inline)) {
        constexpr bool kTrack = decltype(track_c)::value;
        if (partial && tile + 1 == ntiles) mask_tile(tile);
        if (kTrack || tile == 0) {
#pragma unroll
            for (int x = 0; x < X; ++x) {
                float tmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 32; e += 2) tmax = max3(tmax, s[x][e >> 4][e & 15], s[x][(e + 1) >> 4][(e + 1) & 15]);
                tmax *= c;
                if (tile == 0) {
                    m_ref[x] = fmaxf(tmax, swap_halves(tmax)) + (kTrack ? 0.0f : kHeadroom);
                } else if (__any(tmax - m_ref[x] > kThr)) {
                    // every earlier tile's PV is already in O (the matrix phase precedes this): scale O and l only
                    const float m_new = fmaxf(fmaxf(tmax, swap_halves(tmax)), m_ref[x]);
                    const float alpha = fast_exp2(m_ref[x] - m_new);
                    m_ref[x] = m_new;
#pragma unroll
                    for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[x][db][i] *= alpha;
                    l_part[x] *= alpha;
                }
            }
        }
        const f32x2 c2 = {c, c};
#pragma unroll
        for (int x = 0; x < X; ++x) {
            const f32x2 nm = {-m_ref[x], -m_ref[x]};
            float ls0 = 0.0f, ls1 = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int kb = q4 >> 1, b8 = (q4 & 1) * 8;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    f32x2 v = {s[x][kb][b8 + 2 * w], s[x][kb][b8 + 2 * w + 1]};
                    v = __builtin_elementwise_fma(v, c2, nm);
                    pk[x][q4][w] = T::pack2(fast_exp2(v[0]), fast_exp2(v[1]));
                    if (w & 1) ls1 = T::sum2(pk[x][q4][w], ls1);   // sums of the ROUNDED weights (fa_common.hpp)
                    else ls0 = T::sum2(pk[x][q4][w], ls0);
                }
            }
            l_part[x] += ls0 + ls1;
        }
    };

    // Matrix phase of iteration t: S(t+1) (K slot (t+1)&1) when kQK, O += V(t)^T.P(t)^T (V slot t&1) when kPV.
    // Fragment order alternates the two products so that consecutive MFMAs belong to different chains.
    auto matrix_phase = [&](auto qk_c, auto pv_c, int t) __attribute__((always_inline)) {
        constexpr bool kQK = decltype(qk_c)::value, kPV = decltype(pv_c)::value;
        constexpr int nF = (kQK ? nQK : 0) + (kPV ? nPV : 0);
        const unsigned kbuf = ((unsigned)(t + 1) & 1u) * G::kTileBytes, vbuf = ((unsigned)t & 1u) * G::kTileBytes;
        u32x4 frag[kRing];
        // fragment g -> (is_qk, index): alternate QK^T and PV while both remain
        auto is_qk = [](int g) constexpr {
            if (!kQK) return false;
            if (!kPV) return true;
            constexpr int nmin = nQK < nPV ? nQK : nPV;
            return g < 2 * nmin ? (g % 2 == 0) : (nQK > nPV);
        };
        auto idx_of = [](int g) constexpr {
            if (!kQK || !kPV) return g;
            constexpr int nmin = nQK < nPV ? nQK : nPV;
            return g < 2 * nmin ? g / 2 : g - nmin;
        };
        auto read = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < nF) {
                constexpr int i = idx_of(g);
                if constexpr (is_qk(g)) {
                    constexpr int kb = i % 2, ks = i / 2;   // alternate the two key blocks
                    frag[g % kRing] = lds_read16(smem, kbuf + kb * 32u * G::kRowBytes + k_rd_row + (((2u * ks + h) ^ k_rd_swz) << 4));
                } else {
                    constexpr int db = i % G::kDBlocks, ks = i / G::kDBlocks;   // alternate the head-dim blocks
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(smem, vbuf + v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    frag[g % kRing] = vf;
                }
            }
        };
        sfor<kAhead>([&](auto gc) { read(gc); });
        sfor<nF>([&](auto gc) {
            constexpr int g = decltype(gc)::value, i = idx_of(g);
            if constexpr (kQK && kPV && g == nF / 2) {   // land the staged tiles half way through the phase
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    lds_write16(smem, ((unsigned)t & 1u) * G::kTileBytes + k_lds[p], kst[p]);          // K(t+2) over K(t)
                    lds_write16(smem, ((unsigned)(t + 1) & 1u) * G::kTileBytes + v_lds[p], vst[p]);    // V(t+1) over V(t-1)
                }
            }
            if constexpr (is_qk(g)) {
                constexpr int kb = i % 2, ks = i / 2;
#pragma unroll
                for (int x = 0; x < X; ++x) s[x][kb] = T::mfma32(frag[g % kRing], qf[x][ks], ks == 0 ? zero16 : s[x][kb]);
            } else {
                constexpr int db = i % G::kDBlocks, ks = i / G::kDBlocks;
#pragma unroll
                for (int x = 0; x < X; ++x) o[x][db] = T::mfma32(frag[g % kRing], pk[x][ks], o[x][db]);
            }
            read(std::integral_constant<int, g + kAhead>{});
        });
    };

    auto run = [&](auto track_c) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < X; ++x) {
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db) o[x][db] = zero16;
            l_part[x] = 0.0f;
        }
        // ---- prologue: K(0) -> K slot 0, K(1) -> K slot 1, V(0) -> V slot 0; S(0); softmax(0) ----------
#pragma unroll
        for (int p = 0; p < kLoads; ++p) {
            kst[p] = buf_load16(rk, st_goff[p]);
            vst[p] = buf_load16(rv, st_goff[p]);
        }
#pragma unroll
        for (int p = 0; p < kLoads; ++p) {
            lds_write16(smem, k_lds[p], kst[p]);
            lds_write16(smem, v_lds[p], vst[p]);
            kst[p] = buf_load16(rk, G::kTileBytes + st_goff[p]);
        }
#pragma unroll
        for (int p = 0; p < kLoads; ++p) lds_write16(smem, G::kTileBytes + k_lds[p], kst[p]);
        __syncthreads();
        matrix_phase(yes, no, -1);   // S(0) from K slot 0
        softmax(track_c, 0);
        __syncthreads();             // iteration 0 overwrites K slot 0 half way through: everybody has read K(0)

        for (int t = 0; t + 1 < ntiles; ++t) {
            // tiles past the end read zeros through the buffer bounds and land in slots nobody reads again
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                kst[p] = buf_load16(rk, (unsigned)(t + 2) * G::kTileBytes + st_goff[p]);
                vst[p] = buf_load16(rv, (unsigned)(t + 1) * G::kTileBytes + st_goff[p]);
            }
            matrix_phase(yes, yes, t);     // S(t+1), O += V(t)^T.P(t)^T, staged tiles land
            softmax(track_c, t + 1);       // P(t+1)
            __syncthreads();
        }
        matrix_phase(no, yes, ntiles - 1);   // the last PV
    };

    run(no);
    float l_row[X];
    bool bad = false;
    const float lim = T::id == 1 ? 0x1p+96f : 60000.0f;   // a packed p can only have overflowed if the row sum got here
#pragma unroll
    for (int x = 0; x < X; ++x) {
        l_row[x] = l_part[x] + swap_halves(l_part[x]);
        bad = bad || !(l_row[x] < lim);
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        run(yes);
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

template <typename T, int D, int X, bool kOutF32>
static hipError_t launch_w64m(const void* Q, const void* K, const void* V, void* O,
                              int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    constexpr int kRows = 32 * X * w64m::kW;
    const int nqb = (N + kRows - 1) / kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    static const int grid_cap = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus;
    }();
    const unsigned grid = nwg > grid_cap ? (unsigned)grid_cap : (unsigned)nwg;
    const hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(&fa_fwd_w64m_kernel<T, D, X, kOutF32>), G::kLdsBytes);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fa_fwd_w64m_kernel<T, D, X, kOutF32>), dim3(grid), dim3(64 * w64m::kW), G::kLdsBytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, (unsigned)nwg);
    return hipGetLastError();
}

hipError_t w64m_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                         hipStream_t stream)
{
    if (D != 64 && D != 128) return hipErrorInvalidValue;
    if ((unsigned long long)(N + 64 * w64m::kW + 2 * kBlockN) * (unsigned)D * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (D == 64) {
        if (in_dtype == 0)
            return out_dtype == 0 ? launch_w64m<F16, 64, 2, true>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_w64m<F16, 64, 2, false>(Q, K, V, O, BH, N, scale, stream);
        return out_dtype == 0 ? launch_w64m<BF16, 64, 2, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_w64m<BF16, 64, 2, false>(Q, K, V, O, BH, N, scale, stream);
    }
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_w64m<F16, 128, 1, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_w64m<F16, 128, 1, false>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_w64m<BF16, 128, 1, true>(Q, K, V, O, BH, N, scale, stream)
                          : launch_w64m<BF16, 128, 1, false>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
