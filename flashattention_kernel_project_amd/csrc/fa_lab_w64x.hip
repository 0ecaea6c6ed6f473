// fa_lab_w64x.hip -- measurement build of the d=64 fp16 forward stream (fa_fwd_w64x.hip) -- NOT a product path.
//
// Same lane roles, LDS images and MFMA shape (v_mfma_f32_16x16x32_f16, four 16-row blocks per wave) as
// fa_fwd_w64x.hip, instantiated once per experiment:
//   kStruct 0   the shipped order: one barrier per tile, K and V staged together, written half way through PV
//   kStruct 1   two half-iterations per tile, H1 = {stage K(t+1); QK^T(t); softmax of blocks 0,1} and
//               H2 = {stage V(t+1); softmax of blocks 2,3; PV(t)}, a barrier after each.  With kSkew waves 4-7
//               run half an iteration behind waves 0-3 (one extra barrier in front / behind), so on every
//               SIMD a wave in its matrix part always sits beside a wave in its vector part.
//   kAbl        timing ablations (results wrong by construction): 1 no LDS operand reads, 2 no MFMA,
//               4 no softmax VALU, 8 no K/V staging, 16 no barrier.
//   kDiag       s_memtime stamps per phase, summed per wave over all tiles into diag[wg][wave][8].
// The reference's analogue of this file: FlashAttention/flashattn_forward_memory_bound/
// flashattn_stage_latency_breakdown.cu:181-207 (per-stage clock64 stamps) and flashattn_forward_cp_async_stall.cu:93-206.
#include "fa_tile.hpp"

#include <type_traits>
#include <utility>

namespace fa {
namespace lab {

template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

constexpr int kW = 8, kAhead = 2, kRing = kAhead + 1, D = 64, X = 4;

template <int kStruct, bool kSkew, int kAbl, bool kDiag>
__global__ __launch_bounds__(64 * kW, 2)
void lab_w64x_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                     const uint16_t* __restrict__ Vg, float* __restrict__ Og,
                     int N, int nqb, float scale_log2e, unsigned total_wg, unsigned long long* __restrict__ diag)
{
    using T = F16;
    using G = TileGeom<D>;
    constexpr int kRows = 16 * X * kW;
    constexpr int kKS = D / 32, kDB = D / 16;
    constexpr unsigned kRowB = D * 2;
    constexpr unsigned kTile = kBlockN * D * 2;
    constexpr unsigned kBuf = 2 * kTile;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned c16 = lane & 15u, g = lane >> 4;
    const float c = fabsf(scale_log2e);
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool late = kSkew && wave >= kW / 2;

    const unsigned srow = tid / G::kChunks, sch = tid % G::kChunks;
    const unsigned st_goff = srow * kRowB + sch * 16u;
    const unsigned k_lds = G::k_off(srow, sch);
    const unsigned v_lds = kTile + ((srow >> 3) * (unsigned)kDB + (sch >> 1)) * 256u + ((srow & 7u) << 5) + ((sch & 1u) << 4);
    unsigned k_rd[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) k_rd[ks] = c16 * kRowB + (((4u * ks + g) ^ G::k_swz(c16)) << 4);
    const unsigned v_rd = kTile + (g >> 1) * (unsigned)kDB * 256u + ((4u * (g & 1u) + (c16 >> 2)) << 5) + (c16 & 3u) * 8u;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto across_max = [&](float v) -> float {
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        return fmaxf(v, __shfl_xor(v, 32, 64));
    };
    auto across_sum = [&](float v) -> float {
        v += __shfl_xor(v, 16, 64);
        return v + __shfl_xor(v, 32, 64);
    };

    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
#define LAB_STAMP(slot)                                                   \
    if constexpr (kDiag) {                                                \
        __builtin_amdgcn_sched_barrier(0);                                \
        const unsigned long long now_ = stamp();                          \
        acc[slot] += now_ - last;                                         \
        last = now_;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                \
    }
    auto barrier = [&]() {
        if constexpr (!(kAbl & 16)) __syncthreads();
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
    const unsigned q_row0 = qb * kRows + wave * (16u * X) + c16;

    u32x4 qf[X][kKS];
#pragma unroll
    for (int x = 0; x < X; ++x)
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) qf[x][ks] = buf_load16(rq, (q_row0 + 16u * x) * kRowB + (32u * ks + 8u * g) * 2u);

    f32x4 o[X][kDB];
    float m_ref[X] = {}, l_part[X] = {};
    u32x4 kst, vst;
#pragma unroll
    for (int x = 0; x < X; ++x)
#pragma unroll
        for (int db = 0; db < kDB; ++db) o[x][db] = zero4;

    // ---- the three pieces of a tile --------------------------------------------------------------------------
    f32x4 s[X][4];
    u32x4 pk[X][2];
    u32x4 frag[kRing];
    auto qk = [&](unsigned cur) __attribute__((always_inline)) {
        auto read_k = [&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f < 4 * kKS) {
                constexpr int kb = f % 4, ks = f / 4;
                if constexpr (kAbl & 1) frag[f % kRing] = qf[kb][ks];
                else frag[f % kRing] = lds_read16(smem, cur + kb * 16u * kRowB + k_rd[ks]);
            }
        };
        sfor<kAhead>([&](auto fc) { read_k(fc); });
        sfor<4 * kKS>([&](auto fc) {
            constexpr int f = decltype(fc)::value, kb = f % 4, ks = f / 4;
            if constexpr (kAbl & 2) {
                asm volatile("" ::"v"(frag[f % kRing]));
                if constexpr (ks == 0) {
#pragma unroll
                    for (int x = 0; x < X; ++x) {
                        s[x][kb] = zero4;
                        asm volatile("" : "+v"(s[x][kb]));
                    }
                }
            } else {
#pragma unroll
                for (int x = 0; x < X; ++x) s[x][kb] = mfma(frag[f % kRing], qf[x][ks], ks == 0 ? zero4 : s[x][kb]);
            }
            read_k(std::integral_constant<int, f + kAhead>{});
        });
    };
    auto set_ref = [&]() __attribute__((always_inline)) {   // tile 0: reference max = row max + headroom
#pragma unroll
        for (int x = 0; x < X; ++x) {
            float a = max3(s[x][0][0], s[x][0][1], s[x][0][2]), b = max3(s[x][1][0], s[x][1][1], s[x][1][2]);
            float d = max3(s[x][2][0], s[x][2][1], s[x][2][2]), e = max3(s[x][3][0], s[x][3][1], s[x][3][2]);
            float tmax = fmaxf(max3(a, b, s[x][0][3]), max3(d, e, fmaxf(s[x][1][3], fmaxf(s[x][2][3], s[x][3][3])))) * c;
            m_ref[x] = across_max(tmax) + 4.0f;
        }
    };
    auto softmax = [&](auto x0c, auto x1c) __attribute__((always_inline)) {
        constexpr int x0 = decltype(x0c)::value, x1 = decltype(x1c)::value;
        const f32x2 c2 = {c, c};
#pragma unroll
        for (int x = x0; x < x1; ++x) {
            if constexpr (kAbl & 4) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) asm volatile("" ::"v"(s[x][kb]));
                pk[x][0] = __builtin_bit_cast(u32x4, s[x][0]);
                pk[x][1] = __builtin_bit_cast(u32x4, s[x][2]);
            } else {
                const f32x2 nm = {-m_ref[x], -m_ref[x]};
                f32x2 lsv = {0.0f, 0.0f}, lsw = {0.0f, 0.0f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        f32x2 v = {s[x][kb][2 * pr], s[x][kb][2 * pr + 1]};
                        v = __builtin_elementwise_fma(v, c2, nm);
                        const float p0 = fast_exp2(v[0]), p1 = fast_exp2(v[1]);
                        pk[x][kb >> 1][(kb & 1) * 2 + pr] = T::pack2(p0, p1);
                        const f32x2 pv = {p0, p1};
                        if (pr) lsw = lsw + pv;
                        else lsv = lsv + pv;
                    }
                l_part[x] += (lsv[0] + lsv[1]) + (lsw[0] + lsw[1]);
            }
        }
    };
    // PV over V fragments [f0, f1) of the 2*kDB; `mid` runs in front of fragment `at`
    auto pv = [&](unsigned cur, auto f0c, auto f1c, auto&& mid, auto atc) __attribute__((always_inline)) {
        constexpr int f0 = decltype(f0c)::value, f1 = decltype(f1c)::value, at = decltype(atc)::value;
        auto read_v = [&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f < f1) {
                constexpr int db = f % kDB, sk = f / kDB;
                if constexpr (kAbl & 1) {
                    frag[f % kRing] = qf[db][sk];
                } else {
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(smem, cur + v_rd + (4u * sk + 2u * jj) * (unsigned)kDB * 256u + db * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    frag[f % kRing] = vf;
                }
            }
        };
        sfor<kAhead>([&](auto fc) { read_v(std::integral_constant<int, f0 + decltype(fc)::value>{}); });
        sfor<f1 - f0>([&](auto ic) {
            constexpr int f = f0 + decltype(ic)::value, db = f % kDB, sk = f / kDB;
            if constexpr (f == at) mid();
            if constexpr (kAbl & 2) {
                asm volatile("" ::"v"(frag[f % kRing]));
#pragma unroll
                for (int x = 0; x < X; ++x) asm volatile("" ::"v"(pk[x][sk]));
            } else {
#pragma unroll
                for (int x = 0; x < X; ++x) o[x][db] = mfma(frag[f % kRing], pk[x][sk], o[x][db]);
            }
            read_v(std::integral_constant<int, f + kAhead>{});
        });
    };
    const std::integral_constant<int, 0> i0{};
    const std::integral_constant<int, 2> i2{};
    const std::integral_constant<int, 4> i4{};
    const std::integral_constant<int, 2 * kDB> iF{};
    const std::integral_constant<int, kDB> iH{};
    const std::integral_constant<int, -1> iN{};

    // ---- prologue: tile 0 in buffer 0 ----
    if constexpr (!(kAbl & 8)) {
        kst = buf_load16(rk, st_goff);
        vst = buf_load16(rv, st_goff);
        lds_write16(smem, k_lds, kst);
        lds_write16(smem, v_lds, vst);
        if constexpr (kStruct == 1) {   // tile 1 on its way
            kst = buf_load16(rk, kTile + st_goff);
            vst = buf_load16(rv, kTile + st_goff);
        }
    }
    __syncthreads();
    if constexpr (kDiag) last = stamp();

    if constexpr (kStruct == 0) {
        unsigned cur = 0u;
        for (int t = 0; t < ntiles; ++t) {
            const unsigned nxt = kBuf - cur;
            if constexpr (!(kAbl & 8)) {
                kst = buf_load16(rk, (unsigned)(t + 1) * kTile + st_goff);
                vst = buf_load16(rv, (unsigned)(t + 1) * kTile + st_goff);
            }
            LAB_STAMP(0)
            qk(cur);
            LAB_STAMP(1)
            if (t == 0) set_ref();
            softmax(i0, i4);
            LAB_STAMP(2)
            pv(cur, i0, iH, [] {}, iN);
            LAB_STAMP(3)
            if constexpr (!(kAbl & 8)) {
                lds_write16(smem, nxt + k_lds, kst);
                lds_write16(smem, nxt + v_lds, vst);
            }
            LAB_STAMP(4)
            pv(cur, iH, iF, [] {}, iN);
            LAB_STAMP(5)
            barrier();
            LAB_STAMP(6)
            cur = nxt;
        }
    } else {
        if (late) barrier();   // waves 4-7 sit out the first half-iteration
        if constexpr (kDiag) last = stamp();
        unsigned cur = 0u;
        for (int t = 0; t < ntiles; ++t) {
            const unsigned nxt = kBuf - cur;
            // ---- H1: K(t+1) -> LDS, K(t+2) on its way; S = K.Q^T; softmax of blocks 0,1 ----
            if constexpr (!(kAbl & 8)) {
                lds_write16(smem, nxt + k_lds, kst);
                kst = buf_load16(rk, (unsigned)(t + 2) * kTile + st_goff);
            }
            LAB_STAMP(0)
            qk(cur);
            LAB_STAMP(1)
            if (t == 0) set_ref();
            softmax(i0, i2);
            LAB_STAMP(2)
            barrier();
            LAB_STAMP(3)
            // ---- H2: V(t+1) -> LDS, V(t+2) on its way; softmax of blocks 2,3; O^T += V^T.P^T ----
            if constexpr (!(kAbl & 8)) {
                lds_write16(smem, nxt + v_lds, vst);
                vst = buf_load16(rv, (unsigned)(t + 2) * kTile + st_goff);
            }
            LAB_STAMP(4)
            softmax(i2, i4);
            LAB_STAMP(5)
            pv(cur, i0, iF, [] {}, iN);
            LAB_STAMP(6)
            barrier();
            LAB_STAMP(7)
            cur = nxt;
        }
        if (kSkew && !late) barrier();   // waves 0-3 sit out the last one
    }

    float l_row[X];
#pragma unroll
    for (int x = 0; x < X; ++x) l_row[x] = across_sum(l_part[x]);
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * 4u, (unsigned)(head_elems * 4u));
#pragma unroll
    for (int x = 0; x < X; ++x) {
        const float inv = 1.0f / l_row[x];
        const unsigned row = q_row0 + 16u * x;
#pragma unroll
        for (int db = 0; db < kDB; ++db) {
            const unsigned col = 16u * db + 4u * g;
            const f32x4 v = {o[x][db][0] * inv, o[x][db][1] * inv, o[x][db][2] * inv, o[x][db][3] * inv};
            buf_store16(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
        }
    }
    }   // persistent loop
    if constexpr (kDiag) {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) diag[((size_t)blockIdx.x * kW + wave) * 8 + i] = acc[i];
        }
    }
#undef LAB_STAMP
}

template <int kStruct, bool kSkew, int kAbl, bool kDiag>
static hipError_t launch(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale,
                         unsigned long long* diag, hipStream_t stream)
{
    constexpr int lds_bytes = 4 * kBlockN * D * 2;
    constexpr int kRows = 16 * X * kW;
    const int nqb = (N + kRows - 1) / kRows;
    const long long nwg = (long long)BH * nqb;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = nwg > cus ? (unsigned)cus : (unsigned)nwg;
    FA_LAUNCH((lab_w64x_kernel<kStruct, kSkew, kAbl, kDiag>), dim3(grid), dim3(64 * kW), lds_bytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K), static_cast<const uint16_t*>(V),
                       static_cast<float*>(O), N, nqb, scale * kLog2e, (unsigned)nwg, diag);
    return launch_status();
}

}  // namespace lab

// variant = 100*struct(+skew: 2) + (diag ? 50 : 0) ... decoded below; fp16, d = 64, fp32 out only
hipError_t lab_w64x_dispatch(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale,
                             int kstruct, int abl, unsigned long long* diag, hipStream_t stream)
{
    using namespace lab;
    if (N <= 0 || BH <= 0 || (unsigned long long)(N + 512) * 64ull * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
#define LAB_CASE(S, SK, A)                                                                             \
    if (kstruct == (S) + 2 * (SK) && abl == (A))                                                       \
        return diag ? launch<S == 0 ? 0 : 1, SK != 0, A, true>(Q, K, V, O, BH, N, scale, diag, stream)    \
                    : launch<S == 0 ? 0 : 1, SK != 0, A, false>(Q, K, V, O, BH, N, scale, diag, stream);
    LAB_CASE(0, 0, 0) LAB_CASE(1, 0, 0) LAB_CASE(1, 1, 0)
    LAB_CASE(0, 0, 1) LAB_CASE(0, 0, 2) LAB_CASE(0, 0, 4) LAB_CASE(0, 0, 8) LAB_CASE(0, 0, 16)
    LAB_CASE(0, 0, 3) LAB_CASE(0, 0, 5) LAB_CASE(0, 0, 6) LAB_CASE(0, 0, 7) LAB_CASE(0, 0, 9) LAB_CASE(0, 0, 11) LAB_CASE(0, 0, 13) LAB_CASE(0, 0, 14)
    LAB_CASE(0, 0, 15) LAB_CASE(0, 0, 31)
    LAB_CASE(1, 1, 1) LAB_CASE(1, 1, 2) LAB_CASE(1, 1, 4) LAB_CASE(1, 1, 8)
#undef LAB_CASE
    return hipErrorInvalidValue;
}

}  // namespace fa
