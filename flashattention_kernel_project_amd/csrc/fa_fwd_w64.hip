// fa_fwd_w64.hip -- attention forward with 64 query rows per wave, 512-row workgroups (gfx950, d = 64).
//
// The d=64 forward runs at the package power cap (1 380 W, sclk 1.85 GHz), so wall
// time is set by joules per launch, not by cycles.  Sustained-load ablations of the 32-row kernel put
// the energy at roughly 55 % matrix work, 25 % softmax VALU, 10 % LDS operand reads and 10 % K/V
// staging (L2 -> VGPR -> LDS).  The last two are per-wave and per-workgroup overheads of the
// DECOMPOSITION, not of the arithmetic: here every wave owns two 32-row query blocks, so each K and
// V^T fragment read from LDS feeds two MFMAs, and a workgroup of 8 waves covers 512 query rows, so
// each staged K/V tile serves twice the rows -- both overheads halve.  With a single score set per
// wave there is no room (or need: the clock, not the issue stream, is the limit) for the software
// pipeline of fa_fwd_il.hip; the per-tile stream is QK^T (16 MFMA) -> softmax -> PV (16 MFMA).
//
// Everything else is as in fa_fwd_il.hip: S^T = K.Q^T orientation, packed P straight from the
// accumulator registers, optimistic pass without per-tile row max + tracked re-run on overflow,
// fp32 row sums by v_add, packed fma, persistent grid, XCD-aware item order.
#include "fa_tile.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

namespace w64 {
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#ifndef FA_W64_DOT2
#define FA_W64_DOT2 1   // row sums over the ROUNDED p, one v_dot2c per packed pair (bf16: needed for accuracy, fa_common.hpp;
                        // fp16: -1.1 % wall in THIS kernel's separate softmax phase, +2.6 % in the interleaved kernel)
#endif
#ifndef FA_W64_BARRIER_EVERY
#define FA_W64_BARRIER_EVERY 1   // tiles per workgroup barrier (measured: 1 and 2 tie within 0.3 %): 2 = ring of four K/V buffers, tile t+2 staged during
                                 // tile t, one barrier per two tiles (waves drift by up to a tile); 1 = two buffers, one per tile
#endif
#ifndef FA_W64_KV_AUX
#define FA_W64_KV_AUX 0   // cache-policy bits of the K/V staging loads (1 sc0, 2 nt, 16 sc1)
#endif
#ifndef FA_W64_FENCE
#define FA_W64_FENCE 0   // 1: scheduling fence after every {MFMAs of a fragment; read of the fragment kAhead further}
#endif
#ifndef FA_W64_ALT
#define FA_W64_ALT 1   // 1: consecutive MFMAs alternate accumulators (-1.6 % d=64, -2.1 % d=128 against 0); (key blocks in QK^T, head-dim blocks in PV) instead of running each chain to its end
#endif
#ifndef FA_W64_NT
#define FA_W64_NT 0   // 1: non-temporal O stores, 2: also non-temporal Q loads
#endif
#ifndef FA_W64_YOUNG_PRIO
#define FA_W64_YOUNG_PRIO 0
#endif
#ifndef FA_W64_WAVES
#define FA_W64_WAVES 8
#endif
constexpr int kW = FA_W64_WAVES;   // waves per workgroup (4: two independent 256-row workgroups per CU)
#ifndef FA_W64_AHEAD
#define FA_W64_AHEAD 2
#endif
#ifndef FA_W64_STAGE_AT
#define FA_W64_STAGE_AT 2   // the staged tile is written to LDS in front of PV fragment FA_W64_STAGE_AT * kDBlocks
#endif
constexpr int kAhead = FA_W64_AHEAD;   // LDS fragment read-ahead
}  // namespace w64

// D = head dim (64 or 128); X = 32-row query blocks per wave (2 at D = 64; 1 at D = 128, where the
// 128-wide O^T leaves no room for a second block -- that instantiation is the plain tiled kernel plus
// the optimistic pass, packed fma, fragment read-ahead and persistent grid).
// kCausal: query row i attends to keys 0..i.  The workgroup stops at the tile that holds its last row's
// diagonal, a wave skips (but still stages and synchronises) tiles wholly above its rows, crossed
// tiles get the element mask.  Launched one workgroup per item, query blocks last-to-first within a
// head, so the hardware dispatcher balances the unequal items.
template <typename T, int D, int X, bool kOutF32, bool kCausal = false, int kW = w64::kW>
__global__ __launch_bounds__(64 * kW, 2)
void fa_fwd_w64_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                       const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                       int N, int nqb, float scale_log2e, unsigned total_wg)
{
    using namespace w64;
    using G = TileGeom<D>;
    constexpr int kAhead = w64::kAhead, kRing = kAhead + 1;
    constexpr int kRows = 32 * X * kW;                              // query rows per workgroup
    constexpr int kLoads = (kBlockN * G::kChunks) / (64 * kW);      // 16-B K (and V) chunks per thread and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][K tile][V tile]

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned r = lane & 31u, h = lane >> 5;
    const float c = fabsf(scale_log2e);
    const unsigned q_flip = scale_log2e < 0.0f ? 0x80008000u : 0u;
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    // staging: one 16-B chunk of K and one of V per thread and tile
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
    constexpr float kHeadroom = 4.0f;
    const std::true_type yes{};
    const std::false_type no{};

#if FA_W64_YOUNG_PRIO
    // the SIMD arbitrates its two waves by age: give the later-dispatched half a static priority
    if (wave >= kW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    const unsigned nwg = total_wg;
    for (unsigned bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
    if (bid != blockIdx.x) __syncthreads();
    const unsigned xq = nwg >> 3, xr = nwg & 7u, xcd = bid & 7u;
    const unsigned wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const unsigned bh = wgid / (unsigned)nqb;
    const unsigned qb = kCausal ? (unsigned)nqb - 1u - (wgid - bh * (unsigned)nqb) : wgid - bh * (unsigned)nqb;
    const size_t head_elems = (size_t)N * D;
    const unsigned head_bytes = (unsigned)(head_elems * 2);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, head_bytes);
    const unsigned wave_row0 = qb * kRows + wave * (32u * X);    // first query row of this wave
    const unsigned q_row0 = wave_row0 + r;   // row of query block 0; block x is 32x rows further
    int ntiles_wg = ntiles;
    if constexpr (kCausal) ntiles_wg = min(ntiles, (int)(min((unsigned)N - 1u, qb * kRows + kRows - 1u) / kBlockN) + 1);

    u32x4 qf[X][G::kKSteps];
#pragma unroll
    for (int x = 0; x < X; ++x)
#pragma unroll
        for (int s = 0; s < G::kKSteps; ++s) {
            u32x4 raw = FA_W64_NT >= 2 ? buf_load16_nt(rq, (q_row0 + 32u * x) * G::kRowBytes + (16u * s + 8u * h) * 2u)
                                       : buf_load16(rq, (q_row0 + 32u * x) * G::kRowBytes + (16u * s + 8u * h) * 2u);
#pragma unroll
            for (int w = 0; w < 4; ++w) raw[w] ^= q_flip;
            qf[x][s] = raw;
        }

    f32x16 o[X][G::kDBlocks];
    float m_ref[X] = {}, l_part[X] = {};
    u32x4 kst[kLoads], vst[kLoads];

    auto run = [&](auto track_c) __attribute__((always_inline)) {
        constexpr bool kTrack = decltype(track_c)::value;
#pragma unroll
        for (int x = 0; x < X; ++x) {
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db) o[x][db] = zero16;
            l_part[x] = 0.0f;
        }
        // ring of kSlotsR buffers, tile t in slot t % kSlotsR; tile t + kDist is fetched at the top of iteration t and
        // written half way through its PV; a barrier closes every kDist-th iteration:
        //   visibility: tile t+kDist is written in iteration t, first read in iteration t+kDist, a barrier lies between;
        //   reuse: its slot held tile t-kDist, last read in iteration t-kDist, and a barrier lies between as well.
        constexpr int kDist = FA_W64_BARRIER_EVERY, kSlotsR = 2 * kDist;
#pragma unroll
        for (int pt = 0; pt < kDist; ++pt) {
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                kst[p] = buf_load16(rk, (unsigned)pt * G::kTileBytes + st_goff[p]);
                vst[p] = buf_load16(rv, (unsigned)pt * G::kTileBytes + st_goff[p]);
            }
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                lds_write16(smem, (unsigned)pt * G::kBufBytes + k_lds[p], kst[p]);
                lds_write16(smem, (unsigned)pt * G::kBufBytes + v_lds[p], vst[p]);
            }
        }
        __syncthreads();

        for (int t = 0; t < ntiles_wg; ++t) {
            const unsigned cur = (unsigned)t % kSlotsR, land = (unsigned)(t + kDist) % kSlotsR;
            const bool sync = ((t + 1) % kDist) == 0;
            // next tile: tiles past the end read zeros through the buffer bounds, into the free buffer
#pragma unroll
            for (int p = 0; p < kLoads; ++p) {
                kst[p] = buf_load16_cp<FA_W64_KV_AUX>(rk, (unsigned)(t + kDist) * G::kTileBytes + st_goff[p]);
                vst[p] = buf_load16_cp<FA_W64_KV_AUX>(rv, (unsigned)(t + kDist) * G::kTileBytes + st_goff[p]);
            }

            // causal: tiles wholly above this wave's rows contribute nothing (wave-uniform; tile 0 never is)
            if (kCausal && (unsigned)(t * kBlockN) > wave_row0 + 32u * X - 1u) {
#pragma unroll
                for (int p = 0; p < kLoads; ++p) {
                    lds_write16(smem, land * G::kBufBytes + k_lds[p], kst[p]);
                    lds_write16(smem, land * G::kBufBytes + v_lds[p], vst[p]);
                }
                if (sync) __syncthreads();
                continue;
            }

            // ---- S^T = K.Q^T for both query blocks: each K fragment feeds two MFMAs ------------
            f32x16 s[X][2];
            u32x4 frag[kRing];
            auto read_k = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f < 2 * G::kKSteps) {
                    constexpr int kb = FA_W64_ALT ? f % 2 : f / G::kKSteps, ks = FA_W64_ALT ? f / 2 : f % G::kKSteps;
                    frag[f % kRing] = lds_read16(smem, cur * G::kBufBytes + kb * 32u * G::kRowBytes + k_rd_row +
                                                           (((2u * ks + h) ^ k_rd_swz) << 4));
                }
            };
            sfor<kAhead>([&](auto fc) { read_k(fc); });
            sfor<2 * G::kKSteps>([&](auto fc) {
                constexpr int f = decltype(fc)::value, kb = FA_W64_ALT ? f % 2 : f / G::kKSteps, ks = FA_W64_ALT ? f / 2 : f % G::kKSteps;
#pragma unroll
                for (int x = 0; x < X; ++x) s[x][kb] = T::mfma32(frag[f % kRing], qf[x][ks], ks == 0 ? zero16 : s[x][kb]);
                read_k(std::integral_constant<int, f + kAhead>{});
                if constexpr (FA_W64_FENCE) __builtin_amdgcn_sched_barrier(0);
            });

            if (partial && t + 1 == ntiles) {   // keys >= N -> -inf (p = 0)
#pragma unroll
                for (int x = 0; x < X; ++x)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = t * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                            if (key >= N) s[x][kb][i] = -INFINITY;
                        }
            }

            if constexpr (kCausal) {   // tiles the diagonal crosses: keys after the query -> -inf
                if ((unsigned)(t * kBlockN) + (unsigned)kBlockN - 1u > wave_row0) {
#pragma unroll
                    for (int x = 0; x < X; ++x)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const unsigned key = (unsigned)(t * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2)) + 4u * h;
                                if (key > q_row0 + 32u * x) s[x][kb][i] = -INFINITY;
                            }
                }
            }

            // ---- reference max: tile 0 always; later tiles only in the tracked (fallback) pass ----
            if (kTrack || t == 0) {
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    float tmax = -INFINITY;
#pragma unroll
                    for (int e = 0; e < 32; e += 2) tmax = max3(tmax, s[x][e >> 4][e & 15], s[x][(e + 1) >> 4][(e + 1) & 15]);
                    tmax *= c;
                    if (t == 0) {
                        m_ref[x] = fmaxf(tmax, swap_halves(tmax)) + (kTrack ? 0.0f : kHeadroom);
                    } else if (__any(tmax - m_ref[x] > kThr)) {
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

            // ---- P = 2^(c*S - m), row sums (fp32), packed to 16 bit ----------------------------------
            u32x4 pk[X][4];
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
                        const float p0 = fast_exp2(v[0]), p1 = fast_exp2(v[1]);
                        pk[x][q4][w] = T::pack2(p0, p1);
                        if constexpr (T::kSumRounded || FA_W64_DOT2) {
                            if (w & 1) ls1 = T::sum2(pk[x][q4][w], ls1);
                            else ls0 = T::sum2(pk[x][q4][w], ls0);
                        } else {
                            ls0 += p0;
                            ls1 += p1;
                        }
                    }
                }
                l_part[x] += ls0 + ls1;
            }

            // ---- O^T += V^T.P^T for both query blocks: each V^T fragment feeds two MFMAs ----------
            auto read_v = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f < 4 * G::kDBlocks) {
                    constexpr int db = FA_W64_ALT ? f % G::kDBlocks : f / 4, ks = FA_W64_ALT ? f / G::kDBlocks : f % 4;
                    u32x4 vf;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const u32x2 half = lds_read_tr8(
                            smem, cur * G::kBufBytes + v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                        vf[2 * jj] = half[0];
                        vf[2 * jj + 1] = half[1];
                    }
                    frag[f % kRing] = vf;
                }
            };
            sfor<kAhead>([&](auto fc) { read_v(fc); });
            sfor<4 * G::kDBlocks>([&](auto fc) {
                constexpr int f = decltype(fc)::value, db = FA_W64_ALT ? f % G::kDBlocks : f / 4, ks = FA_W64_ALT ? f / G::kDBlocks : f % 4;
                if constexpr (f == FA_W64_STAGE_AT * G::kDBlocks) {   // land the next tile in the other buffer (half way through PV)
#pragma unroll
                    for (int p = 0; p < kLoads; ++p) {
                        lds_write16(smem, land * G::kBufBytes + k_lds[p], kst[p]);
                        lds_write16(smem, land * G::kBufBytes + v_lds[p], vst[p]);
                    }
                }
#pragma unroll
                for (int x = 0; x < X; ++x) o[x][db] = T::mfma32(frag[f % kRing], pk[x][ks], o[x][db]);
                read_v(std::integral_constant<int, f + kAhead>{});
                if constexpr (FA_W64_FENCE) __builtin_amdgcn_sched_barrier(0);
            });
            if (sync) __syncthreads();
        }
        if (kDist > 1) __syncthreads();   // the next pass / item rewrites the ring: everybody is done reading
    };

    run(no);
    float l_row[X];
    bool bad = false;
    // a packed p can only have overflowed if the fp32 row sum reached the 16-bit format's range
    const float lim = T::id == 1 ? 0x1p+96f : 60000.0f;
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
                    if constexpr (FA_W64_NT >= 1) buf_store16_nt(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
                    else buf_store16(ro, (row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
                } else {
                    const u32x2 v = {T::pack2(a, b), T::pack2(cc, d)};
                    if constexpr (FA_W64_NT >= 1) buf_store8_nt(ro, (row * D + col) * 2u, v);
                    else buf_store8(ro, (row * D + col) * 2u, v);
                }
            }
    }
    }   // persistent loop over work items
}

template <typename T, int D, int X, bool kOutF32, bool kCausal = false, int kW = w64::kW>
static hipError_t launch_w64(const void* Q, const void* K, const void* V, void* O,
                             int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    constexpr int kRows = 32 * X * kW;
    const int nqb = (N + kRows - 1) / kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int grid_cap = device_cus();
    const long long cap = (long long)grid_cap * (8 / kW);
    const unsigned grid = (nwg > cap && !kCausal) ? (unsigned)cap : (unsigned)nwg;
    const hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(&fa_fwd_w64_kernel<T, D, X, kOutF32, kCausal, kW>),
                                           2 * FA_W64_BARRIER_EVERY * G::kBufBytes);
    if (attr != hipSuccess) return attr;
    FA_LAUNCH((fa_fwd_w64_kernel<T, D, X, kOutF32, kCausal, kW>), dim3(grid), dim3(64 * kW), 2 * FA_W64_BARRIER_EVERY * G::kBufBytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e, (unsigned)nwg);
    return launch_status();
}

template <bool kCausal>
static hipError_t w64_dispatch_impl(const void* Q, const void* K, const void* V, void* O,
                                    int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                                    hipStream_t stream)
{
    if (D != 64 && D != 128) return hipErrorInvalidValue;
    if ((unsigned long long)(N + 64 * w64::kW) * (unsigned)D * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (D == 64) {
        // under the mask: one 32-row block per wave (256-row workgroups): twice the items to balance, a
        // finer diagonal, and no spills (the two-block causal instantiation is 12 VGPRs over budget)
#ifndef FA_W64_SMALL
#define FA_W64_SMALL 0   // experiment: the 128-row (X = 1, 4 waves) shape without the mask, for small grids
#endif
        constexpr int X = (kCausal || FA_W64_SMALL) ? 1 : 2;
        constexpr int W = (kCausal || FA_W64_SMALL) ? 4 : w64::kW;   // under the mask also 4 waves: 128-row workgroups, two per CU
        if (in_dtype == 0)
            return out_dtype == 0 ? launch_w64<F16, 64, X, true, kCausal, W>(Q, K, V, O, BH, N, scale, stream)
                                  : launch_w64<F16, 64, X, false, kCausal, W>(Q, K, V, O, BH, N, scale, stream);
        return out_dtype == 0 ? launch_w64<BF16, 64, X, true, kCausal, W>(Q, K, V, O, BH, N, scale, stream)
                              : launch_w64<BF16, 64, X, false, kCausal, W>(Q, K, V, O, BH, N, scale, stream);
    }
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_w64<F16, 128, 1, true, kCausal>(Q, K, V, O, BH, N, scale, stream)
                              : launch_w64<F16, 128, 1, false, kCausal>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_w64<BF16, 128, 1, true, kCausal>(Q, K, V, O, BH, N, scale, stream)
                          : launch_w64<BF16, 128, 1, false, kCausal>(Q, K, V, O, BH, N, scale, stream);
}

hipError_t w64_dispatch(const void* Q, const void* K, const void* V, void* O,
                        int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                        hipStream_t stream)
{
    return w64_dispatch_impl<false>(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);
}

hipError_t w64_causal_dispatch(const void* Q, const void* K, const void* V, void* O,
                               int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                               hipStream_t stream)
{
    return w64_dispatch_impl<true>(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);
}

}  // namespace fa
