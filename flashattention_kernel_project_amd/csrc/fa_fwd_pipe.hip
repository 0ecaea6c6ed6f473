// fa_fwd_pipe.hip -- software-pipelined tiled attention forward for gfx950 (MI355X / CDNA4).
//
// Same data layout, MFMA orientation and LDS images as fa_fwd_kernels.hip (see the header there);
// what changes is the schedule.  At d=64 the softmax VALU stream (v_fma + v_exp + v_add + v_cvt +
// v_max3 per score) costs MORE issue cycles than the two MFMA products, so a wave that runs
// QK^T -> softmax -> PV back to back leaves the matrix pipe idle for most of each tile.  Here each
// wave overlaps the two pipes inside its own instruction stream:
//
//   iteration t:   QK^T(t+1)  MFMA   ||   p(t) = 2^(c*S(t) - m)  + pack      VALU   (phase 1)
//                  PV(t)      MFMA   ||   row-sum p(t), row-max S(t+1)       VALU   (phase 2)
//
// i.e. the raw scores of the NEXT tile are produced while the current tile is exponentiated
// (two score register sets, statically named so nothing is indexed at run time).  K therefore
// runs one tile further ahead than V: K(t+2) and V(t+1) are fetched HBM/L2 -> registers at the top
// of iteration t, written to LDS at its end, one barrier per tile; K and V each have a 2-deep ring.
// This is the role the reference gives its loader warp + cp.async ping-pong
// (flashattn_streaming_16x16_mw_v10.cu:156-195, v5_cp_async.cu:221-256), re-derived for a machine
// where every wave has to feed both the matrix and the vector pipe.
#include "fa_tile.hpp"

#include <type_traits>

namespace fa {

template <typename T, int D, bool kOutF32>
__global__ __launch_bounds__(64 * kWaves, 2)
void fa_fwd_pipe_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                        const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                        int N, int nqb, float scale_log2e)
{
    using G = TileGeom<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0][K1][V0][V1]

    // ---- block -> (head, query block): blocks that share K/V sit on one XCD, consecutively ----
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

    const unsigned q_row = qb * kBlockM + wave * 32u + r;

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
    unsigned g_off[G::kLoads], k_lds[G::kLoads], v_lds[G::kLoads];
#pragma unroll
    for (int p = 0; p < G::kLoads; ++p) {
        const unsigned idx = tid + p * 64u * kWaves;
        const unsigned row = idx / G::kChunks, ch = idx % G::kChunks;
        g_off[p] = row * G::kRowBytes + ch * 16u;
        k_lds[p] = G::k_off(row, ch);
        v_lds[p] = 2u * G::kTileBytes + G::v_off(row, ch);
    }
    u32x4 kst[G::kLoads], vst[G::kLoads];
    auto load_k = [&](int tile) {
#pragma unroll
        for (int p = 0; p < G::kLoads; ++p) kst[p] = buf_load16(rk, (unsigned)tile * kBlockN * G::kRowBytes + g_off[p]);
    };
    auto load_v = [&](int tile) {
#pragma unroll
        for (int p = 0; p < G::kLoads; ++p) vst[p] = buf_load16(rv, (unsigned)tile * kBlockN * G::kRowBytes + g_off[p]);
    };
    auto write_k = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < G::kLoads; ++p) lds_write16(smem, buf * G::kTileBytes + k_lds[p], kst[p]);
    };
    auto write_v = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < G::kLoads; ++p) lds_write16(smem, buf * G::kTileBytes + v_lds[p], vst[p]);
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
    float l_part = 0.0f;   // this half-wave's share of the row sum

    const int ntiles = (N + kBlockN - 1) / kBlockN;
    const bool partial = (N % kBlockN) != 0;

    auto qk = [&](unsigned buf, f32x16 (&s)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < G::kKSteps; ++ks) {
                const u32x4 kf = lds_read16(smem, buf * G::kTileBytes + kb * 32u * G::kRowBytes + k_rd_row +
                                                      (((2u * ks + h) ^ k_rd_swz) << 4));
                s[kb] = T::mfma32(kf, qf[ks], ks == 0 ? zero16 : s[kb]);
            }
    };
    auto mask_tail = [&](int tile, f32x16 (&s)[2]) {   // keys >= N -> -inf (p = 0)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = tile * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                if (key >= N) s[kb][i] = -INFINITY;
            }
    };
    auto tile_max = [&](const f32x16 (&s)[2]) -> float {
        float t0 = max3(s[0][0], s[0][1], s[0][2]);
        float t1 = max3(s[1][0], s[1][1], s[1][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) {
            t0 = max3(t0, s[0][i], s[0][i + 1]);
            t1 = max3(t1, s[1][i], s[1][i + 1]);
        }
        return max3(t0, t1, fmaxf(s[0][15], s[1][15]));
    };

    // One iteration: scores of tile t are in s_cur (raw); if kHasNext, scores of tile t+1 are
    // produced into s_nxt.  On exit m_ref covers tile t+1.
    auto iter = [&](auto has_next_c, int t, bool mask_next, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
        constexpr bool kHasNext = decltype(has_next_c)::value;
        const bool more_k = t + 2 < ntiles;
        if (more_k) load_k(t + 2);
        if constexpr (kHasNext) load_v(t + 1);

        // ---- phase 1: QK^T(t+1) on the matrix pipe, exp/pack of tile t on the vector pipe ----
        if constexpr (kHasNext) qk((unsigned)(t + 1) & 1u, s_nxt);
        u32x4 pk[4];
        const float neg_m = -m_ref;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s_cur[kb][i] = fast_exp2(__builtin_fmaf(s_cur[kb][i], c, neg_m));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    pk[kb * 2 + s2][w] = T::pack2(s_cur[kb][8 * s2 + 2 * w], s_cur[kb][8 * s2 + 2 * w + 1]);
        }

        // ---- phase 2: PV(t) on the matrix pipe, row sums of tile t / row max of tile t+1 -----
        const unsigned vbuf = (unsigned)t & 1u;
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
                o[db] = T::mfma32(vf, pk[ks], o[db]);
            }
        float ls0 = 0.0f, ls1 = 0.0f, ls2 = 0.0f, ls3 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            ls0 += s_cur[0][i];
            ls1 += s_cur[0][i + 1];
            ls2 += s_cur[1][i];
            ls3 += s_cur[1][i + 1];
        }
        l_part += (ls0 + ls1) + (ls2 + ls3);

        if constexpr (kHasNext) {
            if (mask_next) mask_tail(t + 1, s_nxt);
            const float tmax = tile_max(s_nxt) * c;
            if (__any(tmax - m_ref > kThr)) {   // rare: raise the reference max, rescale O and l
                const float mx = fmaxf(tmax, swap_halves(tmax));
                const float m_new = fmaxf(mx, m_ref);
                const float alpha = fast_exp2(m_ref - m_new);
                m_ref = m_new;
#pragma unroll
                for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
                l_part *= alpha;
            }
        }

        if (more_k) write_k((unsigned)t & 1u);
        if constexpr (kHasNext) write_v((unsigned)(t + 1) & 1u);
        __syncthreads();
    };

    // ---- prologue: K(0), V(0), K(1) into LDS; S(0); exact row max of tile 0 ---------------------
    f32x16 sA[2], sB[2];
    load_k(0);
    load_v(0);
    write_k(0);
    write_v(0);
    if (ntiles > 1) {
        load_k(1);
        write_k(1);
    }
    __syncthreads();
    qk(0, sA);
    if (ntiles == 1 && partial) mask_tail(0, sA);
    {
        const float tmax = tile_max(sA) * c;
        m_ref = fmaxf(tmax, swap_halves(tmax));
    }

    // ---- main loop, two tiles per trip so the two score sets swap names, not registers ----------
    const int n_plain = partial ? ntiles - 2 : ntiles - 1;   // iterations whose next tile is full
    int t = 0;
    for (; t + 1 < n_plain; t += 2) {
        iter(std::true_type{}, t, false, sA, sB);
        iter(std::true_type{}, t + 1, false, sB, sA);
    }
    // ---- up to three leftover iterations (odd count, masked next tile, last tile) ---------------
    for (; t < ntiles; ++t) {
        if (t + 1 < ntiles) {
            iter(std::true_type{}, t, partial && (t + 2 == ntiles), sA, sB);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sA[kb] = sB[kb];
        } else {
            iter(std::false_type{}, t, false, sA, sB);
        }
    }

    // ---- normalise and store: lane holds O[q_row][db*32 + 8g + 4h + 0..3] in o[db][4g..4g+3] ---
    const float l = l_part + swap_halves(l_part);
    const float inv = 1.0f / l;
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

template <typename T, int D, bool kOutF32>
static hipError_t launch_pipe(const void* Q, const void* K, const void* V, void* O,
                              int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    auto kern = fa_fwd_pipe_kernel<T, D, kOutF32>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes);
    if (e != hipSuccess) return e;
    const int nqb = (N + kBlockM - 1) / kBlockM;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * kWaves), G::kLdsBytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e);
    return hipGetLastError();
}

// D = 64 only for now (the two score sets do not fit next to a 128-wide O^T without spilling).
hipError_t pipe_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                         hipStream_t stream)
{
    if (D != 64) return hipErrorInvalidValue;
    if (in_dtype == 0)
        return out_dtype == 0 ? launch_pipe<F16, 64, true>(Q, K, V, O, BH, N, scale, stream)
                              : launch_pipe<F16, 64, false>(Q, K, V, O, BH, N, scale, stream);
    return out_dtype == 0 ? launch_pipe<BF16, 64, true>(Q, K, V, O, BH, N, scale, stream)
                          : launch_pipe<BF16, 64, false>(Q, K, V, O, BH, N, scale, stream);
}

}  // namespace fa
