// fa_fwd_kernels.hip -- general-shape FlashAttention forward for gfx950 (MI355X / CDNA4).
//
// Replaces the reference's general-shape family
//   flashattn_forward_wmma_kernel{,_v2,_v3,_v4}   FlashAttention/flashattn_forward_wmma/*.cu:49-346
//   flashattn_forward_wmma_v5_cp_async            FlashAttention/flashattn_forward_memory_bound/*v5_cp_async.cu:99
// (signature (Q,K,V,O,BH,N,D,scale); Q,K,V,O [BH,N,D] row-major; self-attention; grid over
// (query block, BH)) with a design made for 64-lane wavefronts and MFMA, not a WMMA translation:
//
//   * one workgroup = 8 waves = 256 query rows of one (batch,head); each wave owns 32 rows.
//   * S^T = K.Q^T with v_mfma_f32_32x32x16 (K tile is the A operand from LDS, Q^T the B operand
//     held in registers for the whole kernel).  The accumulator then has the QUERY on the lane
//     and the KEYS in the 16 registers: the online-softmax row statistics are in-lane, no
//     shuffles per tile, and the same registers (packed to 16 bit) are directly the B operand of
//     O^T += V^T.P^T -- P never goes through LDS or cross-lane moves.
//   * V^T fragments come from a row-major V tile through ds_read_b64_tr_b16 (hardware transpose);
//     the k-order permutation the accumulator imposes is absorbed by which 4-key groups each
//     half-wave reads.
//   * running max: p = 2^(c*S - m_ref) is one v_fma + one v_exp per element (c = scale*log2(e),
//     applied in fp32 to the exact fp32 QK^T sums, as the reference's `acc * scale` does -- folding
//     c into the 16-bit Q would perturb large logits by |s|*2^-11).  The reference max m_ref of a
//     row is only raised when a tile exceeds it by more than 2^kThr (wave-uniform rare branch):
//     the two-term (alpha,beta) renormalisation of flashattn_streaming_16x16_mw.cu:200-229 / the
//     FA-2 form of v12f.cu:193-220, applied lazily.
//   * K/V tiles (64 keys) are staged HBM/L2 -> registers -> LDS with the load issued a tile ahead
//     (replaces the reference's cp.async ping-pong, v5_cp_async.cu:221-256), double-buffered,
//     one barrier per tile; K image XOR-swizzled for ds_read_b128, V image laid out in 256-B
//     [4 keys][32 cols] blocks so each half-wave's transposed read covers all 64 banks once.
//   * buffer_load/buffer_store with a per-head resource descriptor give N-tail handling for free
//     (rows >= N read 0 / are not stored); keys >= N are masked to -inf in the last tile only.
#include "fa_tile.hpp"

namespace fa {

// W = waves per workgroup (32 query rows each), kOcc = workgroups' waves per SIMD the register
// budget is held to (W = 8, kOcc = 2: one 256-row workgroup per CU; W = 4, kOcc = 3: three
// independent 128-row workgroups per CU, so the waves sharing a SIMD are never barrier-coupled).
// kCausal: query row i attends to keys 0..i only.  A workgroup stops at the tile that holds its last
// row's diagonal, a wave skips (but still stages and synchronises) tiles that lie wholly above its 32
// rows, and the tiles the diagonal crosses get a per-element mask; query blocks are walked last to
// first so that the longest ones start first.
template <typename T, int D, bool kOutF32, int W = kWaves, int kOcc = 2, bool kCausal = false>
__global__ __launch_bounds__(64 * W, kOcc)
void fa_fwd_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                   const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                   int N, int nqb, float scale_log2e)
{
    using G = TileGeom<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- block -> (head, query block): blocks that share K/V sit on one XCD, consecutively ----
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned xq = nwg >> 3, xr = nwg & 7u, xcd = bid & 7u;
    const unsigned wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const unsigned bh = wgid / (unsigned)nqb;
    const unsigned qb = kCausal ? (unsigned)nqb - 1u - (wgid - bh * (unsigned)nqb) : wgid - bh * (unsigned)nqb;

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned r = lane & 31u, h = lane >> 5;

    const size_t head_elems = (size_t)N * D;
    const unsigned head_bytes = (unsigned)(head_elems * 2);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, head_bytes);

    constexpr int kLoadsW = (kBlockN * G::kChunks) / (64 * W);
    const unsigned wave_row0 = qb * (32u * W) + wave * 32u;   // first query row of this wave
    const unsigned q_row = wave_row0 + r;

    // ---- Q^T fragments (B operand of S^T = K.Q^T), resident for the whole kernel ---------------
    // c = |scale|*log2(e) is applied to the fp32 scores; a negative scale flips Q's sign bits so
    // that the row max of c*S is always c*max(S).
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

    // ---- staging: each thread moves kLoads 16-B chunks of K and of V per tile -----------------
    unsigned g_off[kLoadsW], k_lds[kLoadsW], v_lds[kLoadsW];
#pragma unroll
    for (int p = 0; p < kLoadsW; ++p) {
        const unsigned idx = tid + p * 64u * W;
        const unsigned row = idx / G::kChunks, c = idx % G::kChunks;
        g_off[p] = row * G::kRowBytes + c * 16u;
        k_lds[p] = G::k_off(row, c);
        v_lds[p] = G::kTileBytes + G::v_off(row, c);
    }
    u32x4 kst[kLoadsW], vst[kLoadsW];
    auto stage_load = [&](unsigned kv0) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
            kst[p] = buf_load16(rk, kv0 * G::kRowBytes + g_off[p]);
            vst[p] = buf_load16(rv, kv0 * G::kRowBytes + g_off[p]);
        }
    };
    auto stage_write = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
            lds_write16(smem, buf * G::kBufBytes + k_lds[p], kst[p]);
            lds_write16(smem, buf * G::kBufBytes + v_lds[p], vst[p]);
        }
    };

    // ---- per-lane LDS read addresses -----------------------------------------------------------
    // K (A operand of QK^T): lane (r,h) reads row kb*32+r, chunk 2s+h.  The swizzle term depends
    // only on r for both key blocks.
    const unsigned k_rd_row = r * G::kRowBytes;
    const unsigned k_rd_swz = G::k_swz(r);
    // V^T (A operand of PV): 16-lane group g covers 16 d-columns; lane 4q+p of the group supplies
    // row q (key), columns 4p..4p+3.  Half-wave h takes the key groups 4h..4h+3 of every 8.
    const unsigned i16 = lane & 15u, vq = i16 >> 2, vp = i16 & 3u, vg = (lane >> 4) & 1u;
    unsigned v_rd[2];   // per parity of the column block (row-slot rotation)
#pragma unroll
    for (int par = 0; par < 2; ++par)
        v_rd[par] = G::kTileBytes + h * G::kDBlocks * 256u + ((vq ^ par) << 6) + vg * 32u + vp * 8u;

    // ---- running state -------------------------------------------------------------------------
    f32x16 o[G::kDBlocks];
#pragma unroll
    for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.0f;
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.0f;
    float m_ref = 0.0f;    // reference max of this lane's query row, in log2 units (c*S)
    float l_part = 0.0f;   // this half-wave's share of the row sum

    int ntiles = (N + kBlockN - 1) / kBlockN;
    if constexpr (kCausal) {   // tiles up to the diagonal of the workgroup's last (existing) row
        const unsigned last_row = min((unsigned)N - 1u, qb * (32u * W) + 32u * W - 1u);
        ntiles = min(ntiles, (int)(last_row / kBlockN) + 1);
    }

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned cur = t & 1u;
        const char* kbuf = smem + cur * G::kBufBytes;
        if (t + 1 < ntiles) stage_load((t + 1) * kBlockN);
        // causal: tiles wholly above this wave's rows contribute nothing (wave-uniform; tile 0 never is)
        if (!kCausal || (unsigned)(t * kBlockN) <= wave_row0 + 31u) {

        // ---- S^T = K.Q^T (raw fp32 scores) ------------------------------------------------
        f32x16 s[2];
        // consecutive MFMAs alternate accumulators (a dependent MFMA issued right behind its producer stalls)
#pragma unroll
        for (int ks = 0; ks < G::kKSteps; ++ks) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const u32x4 kf = lds_read16(kbuf, kb * 32u * G::kRowBytes + k_rd_row +
                                                      (((2u * ks + h) ^ k_rd_swz) << 4));
                s[kb] = T::mfma32(kf, qf[ks], ks == 0 ? zero16 : s[kb]);
            }
        }

        // ---- keys >= N (last tile only): -inf so that p = 0 --------------------------------
        if ((t + 1) * kBlockN > N) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = t * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                    if (key >= N) s[kb][i] = -INFINITY;
                }
        }
        if constexpr (kCausal) {   // tiles the diagonal crosses: keys after the query -> -inf
            if ((unsigned)(t * kBlockN) + (unsigned)kBlockN - 1u > wave_row0) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const unsigned key = (unsigned)(t * kBlockN + kb * 32 + (i & 3) + 8 * (i >> 2)) + 4u * h;
                        if (key > q_row) s[kb][i] = -INFINITY;
                    }
            }
        }

        // ---- tile max vs the reference max; raise the reference only when needed ----------------
        float tmax = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) tmax = max3(tmax, s[0][i], s[0][i + 1]);
        tmax = max3(tmax, s[0][15], s[1][0]);
#pragma unroll
        for (int i = 1; i < 15; i += 2) tmax = max3(tmax, s[1][i], s[1][i + 1]);
        tmax = fmaxf(tmax, s[1][15]) * c;   // log2 units

        if (t == 0 || __any(tmax - m_ref > kThr)) {
            const float mx = fmaxf(tmax, swap_halves(tmax));         // row max, same in both halves
            const float m_new = (t == 0) ? mx : fmaxf(mx, m_ref);
            const float alpha = (t == 0) ? 0.0f : fast_exp2(m_ref - m_new);
            m_ref = m_new;
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            l_part *= alpha;
        }

        // ---- p = 2^(c*S - m_ref), row-sum share, pack to 16 bit (B operand of PV) -----------
        u32x4 pk[4];
        float lsum0 = 0.0f, lsum1 = 0.0f;
        const float neg_m = -m_ref;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = fast_exp2(__builtin_fmaf(s[kb][i], c, neg_m));
            if constexpr (!T::kSumRounded) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    lsum0 += s[kb][i];
                    lsum1 += s[kb][i + 1];
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pk[kb * 2 + s2][w] = T::pack2(s[kb][8 * s2 + 2 * w], s[kb][8 * s2 + 2 * w + 1]);
                    if constexpr (T::kSumRounded) {
                        if (w & 1) lsum1 = T::sum2(pk[kb * 2 + s2][w], lsum1);
                        else lsum0 = T::sum2(pk[kb * 2 + s2][w], lsum0);
                    }
                }
        }
        l_part += lsum0 + lsum1;

        // ---- O^T += V^T.P^T -------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db) {
                u32x4 vf;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const u32x2 half = lds_read_tr8(
                        kbuf, v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                    vf[2 * jj] = half[0];
                    vf[2 * jj + 1] = half[1];
                }
                o[db] = T::mfma32(vf, pk[ks], o[db]);
            }
        }
        }   // wave has work in this tile

        if (t + 1 < ntiles) stage_write(cur ^ 1u);
        __syncthreads();
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
            const float c = o[db][4 * g + 2] * inv, d = o[db][4 * g + 3] * inv;
            if constexpr (kOutF32) {
                const f32x4 v = {a, b, c, d};
                buf_store16(ro, (q_row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
            } else {
                const u32x2 v = {T::pack2(a, b), T::pack2(c, d)};
                buf_store8(ro, (q_row * D + col) * 2u, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic fallback: any D % 16 == 0 (D <= 256), the "single 16x16 MFMA fragment" kernel.
// One wave = 16 query rows, 16 keys per step straight from global memory, v_mfma_f32_16x16x16.
// S^T = K.Q^T again puts the query on the lane (col = lane&15) and 4 keys in the registers of
// each of the four 16-lane groups; the row statistics are combined across the groups with two
// xor-shuffles.  This is the replacement for the reference's 1-warp kernels
// (flashattn_forward_wmma.cu:49-346, flashattn_fused_softmax_tensorcore_16x16.cu:41-136) and the
// correctness path for head dims the tiled kernel is not instantiated for.
// ---------------------------------------------------------------------------------------------
constexpr int kGenMaxD = 256;

template <typename T, bool kOutF32, bool kCausal = false>
__global__ __launch_bounds__(64)
void fa_fwd_generic_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                           const uint16_t* __restrict__ Vg, void* __restrict__ Og,
                           int N, int D, int nqb, float scale_log2e)
{
    const unsigned bh = blockIdx.x / (unsigned)nqb, qb = blockIdx.x % (unsigned)nqb;
    const unsigned lane = threadIdx.x & 63u, c16 = lane & 15u, g4 = lane >> 4;
    const size_t head_elems = (size_t)N * D;
    const unsigned head_bytes = (unsigned)(head_elems * 2);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, head_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, head_bytes);
    const unsigned q_row = qb * 16u + c16;
    const int nks = D / 16;      // k-steps over d for QK^T
    const int ndb = D / 16;      // 16-row blocks of O^T

    // B operand of S^T = K.Q^T: lane (c16,g4) holds Q[q_row][16s + 4g4 + 0..3].
    u32x2 qf[kGenMaxD / 16];
#pragma unroll
    for (int s = 0; s < kGenMaxD / 16; ++s)
        if (s < nks) qf[s] = buf_load8(rq, (q_row * D + 16u * s + 4u * g4) * 2u);
    f32x4 o[kGenMaxD / 16];
#pragma unroll
    for (int db = 0; db < kGenMaxD / 16; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l_part = 0.0f;

    const int kv_end = kCausal ? min(N, (int)(qb * 16u) + 16) : N;   // causal: up to the wave's last diagonal
    for (int kv0 = 0; kv0 < kv_end; kv0 += 16) {
        // A operand: lane (c16,g4) holds K[kv0 + c16][16s + 4g4 + 0..3]
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < kGenMaxD / 16; ++s) {
            if (s < nks) {
                const u32x2 kf = buf_load8(rk, ((kv0 + c16) * D + 16u * s + 4u * g4) * 2u);
                s4 = T::mfma16(kf, qf[s], s4);
            }
        }
        // s4[i] = S[q_row][kv0 + 4*g4 + i]; to the log2 domain in fp32
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s4[i] = (kv0 + 4 * (int)g4 + i >= N || (kCausal && (unsigned)(kv0 + 4 * (int)g4 + i) > q_row))
                        ? -INFINITY : s4[i] * scale_log2e;
        float tmax = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const float alpha = fast_exp2(m - m_new);   // m = -inf on the first tile -> 0
        m = m_new;
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = fast_exp2(s4[i] - m_new);
        l_part = l_part * alpha + (p[0] + p[1]) + (p[2] + p[3]);
        // B operand of O^T += V^T.P^T: lane (c16,g4) holds P^T[k = 4g4 + i][q = c16] = p[i]
        const u32x2 pf = {T::pack2(p[0], p[1]), T::pack2(p[2], p[3])};
#pragma unroll
        for (int db = 0; db < kGenMaxD / 16; ++db) {
            if (db < ndb) {
                // A operand: V^T[d = 16db + c16][k = 4g4 + i] = V[kv0 + 4g4 + i][16db + c16]
                uint16_t e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned off = ((kv0 + 4u * g4 + i) * D + 16u * db + c16) * 2u;
                    e[i] = __builtin_amdgcn_raw_buffer_load_b16(rv, off, 0, 0);
                }
                const u32x2 vf = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
#pragma unroll
                for (int i = 0; i < 4; ++i) o[db][i] *= alpha;
                o[db] = T::mfma16(vf, pf, o[db]);
            }
        }
    }
    float l = l_part + __shfl_xor(l_part, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    constexpr unsigned es = kOutF32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t ro =
        make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * head_elems * es, (unsigned)(head_elems * es));
    // o[db][i] = O[q_row][16db + 4g4 + i]
#pragma unroll
    for (int db = 0; db < kGenMaxD / 16; ++db) {
        if (db < ndb) {
            const unsigned col = 16u * db + 4u * g4;
            if constexpr (kOutF32) {
                const f32x4 v = {o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv};
                buf_store16(ro, (q_row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
            } else {
                const u32x2 v = {T::pack2(o[db][0] * inv, o[db][1] * inv), T::pack2(o[db][2] * inv, o[db][3] * inv)};
                buf_store8(ro, (q_row * D + col) * 2u, v);
            }
        }
    }
}

}  // namespace fa

// ---------------------------------------------------------------------------------------------
// host-side dispatch (C++ linkage; the C-ABI lives in fa_capi.hip)
// ---------------------------------------------------------------------------------------------
namespace fa {

template <typename T, int D, bool kOutF32, int W = kWaves, int kOcc = 2, bool kCausal = false>
static hipError_t launch_tiled(const void* Q, const void* K, const void* V, void* O,
                               int BH, int N, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    auto kern = fa_fwd_kernel<T, D, kOutF32, W, kOcc, kCausal>;
    const hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(kern), G::kLdsBytes);
    if (attr != hipSuccess) return attr;
    const int nqb = (N + 32 * W - 1) / (32 * W);
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    FA_LAUNCH(kern, dim3((unsigned)nwg), dim3(64 * W), G::kLdsBytes, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, nqb, scale * kLog2e);
    return launch_status();
}

template <typename T, bool kOutF32, bool kCausal = false>
static hipError_t launch_generic(const void* Q, const void* K, const void* V, void* O,
                                 int BH, int N, int D, float scale, hipStream_t stream)
{
    const int nqb = (N + 15) / 16;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    FA_LAUNCH((fa_fwd_generic_kernel<T, kOutF32, kCausal>), dim3((unsigned)nwg), dim3(64), 0, stream,
                       static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K),
                       static_cast<const uint16_t*>(V), O, N, D, nqb, scale * kLog2e);
    return launch_status();
}

template <typename T, bool kOutF32>
static hipError_t dispatch_d(const void* Q, const void* K, const void* V, void* O,
                             int BH, int N, int D, float scale, int algo, hipStream_t stream)
{
    if (algo != 1) {   // 0 = auto, 2 = force tiled
        if (D == 64)  return launch_tiled<T, 64, kOutF32>(Q, K, V, O, BH, N, scale, stream);
        if (D == 128) return launch_tiled<T, 128, kOutF32>(Q, K, V, O, BH, N, scale, stream);
        if (algo == 2) return hipErrorInvalidValue;
    }
    return launch_generic<T, kOutF32>(Q, K, V, O, BH, N, D, scale, stream);
}

template <typename T, bool kOutF32>
static hipError_t dispatch_causal_d(const void* Q, const void* K, const void* V, void* O,
                                    int BH, int N, int D, float scale, int algo, hipStream_t stream)
{
    // 128-row workgroups, two per CU: finer diagonal, prologues overlap the neighbour's main loop.  Measured
    // B8 H16 N4096 d64: 0.375 ms against 0.442 (256-row) -> AUTO at d=64; N8192 d128: 2.40 vs 2.31 ms -> not at d=128.
    if (algo == 6 || (algo == 0 && D == 64)) {
        if (D == 64)  return launch_tiled<T, 64, kOutF32, 4, 2, true>(Q, K, V, O, BH, N, scale, stream);
        if (D == 128) return launch_tiled<T, 128, kOutF32, 4, 2, true>(Q, K, V, O, BH, N, scale, stream);
        return hipErrorInvalidValue;
    }
    if (algo != 1) {
        if (D == 64)  return launch_tiled<T, 64, kOutF32, kWaves, 2, true>(Q, K, V, O, BH, N, scale, stream);
        if (D == 128) return launch_tiled<T, 128, kOutF32, kWaves, 2, true>(Q, K, V, O, BH, N, scale, stream);
        if (algo == 2) return hipErrorInvalidValue;
    }
    return launch_generic<T, kOutF32, true>(Q, K, V, O, BH, N, D, scale, stream);
}

hipError_t il_dispatch(const void* Q, const void* K, const void* V, void* O,
                       int BH, int N, int D, float scale, int in_dtype, int out_dtype, int waves,
                       hipStream_t stream);

hipError_t w64p_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                         hipStream_t stream);
hipError_t w64_dispatch(const void* Q, const void* K, const void* V, void* O,
                        int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                        hipStream_t stream);
hipError_t w64x_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                         hipStream_t stream);
hipError_t rp16_causal_dispatch(const void* Q, const void* K, const void* V, void* O,
                                int BH, int N, int D, float scale, int in_dtype, int out_dtype, hipStream_t stream);
hipError_t rp16_causal_dispatch_1w(const void* Q, const void* K, const void* V, void* O,
                                int BH, int N, int D, float scale, int in_dtype, int out_dtype, hipStream_t stream);
hipError_t rp16_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype, int fold,
                         hipStream_t stream);
hipError_t rp_dispatch(const void* Q, const void* K, const void* V, void* O,
                       int BH, int N, int D, float scale, int in_dtype, int out_dtype, int fold,
                       hipStream_t stream);
hipError_t sk_dispatch(const void* Q, const void* K, const void* V, void* O,
                       int BH, int N, int D, float scale, int in_dtype, int out_dtype, int variant,
                       hipStream_t stream);
hipError_t w64_causal_dispatch(const void* Q, const void* K, const void* V, void* O,
                               int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                               hipStream_t stream);
// AUTO: the explicit algo id a shape resolves to (one rule for the dispatcher and for fa_selected_kernel()).
//   d = 64, N > 256 and at least one 512-row workgroup per CU: the rolling half-tile pipeline on 16x16x32 with the folded
//     fast pass (fa_fwd_rp16.hip, 24), fp16 and bf16.  Round 2, one device, interleaved A/B, B8 H16 N4096, ms per launch:
//     fp16 0.489 (24) / 0.524 (22, the same on 32x32x16) / 0.526 (23, exact) / 0.541 (fa_fwd_w64x);
//     bf16 0.487 (24) / 0.502 (23) / 0.528 (21, fa_fwd_rp) / 0.533 (fa_fwd_w64);
//   d = 64, smaller grids or N <= 256: the interleaved kernel with 256-row workgroups, or 128-row ones (two per CU);
//   d = 128: the same pipeline with two 16-row blocks per wave (256-row workgroups), B8 H16 N8192: fp16 3.74 ms against 3.94
//     for fa_fwd_w64x, bf16 3.63 against 3.84 for fa_fwd_w64; from N = 4096 with one wave per SIMD (28);
//   anything else: the generic single-fragment kernel.
// The CU count is read from the current device per call.
int auto_algo(int BH, int N, int D, int in_dtype)
{
    if (D != 64 && D != 128) return 1;
    const long long cus = device_cus();
    if (D == 64 && N <= 256) {   // a handful of tiles per item: the 16x16x16 interleaved kernels' short prologue wins
        const long long nwg256 = (long long)BH * ((N + 255) / 256);
        return nwg256 >= 2 * cus ? 5 : 6;
    }
    // The rolling pipeline on the widest waves whose grid still covers the device: rounds of the persistent grid x rows per
    // workgroup / efficiency of that width (LDS fragment reuse: 1.0 / 0.9 / 0.7 for 64- / 32- / 16-row waves at d=64; measured on
    // B*H x N sweeps, tools/mid_grid_sweep.py -> profiles/r02_mid_grid.txt)
    const int wide = D == 64 ? 512 : 256;
    const int ids[3] = {24, 26, 27};
    const double eff[3] = {1.0, D == 64 ? 0.9 : 0.72, 0.7};   // (d=128's half width is the 16-row wave)
    int best = 24;
    double best_cost = 0.0;
    for (int w = 0; w < (D == 64 ? 3 : 2); ++w) {
        const int rows = wide >> w;
        const long long nwg = (long long)BH * ((N + rows - 1) / rows);
        const double cost = (double)((nwg + cus - 1) / cus) * rows / eff[w];
        if (w == 0 || cost < best_cost) { best = ids[w]; best_cost = cost; }
    }
    // 128-row workgroups on long sequences: the same rows as 32-row waves on half the keys each (key split, 29) -- every LDS
    // fragment feeds two matrix instructions; B1 H16 N2048 0.0268 vs 0.0280 ms, B2 H8 N4096 0.0821 vs 0.0882 (profiles/
    // r03_cfg3_variants.txt); at N = 1024 it loses 2 %
    if (best == 27 && D == 64 && N >= 2048 && N % 128 == 0) best = 29;
    // d = 128 from N = 4096: the same 256-row workgroups as FOUR 64-row waves, one per SIMD with the whole register file (28): every LDS
    // fragment feeds four matrix instructions instead of two (the two-wave kernel's fragment reads need the LDS's whole bandwidth).
    // Same device, interleaved (profiles/r03_d128_variants.txt, last table): B8 H16 N8192 3.49 vs 3.72 ms, B1 H32 N16384 3.40 vs 3.61,
    // B4 H16 N4096 0.500 vs 0.513, B1 H8 N8192 (one round) 0.250 vs 0.256; at N = 2048 they tie, N = 1024 loses 1-2 % (nothing hides the
    // prologue of a lone wave)
    if (best == 24 && D == 128 && N >= 4096) best = 28;
    return best;
}

// Name of the kernel template an explicit algo id launches (rocprofv3 kernel-trace names start with it).
const char* algo_kernel_name(int algo, int D)
{
    switch (algo) {
        case 1: return "fa::fa_fwd_generic_kernel";
        case 2: return (D == 64 || D == 128) ? "fa::fa_fwd_kernel" : "fa::fa_fwd_generic_kernel";
        case 5: case 6: return "fa::fa_fwd_il_kernel";
#ifdef FA_EXPERIMENTS
        case 13: return "fa::fa_fwd_w64_kernel";
        case 16: return "fa::fa_fwd_w64x_kernel";
        case 21: case 22: return "fa::fa_fwd_rp_kernel";
#endif
        case 23: case 24: case 25: case 26: case 27: case 28: case 29: return "fa::fa_fwd_rp16_kernel";
        default: return "";
    }
}

// algo: 0 auto, 1 generic single-fragment kernel, 2 tiled kernel (D in {64,128} only),
//       3 software-pipelined tiled kernel (D = 64 only), 4 ping-pong tiled kernel (D in {64,128})
hipError_t forward_dispatch(const void* Q, const void* K, const void* V, void* O,
                            int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                            int algo, hipStream_t stream)
{
    if (!Q || !K || !V || !O) return hipErrorInvalidValue;
    if (BH <= 0 || N <= 0 || D <= 0 || D % 16 != 0 || D > kGenMaxD) return hipErrorInvalidValue;
    // per-head byte offsets are 32 bit, including the rows a partial last query block overhangs
    if ((unsigned long long)(N + kBlockM) * D * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (in_dtype != 0 && in_dtype != 1) return hipErrorInvalidValue;
    if (algo == 0) algo = auto_algo(BH, N, D, in_dtype);
    if (algo == 5) return il_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 8, stream);
    if (algo == 6) return il_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 4, stream);
    if (algo == 23) return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 0, stream);
    if (algo == 24) return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 1, stream);
    if (algo == 29) {   // 32-row waves, keys split over two groups of four waves per 128-row workgroup (d = 64); the split needs
        if (D != 64) return hipErrorInvalidValue;   // whole tiles per group: other N run the 16-row waves (same workgroup size)
        if (N % 128 != 0) return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 1 | 8, stream);
        return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 1 | 4 | 16, stream);
    }
    if (algo == 28) {   // d = 128 with one wave per SIMD (four 64-row waves, the whole register file each)
        if (D != 128) return hipErrorInvalidValue;
        return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 1 | 12, stream);
    }
    if (algo == 26 || algo == 27) {   // the pipeline on half-width / quarter-width waves (32 / 16 rows at D = 64, 16 at D = 128)
        if (D != 64 && !(D == 128 && algo == 26)) return hipErrorInvalidValue;
        return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 1 | (algo == 26 ? 4 : 8), stream);
    }
#ifdef FA_EXPERIMENTS
    if (algo == 13) return w64_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);    // round 1's defaults and the
    if (algo == 16) return w64x_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);   // 32x32x16 pipeline: A/B baselines
    if (algo == 21) return rp_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 0, stream);
    if (algo == 22) return rp_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 1, stream);
    if (algo == 25) return rp16_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, 3, stream);   // 24 with LDS-DMA staging
    if (algo >= 17 && algo <= 20) return sk_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, algo - 17, stream);   // A/B kernels AUTO never selects: only in libfa_mi355_exp.so (make experimental)
    if (algo == 7 || algo == 8) {   // occupancy variants of the plain tiled kernel, fp16 d=64 fp32-out
        if (D != 64 || in_dtype != 0 || out_dtype != 0) return hipErrorInvalidValue;
        return algo == 7 ? launch_tiled<F16, 64, true, 4, 3>(Q, K, V, O, BH, N, scale, stream)
                         : launch_tiled<F16, 64, true, 4, 2>(Q, K, V, O, BH, N, scale, stream);
    }
    if (algo == 14) return w64p_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);
#else
    if (algo == 3 || algo == 4 || (algo >= 7 && algo <= 22) || algo == 25 || algo > 29) return hipErrorInvalidValue;
#endif
    if (algo == 3 || algo == 4 || (algo >= 9 && algo <= 12) || algo == 15) return hipErrorInvalidValue;   // ids of removed A/B kernels
    if (in_dtype == 0)
        return out_dtype == 0 ? dispatch_d<F16, true>(Q, K, V, O, BH, N, D, scale, algo, stream)
                              : dispatch_d<F16, false>(Q, K, V, O, BH, N, D, scale, algo, stream);
    if (in_dtype == 1)
        return out_dtype == 0 ? dispatch_d<BF16, true>(Q, K, V, O, BH, N, D, scale, algo, stream)
                              : dispatch_d<BF16, false>(Q, K, V, O, BH, N, D, scale, algo, stream);
    return hipErrorInvalidValue;
}

// Causal forward (not a reference entry point).  algo: 0 auto, 1 generic, 2 tiled, 13 w64, 24 the pipeline, 28 the pipeline with one wave per SIMD (d = 128).
hipError_t forward_causal_dispatch(const void* Q, const void* K, const void* V, void* O,
                                   int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                                   int algo, hipStream_t stream)
{
    if (!Q || !K || !V || !O) return hipErrorInvalidValue;
    if (BH <= 0 || N <= 0 || D <= 0 || D % 16 != 0 || D > kGenMaxD) return hipErrorInvalidValue;
    if ((unsigned long long)(N + kBlockM) * D * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (algo != 0 && algo != 1 && algo != 2 && algo != 6 && algo != 13 && algo != 24 && algo != 28) return hipErrorInvalidValue;
    if (in_dtype != 0 && in_dtype != 1) return hipErrorInvalidValue;
    // algo 13: the 64-rows-per-wave kernel with the mask; measured 3-4 % SLOWER than the plain tiled kernel
    // under the mask (B8 H16 N4096 d64: 0.462 vs 0.443 ms; N8192 d128: 2.41 vs 2.36 ms), so AUTO stays tiled.
#ifdef FA_EXPERIMENTS
    if (algo == 13)
        return w64_causal_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);
#else
    if (algo == 13) return hipErrorInvalidValue;
#endif
    // the pipeline under the mask (fa_fwd_rp16.hip): B8 H16 N4096 d64 fp16 0.309 ms against 0.369 for the tiled kernel (bf16 0.349 /
    // 0.364), N8192 d128 2.00 against 2.29 ms -- AUTO wherever the grid gives every CU a workgroup, else the tiled kernel
    const bool algo_was_auto = algo == 0;
    if (algo == 0 && (D == 64 || D == 128) && N > 256 &&
        (long long)BH * ((N + (D == 64 ? 511 : 255)) / (D == 64 ? 512 : 256)) >= device_cus())
        algo = 24;
    // d = 128, long sequences on grids of four rounds or more: one wave per SIMD as in the plain forward (B8 H16 N8192 2.005 vs 2.063 ms,
    // B1 H32 N16384 2.55 vs 2.68; N = 4096 loses 2 %; profiles/r03_d128_variants.txt)
    if (algo_was_auto && algo == 24 && D == 128 && N >= 8192 && (long long)BH * ((N + 255) / 256) >= 4 * (long long)device_cus())
        algo = 28;
    if (algo == 28) return rp16_causal_dispatch_1w(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);   // (d = 128 only)
    if (algo == 24) return rp16_causal_dispatch(Q, K, V, O, BH, N, D, scale, in_dtype, out_dtype, stream);
    if (in_dtype == 0)
        return out_dtype == 0 ? dispatch_causal_d<F16, true>(Q, K, V, O, BH, N, D, scale, algo, stream)
                              : dispatch_causal_d<F16, false>(Q, K, V, O, BH, N, D, scale, algo, stream);
    if (in_dtype == 1)
        return out_dtype == 0 ? dispatch_causal_d<BF16, true>(Q, K, V, O, BH, N, D, scale, algo, stream)
                              : dispatch_causal_d<BF16, false>(Q, K, V, O, BH, N, D, scale, algo, stream);
    return hipErrorInvalidValue;
}

}  // namespace fa
