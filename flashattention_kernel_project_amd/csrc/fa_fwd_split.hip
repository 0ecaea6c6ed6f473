// fa_fwd_split.hip -- attention forward with Nq != Nk and a split over the keys ("flash-decoding").
//
// SURVEY.md 8(f) rank 1, third part; not a reference entry point (the reference's kernels are
// self-attention, Nq == Nk; its single-query experiment is
// flashattn_warp_spc/../flashattn_streaming_16x16_mw_v7_5*.cu).  Use: few query rows against a long
// K/V (decode), where one workgroup per (head, query block) cannot fill 256 CUs and the job is to
// stream K and V once at HBM rate.
//
//   pass 1  fa_fwd_split_kernel: grid = BH x query blocks x S.  Workgroup (bh, qb, s) runs the tiled
//           stream of fa_fwd_kernels.hip (same LDS images, MFMA orientation, lazy running max) over
//           the keys [s*chunk, (s+1)*chunk) and writes the UNNORMALISED O^T (fp32), the reference max
//           m (log2 units) and the row sum l of its split to the caller's workspace
//           [BH][S][Nq][D + 2] fp32 (m and l in the two trailing slots of a row).
//   pass 2  fa_split_combine_kernel: per (bh, row): M = max_s m_s, L = sum_s l_s 2^(m_s - M),
//           O = sum_s O_s 2^(m_s - M) / L, written in the output dtype.
//   S == 1  pass 1 normalises and writes O directly; no workspace, no pass 2.
//
// 128-row workgroups (4 waves x 32 rows): query rows >= Nq read zeros and are not stored, so for a
// handful of query rows most MFMA work is idle lanes -- irrelevant here, the path is HBM-bound.
#include "fa_tile.hpp"

#include <cstdlib>

namespace fa {

#ifndef FA_SPLIT_ROTATE
#define FA_SPLIT_ROTATE 1
#endif
#ifndef FA_SPLIT_NT
#define FA_SPLIT_NT 1   // K/V are read exactly once: non-temporal loads (5.97 -> 6.75 TB/s at B8 H16 Nq1 Nk32768 d128)
#endif
namespace split {
constexpr int kW = 4;                 // waves per workgroup
constexpr int kRows = 32 * kW;        // query rows per workgroup
}  // namespace split

// kPartial: write (O^T unnormalised, m, l) to the workspace instead of the normalised output.
template <typename T, int D, bool kOutF32, bool kPartial>
__global__ __launch_bounds__(64 * split::kW, D == 64 ? 4 : 2)
void fa_fwd_split_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                         const uint16_t* __restrict__ Vg, void* __restrict__ Og, float* __restrict__ ws,
                         int Nq, int Nk, int nqb, int S, int chunk, float scale_log2e)
{
    using namespace split;
    using G = TileGeom<D>;
    constexpr int W = kW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // block -> (head, query block, split): the splits of one (head, query block) are consecutive
    const unsigned sp = blockIdx.x % (unsigned)S;
    const unsigned rest = blockIdx.x / (unsigned)S;
    const unsigned bh = rest / (unsigned)nqb, qb = rest % (unsigned)nqb;

    const unsigned tid  = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane = tid & 63u;
    const unsigned r = lane & 31u, h = lane >> 5;

    const unsigned key0 = sp * (unsigned)chunk;                       // multiple of kBlockN
    const unsigned key1 = min((unsigned)Nk, key0 + (unsigned)chunk);  // exclusive
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + (size_t)bh * Nq * D, (unsigned)((size_t)Nq * D * 2));
    // K/V descriptors end at this split's last key: rows beyond it read 0 and are masked below
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + (size_t)bh * Nk * D, key1 * (unsigned)D * 2u);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + (size_t)bh * Nk * D, key1 * (unsigned)D * 2u);

    constexpr int kLoadsW = (kBlockN * G::kChunks) / (64 * W);
    const unsigned q_row = qb * (unsigned)kRows + wave * 32u + r;
    const bool has_rows = qb * (unsigned)kRows + wave * 32u < (unsigned)Nq;   // wave-uniform

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
    auto stage_load = [&](unsigned kv0) {
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
#if FA_SPLIT_NT
            kst[p] = buf_load16_nt(rk, kv0 * G::kRowBytes + g_off[p]);
            vst[p] = buf_load16_nt(rv, kv0 * G::kRowBytes + g_off[p]);
#else
            kst[p] = buf_load16(rk, kv0 * G::kRowBytes + g_off[p]);
            vst[p] = buf_load16(rv, kv0 * G::kRowBytes + g_off[p]);
#endif
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

    f32x16 o[G::kDBlocks];
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.0f;
#pragma unroll
    for (int db = 0; db < G::kDBlocks; ++db) o[db] = zero16;
    float m_ref = 0.0f, l_part = 0.0f;

    const int ntiles = (int)((key1 - key0 + kBlockN - 1) / kBlockN);   // >= 1 by construction of S
    // Tiles are visited in a rotated order that differs per (head, split): the chunks of one head start a
    // power-of-two stride apart, and workgroups marching through them in lockstep would keep hitting the
    // same few HBM channels.  The streaming softmax does not care about the order.
    const unsigned rot = FA_SPLIT_ROTATE ? (sp * 5u + bh * 3u) % (unsigned)ntiles : 0u;
    auto tile_of = [&](int t) { const unsigned ti = (unsigned)t + rot; return ti >= (unsigned)ntiles ? ti - (unsigned)ntiles : ti; };
    stage_load(key0 + tile_of(0) * kBlockN);
    stage_write(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned cur = t & 1u;
        const char* kbuf = smem + cur * G::kBufBytes;
        const unsigned kv0 = key0 + tile_of(t) * kBlockN;
        if (t + 1 < ntiles) stage_load(key0 + tile_of(t + 1) * kBlockN);

        // a wave whose 32 rows all lie past Nq (the usual case for a handful of query rows) only stages
        if (has_rows) {
        f32x16 s[2];
#pragma unroll
        for (int ks = 0; ks < G::kKSteps; ++ks)   // consecutive MFMAs alternate accumulators
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const u32x4 kf = lds_read16(kbuf, kb * 32u * G::kRowBytes + k_rd_row + (((2u * ks + h) ^ k_rd_swz) << 4));
                s[kb] = T::mfma32(kf, qf[ks], ks == 0 ? zero16 : s[kb]);
            }
        if (kv0 + kBlockN > key1) {   // keys past the split's end -> -inf (p = 0)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned key = kv0 + (unsigned)(kb * 32 + (i & 3) + 8 * (i >> 2)) + 4u * h;
                    if (key >= key1) s[kb][i] = -INFINITY;
                }
        }

        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 32; e += 2) tmax = max3(tmax, s[e >> 4][e & 15], s[(e + 1) >> 4][(e + 1) & 15]);
        tmax *= c;
        if (t == 0 || __any(tmax - m_ref > kThr)) {
            const float mx = fmaxf(tmax, swap_halves(tmax));
            const float m_new = (t == 0) ? mx : fmaxf(mx, m_ref);
            const float alpha = (t == 0) ? 0.0f : fast_exp2(m_ref - m_new);
            m_ref = m_new;
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            l_part *= alpha;
        }

        u32x4 pk[4];
        float ls0 = 0.0f, ls1 = 0.0f;
        const float neg_m = -m_ref;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int kb = q4 >> 1, b8 = (q4 & 1) * 8;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float p0 = fast_exp2(__builtin_fmaf(s[kb][b8 + 2 * w], c, neg_m));
                const float p1 = fast_exp2(__builtin_fmaf(s[kb][b8 + 2 * w + 1], c, neg_m));
                pk[q4][w] = T::pack2(p0, p1);
                if (w & 1) ls1 = T::sum2(pk[q4][w], ls1);   // sums of the ROUNDED weights (fa_common.hpp)
                else ls0 = T::sum2(pk[q4][w], ls0);
            }
        }
        l_part += ls0 + ls1;

#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < G::kDBlocks; ++db) {
                u32x4 vf;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const u32x2 half = lds_read_tr8(kbuf, v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                    vf[2 * jj] = half[0];
                    vf[2 * jj + 1] = half[1];
                }
                o[db] = T::mfma32(vf, pk[ks], o[db]);
            }
        }   // has_rows

        if (t + 1 < ntiles) stage_write(cur ^ 1u);
        __syncthreads();
    }

    const float l = l_part + swap_halves(l_part);
    if constexpr (kPartial) {
        // workspace row (bh, sp, q_row): D floats of O^T (unnormalised), then m, then l
        const size_t rows = (size_t)Nq;
        const unsigned rs = (unsigned)(D + 2) * 4u;
        const __amdgpu_buffer_rsrc_t rw =
            make_rsrc(ws + ((size_t)bh * S + sp) * rows * (D + 2), (unsigned)(rows * rs));
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned col = db * 32u + 8u * g + 4u * h;
                // scalar copies first: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0
                const float a = o[db][4 * g], b = o[db][4 * g + 1], cc = o[db][4 * g + 2], d = o[db][4 * g + 3];
                // rows are (D+2)*4 bytes: 8-byte aligned, not 16 -> two 8-byte stores
                buf_store8(rw, q_row * rs + col * 4u, u32x2{__float_as_uint(a), __float_as_uint(b)});
                buf_store8(rw, q_row * rs + col * 4u + 8u, u32x2{__float_as_uint(cc), __float_as_uint(d)});
            }
        if (h == 0)
            buf_store8(rw, q_row * rs + (unsigned)D * 4u, u32x2{__float_as_uint(m_ref), __float_as_uint(l)});
    } else {
        const float inv = 1.0f / l;
        constexpr unsigned es = kOutF32 ? 4u : 2u;
        const __amdgpu_buffer_rsrc_t ro =
            make_rsrc(reinterpret_cast<char*>(Og) + (size_t)bh * Nq * D * es, (unsigned)((size_t)Nq * D * es));
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned col = db * 32u + 8u * g + 4u * h;
                const float a = o[db][4 * g] * inv, b = o[db][4 * g + 1] * inv;
                const float cc = o[db][4 * g + 2] * inv, d = o[db][4 * g + 3] * inv;
                if constexpr (kOutF32) {
                    const f32x4 v = {a, b, cc, d};
                    buf_store16(ro, (q_row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
                } else {
                    buf_store8(ro, (q_row * D + col) * 2u, u32x2{T::pack2(a, b), T::pack2(cc, d)});
                }
            }
    }
}

// One wave per (bh, row).  Lane = (slice, 4 output columns): the D/4 column groups times 64/(D/4) slices of
// the split axis, so that the S partials of a row are read by parallel lanes with independent loads instead
// of one lane walking them (a serial walk costs ~0.25 us per partial: 64 us at S = 128).
template <typename T, bool kOutF32>
__global__ __launch_bounds__(64)
void fa_split_combine_kernel(const float* __restrict__ ws, void* __restrict__ Og, int BH, int Nq, int D, int S)
{
    const int cols4 = D / 4;             // 16 or 32
    const int nsl = 64 / cols4;          // 4 or 2 slices of the split axis
    const unsigned lane = threadIdx.x;
    const int c4 = (int)lane % cols4, sl = (int)lane / cols4;
    const long long rowi = blockIdx.x;   // bh * Nq + row
    const int row = (int)(rowi % Nq);
    const int bh = (int)(rowi / Nq);
    const size_t stride_s = (size_t)Nq * (D + 2);
    const float* base = ws + (size_t)bh * S * stride_s + (size_t)row * (D + 2);
    // global reference max: every lane takes splits lane, lane+64, ...
    float M = -INFINITY;
    for (int s = (int)lane; s < S; s += 64) M = fmaxf(M, base[(size_t)s * stride_s + D]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
    float L = 0.0f, acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int s = sl; s < S; s += nsl) {
        const float* p = base + (size_t)s * stride_s;
        const float w = fast_exp2(p[D] - M);
        L += p[D + 1] * w;
        // workspace rows are (D+2)*4 bytes: 8-byte aligned, so two 8-byte loads
        const float2 v0 = *reinterpret_cast<const float2*>(p + 4 * c4), v1 = *reinterpret_cast<const float2*>(p + 4 * c4 + 2);
        acc[0] += v0.x * w;
        acc[1] += v0.y * w;
        acc[2] += v1.x * w;
        acc[3] += v1.y * w;
    }
    // sum the slices: lanes that differ only in `sl` are cols4, 2*cols4, ... apart
    for (int o = cols4; o < 64; o <<= 1) {
        L += __shfl_xor(L, o, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    if (sl != 0) return;
    const float inv = 1.0f / L;
    const size_t off = ((size_t)bh * Nq + row) * D + 4 * c4;
    if constexpr (kOutF32) {
        float* o = static_cast<float*>(Og) + off;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = acc[i] * inv;
    } else {
        unsigned* o = reinterpret_cast<unsigned*>(static_cast<uint16_t*>(Og) + off);
        o[0] = T::pack2(acc[0] * inv, acc[1] * inv);
        o[1] = T::pack2(acc[2] * inv, acc[3] * inv);
    }
}

// ---- host side -------------------------------------------------------------------------------------
// Number of key splits: enough workgroups for about four per CU, at least 4 tiles per split.
int split_count(int BH, int Nq, int Nk)
{
    const long long base = (long long)BH * ((Nq + split::kRows - 1) / split::kRows);
    const int tiles = (Nk + kBlockN - 1) / kBlockN;
    constexpr int target = 1024, min_tiles = 4;   // workgroups in total, fewest tiles per split
    long long s = (target + base - 1) / base;   // about four workgroups per CU
    if (s > tiles / min_tiles) s = tiles / min_tiles;
    if (s < 1) s = 1;
    // every split must hold at least one key: chunk = ceil(tiles / s) tiles, recompute s from the chunk
    const int chunk_tiles = (int)((tiles + s - 1) / s);
    return (tiles + chunk_tiles - 1) / chunk_tiles;
}

size_t split_workspace_bytes(int BH, int Nq, int Nk, int D)
{
    const int S = split_count(BH, Nq, Nk);
    return S <= 1 ? 0 : (size_t)BH * S * Nq * (D + 2) * sizeof(float);
}

template <typename T, int D, bool kOutF32>
static hipError_t launch_split(const void* Q, const void* K, const void* V, void* O, void* ws, size_t ws_bytes,
                               int BH, int Nq, int Nk, float scale, hipStream_t stream)
{
    using G = TileGeom<D>;
    const int S = split_count(BH, Nq, Nk);
    const int tiles = (Nk + kBlockN - 1) / kBlockN;
    const int chunk = ((tiles + S - 1) / S) * kBlockN;
    const int nqb = (Nq + split::kRows - 1) / split::kRows;
    const long long nwg = (long long)BH * nqb * S;
    if (nwg > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (S > 1 && (!ws || ws_bytes < split_workspace_bytes(BH, Nq, Nk, D))) return hipErrorInvalidValue;
    const uint16_t *q = static_cast<const uint16_t*>(Q), *k = static_cast<const uint16_t*>(K), *v = static_cast<const uint16_t*>(V);
    hipError_t attr = ensure_dyn_lds(reinterpret_cast<const void*>(&fa_fwd_split_kernel<T, D, kOutF32, false>), G::kLdsBytes);
    if (attr == hipSuccess) attr = ensure_dyn_lds(reinterpret_cast<const void*>(&fa_fwd_split_kernel<T, D, kOutF32, true>), G::kLdsBytes);
    if (attr != hipSuccess) return attr;
    if (S == 1) {
        FA_LAUNCH((fa_fwd_split_kernel<T, D, kOutF32, false>), dim3((unsigned)nwg), dim3(64 * split::kW), G::kLdsBytes,
                           stream, q, k, v, O, static_cast<float*>(nullptr), Nq, Nk, nqb, S, chunk, scale * kLog2e);
        return launch_status();
    }
    FA_LAUNCH((fa_fwd_split_kernel<T, D, kOutF32, true>), dim3((unsigned)nwg), dim3(64 * split::kW), G::kLdsBytes,
                       stream, q, k, v, O, static_cast<float*>(ws), Nq, Nk, nqb, S, chunk, scale * kLog2e);
    hipError_t e = launch_status();
    if (e != hipSuccess) return e;
    const long long rows = (long long)BH * Nq;   // one wave per output row
    if (rows > 0x7FFFFFFFll) return hipErrorInvalidValue;
    FA_LAUNCH((fa_split_combine_kernel<T, kOutF32>), dim3((unsigned)rows), dim3(64), 0, stream,
                       static_cast<const float*>(ws), O, BH, Nq, D, S);
    return launch_status();
}

hipError_t split_dispatch(const void* Q, const void* K, const void* V, void* O, void* ws, size_t ws_bytes,
                          int BH, int Nq, int Nk, int D, float scale, int in_dtype, int out_dtype, hipStream_t stream)
{
    if (!Q || !K || !V || !O) return hipErrorInvalidValue;
    if (BH <= 0 || Nq <= 0 || Nk <= 0 || (D != 64 && D != 128)) return hipErrorInvalidValue;
    if (in_dtype != 0 && in_dtype != 1) return hipErrorInvalidValue;
    if (out_dtype != 0 && out_dtype != 1) return hipErrorInvalidValue;
    // per-head byte offsets are 32 bit (fp32 output / workspace rows of D+2 floats)
    if ((unsigned long long)(Nq + split::kRows) * (unsigned)(D + 2) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if ((unsigned long long)(Nk + kBlockN) * (unsigned)D * 2ull >= (1ull << 32)) return hipErrorInvalidValue;
#define FA_SPLIT_GO(TT, DD, OF) return launch_split<TT, DD, OF>(Q, K, V, O, ws, ws_bytes, BH, Nq, Nk, scale, stream)
    if (D == 64) {
        if (in_dtype == 0) { if (out_dtype == 0) FA_SPLIT_GO(F16, 64, true); FA_SPLIT_GO(F16, 64, false); }
        if (out_dtype == 0) FA_SPLIT_GO(BF16, 64, true);
        FA_SPLIT_GO(BF16, 64, false);
    }
    if (in_dtype == 0) { if (out_dtype == 0) FA_SPLIT_GO(F16, 128, true); FA_SPLIT_GO(F16, 128, false); }
    if (out_dtype == 0) FA_SPLIT_GO(BF16, 128, true);
    FA_SPLIT_GO(BF16, 128, false);
#undef FA_SPLIT_GO
}

}  // namespace fa
