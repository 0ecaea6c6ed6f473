// fa_streaming16.hip -- the reference's 16x16 streaming family on gfx950.
//
// Replaces flashattn_streaming_16x16_kernel_mw* (Streaming_FlashAttention_Forward_Kernel/
// flashattn_streaming_16x16_mw.cu:73-248 ... flashattn_warp_spc/flashattn_streaming_16x16_mw_v11.cu:101-280):
// M = Kdim = Dv = 16, K/V streamed in 16-key tiles, one problem per "batch" element,
// signature (Q,K,V,O,num_batches,seq_len,scale); Q [B,16,16], K [B,16,L] (k-major) -- or the
// v8+ pre-transposed K_T [B,L,16] (flashattn_streaming_16x16_mw_v8.cu:103-111,344-359) --
// V [B,L,16], O [B,16,16] fp32, O = y / (l + 1e-6) (mw.cu:237-247).
//
// The reference spends its time in 18 five-step shuffle all-reduces per row per tile and keeps
// (m,l,y) in one lane.  Here one 64-lane wave owns one batch element: S^T = K.Q^T on
// v_mfma_f32_16x16x16 leaves the query on the lane (col = lane&15) and 4 keys in each 16-lane
// group's registers, so a row's statistics need two xor-shuffles per tile, every lane keeps its
// own (m,l), and PV is a second MFMA whose B operand is the packed P registers as they stand.
// These shapes are launch/latency bound (B=1024,L=128 is 134 MFLOP, 10 MB): no LDS staging, the
// 512-B tiles come straight from L2 with the next tile's loads issued before the current math.
#include "fa_common.hpp"

namespace fa {

constexpr int kS16WavesPerBlock = 4;

template <bool kKT>
__global__ __launch_bounds__(64 * kS16WavesPerBlock)
void fa_streaming16_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg,
                           const uint16_t* __restrict__ Vg, float* __restrict__ Og,
                           int num_batches, int seq_len, float scale_log2e)
{
    using T = F16;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned b = blockIdx.x * kS16WavesPerBlock + wave;
    if (b >= (unsigned)num_batches) return;   // wave-uniform
    const unsigned lane = threadIdx.x & 63u, c16 = lane & 15u, g4 = lane >> 4;
    const unsigned L = (unsigned)seq_len;

    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + (size_t)b * 256, 512u);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + (size_t)b * 16 * L, 32u * L);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + (size_t)b * 16 * L, 32u * L);

    // B operand of S^T = K.Q^T: Q[q = c16][k = 4g4 + j]
    const u32x2 qf = buf_load8(rq, (c16 * 16u + 4u * g4) * 2u);

    auto load_k = [&](unsigned kv0) -> u32x2 {   // A operand: K[key = kv0 + c16][k = 4g4 + j]
        if constexpr (kKT) {
            return buf_load8(rk, ((kv0 + c16) * 16u + 4u * g4) * 2u);
        } else {
            unsigned e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                e[j] = __builtin_amdgcn_raw_buffer_load_b16(rk, ((4u * g4 + j) * L + kv0 + c16) * 2u, 0, 0);
            return u32x2{e[0] | (e[1] << 16), e[2] | (e[3] << 16)};
        }
    };
    auto load_v = [&](unsigned kv0) -> u32x2 {   // A operand of PV: V^T[d = c16][k = 4g4 + j]
        unsigned e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            e[j] = __builtin_amdgcn_raw_buffer_load_b16(rv, ((kv0 + 4u * g4 + j) * 16u + c16) * 2u, 0, 0);
        return u32x2{e[0] | (e[1] << 16), e[2] | (e[3] << 16)};
    };

    f32x4 y = {0.f, 0.f, 0.f, 0.f};   // y[i] = (unnormalised) O[q = c16][d = 4g4 + i]
    float m = -INFINITY, l_part = 0.0f;

    u32x2 kf = load_k(0), vf = load_v(0);
    for (unsigned kv0 = 0; kv0 < L; kv0 += 16) {
        const u32x2 kc = kf, vc = vf;
        if (kv0 + 16 < L) { kf = load_k(kv0 + 16); vf = load_v(kv0 + 16); }
        f32x4 s4 = T::mfma16(kc, qf, f32x4{0.f, 0.f, 0.f, 0.f});   // S[q=c16][kv0 + 4g4 + i]
#pragma unroll
        for (int i = 0; i < 4; ++i)   // `acc * scale` in fp32 (mw.cu:283), to log2 units; ragged tail masked
            s4[i] = (kv0 + 4u * g4 + i >= L) ? -INFINITY : s4[i] * scale_log2e;
        float tmax = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const float alpha = fast_exp2(m - m_new);
        m = m_new;
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = fast_exp2(s4[i] - m_new);
        l_part = l_part * alpha + (p[0] + p[1]) + (p[2] + p[3]);
        const u32x2 pf = {T::pack2(p[0], p[1]), T::pack2(p[2], p[3])};
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] *= alpha;
        y = T::mfma16(vc, pf, y);
    }
    float l = l_part + __shfl_xor(l_part, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / (l + 1e-6f);   // EPS of the reference family (mw.cu:45,239)
    const f32x4 out = {y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv};
    *reinterpret_cast<f32x4*>(Og + (size_t)b * 256 + c16 * 16u + 4u * g4) = out;
}

hipError_t streaming16_dispatch(const void* Q, const void* K, const void* V, float* O,
                                int num_batches, int seq_len, float scale, bool k_transposed,
                                hipStream_t stream)
{
    if (!Q || !K || !V || !O || num_batches <= 0 || seq_len <= 0) return hipErrorInvalidValue;
    if ((long long)seq_len * 32 >= (1ll << 31)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((num_batches + kS16WavesPerBlock - 1) / kS16WavesPerBlock);
    const float c = scale * kLog2e;
    if (k_transposed)
        FA_LAUNCH(fa_streaming16_kernel<true>, dim3(grid), dim3(64 * kS16WavesPerBlock), 0, stream,
                           (const uint16_t*)Q, (const uint16_t*)K, (const uint16_t*)V, O, num_batches, seq_len, c);
    else
        FA_LAUNCH(fa_streaming16_kernel<false>, dim3(grid), dim3(64 * kS16WavesPerBlock), 0, stream,
                           (const uint16_t*)Q, (const uint16_t*)K, (const uint16_t*)V, O, num_batches, seq_len, c);
    return launch_status();
}

}  // namespace fa
