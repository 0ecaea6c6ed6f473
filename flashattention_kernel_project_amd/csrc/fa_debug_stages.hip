// fa_debug_stages.hip -- the three stages of the tiled forward, one at a time, with their
// intermediates in memory (the reference's counterparts are the single-stage
// experiments FlashAttention/t16/*debug*.cu).  NOT on the product path: these exist to localise a
// layout bug to "K image / Q fragment / accumulator map", "softmax arithmetic" or "P packing / V^T
// image / transposed reads" in one run.  Each stage uses the SAME LDS images, fragment loads and
// accumulator-to-(row,key) maps as fa_fwd_kernels.hip; only the data flow between stages goes
// through global memory instead of registers.
//
//   stage 1  S = scale * Q.K^T                     Q,K [BH,N,D] 16-bit  ->  S [BH,N,N] fp32
//   stage 2  P = softmax rows of S, rounded         S [BH,N,N] fp32      ->  P [BH,N,N] 16-bit
//            (2^(c*s - max) / sum, c = log2 e, sum over the un-rounded terms, as the tiled kernels' fp16 path)
//   stage 3  O = P.V                                P [BH,N,N], V [BH,N,D] 16-bit -> O [BH,N,D] fp32
//
// One workgroup = 4 waves x 32 query rows; D in {64,128}; any N (ragged tails handled as in the
// product kernels: rows/keys past N read as zero and are not stored).
#include "fa_tile.hpp"

namespace fa {

namespace dbg {
constexpr int kW = 4;
constexpr int kRows = 32 * kW;
}  // namespace dbg

// ---- stage 1: S = scale * Q.K^T through the K LDS image and the swapped MFMA --------------------------
template <typename T, int D>
__global__ __launch_bounds__(64 * dbg::kW)
void fa_debug_qk_kernel(const uint16_t* __restrict__ Qg, const uint16_t* __restrict__ Kg, float* __restrict__ Sg,
                        int N, int nqb, float scale)
{
    using G = TileGeom<D>;
    constexpr int W = dbg::kW;
    __shared__ __attribute__((aligned(16))) char smem[G::kTileBytes];
    const unsigned bh = blockIdx.x / (unsigned)nqb, qb = blockIdx.x % (unsigned)nqb;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, r = lane & 31u, h = lane >> 5;
    const size_t head_elems = (size_t)N * D;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(Qg + bh * head_elems, (unsigned)(head_elems * 2));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(Kg + bh * head_elems, (unsigned)(head_elems * 2));
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(Sg + (size_t)bh * N * N, (unsigned)((size_t)N * N * 4));
    const unsigned q_row = qb * dbg::kRows + wave * 32u + r;
    u32x4 qf[G::kKSteps];
#pragma unroll
    for (int s = 0; s < G::kKSteps; ++s) qf[s] = buf_load16(rq, q_row * G::kRowBytes + (16u * s + 8u * h) * 2u);
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.0f;
    constexpr int kLoadsW = (kBlockN * G::kChunks) / (64 * W);
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
            const unsigned idx = tid + p * 64u * W, row = idx / G::kChunks, ch = idx % G::kChunks;
            lds_write16(smem, G::k_off(row, ch), buf_load16(rk, ((unsigned)t * kBlockN + row) * G::kRowBytes + ch * 16u));
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s = zero16;
#pragma unroll
            for (int ks = 0; ks < G::kKSteps; ++ks) {
                const u32x4 kf = lds_read16(smem, kb * 32u * G::kRowBytes + r * G::kRowBytes + (((2u * ks + h) ^ G::k_swz(r)) << 4));
                s = T::mfma32(kf, qf[ks], s);
            }
            // accumulator register i of lane (r,h): query row q_row, key 32kb + 8(i>>2) + 4h + (i&3)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned key = (unsigned)t * kBlockN + kb * 32u + 8u * (i >> 2) + 4u * h + (i & 3);
                const float val = s[i] * scale;
                if (q_row < (unsigned)N && key < (unsigned)N) buf_store4(rs, (q_row * (unsigned)N + key) * 4u, __float_as_uint(val));
            }
        }
    }
}

// ---- stage 2: row softmax of S, packed to 16 bit (one wave per row; the arithmetic of the product path)
template <typename T>
__global__ __launch_bounds__(64)
void fa_debug_softmax_kernel(const float* __restrict__ Sg, uint16_t* __restrict__ Pg, int N)
{
    const size_t row = blockIdx.x;
    const float* s = Sg + row * (size_t)N;
    uint16_t* p = Pg + row * (size_t)N;
    const unsigned lane = threadIdx.x;
    float mx = -INFINITY;
    for (unsigned j = lane; j < (unsigned)N; j += 64) mx = fmaxf(mx, s[j] * kLog2e);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.0f;
    for (unsigned j = lane; j < (unsigned)N; j += 64) sum += fast_exp2(__builtin_fmaf(s[j], kLog2e, -mx));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (unsigned j = lane; j < (unsigned)N; j += 64) {
        const float e = fast_exp2(__builtin_fmaf(s[j], kLog2e, -mx)) * inv;
        p[j] = (uint16_t)(T::pack2(e, 0.0f) & 0xFFFFu);
    }
}

// ---- stage 3: O = P.V through the V LDS image, transposed reads and P as the B operand ----------------
template <typename T, int D>
__global__ __launch_bounds__(64 * dbg::kW)
void fa_debug_pv_kernel(const uint16_t* __restrict__ Pg, const uint16_t* __restrict__ Vg, float* __restrict__ Og,
                        int N, int nqb)
{
    using G = TileGeom<D>;
    constexpr int W = dbg::kW;
    __shared__ __attribute__((aligned(16))) char smem[G::kBufBytes];   // [K tile (unused here)][V tile]: the product kernels' buffer shape
    const unsigned bh = blockIdx.x / (unsigned)nqb, qb = blockIdx.x % (unsigned)nqb;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, r = lane & 31u, h = lane >> 5;
    const size_t head_elems = (size_t)N * D;
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(Vg + bh * head_elems, (unsigned)(head_elems * 2));
    const uint16_t* P = Pg + (size_t)bh * N * N;
    const unsigned q_row = qb * dbg::kRows + wave * 32u + r;
    const unsigned i16 = lane & 15u, vq = i16 >> 2, vp = i16 & 3u, vg = (lane >> 4) & 1u;
    unsigned v_rd[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        v_rd[par] = G::kTileBytes + h * G::kDBlocks * 256u + ((vq ^ par) << 6) + vg * 32u + vp * 8u;
    f32x16 o[G::kDBlocks];
#pragma unroll
    for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.0f;
    constexpr int kLoadsW = (kBlockN * G::kChunks) / (64 * W);
    const int ntiles = (N + kBlockN - 1) / kBlockN;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kLoadsW; ++p) {
            const unsigned idx = tid + p * 64u * W, row = idx / G::kChunks, ch = idx % G::kChunks;
            lds_write16(smem, G::kTileBytes + G::v_off(row, ch), buf_load16(rv, ((unsigned)t * kBlockN + row) * G::kRowBytes + ch * 16u));
        }
        __syncthreads();
        // B operand of k-step ks: element j of lane (r,h) is P[q_row][key 16ks + 8(j>>2) + 4h + (j&3)]
        u32x4 pk[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned e[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = 2 * w + u;
                    const unsigned key = (unsigned)t * kBlockN + 16u * ks + 8u * (j >> 2) + 4u * h + (j & 3);
                    e[u] = (q_row < (unsigned)N && key < (unsigned)N) ? P[(size_t)q_row * N + key] : 0u;
                }
                pk[ks][w] = e[0] | (e[1] << 16);
            }
#pragma unroll
        for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 vf;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const u32x2 half = lds_read_tr8(smem, v_rd[db & 1] + ((4u * ks + 2u * jj) * G::kDBlocks + db) * 256u);
                    vf[2 * jj] = half[0];
                    vf[2 * jj + 1] = half[1];
                }
                o[db] = T::mfma32(vf, pk[ks], o[db]);
            }
    }
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(Og + bh * head_elems, (unsigned)(head_elems * 4));
#pragma unroll
    for (int db = 0; db < G::kDBlocks; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned col = db * 32u + 8u * g + 4u * h;
            const f32x4 v = {o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
            buf_store16(ro, (q_row * D + col) * 4u, __builtin_bit_cast(u32x4, v));
        }
}

// ---- host side -------------------------------------------------------------------------------------
template <typename T>
static hipError_t debug_stage_t(int stage, const void* A, const void* B, void* Out, int BH, int N, int D, float scale,
                                hipStream_t stream)
{
    const int nqb = (N + dbg::kRows - 1) / dbg::kRows;
    const long long nwg = (long long)BH * nqb;
    if (nwg > 0x7FFFFFFFll || (long long)BH * N > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(B);
    if (stage == 1) {
        if (D == 64) FA_LAUNCH((fa_debug_qk_kernel<T, 64>), dim3((unsigned)nwg), dim3(256), 0, stream, a, b, static_cast<float*>(Out), N, nqb, scale);
        else         FA_LAUNCH((fa_debug_qk_kernel<T, 128>), dim3((unsigned)nwg), dim3(256), 0, stream, a, b, static_cast<float*>(Out), N, nqb, scale);
    } else if (stage == 2) {
        FA_LAUNCH((fa_debug_softmax_kernel<T>), dim3((unsigned)((long long)BH * N)), dim3(64), 0, stream,
                           static_cast<const float*>(A), static_cast<uint16_t*>(Out), N);
    } else {
        if (D == 64) FA_LAUNCH((fa_debug_pv_kernel<T, 64>), dim3((unsigned)nwg), dim3(256), 0, stream, a, b, static_cast<float*>(Out), N, nqb);
        else         FA_LAUNCH((fa_debug_pv_kernel<T, 128>), dim3((unsigned)nwg), dim3(256), 0, stream, a, b, static_cast<float*>(Out), N, nqb);
    }
    return launch_status();
}

// stage 1: A = Q, B = K, Out = S fp32 [BH,N,N].  stage 2: A = S fp32, B unused, Out = P 16-bit [BH,N,N].
// stage 3: A = P 16-bit, B = V, Out = O fp32 [BH,N,D].
hipError_t debug_stage_dispatch(int stage, const void* A, const void* B, void* Out, int BH, int N, int D, float scale,
                                int dtype, hipStream_t stream)
{
    if (stage < 1 || stage > 3 || !A || !Out || (stage != 2 && !B)) return hipErrorInvalidValue;
    if (BH <= 0 || N <= 0 || (D != 64 && D != 128) || (dtype != 0 && dtype != 1)) return hipErrorInvalidValue;
    if ((unsigned long long)N * N * 4ull >= (1ull << 32)) return hipErrorInvalidValue;   // S per head behind one descriptor
    return dtype == 0 ? debug_stage_t<F16>(stage, A, B, Out, BH, N, D, scale, stream)
                      : debug_stage_t<BF16>(stage, A, B, Out, BH, N, D, scale, stream);
}

}  // namespace fa
