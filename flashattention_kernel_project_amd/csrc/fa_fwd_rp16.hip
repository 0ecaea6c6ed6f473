// fa_fwd_rp16.hip -- entry points of the rolling half-tile pipeline (fa_fwd_rp16_kernel.hpp); the kernel families are
// instantiated in fa_fwd_rp16_{d64,d64n,d128,c}.hip.
#include "fa_tile.hpp"

namespace fa {

#define RP16_FAMILY_DECL(name) \
    hipError_t name(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype, \
                    bool fold, hipStream_t stream)
RP16_FAMILY_DECL(rp16_d64x4);    // 64-row waves, 512-row workgroups (+ the half-width redo kernel)
RP16_FAMILY_DECL(rp16_d64x2);    // 32-row waves
RP16_FAMILY_DECL(rp16_d64x1);    // 16-row waves
RP16_FAMILY_DECL(rp16_d64x2ks2); // 32-row waves, keys split over two groups of FOUR waves per 128-row workgroup (N % 128 == 0)
RP16_FAMILY_DECL(rp16_d128x2);   // d = 128: 32-row waves, 256-row workgroups (+ redo kernel on 16-row waves)
RP16_FAMILY_DECL(rp16_d128x1);
RP16_FAMILY_DECL(rp16_d128x4w4); // d = 128, one wave per SIMD: four 64-row waves, 256-row workgroups, 512 registers per wave
RP16_FAMILY_DECL(rp16_c_d64);    // causal
RP16_FAMILY_DECL(rp16_c_d128);
RP16_FAMILY_DECL(rp16_c_d128w4);  // causal, d = 128, one wave per SIMD
#ifdef FA_EXPERIMENTS
RP16_FAMILY_DECL(rp16_d64x4_dma);
hipError_t rp16_set_pass_ids_d64(unsigned*);
hipError_t rp16_set_pass_ids_d64n(unsigned*);
hipError_t rp16_set_pass_ids_d128(unsigned*);
hipError_t rp16_set_pass_ids_c(unsigned*);
hipError_t rp16_set_pass_ids_d128w(unsigned*);
hipError_t rp16_set_pass_ids_cw(unsigned*);
hipError_t rp16_set_pass_ids_d64ks(unsigned*);
hipError_t rp16_set_pass_ids(unsigned* dev_ptr)
{
    hipError_t e = rp16_set_pass_ids_d64(dev_ptr);
    if (e == hipSuccess) e = rp16_set_pass_ids_d64n(dev_ptr);
    if (e == hipSuccess) e = rp16_set_pass_ids_d128(dev_ptr);
    if (e == hipSuccess) e = rp16_set_pass_ids_c(dev_ptr);
    if (e == hipSuccess) e = rp16_set_pass_ids_d128w(dev_ptr);
    if (e == hipSuccess) e = rp16_set_pass_ids_cw(dev_ptr);
    if (e == hipSuccess) e = rp16_set_pass_ids_d64ks(dev_ptr);
    return e;
}
#endif
#undef RP16_FAMILY_DECL

static bool rp16_shape_ok(int N, int D)
{
    if (D != 64 && D != 128) return false;
    return (unsigned long long)(N + 64 * 8 + 3 * kBlockN) * (unsigned)D * 4ull < (1ull << 32);   // per-head byte offsets are 32 bit
}

// fold: 1 = folded fast pass first, 0 = exact passes only; +2 = K/V staging by LDS-DMA (experimental build); bits 2-3: 1 = half-width
// waves (32 rows at D = 64, 16 at D = 128), 2 = quarter-width (16 rows, D = 64), 3 = one wave per SIMD (64-row waves, D = 128); +16 with 1: keys split over two groups of four 32-row waves (N % 128 == 0)
hipError_t rp16_dispatch(const void* Q, const void* K, const void* V, void* O,
                         int BH, int N, int D, float scale, int in_dtype, int out_dtype, int fold,
                         hipStream_t stream)
{
    if (!rp16_shape_ok(N, D)) return hipErrorInvalidValue;
    const bool dma = (fold & 2) != 0;
    const int narrow = (fold >> 2) & 3;
    bool f = (fold & 1) != 0;
    if (!(scale == scale) || scale * kLog2e == 0.0f) f = false;   // NaN / zero scale: the exact passes define the result
    if (D == 128) {
        if (narrow == 3 && !dma) return rp16_d128x4w4(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream);
        if (dma || narrow > 1) return hipErrorInvalidValue;
        return narrow ? rp16_d128x1(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream)
                      : rp16_d128x2(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream);
    }
    if (narrow == 1 && (fold & 16) && !dma) return rp16_d64x2ks2(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream);
    if (narrow) {   // small grids: the same pipeline on narrower waves
        if (dma || narrow > 2) return hipErrorInvalidValue;
        return narrow == 1 ? rp16_d64x2(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream)
                           : rp16_d64x1(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream);
    }
    if (dma) {
#ifdef FA_EXPERIMENTS
        return rp16_d64x4_dma(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream);
#else
        return hipErrorInvalidValue;   // the LDS-DMA variant lost the A/B (0.552 vs 0.508 ms): experimental build only
#endif
    }
    return rp16_d64x4(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, f, stream);
}

// Causal forward on the pipeline (folded fast pass first for both input types), D in {64, 128}.
hipError_t rp16_causal_dispatch(const void* Q, const void* K, const void* V, void* O,
                                int BH, int N, int D, float scale, int in_dtype, int out_dtype, hipStream_t stream)
{
    if (!rp16_shape_ok(N, D)) return hipErrorInvalidValue;
    const bool fold = (scale == scale) && scale * kLog2e != 0.0f;
    return D == 64 ? rp16_c_d64(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream)
                   : rp16_c_d128(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

// ... with one wave per SIMD (d = 128 only)
hipError_t rp16_causal_dispatch_1w(const void* Q, const void* K, const void* V, void* O,
                                   int BH, int N, int D, float scale, int in_dtype, int out_dtype, hipStream_t stream)
{
    if (D != 128 || !rp16_shape_ok(N, D)) return hipErrorInvalidValue;
    const bool fold = (scale == scale) && scale * kLog2e != 0.0f;
    return rp16_c_d128w4(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

}  // namespace fa
