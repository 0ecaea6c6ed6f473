// fa_fwd_rp16_d64.hip -- the pipeline at d = 64 on 64-row waves (512-row workgroups) and its half-width redo kernel (fa_fwd_rp16_kernel.hpp).
#include "fa_fwd_rp16_kernel.hpp"

namespace fa {

hipError_t rp16_d64x4(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                      bool fold, hipStream_t stream)
{
    return rp16_family<64, 4, false, false>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

#ifdef FA_EXPERIMENTS
hipError_t rp16_d64x4_dma(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                          bool fold, hipStream_t stream)
{
    return rp16_family<64, 4, true, false>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}
#endif

#ifdef FA_EXPERIMENTS
hipError_t rp16_set_pass_ids_d64(unsigned* p) { return rp16_set_pass_ids_tu(p); }
#endif

}  // namespace fa
