// fa_fwd_rp16_c.hip -- the pipeline under the causal mask, d = 64 and d = 128 (fa_fwd_rp16_kernel.hpp).
#include "fa_fwd_rp16_kernel.hpp"

namespace fa {

hipError_t rp16_c_d64(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                      bool fold, hipStream_t stream)
{
    return rp16_family<64, 4, false, true>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

hipError_t rp16_c_d128(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                       bool fold, hipStream_t stream)
{
    return rp16_family<128, 2, false, true>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

#ifdef FA_EXPERIMENTS
hipError_t rp16_set_pass_ids_c(unsigned* p) { return rp16_set_pass_ids_tu(p); }
#endif

}  // namespace fa
