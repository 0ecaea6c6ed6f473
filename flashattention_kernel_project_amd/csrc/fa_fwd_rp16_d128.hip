// fa_fwd_rp16_d128.hip -- the pipeline at d = 128 on 32-row waves (256-row workgroups, with the redo kernel on 16-row waves) and on 16-row waves (fa_fwd_rp16_kernel.hpp).
#include "fa_fwd_rp16_kernel.hpp"

namespace fa {

hipError_t rp16_d128x2(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                       bool fold, hipStream_t stream)
{
    return rp16_family<128, 2, false, false>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

hipError_t rp16_d128x1(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                       bool fold, hipStream_t stream)
{
    return rp16_family<128, 1, false, false>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

#ifdef FA_EXPERIMENTS
hipError_t rp16_set_pass_ids_d128(unsigned* p) { return rp16_set_pass_ids_tu(p); }
#endif

}  // namespace fa
