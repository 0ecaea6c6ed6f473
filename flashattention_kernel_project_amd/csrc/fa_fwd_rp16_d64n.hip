// fa_fwd_rp16_d64n.hip -- the pipeline at d = 64 on 32- and 16-row waves (small grids) (fa_fwd_rp16_kernel.hpp).
#include "fa_fwd_rp16_kernel.hpp"

namespace fa {

hipError_t rp16_d64x2(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                      bool fold, hipStream_t stream)
{
    return rp16_family<64, 2, false, false>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

hipError_t rp16_d64x1(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                      bool fold, hipStream_t stream)
{
    return rp16_family<64, 1, false, false>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

#ifdef FA_EXPERIMENTS
hipError_t rp16_set_pass_ids_d64n(unsigned* p) { return rp16_set_pass_ids_tu(p); }
#endif

}  // namespace fa
