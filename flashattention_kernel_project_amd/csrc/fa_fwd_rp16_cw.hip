// fa_fwd_rp16_cw.hip -- the pipeline under the causal mask at d = 128 with ONE wave per SIMD (four 64-row waves per 256-row
// workgroup, fa_fwd_rp16_kernel.hpp, kWv = 4): the causal twin of fa_fwd_rp16_d128w.hip.
#include "fa_fwd_rp16_kernel.hpp"

namespace fa {

hipError_t rp16_c_d128w4(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                         bool fold, hipStream_t stream)
{
    return rp16_family<128, 4, false, true, 4>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

#ifdef FA_EXPERIMENTS
hipError_t rp16_set_pass_ids_cw(unsigned* p) { return rp16_set_pass_ids_tu(p); }
#endif

}  // namespace fa
