// fa_fwd_rp16_d64ks.hip -- the pipeline at d = 64 with the keys split over two groups of four 32-row waves per 128-row
// workgroup (fa_fwd_rp16_kernel.hpp, kKeySplit = 2): few heads, long sequences.  (The 2 x 8 waves of 16 rows form of the same
// split was measured too and lost everywhere: 24.7 against 17.3 us at B4 H8 N1024.)
#include "fa_fwd_rp16_kernel.hpp"

namespace fa {

hipError_t rp16_d64x2ks2(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int in_dtype, int out_dtype,
                         bool fold, hipStream_t stream)
{
    return rp16_family<64, 2, false, false, 4, 2>(Q, K, V, O, BH, N, scale, in_dtype, out_dtype, fold, stream);
}

#ifdef FA_EXPERIMENTS
hipError_t rp16_set_pass_ids_d64ks(unsigned* p) { return rp16_set_pass_ids_tu(p); }
#endif

}  // namespace fa
