// fa_capi.hip -- extern "C" launchers declared in include/fa_mi355.h.
#include <hip/hip_runtime.h>
#include "../../include/fa_mi355.h"

namespace fa {
hipError_t forward_dispatch(const void* Q, const void* K, const void* V, void* O,
                            int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                            int algo, hipStream_t stream);
hipError_t forward_causal_dispatch(const void* Q, const void* K, const void* V, void* O,
                                   int BH, int N, int D, float scale, int in_dtype, int out_dtype,
                                   int algo, hipStream_t stream);
hipError_t split_dispatch(const void* Q, const void* K, const void* V, void* O, void* ws, size_t ws_bytes,
                          int BH, int Nq, int Nk, int D, float scale, int in_dtype, int out_dtype, hipStream_t stream);
size_t split_workspace_bytes(int BH, int Nq, int Nk, int D);
hipError_t debug_stage_dispatch(int stage, const void* A, const void* B, void* Out, int BH, int N, int D, float scale,
                                int dtype, hipStream_t stream);
hipError_t streaming16_dispatch(const void* Q, const void* K, const void* V, float* O,
                                int num_batches, int seq_len, float scale, bool k_transposed,
                                hipStream_t stream);
int auto_algo(int BH, int N, int D, int in_dtype);
const char* algo_kernel_name(int algo, int D);
#ifdef FA_EXPERIMENTS
hipError_t il_diag_dispatch(const void* Q, const void* K, const void* V, void* O,
                            int BH, int N, float scale, unsigned long long* diag, int waves, hipStream_t stream);
hipError_t sk_diag_dispatch(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int variant,
                            unsigned long long* diag, hipStream_t stream);
hipError_t rp16_set_pass_ids(unsigned* dev_ptr);
hipError_t lab_w64x_dispatch(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale,
                             int kstruct, int abl, unsigned long long* diag, hipStream_t stream);
#endif
}  // namespace fa

// The library is built with -fvisibility=hidden: only the entry points below leave it.
#define FA_EXPORT __attribute__((visibility("default")))

extern "C" {

#ifdef FA_EXPERIMENTS
// ---- libfa_mi355_exp.so only (make experimental): measurement entry points, not in the public header ----
// fa_fwd_w64x stream, instantiated per experiment (fa_lab_w64x.hip): kstruct 0 shipped order, 1 two half-iterations
// per tile, 3 the same with waves 4-7 half an iteration behind; abl = timing-ablation bits; diag = per-phase stamps.
FA_EXPORT int fa_lab_w64x(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale,
                int kstruct, int abl, unsigned long long* diag, void* stream)
{
    return (int)fa::lab_w64x_dispatch(Q, K, V, O, BH, N, scale, kstruct, abl, diag, static_cast<hipStream_t>(stream));
}

// per-block pass ids of the fa_fwd_rp16 kernels (see g_rp16_pass_ids); nullptr switches the recording off again
FA_EXPORT int fa_lab_rp16_pass_ids(unsigned* dev_ids) { return (int)fa::rp16_set_pass_ids(dev_ids); }

// fa_fwd_sk (fp16 -> fp32, d = 64) with per-phase s_memtime stamps: diag[wg][wave][8]; variant 0 shipped, 1 no fold, 2 no skew, 3 neither
FA_EXPORT int fa_lab_sk(const void* Q, const void* K, const void* V, void* O, int BH, int N, float scale, int variant,
              unsigned long long* diag, void* stream)
{
    return (int)fa::sk_diag_dispatch(Q, K, V, O, BH, N, scale, variant, diag, static_cast<hipStream_t>(stream));
}

// compute / stage-wait / barrier time of the interleaved kernel.
FA_EXPORT int fa_debug_il_times(const void* Q, const void* K, const void* V, void* O,
                      int BH, int N, float scale, unsigned long long* diag, int waves, void* stream)
{
    return (int)fa::il_diag_dispatch(Q, K, V, O, BH, N, scale, diag, waves, static_cast<hipStream_t>(stream));
}

#endif  // FA_EXPERIMENTS

// 1 when this build carries the experimental A/B kernels (explicit algo ids 3, 4, 7-12, 14, 15), else 0.
FA_EXPORT int fa_mi355_has_experiments(void)
{
#ifdef FA_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

FA_EXPORT int flashattn_forward_wmma(const void* Q, const void* K, const void* V, float* O,
                           int BH, int N, int D, float scale, void* stream)
{
    return (int)fa::forward_dispatch(Q, K, V, O, BH, N, D, scale, FA_DTYPE_F16, FA_OUT_F32, FA_ALGO_AUTO,
                                     static_cast<hipStream_t>(stream));
}

FA_EXPORT int fa_forward_ex(const void* Q, const void* K, const void* V, void* O,
                  int B, int H, int N, int d, float scale,
                  int in_dtype, int out_dtype, int algo, void* stream)
{
    if (B <= 0 || H <= 0 || (long long)B * H > 0x7FFFFFFFll) return (int)hipErrorInvalidValue;
    if (algo < FA_ALGO_AUTO || algo > 29) return (int)hipErrorInvalidValue;
    if (out_dtype != FA_OUT_F32 && out_dtype != FA_OUT_SAME) return (int)hipErrorInvalidValue;
    return (int)fa::forward_dispatch(Q, K, V, O, B * H, N, d, scale, in_dtype, out_dtype, algo,
                                     static_cast<hipStream_t>(stream));
}

FA_EXPORT int fa_forward(const void* Q, const void* K, const void* V, void* O,
               int B, int H, int N, int d, float scale,
               int in_dtype, int out_dtype, void* stream)
{
    return fa_forward_ex(Q, K, V, O, B, H, N, d, scale, in_dtype, out_dtype, FA_ALGO_AUTO, stream);
}

FA_EXPORT int fa_forward_causal(const void* Q, const void* K, const void* V, void* O,
                      int B, int H, int N, int d, float scale,
                      int in_dtype, int out_dtype, int algo, void* stream)
{
    if (B <= 0 || H <= 0 || (long long)B * H > 0x7FFFFFFFll) return (int)hipErrorInvalidValue;
    if (out_dtype != FA_OUT_F32 && out_dtype != FA_OUT_SAME) return (int)hipErrorInvalidValue;
    return (int)fa::forward_causal_dispatch(Q, K, V, O, B * H, N, d, scale, in_dtype, out_dtype, algo,
                                            static_cast<hipStream_t>(stream));
}

FA_EXPORT size_t fa_forward_splitkv_workspace_bytes(int B, int H, int Nq, int Nk, int d)
{
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || (long long)B * H > 0x7FFFFFFFll) return 0;
    return fa::split_workspace_bytes(B * H, Nq, Nk, d);
}

FA_EXPORT int fa_forward_splitkv(const void* Q, const void* K, const void* V, void* O,
                       int B, int H, int Nq, int Nk, int d, float scale,
                       int in_dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream)
{
    if (B <= 0 || H <= 0 || (long long)B * H > 0x7FFFFFFFll) return (int)hipErrorInvalidValue;
    return (int)fa::split_dispatch(Q, K, V, O, workspace, workspace_bytes, B * H, Nq, Nk, d, scale, in_dtype, out_dtype,
                                   static_cast<hipStream_t>(stream));
}

FA_EXPORT int fa_debug_stage(int stage, const void* A, const void* B, void* Out, int BH, int N, int d, float scale,
                   int dtype, void* stream)
{
    return (int)fa::debug_stage_dispatch(stage, A, B, Out, BH, N, d, scale, dtype, static_cast<hipStream_t>(stream));
}

FA_EXPORT int flashattn_streaming_16x16_mw(const void* Q, const void* K, const void* V, float* O,
                                 int num_batches, int seq_len, float scale, void* stream)
{
    return (int)fa::streaming16_dispatch(Q, K, V, O, num_batches, seq_len, scale, false,
                                         static_cast<hipStream_t>(stream));
}

FA_EXPORT int flashattn_streaming_16x16_mw_kt(const void* Q, const void* K_T, const void* V, float* O,
                                    int num_batches, int seq_len, float scale, void* stream)
{
    return (int)fa::streaming16_dispatch(Q, K_T, V, O, num_batches, seq_len, scale, true,
                                         static_cast<hipStream_t>(stream));
}

FA_EXPORT int fa_selected_algo(int B, int H, int N, int d, int in_dtype)
{
    if (B <= 0 || H <= 0 || N <= 0 || d <= 0 || (long long)B * H > 0x7FFFFFFFll) return -1;
    return fa::auto_algo(B * H, N, d, in_dtype);
}

FA_EXPORT const char* fa_selected_kernel(int B, int H, int N, int d, int in_dtype, int algo)
{
    if (algo == FA_ALGO_AUTO) algo = fa_selected_algo(B, H, N, d, in_dtype);
    return algo < 0 ? "" : fa::algo_kernel_name(algo, d);
}

FA_EXPORT const char* fa_mi355_version(void) { return "fa_mi355 0.3.1 gfx950"; }

}  // extern "C"
