/*
 * fa_mi355.h -- C ABI of libfa_mi355.so: FlashAttention forward for AMD Instinct MI355X (gfx950).
 *
 * The reference (jeehun98/FlashAttention_Kernel_Project) has no library and nothing `extern "C"`:
 * every kernel is a C++ `__global__` launched with <<<grid,block[,smem]>>> from its own main().
 * What the reference fixes is each kernel's ARGUMENT LIST, tensor layouts, dtypes and grid
 * convention.  Each entry point below keeps one of those argument lists in the same order and
 * adds a trailing stream handle (the reference always uses the null stream: pass NULL).
 *
 * All pointers are DEVICE pointers owned by the caller; nothing is allocated or freed here; O is
 * fully overwritten (no need to pre-zero).  Launches are asynchronous with respect to the host.
 * Return value: a hipError_t as int (0 = hipSuccess).  Unsupported shapes return
 * hipErrorInvalidValue (1) -- the reference kernels silently `return` instead
 * (flashattn_forward_wmma.cu:59-63); the library never calls exit().
 *
 * `stream` is a hipStream_t passed as void*.
 */
#ifndef FA_MI355_H
#define FA_MI355_H

#include <stddef.h>   /* size_t (split-KV workspace) */

#ifdef __cplusplus
extern "C" {
#endif

#define FA_DTYPE_F16  0
#define FA_DTYPE_BF16 1
#define FA_OUT_F32    0   /* the reference's output type */
#define FA_OUT_SAME   1   /* output in the input's 16-bit type */

/* Kernel selection for fa_forward_ex().  Ids present in the product library: */
#define FA_ALGO_AUTO            0 /* d=64, N > 256 and d=128: RP16_FOLD on the widest waves whose grid still covers the device
                                     (24, else _HALF 26, else _QUARTER 27 / _KS2 29, by rounds x rows / efficiency); d=64, N <= 256:
                                     INTERLEAVED / _2WG; else GENERIC -- fa_selected_algo() */
#define FA_ALGO_GENERIC         1 /* single 16x16 MFMA fragment per wave, any D % 16 == 0, D <= 256 */
#define FA_ALGO_TILED           2 /* LDS-staged 256-row workgroups, QK^T -> softmax -> PV per tile, D in {64,128} */
#define FA_ALGO_INTERLEAVED     5 /* QK^T one tile ahead, PV one tile behind, one MFMA per slice of softmax VALU, D = 64 */
#define FA_ALGO_INTERLEAVED_2WG 6 /* the same with 128-row workgroups, two per CU, D = 64 */
#define FA_ALGO_RP16           23 /* rolling half-tile pipeline on v_mfma_f32_16x16x32 (four 16-row blocks per wave at D = 64, two at D = 128),
                                     branch-free steady state, single-instruction fp32 vector work, exact passes */
#define FA_ALGO_RP16_FOLD      24 /* RP16 with the folded fast pass: scale folded into a rounded Q, the wave's reference max as the
                                     accumulators' start value (bf16: K converted to fp16 while it is staged), row sums on the matrix
                                     pipe; per-workgroup fallback chain: exact optimistic pass, then the tracked pass */
#define FA_ALGO_RP16_FOLD_HALF 26 /* RP16_FOLD on half-width waves (256-row workgroups at D = 64, 128-row at D = 128): grids too
                                     small to cover the device with the full-width ones */
#define FA_ALGO_RP16_FOLD_QUARTER 27 /* ... on quarter-width waves (128-row workgroups), D = 64 */
#define FA_ALGO_RP16_FOLD_KS2  29 /* RP16_FOLD, D = 64, 128-row workgroups: two groups of four 32-row waves, each on half the keys, merged through LDS
                                     (N % 128 == 0; other N run _QUARTER); AUTO for few heads and N >= 2048 */
#define FA_ALGO_RP16_FOLD_1W   28 /* RP16_FOLD at D = 128 with ONE wave per SIMD: four 64-row waves per 256-row workgroup, 512 registers each,
                                     every LDS fragment feeding four matrix instructions; AUTO at D = 128 from N = 4096; also accepted by fa_forward_causal (D = 128; AUTO there from N = 8192 on grids of >= 4 rounds) */
/* Only in the experimental build (`make experimental`, fa_mi355_has_experiments() == 1; hipErrorInvalidValue otherwise):
 * A/B kernels that AUTO never selects. */
#define FA_ALGO_W64            13 /* round 1's default for bf16: 64 query rows per wave, phase-ordered stream on 32x32x16, packed fp32 */
#define FA_ALGO_W64X           16 /* round 1's default for fp16: the W64 stream on v_mfma_f32_16x16x32 */
#define FA_ALGO_RP             21 /* the rolling pipeline on 32x32x16 (two 32-row blocks per wave), exact passes */
#define FA_ALGO_RP_FOLD        22 /* RP with the folded fast pass (fp16, D = 64) */
#define FA_ALGO_W64P           14 /* W64 with a half-tile rolling pipeline, packed fp32 (round 1's form of RP), D in {64,128} */
#define FA_ALGO_RP16_DMA       25 /* RP16_FOLD with K/V staged by LDS-DMA (buffer_load ... lds) instead of through registers */
#define FA_ALGO_SK             17 /* skewed halves: waves 4-7 half an iteration behind waves 0-3, folded fast pass; 18 exact, 19/20 lock-step */
/* 7, 8: occupancy variants of TILED (fp16, d=64).  3, 4, 9-12, 15 (round 1 / 2 A/B kernels two generations stale) were removed in round 3. */

/* General-shape forward.  Replaces
 *   flashattn_forward_wmma_kernel(const half* Q, const half* K, const half* V, float* O,
 *                                 int BH, int N, int D, float scale)
 *   FlashAttention/flashattn_forward_wmma/flashattn_forward_wmma.cu:49-58 (and _v2.cu:53, _v3.cu:52,
 *   _v4.cu:52, flashattn_forward_memory_bound/flashattn_forward_wmma_v5_cp_async.cu:99,
 *   flashattn_forward_wmma_memprofile.cu:60), launched as <<<(ceil(N/BLOCK_M), BH), block, smem>>>
 *   (flashattn_forward_wmma.cu:389-413).
 * Q,K,V [BH,N,D] row-major fp16, O [BH,N,D] fp32, self-attention, no mask.
 * D % 16 == 0 (as the reference requires, :63), D <= 256; any N >= 1 (tail rows/keys handled). */
int flashattn_forward_wmma(const void* Q, const void* K, const void* V, float* O,
                           int BH, int N, int D, float scale, void* stream);

/* The same operation with the dtype / output / kernel choices BASELINE's configs need
 * (bf16 inputs, 16-bit outputs, B and H separate).  BH = B*H.  Same layouts as above. */
int fa_forward(const void* Q, const void* K, const void* V, void* O,
               int B, int H, int N, int d, float scale,
               int in_dtype, int out_dtype, void* stream);
int fa_forward_ex(const void* Q, const void* K, const void* V, void* O,
                  int B, int H, int N, int d, float scale,
                  int in_dtype, int out_dtype, int algo, void* stream);

/* Causal (lower-triangular) self-attention: query row i attends to keys 0..i.  Same layouts and
 * dtypes as fa_forward.  NOT a reference entry point: the reference has no mask; this is the first
 * "next" row of SURVEY.md 8(f) (cf. the runtime-M tail masking of
 * flashattn_warp_spc/flashattn_streaming_16x16_mw_v12d.cu:100-135).  algo: FA_ALGO_AUTO,
 * FA_ALGO_GENERIC, FA_ALGO_TILED (256-row workgroups), 6 (the tiled kernel with 128-row workgroups, two
 * per CU), FA_ALGO_RP16_FOLD (the pipeline under the mask; AUTO's choice whenever the grid gives every CU a
 * workgroup) or FA_ALGO_RP16_FOLD_1W (the same with one wave per SIMD, D = 128 only; AUTO's choice there from N = 8192 on
 * grids of >= 4 rounds); all but the first two need D in {64,128}.  (FA_ALGO_W64 under the mask: experimental build only.) */
int fa_forward_causal(const void* Q, const void* K, const void* V, void* O,
                      int B, int H, int N, int d, float scale,
                      int in_dtype, int out_dtype, int algo, void* stream);

/* Nq != Nk with a split over the keys ("flash-decoding"): Q [B*H,Nq,d], K,V [B*H,Nk,d], O [B*H,Nq,d],
 * no mask, d in {64,128}.  Meant for few query rows against a long K/V, where one workgroup per
 * (head, query block) cannot fill the chip: the keys are cut into S chunks (S chosen by the library
 * from B*H, Nq, Nk), each chunk leaves an unnormalised partial result in `workspace`, and a second
 * small kernel merges them.  The caller owns the workspace (size from
 * fa_forward_splitkv_workspace_bytes(); 0 means S = 1 and `workspace` may be NULL).
 * Grouped-query attention needs no separate entry: with Q [B,Hq,Nq,d] and K,V [B,Hkv,Nk,d], the
 * G = Hq/Hkv query heads of a group are contiguous, so pass H = Hkv and Nq = G*Nq -- the group's K/V is
 * then streamed once for all G heads.
 * NOT a reference entry point (cf. the single-query experiment
 * flashattn_warp_spc_2/flashattn_streaming_16x16_mw_v7_5*.cu). */
size_t fa_forward_splitkv_workspace_bytes(int B, int H, int Nq, int Nk, int d);
int fa_forward_splitkv(const void* Q, const void* K, const void* V, void* O,
                       int B, int H, int Nq, int Nk, int d, float scale,
                       int in_dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Stage-level debug entry (cf. the reference's single-stage experiments
 * FlashAttention/t16/ *debug*.cu): one stage of the tiled forward with its result in memory, through
 * the same LDS images, fragment loads and accumulator maps as the product kernels.  d in {64,128}.
 *   stage 1: A = Q, B = K [BH,N,d] 16-bit          -> Out = S = scale*Q.K^T  [BH,N,N] fp32
 *   stage 2: A = S [BH,N,N] fp32, B = NULL         -> Out = P = softmax rows of S, 16-bit [BH,N,N]
 *   stage 3: A = P [BH,N,N] 16-bit, B = V [BH,N,d] -> Out = O = P.V  [BH,N,d] fp32
 * Not a product path and not tuned. */
int fa_debug_stage(int stage, const void* A, const void* B, void* Out, int BH, int N, int d, float scale,
                   int dtype, void* stream);

/* 16x16 streaming family.  Replaces
 *   flashattn_streaming_16x16_kernel_mw(const __half* Q, const __half* K, const __half* V, float* O,
 *                                       int num_batches, int seq_len, float scale)
 *   Streaming_FlashAttention_Forward_Kernel/flashattn_streaming_16x16_mw.cu:73-81 (same list in
 *   _mw_fixed.cu:78, _mw_v2.cu:75, _mw_cpasync.cu:73, flashattn_warp_spc/..._v3..v7), launched as
 *   <<<num_batches, 64>>> (mw.cu:368-377).
 * Q [B,16,16], K [B,16,L] (k-major), V [B,L,16] fp16; O [B,16,16] fp32; O = y/(l+1e-6). */
int flashattn_streaming_16x16_mw(const void* Q, const void* K, const void* V, float* O,
                                 int num_batches, int seq_len, float scale, void* stream);

/* v8+ ABI of the same family: second pointer is K_T [B,L,16], produced by the host pre-transpose
 *   flashattn_streaming_16x16_kernel_mw_v8(const __half* Q, const __half* K_T, const __half* V, float* O,
 *                                          int num_batches, int seq_len, float scale)
 *   flashattn_warp_spc/flashattn_streaming_16x16_mw_v8.cu:103-111 (v10.cu:104, v11.cu:101). */
int flashattn_streaming_16x16_mw_kt(const void* Q, const void* K_T, const void* V, float* O,
                                    int num_batches, int seq_len, float scale, void* stream);

/* Library identification: "fa_mi355 <version> gfx950". */
const char* fa_mi355_version(void);

/* What FA_ALGO_AUTO resolves to for a shape on the CURRENT device (an FA_ALGO_* id; -1 for bad arguments), and the
 * name of the kernel template an algo id launches there (prefix of its rocprofv3 kernel-trace name; "" if unknown).
 * bench.py names its dominant kernel with these instead of hard-coding it. */
int fa_selected_algo(int B, int H, int N, int d, int in_dtype);
const char* fa_selected_kernel(int B, int H, int N, int d, int in_dtype, int algo);

/* 1 when the library was built with the experimental A/B kernels (`make experimental`: explicit algo ids
 * 3, 4, 7-12, 14, 15 and the measurement entry points), 0 for the product build, where those ids return
 * hipErrorInvalidValue. */
int fa_mi355_has_experiments(void);

#ifdef __cplusplus
}
#endif
#endif
