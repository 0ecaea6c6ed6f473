/*
 * oracle/attention_cpu.h -- CPU restatement of the reference's attention-forward oracles.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The shipped path is libfa_mi355.so (HIP, gfx950) and it never calls in here.
 *
 * Parity pin: fa_oracle_forward() is checked against the reference's own general-shape CPU
 * oracle, compiled from where it lies (oracle/build_ref.sh -> oracle/_ref/libref_cpu.so), and
 * against the committed vectors in tests/golden/ that were produced by that build.
 */
#ifndef FA_ORACLE_ATTENTION_CPU_H
#define FA_ORACLE_ATTENTION_CPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* General-shape naive 3-loop attention forward, Q,K,V,O: [BH, N, D] row-major fp32 (the caller
 * has already rounded Q/K/V to fp16/bf16-representable values).  accum: 0 = fp32 accumulators
 * (north-star's "naive 3-loop fp32 CPU reference"), 1 = double accumulators (the reference's
 * own choice, GEMM/FlashAttention Forward Fused/flashattn_forward_fused_5_4_2.cu:224-272).
 * nthreads <= 1 runs single-threaded; otherwise OpenMP over (bh,row). */
void fa_oracle_forward(const float* Q, const float* K, const float* V, float* O,
                       int BH, int N, int D, float scale, int accum, int nthreads);

/* Same, but only rows [row0,row1) of heads [bh0,bh1) are computed (O is still indexed as the
 * full [BH,N,D] tensor).  Used for bounded CPU-baseline samples and big-shape spot checks. */
void fa_oracle_forward_rows(const float* Q, const float* K, const float* V, float* O,
                            int BH, int N, int D, float scale, int accum, int nthreads,
                            int bh0, int bh1, int row0, int row1);

/* Causal variant (query row i sees keys 0..i): the row oracle above over the key prefix i+1.
 * Not a reference function; see the .c file. */
void fa_oracle_forward_causal_rows(const float* Q, const float* K, const float* V, float* O,
                                   int BH, int N, int D, float scale, int accum, int nthreads,
                                   int bh0, int bh1, int row0, int row1);

/* Nq != Nk, no mask: Q,O [BH,Nq,D]; K,V [BH,Nk,D] (the row oracle over all Nk keys; see the .c file). */
void fa_oracle_forward_cross(const float* Q, const float* K, const float* V, float* O,
                             int BH, int Nq, int Nk, int D, float scale, int accum, int nthreads);

/* 16x16 streaming family oracle: Q [B,16,16], K [B,16,L] (k-major), V [B,L,16], O [B,16,16];
 * softmax normalised as 1/(sum + 1e-6), running max seeded with -1e30
 * (Streaming_FlashAttention_Forward_Kernel/flashattn_streaming_16x16_mw.cu:252-317). */
void fa_oracle_streaming_16x16(const float* Q, const float* K, const float* V, float* O,
                               int num_batches, int seq_len, float scale);

/* Host-side K pre-transpose of the v8+ ABI: K [B,16,L] -> K_T [B,L,16]
 * (flashattn_warp_spc/flashattn_streaming_16x16_mw_v8.cu:344-359). */
void fa_oracle_transpose_k_16(const float* K, float* K_T, int num_batches, int seq_len);

/* Portable synthetic inputs: counter-based splitmix64 + Box-Muller N(0,1) / U(-1,1).
 * Element i of the stream `seed` is a pure function of (seed, i): Q,K,V are drawn as
 * consecutive ranges of one stream (order Q -> K -> V as in the reference drivers,
 * flashattn_streaming_16x16_mw.cu:332-349). dist: 0 = N(0,1), 1 = U(-1,1). */
void fa_oracle_fill(float* dst, size_t n, uint64_t seed, uint64_t offset, int dist);

/* Round-to-nearest-even through IEEE binary16 / bfloat16 and back to fp32 (in place).
 * fmt: 0 = fp16, 1 = bf16. Also the raw 16-bit encoders used to build device inputs. */
void fa_oracle_round_through(float* x, size_t n, int fmt);
void fa_oracle_encode16(const float* x, uint16_t* out, size_t n, int fmt);
void fa_oracle_decode16(const uint16_t* in, float* out, size_t n, int fmt);

/* Error metrics the reference prints: relative L2 (flashattn_streaming_16x16_mw.cu:383-391)
 * and max-abs (flashattn_forward_fused_5_4_2.cu:366-370). */
double fa_oracle_rel_l2(const float* got, const float* ref, size_t n);
double fa_oracle_max_abs(const float* got, const float* ref, size_t n);

#ifdef __cplusplus
}
#endif
#endif
