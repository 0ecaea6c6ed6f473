/*
 * oracle/attention_cpu.c -- CPU restatement of the reference's attention-forward oracles.
 *
 * TEST INFRASTRUCTURE ONLY (see attention_cpu.h).  Plain C11 + optional OpenMP.
 *
 * What is restated, and from where:
 *   fa_oracle_forward*        general [BH,N,D] softmax(QK^T*scale)V, one query row at a time:
 *                             scores -> row max -> exp/sum -> weighted V sum.  Follows
 *                             GEMM/FlashAttention Forward Fused/flashattn_forward_fused_5_4_2.cu:224-272
 *                             (which hard-codes scale = 1/sqrt(D) and double accumulators; here both
 *                             are parameters because the WMMA kernels take `scale` as an argument,
 *                             FlashAttention/flashattn_forward_wmma/flashattn_forward_wmma.cu:49-58).
 *   fa_oracle_streaming_16x16 M=Kdim=Dv=16 family with K stored [B,16,L], fp32 accumulators,
 *                             running max seeded with -1e30 and 1/(sum+1e-6):
 *                             Streaming_FlashAttention_Forward_Kernel/flashattn_streaming_16x16_mw.cu:252-317
 *                             (constants :38-45).
 *   fa_oracle_transpose_k_16  flashattn_warp_spc/flashattn_streaming_16x16_mw_v8.cu:344-359.
 *   rel-L2 / max-abs          flashattn_streaming_16x16_mw.cu:383-391 / flashattn_forward_fused_5_4_2.cu:366-370.
 */
#include "attention_cpu.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- general [BH,N,D] */

static void one_row_f32(const float* q, const float* Kb, const float* Vb, float* o,
                        int N, int D, float scale, float* w /* [N] scratch */)
{
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) {
        const float* kj = Kb + (size_t)j * D;
        float s = 0.0f;
        for (int c = 0; c < D; ++c) s += q[c] * kj[c];
        s *= scale;
        w[j] = s;
        if (s > mx) mx = s;
    }
    float den = 0.0f;
    for (int j = 0; j < N; ++j) {
        float e = expf(w[j] - mx);
        w[j] = e;
        den += e;
    }
    for (int c = 0; c < D; ++c) o[c] = 0.0f;
    for (int j = 0; j < N; ++j) {
        const float p = w[j] / den;
        const float* vj = Vb + (size_t)j * D;
        for (int c = 0; c < D; ++c) o[c] += p * vj[c];
    }
}

static void one_row_f64(const float* q, const float* Kb, const float* Vb, float* o,
                        int N, int D, float scale, float* w, double* od /* [D] scratch */)
{
    /* The reference rounds each score to float before the max / exp (5_4_2.cu:243-251). */
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) {
        const float* kj = Kb + (size_t)j * D;
        double s = 0.0;
        for (int c = 0; c < D; ++c) s += (double)q[c] * kj[c];
        s *= scale;
        w[j] = (float)s;
        if (w[j] > mx) mx = w[j];
    }
    double den = 0.0;
    for (int j = 0; j < N; ++j) den += exp((double)(w[j] - mx));
    for (int c = 0; c < D; ++c) od[c] = 0.0;
    for (int j = 0; j < N; ++j) {
        const double p = exp((double)(w[j] - mx)) / den;
        const float* vj = Vb + (size_t)j * D;
        for (int c = 0; c < D; ++c) od[c] += p * vj[c];
    }
    for (int c = 0; c < D; ++c) o[c] = (float)od[c];
}

static void forward_rows_impl(const float* Q, const float* K, const float* V, float* O,
                              int BH, int N, int D, float scale, int accum, int nthreads,
                              int bh0, int bh1, int row0, int row1, int causal);

void fa_oracle_forward_rows(const float* Q, const float* K, const float* V, float* O,
                            int BH, int N, int D, float scale, int accum, int nthreads,
                            int bh0, int bh1, int row0, int row1)
{
    forward_rows_impl(Q, K, V, O, BH, N, D, scale, accum, nthreads, bh0, bh1, row0, row1, 0);
}

/* Causal (lower-triangular) variant: query row i attends to keys 0..i, i.e. it is the
 * non-causal row oracle over the key prefix of length i+1.  Not in the reference (SURVEY 8(f)
 * rank 1): pinned only through that identity with the pinned non-causal oracle. */
void fa_oracle_forward_causal_rows(const float* Q, const float* K, const float* V, float* O,
                                   int BH, int N, int D, float scale, int accum, int nthreads,
                                   int bh0, int bh1, int row0, int row1)
{
    forward_rows_impl(Q, K, V, O, BH, N, D, scale, accum, nthreads, bh0, bh1, row0, row1, 1);
}

static void forward_rows_impl(const float* Q, const float* K, const float* V, float* O,
                              int BH, int N, int D, float scale, int accum, int nthreads,
                              int bh0, int bh1, int row0, int row1, int causal)
{
    if (BH <= 0 || N <= 0 || D <= 0) return;
    if (bh0 < 0) bh0 = 0;
    if (bh1 > BH) bh1 = BH;
    if (row0 < 0) row0 = 0;
    if (row1 > N) row1 = N;
    if (bh1 <= bh0 || row1 <= row0) return;
    const long nrows = (long)(bh1 - bh0) * (row1 - row0);
    const int rows_per = row1 - row0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* w = (float*)malloc((size_t)N * sizeof(float));
        double* od = (double*)malloc((size_t)D * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (long r = 0; r < nrows; ++r) {
            const int bh = bh0 + (int)(r / rows_per);
            const int i = row0 + (int)(r % rows_per);
            const size_t base = (size_t)bh * N * D;
            const float* q = Q + base + (size_t)i * D;
            float* o = O + base + (size_t)i * D;
            const int nkeys = causal ? i + 1 : N;
            if (accum == 1) one_row_f64(q, K + base, V + base, o, nkeys, D, scale, w, od);
            else            one_row_f32(q, K + base, V + base, o, nkeys, D, scale, w);
        }
        free(w);
        free(od);
    }
}

/* Nq != Nk (no mask): the same row oracle, every query row over all Nk keys.  Q,O [BH,Nq,D];
 * K,V [BH,Nk,D].  Not a reference function (the reference is self-attention); it IS the pinned row
 * oracle applied with a different row count, and equals fa_oracle_forward when Nq == Nk. */
void fa_oracle_forward_cross(const float* Q, const float* K, const float* V, float* O,
                             int BH, int Nq, int Nk, int D, float scale, int accum, int nthreads)
{
    if (BH <= 0 || Nq <= 0 || Nk <= 0 || D <= 0) return;
    const long nrows = (long)BH * Nq;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* w = (float*)malloc((size_t)Nk * sizeof(float));
        double* od = (double*)malloc((size_t)D * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (long r = 0; r < nrows; ++r) {
            const int bh = (int)(r / Nq), i = (int)(r % Nq);
            const float* q = Q + ((size_t)bh * Nq + i) * D;
            float* o = O + ((size_t)bh * Nq + i) * D;
            const size_t kb = (size_t)bh * Nk * D;
            if (accum == 1) one_row_f64(q, K + kb, V + kb, o, Nk, D, scale, w, od);
            else            one_row_f32(q, K + kb, V + kb, o, Nk, D, scale, w);
        }
        free(w);
        free(od);
    }
}

void fa_oracle_forward(const float* Q, const float* K, const float* V, float* O,
                       int BH, int N, int D, float scale, int accum, int nthreads)
{
    fa_oracle_forward_rows(Q, K, V, O, BH, N, D, scale, accum, nthreads, 0, BH, 0, N);
}

/* ---------------------------------------------------------------- 16x16 streaming family */

enum { FA16_M = 16, FA16_K = 16, FA16_DV = 16 };
static const float FA16_NEG_LARGE = -1e30f;
static const float FA16_EPS = 1e-6f;

void fa_oracle_streaming_16x16(const float* Q, const float* K, const float* V, float* O,
                               int num_batches, int seq_len, float scale)
{
    if (num_batches <= 0 || seq_len <= 0) return;
    float* p = (float*)malloc((size_t)seq_len * sizeof(float));
    for (int b = 0; b < num_batches; ++b) {
        const float* Qb = Q + (size_t)b * FA16_M * FA16_K;
        const float* Kb = K + (size_t)b * FA16_K * seq_len;   /* [Kdim, L] */
        const float* Vb = V + (size_t)b * seq_len * FA16_DV;  /* [L, Dv]   */
        float* Ob = O + (size_t)b * FA16_M * FA16_DV;
        for (int i = 0; i < FA16_M; ++i) {
            float mx = FA16_NEG_LARGE;
            for (int j = 0; j < seq_len; ++j) {
                float s = 0.0f;
                for (int k = 0; k < FA16_K; ++k) s += Qb[i * FA16_K + k] * Kb[(size_t)k * seq_len + j];
                s *= scale;
                p[j] = s;
                if (s > mx) mx = s;
            }
            float den = 0.0f;
            for (int j = 0; j < seq_len; ++j) {
                p[j] = expf(p[j] - mx);
                den += p[j];
            }
            const float inv = 1.0f / (den + FA16_EPS);
            for (int j = 0; j < seq_len; ++j) p[j] *= inv;
            for (int d = 0; d < FA16_DV; ++d) {
                float acc = 0.0f;
                for (int j = 0; j < seq_len; ++j) acc += p[j] * Vb[(size_t)j * FA16_DV + d];
                Ob[i * FA16_DV + d] = acc;
            }
        }
    }
    free(p);
}

void fa_oracle_transpose_k_16(const float* K, float* K_T, int num_batches, int seq_len)
{
    for (int b = 0; b < num_batches; ++b)
        for (int k = 0; k < FA16_K; ++k)
            for (int j = 0; j < seq_len; ++j)
                K_T[((size_t)b * seq_len + j) * FA16_K + k] = K[((size_t)b * FA16_K + k) * seq_len + j];
}

/* ---------------------------------------------------------------- synthetic inputs */

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline double u01(uint64_t bits) /* (0,1) */
{
    return ((double)(bits >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

void fa_oracle_fill(float* dst, size_t n, uint64_t seed, uint64_t offset, int dist)
{
    const uint64_t key = splitmix64(seed ^ 0xA5A5A5A55A5A5A5Aull);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t ctr = offset + i;
        const uint64_t a = splitmix64(key + 2 * ctr);
        if (dist == 1) {
            dst[i] = (float)(2.0 * u01(a) - 1.0);
        } else {
            const uint64_t b = splitmix64(key + 2 * ctr + 1);
            const double r = sqrt(-2.0 * log(u01(a)));
            dst[i] = (float)(r * cos(6.283185307179586476925 * u01(b)));
        }
    }
}

/* ---------------------------------------------------------------- 16-bit float formats */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static uint16_t enc_bf16(float f)
{
    uint32_t u = f2u(f);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u); /* NaN */
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float dec_bf16(uint16_t h) { return u2f((uint32_t)h << 16); }

static uint16_t enc_f16(float f)
{
    const uint32_t u = f2u(f);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t ax = u & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | (ax > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);     /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                   /* < 2^-25 (or == 2^-25 tie) -> 0 */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;                 /* 24-bit significand */
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                 /* bits dropped */
    uint32_t half_m = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
    uint32_t out;
    if (e < -14) out = half_m;                                     /* subnormal (may carry into normal) */
    else out = ((uint32_t)(e + 15) << 10) + (half_m - 0x400u);     /* carry propagates into exponent */
    return (uint16_t)(sign | out);
}
static float dec_f16(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0x1Fu) return u2f(sign | 0x7F800000u | (m << 13));
    if (e == 0) {
        if (!m) return u2f(sign);
        const float v = (float)m * (1.0f / 16777216.0f);           /* m * 2^-24 */
        return sign ? -v : v;
    }
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

void fa_oracle_encode16(const float* x, uint16_t* out, size_t n, int fmt)
{
    for (size_t i = 0; i < n; ++i) out[i] = fmt == 1 ? enc_bf16(x[i]) : enc_f16(x[i]);
}
void fa_oracle_decode16(const uint16_t* in, float* out, size_t n, int fmt)
{
    for (size_t i = 0; i < n; ++i) out[i] = fmt == 1 ? dec_bf16(in[i]) : dec_f16(in[i]);
}
void fa_oracle_round_through(float* x, size_t n, int fmt)
{
    for (size_t i = 0; i < n; ++i) x[i] = fmt == 1 ? dec_bf16(enc_bf16(x[i])) : dec_f16(enc_f16(x[i]));
}

/* ---------------------------------------------------------------- metrics */

double fa_oracle_rel_l2(const float* got, const float* ref, size_t n)
{
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = (double)got[i] - (double)ref[i];
        num += d * d;
        den += (double)ref[i] * (double)ref[i];
    }
    return sqrt(num / (den + 1e-12));
}

double fa_oracle_max_abs(const float* got, const float* ref, size_t n)
{
    double m = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = fabs((double)got[i] - (double)ref[i]);
        if (d != d) return NAN; /* a NaN difference poisons the result */
        if (d > m) m = d;
    }
    return m;
}
