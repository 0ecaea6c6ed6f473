// valu_rate.hip -- gfx950 micro-benchmark: issue cost (cycles per wave-instruction) of the VALU
// operations in the softmax stream, for one wave alone on its SIMD and for two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    unsigned y[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    h8 af, bf;
    for (int i = 0; i < 8; ++i) { af[i] = (_Float16)(a + i); bf[i] = (_Float16)(b - i); }
    f16v acc16 = {0};
    f16v accs[4] = {};
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc4[4] = {};
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) char lds[16384];
    u4 ld[4] = {};
    u2 ld2[8] = {};
    const unsigned lane = threadIdx.x & 63;
    // conflict-free patterns: b128: lane*16 ; tr_b64: 16-lane group reads a 4x16 block, half-wave a 256-B block
    const unsigned lds_addr = (unsigned)(size_t)lds + ((KIND == 18) ? lane * 16 : ((lane >> 5) * 256 + ((lane >> 2) & 3) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8));
    if (threadIdx.x < 1024) ((unsigned*)lds)[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {   // independent v_fma_f32 (VOP3, 3 VGPR operands)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 1) {   // v_fma_f32 with SGPR multiplier and negated VGPR addend (as emitted)
#define X(i) asm volatile("v_fma_f32 %0, %0, |%1|, -%2" : "+v"(x[i]) : "s"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 2) {   // v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 3) {   // v_add_f32 e32, 4 dependent chains
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i & 3]) : "v"(x[4 + (i & 7)]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 4) {   // v_max3_f32, 4 chains
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "v"(x[4 + (i & 7)]), "v"(x[12 + (i & 3)]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 5) {   // v_cvt_pk_f16_f32
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y[i & 7]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 27) {  // v_cvt_pkrtz_f16_f32 (round toward zero: the pre-gfx950 pack)
#define X(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(y[i & 7]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 28) {  // v_pack_b32_f16 of two fp16 halves
#define X(i) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(y[i & 7]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 6) {   // v_add_f32 fully independent destinations
#define X(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 7) {   // v_mul_f32 e32 independent
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 8) {   // softmax-like mix per 2 elements: 2 fma, 2 exp, 2 add, 1 cvt, 1 max3
#define X(i) asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t" \
                          "v_cvt_pk_f16_f32 %2, %0, %1" : "+v"(x[i]), "+v"(x[(i + 8) & 15]), "=v"(y[i & 7]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if constexpr (KIND == 9) {   // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&x[(2 * i) & 14]) : "v"(*(double*)&x[(2 * i) & 14]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 10) {  // v_exp_f16 (packed halves not available; single)
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 11) {  // v_pk_add_f16
#define X(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 12) {  // v_dot2_f32_f16 accumulate
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[i & 3]) : "v"(x[4 + (i & 7)]), "v"(a));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 14 || KIND == 15 || KIND == 16) {
            // one MFMA 32x32x16 f16 + the softmax VALU slice of one MFMA slot:
            // 2 fma, 2 exp, 1 cvt, 2 add, 1 max3 (KIND 14), the same without the MFMA (15), MFMA only (16)
#define X(i) \
            if (KIND != 15) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16) : "v"(af), "v"(bf)); \
            if (KIND != 16) asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\t" \
                         "v_cvt_pk_f16_f32 %2, %7, %8\n\tv_add_f32 %9, %9, %7\n\tv_add_f32 %10, %10, %8\n\tv_max3_f32 %11, %11, %0, %1" \
                         : "+v"(x[(i) & 3]), "+v"(x[4 + ((i) & 3)]), "=v"(y[(i) & 7]) : "v"(a), "v"(b), \
                           "v"(x[8 + ((i) & 1)]), "v"(x[10 + ((i) & 1)]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]), "v"(x[12 + ((i) & 3)]));
            REP16(X)
#undef X
        } else if constexpr (KIND >= 17 && KIND <= 20) {
            // slot = MFMA + LDS operand reads + 8 VALU; reads are only waited for once per 16 slots
            // 17: 2x ds_read_b64_tr_b16, 18: 1x ds_read_b128, 19: reads only (no MFMA, no VALU), 20: 17 without VALU
#define X(i) \
            if (KIND != 19) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16) : "v"(af), "v"(bf)); \
            if (KIND == 18) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(i) & 3]) : "v"(lds_addr), "i"(((i) & 7) * 1024)); \
            else asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" \
                              : "=v"(ld2[(2 * (i)) & 7]), "=v"(ld2[(2 * (i) + 1) & 7]) : "v"(lds_addr), "i"(((i) & 7) * 1024), "i"(((i) & 7) * 1024 + 512)); \
            if (KIND == 17 || KIND == 18) asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\t" \
                         "v_cvt_pk_f16_f32 %2, %7, %8\n\tv_add_f32 %9, %9, %7\n\tv_add_f32 %10, %10, %8\n\tv_max3_f32 %11, %11, %0, %1" \
                         : "+v"(x[(i) & 3]), "+v"(x[4 + ((i) & 3)]), "=v"(y[(i) & 7]) : "v"(a), "v"(b), \
                           "v"(x[8 + ((i) & 1)]), "v"(x[10 + ((i) & 1)]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]), "v"(x[12 + ((i) & 3)]));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 23 || KIND == 24) {
            // the MFMA + 8 VALU slot again, accumulator in the AGPR half of the file (23) and, for
            // reference, VALU operands spread over many VGPRs as in the real kernel (24, VGPR accumulator)
#define X(i) \
            if (KIND == 23) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc16) : "v"(af), "v"(bf)); \
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accs[(i) & 3]) : "v"(af), "v"(bf)); \
            asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\t" \
                         "v_cvt_pk_f16_f32 %2, %7, %8\n\tv_add_f32 %9, %9, %7\n\tv_add_f32 %10, %10, %8\n\tv_max3_f32 %11, %11, %0, %1" \
                         : "+v"(x[(i) & 3]), "+v"(x[4 + ((i) & 3)]), "=v"(y[(i) & 7]) : "v"(a), "v"(b), \
                           "v"(x[8 + ((i) & 1)]), "v"(x[10 + ((i) & 1)]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]), "v"(x[12 + ((i) & 3)]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 25 || KIND == 26) {
            // 16-row design: slot = one 16x16x32 MFMA + LDS operand reads + its share of the softmax VALU
            // (per 16x64 wave-tile: 18 MFMA, 24 reads, 8 pk_fma + 16 exp + 8 cvt); 25: 2 tr reads per slot,
            // 26: one b128 read per slot
#define X(i) \
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[(i) & 3]) : "v"(af), "v"(bf)); \
            if (KIND == 25) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" \
                              : "=v"(ld2[(2 * (i)) & 7]), "=v"(ld2[(2 * (i) + 1) & 7]) : "v"(lds_addr), "i"(((i) & 7) * 1024), "i"(((i) & 7) * 1024 + 512)); \
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(i) & 3]) : "v"(lds_addr), "i"(((i) & 7) * 1024)); \
            if ((i) & 1) asm volatile("v_exp_f32 %0, %0\n\tv_cvt_pk_f16_f32 %1, %2, %3" : "+v"(x[(i) & 7]), "=v"(y[(i) & 7]) : "v"(x[8 + ((i) & 3)]), "v"(x[12 + ((i) & 3)])); \
            else asm volatile("v_exp_f32 %0, %0\n\tv_pk_fma_f32 %1, %1, %2, %2" : "+v"(x[(i) & 7]), "+v"(*(double*)&x[8 + 2 * ((i) & 3)]) : "v"(*(double*)&x[8 + 2 * ((i) & 3)]));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 21) {   // MFMA, 4 independent accumulators
#define X(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accs[(i) & 3]) : "v"(af), "v"(bf));
            REP16(X)
#undef X
        } else if constexpr (KIND == 22) {   // MFMA 16x16x32, 4 independent accumulators
#define X(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[(i) & 3]) : "v"(af), "v"(bf));
            REP16(X)
#undef X
        } else if constexpr (KIND == 13) {  // v_exp_f32 interleaved 1:1 with independent v_fma (co-issue test)
#define X(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x[i & 7]), "+v"(x[8 + (i & 7)]) : "v"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += (float)y[i];
    for (int i = 0; i < 16; ++i) acc += acc16[i] + accs[0][i] + accs[1][i] + accs[2][i] + accs[3][i];
    for (int i = 0; i < 4; ++i) acc += acc4[0][i] + acc4[1][i] + acc4[2][i] + acc4[3][i];
    for (int i = 0; i < 4; ++i) acc += (float)ld[i][0];
    for (int i = 0; i < 8; ++i) acc += (float)ld2[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
        cyc[4096 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0;
    }
}

template <int KIND>
void run(const char* name, int per_iter)
{
    const int iters = 2000, nblk = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nblk * 1024 * 4);
    hipMalloc(&cyc, 8192 * 8);
    for (int threads : {256, 512, 768, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(8192);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        const int nw = nblk * threads / 64;
        double s = 0, rt = 0;
        for (int i = 0; i < nw; ++i) { s += (double)h[i]; rt += (double)h[4096 + i]; }
        s /= nw; rt /= nw;
        printf("%-44s %d waves/SIMD: %6.2f ticks per instruction (per wave); memtime/memrealtime = %.2f -> %.0f MHz if realtime is 100 MHz\n",
               name, threads / 256, s / ((double)iters * per_iter), s / rt, s / rt * 100.0);
    }
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0>("v_fma_f32 (3 VGPR)", 32);
    run<1>("v_fma_f32 v, |s|, -v", 32);
    run<2>("v_exp_f32", 32);
    run<3>("v_add_f32 (4 chains)", 32);
    run<4>("v_max3_f32 (4 chains)", 32);
    run<5>("v_cvt_pk_f16_f32", 32);
    run<27>("v_cvt_pkrtz_f16_f32", 32);
    run<28>("v_pack_b32_f16", 32);
    run<6>("v_add_f32 (independent)", 32);
    run<7>("v_mul_f32", 32);
    run<8>("mix: 2 fma + 2 exp + 1 cvt (per 5 instr)", 80);
    run<9>("v_pk_mul_f32", 32);
    run<10>("v_exp_f16", 32);
    run<11>("v_pk_add_f16", 32);
    run<12>("v_dot2_f32_f16 (4 chains)", 32);
    run<13>("v_exp_f32 + v_fma_f32 interleaved (per 2)", 64);
    run<14>("slot: 1 MFMA32x32x16 + 8 VALU  (per slot)", 16);
    run<15>("slot: 8 VALU only              (per slot)", 16);
    run<16>("slot: 1 MFMA only              (per slot)", 16);
    run<21>("MFMA 32x32x16 f16, 4 independent accumulators", 16);
    run<22>("MFMA 16x16x32 f16, 4 independent accumulators", 16);
    run<23>("slot: MFMA (AGPR acc) + 8 VALU", 16);
    run<24>("slot: MFMA (4 VGPR accs) + 8 VALU", 16);
    run<25>("slot16: MFMA16x16x32 + 2 tr reads + 2 VALU", 16);
    run<26>("slot16: MFMA16x16x32 + 1 b128 read + 2 VALU", 16);
    run<17>("slot: MFMA + 2 ds_read_tr_b64 + 8 VALU", 16);
    run<18>("slot: MFMA + 1 ds_read_b128 + 8 VALU", 16);
    run<19>("slot: 2 ds_read_tr_b64 only", 16);
    run<20>("slot: MFMA + 2 ds_read_tr_b64", 16);
    return 0;
}
