// slot_model.hip -- gfx950 micro-benchmark: what one MFMA "slot" of the d=64 attention loop costs (cycles) for
// candidate instruction mixes, at 1..4 waves per SIMD and for both fp16 MFMA shapes.  A slot is one
// v_mfma_f32_32x32x16_f16 (or two v_mfma_f32_16x16x32_f16) plus the softmax vector work of the two scores per lane
// that MFMA pair stands for at d = 64: exp2, row-sum add, half a cvt_pk, and the scale-subtract fma unless it is
// folded into Q / the accumulator's initial value.  No memory traffic, operands in registers: the issue-port and
// matrix-pipe ceiling of each arrangement, i.e. the number a real loop of that shape cannot beat.
// (reference analogues flashattn_tensorcore_util_profile.cu:69 and
// flashattn_forward_softmax_bottleneck.cu:66.)
// Build: hipcc --offload-arch=gfx950 -O2 slot_model.hip -o slot_model ; run: ./slot_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <cstring>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16(X) REP8(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// vector work of one slot (two scores): e0,e1 exp inputs/outputs; l0,l1 row-sum chains; w packed result
#define V_FOLD(i)   "v_exp_f32 %[e0], %[s0]\n\tv_exp_f32 %[e1], %[s1]\n\tv_add_f32 %[l0], %[l0], %[e0]\n\tv_add_f32 %[l1], %[l1], %[e1]\n\tv_cvt_pk_f16_f32 %[w], %[e0], %[e1]\n\t"
#define V_FMA(i)    "v_fma_f32 %[e0], %[s0], %[c], %[m]\n\tv_fma_f32 %[e1], %[s1], %[c], %[m]\n\tv_exp_f32 %[e0], %[e0]\n\tv_exp_f32 %[e1], %[e1]\n\tv_add_f32 %[l0], %[l0], %[e0]\n\tv_add_f32 %[l1], %[l1], %[e1]\n\tv_cvt_pk_f16_f32 %[w], %[e0], %[e1]\n\t"
#define V_PK(i)     "v_pk_fma_f32 %[ep], %[sp], %[cp], %[mp]\n\tv_exp_f32 %[e0], %[e0]\n\tv_exp_f32 %[e1], %[e1]\n\tv_pk_add_f32 %[lp], %[lp], %[ep]\n\tv_cvt_pk_f16_f32 %[w], %[e0], %[e1]\n\t"
#define V_FOLDPK(i) "v_exp_f32 %[e0], %[s0]\n\tv_exp_f32 %[e1], %[s1]\n\tv_pk_add_f32 %[lp], %[lp], %[ep]\n\tv_cvt_pk_f16_f32 %[w], %[e0], %[e1]\n\t"

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, float a, float b)
{
    // exp inputs (stand-ins for scores), kept constant; separate outputs so no chain forms through exp
    float s0[8], s1[8];
    union { float f[2]; double d; } ep[8], sp[8], lp[2], cp, mp;
    float l0[2] = {0.f, 0.f}, l1[2] = {0.f, 0.f};
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s0[i] = -1.0f - 0.01f * (threadIdx.x & 63) - i;
        s1[i] = -2.0f - 0.02f * (threadIdx.x & 63) - i;
        sp[i].f[0] = s0[i];
        sp[i].f[1] = s1[i];
        ep[i].f[0] = 0.f;
        ep[i].f[1] = 0.f;
        w[i] = 0;
    }
    lp[0].f[0] = lp[0].f[1] = lp[1].f[0] = lp[1].f[1] = 0.f;
    cp.f[0] = cp.f[1] = a;
    mp.f[0] = mp.f[1] = b;
    h8 af, bf;
    for (int i = 0; i < 8; ++i) { af[i] = (_Float16)(a + i); bf[i] = (_Float16)(b - i); }
    f16v acc16[4] = {};
    f4v acc4[8] = {};
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        // KIND / 10: vector mix (0 none, 1 folded, 2 scalar fma, 3 packed fma+add, 4 folded + packed add)
        // KIND % 10: matrix part (0 none, 1 one 32x32x16, 2 two 16x16x32 back to back, 3 two 16x16x32 split around the vector work)
#define SLOT(i)                                                                                                               \
        {                                                                                                                     \
            constexpr int M = KIND % 10, V = KIND / 10;                                                                       \
            if constexpr (M == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[(i) & 3]) : "v"(af), "v"(bf)); \
            if constexpr (M == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1" \
                                               : "+v"(acc4[(2 * (i)) & 7]), "+v"(acc4[(2 * (i) + 1) & 7]) : "v"(af), "v"(bf)); \
            if constexpr (M == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[(2 * (i)) & 7]) : "v"(af), "v"(bf)); \
            if constexpr (V == 1) asm volatile(V_FOLD(i) : [e0] "=&v"(ep[(i) & 7].f[0]), [e1] "=&v"(ep[(i) & 7].f[1]), [l0] "+v"(l0[(i) & 1]), [l1] "+v"(l1[(i) & 1]), [w] "=v"(w[(i) & 7]) \
                                               : [s0] "v"(s0[(i) & 7]), [s1] "v"(s1[(i) & 7]));                               \
            if constexpr (V == 2) asm volatile(V_FMA(i) : [e0] "=&v"(ep[(i) & 7].f[0]), [e1] "=&v"(ep[(i) & 7].f[1]), [l0] "+v"(l0[(i) & 1]), [l1] "+v"(l1[(i) & 1]), [w] "=v"(w[(i) & 7]) \
                                               : [s0] "v"(s0[(i) & 7]), [s1] "v"(s1[(i) & 7]), [c] "v"(a), [m] "v"(b));       \
            if constexpr (V == 3) asm volatile(V_PK(i) : [ep] "=&v"(ep[(i) & 7].d), [lp] "+v"(lp[(i) & 1].d), [w] "=v"(w[(i) & 7]), [e0] "=&v"(s0[(i) & 7]), [e1] "=&v"(s1[(i) & 7]) \
                                               : [sp] "v"(sp[(i) & 7].d), [cp] "v"(cp.d), [mp] "v"(mp.d));                    \
            if constexpr (V == 4) asm volatile(V_FOLDPK(i) : [e0] "=&v"(s0[(i) & 7]), [e1] "=&v"(s1[(i) & 7]), [lp] "+v"(lp[(i) & 1].d), [w] "=v"(w[(i) & 7]) \
                                               : [s0] "v"(sp[(i) & 7].f[0]), [s1] "v"(sp[(i) & 7].f[1]), [ep] "v"(ep[(i) & 7].d)); \
            if constexpr (M == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[(2 * (i) + 1) & 7]) : "v"(af), "v"(bf)); \
        }
        REP16(SLOT)
#undef SLOT
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float acc = l0[0] + l0[1] + l1[0] + l1[1] + lp[0].f[0] + lp[0].f[1] + lp[1].f[0] + lp[1].f[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += s0[i] + s1[i] + ep[i].f[0] + ep[i].f[1] + (float)w[i];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc += acc16[j][i];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc += acc4[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
        cyc[8192 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0;
    }
}


// ---- granularity / role experiments (two waves per SIMD, 512 threads) -----------------------------------------------
// GRAN: G matrix instructions back to back, then the vector work of G slots (folded mix), per iteration 16 slots in all.
// MODE 0: every wave the same program; 1: waves 4-7 start with the vector part (half a group out of phase);
//      2: fixed roles -- waves 0-3 issue only the matrix part, waves 4-7 only the vector part (both sets of 16 slots)
template <int GRAN, int MODE>
__global__ __launch_bounds__(512) void kg(float* out, unsigned long long* cyc, int iters, float a, float b)
{
    float s0[8], s1[8], e0[8], e1[8];
    float l0[2] = {0.f, 0.f}, l1[2] = {0.f, 0.f};
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s0[i] = -1.0f - 0.01f * (threadIdx.x & 63) - i;
        s1[i] = -2.0f - 0.02f * (threadIdx.x & 63) - i;
        e0[i] = e1[i] = 0.f;
        w[i] = 0;
    }
    h8 af, bf;
    for (int i = 0; i < 8; ++i) { af[i] = (_Float16)(a + i); bf[i] = (_Float16)(b - i); }
    f16v acc16[4] = {};
    const bool young = threadIdx.x >= 256;
    unsigned long long t0, t1;
#define MF(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[(i) & 3]) : "v"(af), "v"(bf));
#define VF(i) asm volatile(V_FOLD(i) : [e0] "=&v"(e0[(i) & 7]), [e1] "=&v"(e1[(i) & 7]), [l0] "+v"(l0[(i) & 1]), [l1] "+v"(l1[(i) & 1]), [w] "=v"(w[(i) & 7]) \
                                     : [s0] "v"(s0[(i) & 7]), [s1] "v"(s1[(i) & 7]));
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (MODE == 1 && young) {   // half a group out of phase: the vector part of one group first
#pragma unroll
        for (int i = 0; i < GRAN; ++i) { VF(i) }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int grp = 0; grp < 16 / GRAN; ++grp) {
            if (MODE != 2 || !young) {
#pragma unroll
                for (int i = 0; i < GRAN; ++i) { MF(i) }
            }
            if (MODE != 2 || young) {
#pragma unroll
                for (int i = 0; i < GRAN; ++i) { VF(i) }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#undef MF
#undef VF
    float acc = l0[0] + l0[1] + l1[0] + l1[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += e0[i] + e1[i] + (float)w[i];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc += acc16[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

template <int GRAN, int MODE>
void run_g(const char* name)
{
    const int iters = 4000, nblk = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nblk * 512 * 4);
    hipMalloc(&cyc, nblk * 8 * 8);
    hipLaunchKernelGGL((kg<GRAN, MODE>), dim3(nblk), dim3(512), 0, 0, out, cyc, iters, 1.0001f, -0.5f);
    hipLaunchKernelGGL((kg<GRAN, MODE>), dim3(nblk), dim3(512), 0, 0, out, cyc, iters, 1.0001f, -0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double old_ = 0, young = 0;
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? old_ : young) += (double)h[b * 8 + w];
    old_ /= nblk * 4; young /= nblk * 4;
    const double slots = (double)iters * 16;   // per wave; a SIMD runs two waves
    if (MODE == 2)
        printf("%-58s  matrix waves %6.2f cyc per MFMA, vector waves %6.2f cyc per slot's vector work\n", name, old_ / slots, young / slots);
    else
        printf("%-58s  waves 0-3 %6.2f, waves 4-7 %6.2f cyc per own slot -> %6.2f cyc per slot per SIMD\n", name, old_ / slots, young / slots,
               (old_ > young ? old_ : young) / (2 * slots));
    hipFree(out);
    hipFree(cyc);
}

template <int KIND>
void run(const char* name)
{
    const int iters = 4000, nblk = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nblk * 1024 * 4);
    hipMalloc(&cyc, 16384 * 8);
    printf("%-58s", name);
    for (int threads : {256, 512, 768, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, -0.5f);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, -0.5f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(16384);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        const int nw = nblk * threads / 64;
        double mx = 0, rt = 0;
        for (int i = 0; i < nw; ++i) { if ((double)h[i] > mx) mx = (double)h[i]; rt += (double)h[8192 + i]; }
        rt /= nw;
        // slots issued per SIMD = waves/SIMD * iters * 16; the slowest wave's span is the SIMD's busy time
        const int wps = threads / 256;
        const double per_slot = mx / ((double)iters * 16 * wps);
        printf("  %dw: %6.2f cyc/slot (%3.0f%% MFMA) %.2f ms", wps, per_slot, KIND % 10 ? 3200.0 / per_slot : 0.0, ms);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}


template <int KIND>
void sustain(double seconds)
{
    const int iters = 4000, nblk = 256, threads = 512;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nblk * 1024 * 4);
    hipMalloc(&cyc, 16384 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    hipEventRecord(e0, 0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, -0.5f);
        launches += 20;
        hipDeviceSynchronize();
    }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(16384);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < nblk * threads / 64; ++i) if ((double)h[i] > mx) mx = (double)h[i];
    const double slots_per_simd = (double)launches * iters * 16 * 2;
    printf("SUSTAIN kind %d: %.1f ms, %ld launches, %.3f ns per slot per SIMD, %.2f cycles per slot per SIMD (last launch)\n", KIND, ms, launches,
           ms * 1e6 / slots_per_simd, mx / ((double)iters * 16 * 2));
}

int main(int argc, char** argv)
{
    if (argc >= 4 && !strcmp(argv[1], "sustain")) {
        const int kind = atoi(argv[2]);
        const double sec = atof(argv[3]);
        switch (kind) {
            case 1: sustain<1>(sec); break;
            case 2: sustain<2>(sec); break;
            case 10: sustain<10>(sec); break;
            case 20: sustain<20>(sec); break;
            case 30: sustain<30>(sec); break;
            case 11: sustain<11>(sec); break;
            case 21: sustain<21>(sec); break;
            case 31: sustain<31>(sec); break;
            case 12: sustain<12>(sec); break;
            case 22: sustain<22>(sec); break;
            case 32: sustain<32>(sec); break;
            default: printf("unknown kind\n"); return 1;
        }
        return 0;
    }
    printf("# cycles per slot PER SIMD (slowest wave's span / slots issued on its SIMD); %%MFMA = 32 / that; ms = wall of the launch\n");
    run<1>("MFMA 32x32x16 only");
    run<2>("2 x MFMA 16x16x32 only");
    run<10>("vector only: folded (2 exp, 2 add, 1 cvt)");
    run<20>("vector only: scalar fma (2 fma, 2 exp, 2 add, 1 cvt)");
    run<30>("vector only: packed (1 pk_fma, 2 exp, 1 pk_add, 1 cvt)");
    run<40>("vector only: folded + pk_add (2 exp, 1 pk_add, 1 cvt)");
    run<11>("32x32x16 + folded");
    run<21>("32x32x16 + scalar fma");
    run<31>("32x32x16 + packed");
    run<41>("32x32x16 + folded + pk_add");
    run<12>("2 x 16x16x32 + folded");
    run<22>("2 x 16x16x32 + scalar fma");
    run<32>("2 x 16x16x32 + packed");
    run<42>("2 x 16x16x32 + folded + pk_add");
    run<13>("16x16x32, folded, 16x16x32 (split)");
    run<23>("16x16x32, scalar fma, 16x16x32 (split)");
    run<33>("16x16x32, packed, 16x16x32 (split)");
    printf("# two waves per SIMD, 32x32x16 + folded vector mix, by interleave granularity (G matrix instructions, then G slots of vector work)\n");
    run_g<1, 0>("G=1  same program");
    run_g<2, 0>("G=2  same program");
    run_g<4, 0>("G=4  same program");
    run_g<8, 0>("G=8  same program");
    run_g<16, 0>("G=16 same program (phase-ordered, lock-step)");
    run_g<4, 1>("G=4  waves 4-7 half a group out of phase");
    run_g<8, 1>("G=8  waves 4-7 half a group out of phase");
    run_g<16, 1>("G=16 waves 4-7 half a group out of phase");
    run_g<16, 2>("fixed roles: waves 0-3 matrix only, waves 4-7 vector only");
    return 0;
}
