// mfma_power.hip -- sustained fp16 MFMA throughput at the package power cap, 32x32x16 vs 16x16x32,
// random operands (DVFS depends on the data), two waves per SIMD, operands held in registers.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_power.hip -o mfma_power ; run: ./mfma_power [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const h8* ab, float* out, int iters)
{
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ab[(threadIdx.x * 8 + i) & 4095]; b[i] = ab[(threadIdx.x * 8 + 4 + i) & 4095]; }
    f16v c32[4] = {};
    f4v c16[8] = {};
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], c32[i & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) c16[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], c16[i & 7], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c32[i][j];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += c16[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    std::vector<_Float16> h(4096 * 8);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 16) % 2001 - 1000) / 500.0f); }
    h8* ab; float* out;
    hipMalloc(&ab, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(ab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int shape : {32, 16, 32, 16}) {
        auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            for (int r = 0; r < 20; ++r) {
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, ab, out, iters);
                else hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, ab, out, iters);
            }
            launches += 20;
            hipDeviceSynchronize();
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)launches * 256 * 8 * iters * 16 * 32768.0;   // 16 MFMA32 == 32 MFMA16 in flops
        printf("mfma %dx%d: %.1f TFLOP/s sustained over %.1f s\n", shape, shape, flops / (ms * 1e-3) / 1e12, ms * 1e-3);
        fflush(stdout);
    }
    return 0;
}
