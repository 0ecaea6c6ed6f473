// kv_stream.hip -- the K/V tile stream of the attention forward ALONE: global (L2) -> registers -> LDS ring -> one barrier per
// tile, with no matrix and no softmax work.  The CDNA4 counterpart of the reference's stream-only profiling kernel
// (FlashAttention/flashattn_forward_memory_bound/flashattn_forward_cp_async_stall.cu:93-206: cp.async double buffer + wait +
// barrier, nothing else).  Same geometry as fa_fwd_rp16 at d = 64: 512-thread workgroups, one per CU, persistent over
// (head, 512-row query block) items, each item streams ALL tiles of its head (64 keys x 64 dims, fp16) through a ring of four
// [K tile][V tile] slots, tile j+2 requested at the top of iteration j, landed mid-iteration, K image XOR-swizzled, V image in
// 256-B blocks written by 4-key x 2-chunk lane groups (conflict-free ds_write_b128).
//   mode 0: stage only            mode 1: + every wave reads every fragment of the tile (8 ds_read_b128 + 16 ds_read_b64_tr_b16),
//                                          the LDS operand traffic of the real kernel
// Prints ms per launch and TB/s: bytes staged (items x tiles x 16 KB; L2 -> LDS) and bytes read back from LDS.
// Build: make -C tools/microbench ; run: ./kv_stream [BH N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int kD = 64, kBlockN = 64, kW = 8;
constexpr unsigned kRowB = kD * 2, kTile = kBlockN * kD * 2, kSlot = 2 * kTile;

template <int MODE>
__global__ __launch_bounds__(64 * kW, 2) void kv_stream(const uint16_t* __restrict__ K, const uint16_t* __restrict__ V,
                                                         unsigned* __restrict__ sink, int N, int nqb, unsigned items)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, c16 = lane & 15u, g = lane >> 4;
    const unsigned srow = tid >> 3, sch = tid & 7u;
    const unsigned k_goff = srow * kRowB + sch * 16u;
    const unsigned k_lds = srow * kRowB + ((sch ^ ((srow >> 1) & 7u)) << 4);
    const unsigned t = lane >> 3;
    const unsigned vrow = wave * 8u + 4u * (t >> 2) + ((lane >> 1) & 3u), vch = 2u * (t & 3u) + (lane & 1u);
    const unsigned v_goff = vrow * kRowB + vch * 16u;
    const unsigned v_lds = kTile + ((vrow >> 3) * 4u + (vch >> 1)) * 256u + ((vrow & 7u) << 5) + ((vch & 1u) << 4);
    const unsigned k_rd0 = c16 * kRowB + ((g ^ ((c16 >> 1) & 7u)) << 4), k_rd1 = c16 * kRowB + (((4u + g) ^ ((c16 >> 1) & 7u)) << 4);
    const unsigned v_rd = kTile + (g >> 1) * 4u * 256u + ((4u * (g & 1u) + (c16 >> 2)) << 5) + (c16 & 3u) * 8u;
    const int ntiles = N / kBlockN;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
        const unsigned bh = item / (unsigned)nqb;
        const char* kb = reinterpret_cast<const char*>(K) + (size_t)bh * N * kRowB;
        const char* vb = reinterpret_cast<const char*>(V) + (size_t)bh * N * kRowB;
        u32x4 kst, vst;
        // prologue: tiles 0 and 1
        for (int pt = 0; pt < 2; ++pt) {
            kst = *reinterpret_cast<const u32x4*>(kb + (size_t)pt * kTile + k_goff);
            vst = *reinterpret_cast<const u32x4*>(vb + (size_t)pt * kTile + v_goff);
            *reinterpret_cast<u32x4*>(smem + pt * kSlot + k_lds) = kst;
            *reinterpret_cast<u32x4*>(smem + pt * kSlot + v_lds) = vst;
        }
        __syncthreads();
        for (int j = 0; j < ntiles; ++j) {
            const int jn = j + 2 < ntiles ? j + 2 : ntiles - 1;   // (the real kernel reads zeros past the end through the buffer bounds)
            kst = *reinterpret_cast<const u32x4*>(kb + (size_t)jn * kTile + k_goff);
            vst = *reinterpret_cast<const u32x4*>(vb + (size_t)jn * kTile + v_goff);
            const unsigned so = ((unsigned)j & 3u) * kSlot, so2 = ((unsigned)(j + 2) & 3u) * kSlot;
            if constexpr (MODE == 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int kbl = 0; kbl < 2; ++kbl) {
                        acc ^= *reinterpret_cast<const u32x4*>(smem + so + (2 * h + kbl) * 16u * kRowB + k_rd0);
                        acc ^= *reinterpret_cast<const u32x4*>(smem + so + (2 * h + kbl) * 16u * kRowB + k_rd1);
                    }
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                            const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (lds_s16x4*)(__attribute__((address_space(3))) void*)(smem + so + v_rd + (4u * h + 2u * jj) * 4u * 256u + db * 256u));
                            acc[0] ^= (unsigned)r[0] | ((unsigned)r[1] << 16);
                            acc[1] ^= (unsigned)r[2] | ((unsigned)r[3] << 16);
                        }
                }
            }
            *reinterpret_cast<u32x4*>(smem + so2 + k_lds) = kst;
            *reinterpret_cast<u32x4*>(smem + so2 + v_lds) = vst;
            __syncthreads();
        }
        if constexpr (MODE == 0) acc ^= kst ^ vst;
    }
    if (acc[0] == 0x12345u && acc[1] == 0x54321u) sink[blockIdx.x * blockDim.x + tid] = acc[2] ^ acc[3];   // (keeps everything live)
}

int main(int argc, char** argv)
{
    const int BH = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 4096;
    if (N % 512 != 0 || BH <= 0) { fprintf(stderr, "N must be a multiple of 512\n"); return 2; }
    const int nqb = N / 512;
    const unsigned items = (unsigned)(BH * nqb);
    const size_t elems = (size_t)BH * N * kD;
    std::vector<uint16_t> h(elems);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(x >> 16); }
    uint16_t *K, *V; unsigned* sink;
    hipMalloc(&K, elems * 2); hipMalloc(&V, elems * 2); hipMalloc(&sink, 256 * 512 * 4);
    hipMemcpy(K, h.data(), elems * 2, hipMemcpyHostToDevice);
    hipMemcpy(V, h.data(), elems * 2, hipMemcpyHostToDevice);
    int dev = 0, cus = 256;
    hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = items < (unsigned)cus ? items : (unsigned)cus;
    const double staged = (double)items * (N / kBlockN) * kSlot;          // bytes global -> LDS per launch
    const double ldsread = staged * kW;                                  // every wave reads every tile
    for (int mode = 0; mode < 2; ++mode) {
        auto kern = mode == 0 ? kv_stream<0> : kv_stream<1>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kSlot);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kW), 4 * kSlot, 0, K, V, sink, N, nqb, items);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 50;
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kW), 4 * kSlot, 0, K, V, sink, N, nqb, items);
        hipEventRecord(e1); hipEventSynchronize(e1);
        if (hipGetLastError() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
        printf("kv_stream mode %d (%s): BH %d N %d: %.4f ms per launch, staged %.2f TB/s (global/L2 -> LDS)%s\n", mode,
               mode == 0 ? "stage + barrier" : "stage + barrier + all fragment reads", BH, N, ms, staged / ms / 1e9,
               mode == 1 ? "" : "");
        if (mode == 1) printf("kv_stream mode 1: LDS fragment reads %.2f TB/s\n", ldsread / ms / 1e9);
    }
    return 0;
}
