// fa_bench.cpp -- native host driver for libfa_mi355.so.
//
// The reference has no library: each kernel lives in a stand-alone .cu whose main() seeds
// mt19937(42), fills Q,K,V with N(0,1) halves, copies them to the device, launches once for a
// check, then times 50 launches between events and prints TFLOPS
// (Streaming_FlashAttention_Forward_Kernel/flashattn_streaming_16x16_mw.cu:319-450,
//  FlashAttention/flashattn_forward_memory_bound/flashattn_forward_wmma_memprofile.cu:407-531).
// This is that driver written once, in C++, against the C ABI (include/fa_mi355.h): same sequence
// (seed -> fill -> H2D -> warm-up -> timed launches between events -> TFLOP/s with the reference's
// 4*BH*N^2*D FLOP model, memprofile.cu:508) with the shapes as flags, plus what the reference never
// had: a multi-GPU mode that shards B*H contiguously over the visible devices, one host thread and
// one stream per device, no collective (every (b,h) is an independent problem).
//
// `--check` reproduces the reference driver's check step (flashattn_streaming_16x16_mw.cu:352 CPU reference call,
// :383-402 rel-L2 print; flashattn_forward_fused_5_4_2.cu:366-370 max-abs): ONE launch before the warm-up, D2H, and
// "rel_l2 = ... max_abs = ..." against the CPU checker, which is oracle/liboracle_cpu.so loaded with dlopen ONLY under
// that flag and never inside the timed loop (test infrastructure: the harness itself links only the C ABI).  The general
// family checks the first --check-heads (b,h) heads (all of them when the problem is small).
// `--dump FILE` writes O of device 0 for external comparison.
//
//   bench/fa_bench --B 8 --H 16 --N 4096 --d 64 --dtype f16 --iters 50 --warmup 10 [--gpus 8 [--one-device]] [--check]
//   bench/fa_bench --family s16 --B 1024 --N 128 [--check]  (16x16 streaming family; N = seq_len)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/fa_mi355.h"

#define HIP_OK(cmd)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (cmd);                                                               \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

namespace {

struct Args {
    std::string family = "general", dtype = "f16", out = "f32", dump, checker;
    int B = 8, H = 16, N = 4096, d = 64, iters = 50, warmup = 10, gpus = 1, algo = 0, check_heads = 0;
    bool one_device = false;   // --one-device: rehearse --gpus G on a one-GPU box (every shard's thread and stream on device 0)
    bool check = false;
    uint64_t seed = 42;
};

// The CPU checker (oracle/liboracle_cpu.so), resolved at run time and only under --check.
struct Checker {
    void* h = nullptr;
    void (*forward_rows)(const float*, const float*, const float*, float*, int, int, int, float, int, int, int, int, int, int) = nullptr;
    void (*streaming16)(const float*, const float*, const float*, float*, int, int, float) = nullptr;
    void (*decode16)(const uint16_t*, float*, size_t, int) = nullptr;
    double (*rel_l2)(const float*, const float*, size_t) = nullptr;
    double (*max_abs)(const float*, const float*, size_t) = nullptr;
    bool open(const std::string& path) {
        h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { std::fprintf(stderr, "--check: cannot load %s (%s); build it with `make -C oracle`\n", path.c_str(), dlerror()); return false; }
        forward_rows = reinterpret_cast<decltype(forward_rows)>(dlsym(h, "fa_oracle_forward_rows"));
        streaming16 = reinterpret_cast<decltype(streaming16)>(dlsym(h, "fa_oracle_streaming_16x16"));
        decode16 = reinterpret_cast<decltype(decode16)>(dlsym(h, "fa_oracle_decode16"));
        rel_l2 = reinterpret_cast<decltype(rel_l2)>(dlsym(h, "fa_oracle_rel_l2"));
        max_abs = reinterpret_cast<decltype(max_abs)>(dlsym(h, "fa_oracle_max_abs"));
        return forward_rows && streaming16 && decode16 && rel_l2 && max_abs;
    }
};
std::string default_checker(const char* argv0) {
    std::string p = argv0;
    const size_t k = p.find_last_of('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/../oracle/liboracle_cpu.so";
}

// counter-based N(0,1): splitmix64 + Box-Muller (portable, unlike std::normal_distribution)
inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline float normal_at(uint64_t seed, uint64_t i) {
    const double u1 = ((double)(mix(seed + 2 * i) >> 11) + 0.5) / 9007199254740992.0;
    const double u2 = ((double)(mix(seed + 2 * i + 1) >> 11) + 0.5) / 9007199254740992.0;
    return (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
}
inline uint16_t to_f16(float f) {   // round-to-nearest-even, normal range is all N(0,1) needs
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, ax = u & 0x7FFFFFFFu;
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);
    if (ax < 0x38800000u) {   // subnormal half or zero
        if (ax < 0x33000001u) return (uint16_t)sign;
        const int e = (int)(ax >> 23) - 127, shift = 13 + (-14 - e);
        const uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) ++hm;
        return (uint16_t)(sign | hm);
    }
    uint32_t r = ax + 0xFFFu + ((ax >> 13) & 1u);
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
inline uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct Shard {
    int dev = 0, bh0 = 0, bh1 = 0;
    double ms = 0.0;
    int rc = 0;
};

int run_general(const Args& a, Shard& s, double* checksum)
{
    HIP_OK(hipSetDevice(s.dev));
    const int bh = s.bh1 - s.bh0;
    const size_t n = (size_t)bh * a.N * a.d;
    const bool bf = a.dtype == "bf16", o32 = a.out == "f32";
    std::vector<uint16_t> hq(n), hk(n), hv(n);
    const size_t total = (size_t)a.B * a.H * a.N * a.d, base = (size_t)s.bh0 * a.N * a.d;
    for (size_t i = 0; i < n; ++i) {   // one stream, drawn Q -> K -> V like the reference drivers
        const float q = normal_at(a.seed, base + i), k = normal_at(a.seed, total + base + i),
                    v = normal_at(a.seed, 2 * total + base + i);
        hq[i] = bf ? to_bf16(q) : to_f16(q);
        hk[i] = bf ? to_bf16(k) : to_f16(k);
        hv[i] = bf ? to_bf16(v) : to_f16(v);
    }
    void *dq, *dk, *dv, *dout;
    const size_t obytes = n * (o32 ? 4 : 2);
    HIP_OK(hipMalloc(&dq, n * 2));
    HIP_OK(hipMalloc(&dk, n * 2));
    HIP_OK(hipMalloc(&dv, n * 2));
    HIP_OK(hipMalloc(&dout, obytes));
    HIP_OK(hipMemcpy(dq, hq.data(), n * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dk, hk.data(), n * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dv, hv.data(), n * 2, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const float scale = 1.0f / std::sqrt((float)a.d);
    auto launch = [&]() {
        return fa_forward_ex(dq, dk, dv, dout, 1, bh, a.N, a.d, scale, bf ? FA_DTYPE_BF16 : FA_DTYPE_F16,
                             o32 ? FA_OUT_F32 : FA_OUT_SAME, a.algo, st);
    };
    if (a.check && s.dev == 0) {   // the reference driver's order: CPU reference, ONE check launch, error print, then timing
        Checker ck;
        if (!ck.open(a.checker)) return 1;
        const double fl_head = 4.0 * (double)a.N * a.N * a.d;
        int heads = a.check_heads > 0 ? a.check_heads : (fl_head * bh <= 2.0e10 ? bh : std::max(1, (int)(2.0e10 / fl_head)));
        heads = std::min(heads, bh);
        const size_t hn = (size_t)heads * a.N * a.d;
        std::vector<float> fq(hn), fk(hn), fv(hn), want(hn), got(hn);
        ck.decode16(hq.data(), fq.data(), hn, bf ? 1 : 0);
        ck.decode16(hk.data(), fk.data(), hn, bf ? 1 : 0);
        ck.decode16(hv.data(), fv.data(), hn, bf ? 1 : 0);
        const int nt = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        ck.forward_rows(fq.data(), fk.data(), fv.data(), want.data(), heads, a.N, a.d, scale, 0, nt, 0, heads, 0, a.N);
        if (int rc = launch()) { std::fprintf(stderr, "fa_forward_ex -> %d\n", rc); return 1; }
        HIP_OK(hipStreamSynchronize(st));
        if (o32) {
            HIP_OK(hipMemcpy(got.data(), dout, hn * 4, hipMemcpyDeviceToHost));
        } else {
            std::vector<uint16_t> h16(hn);
            HIP_OK(hipMemcpy(h16.data(), dout, hn * 2, hipMemcpyDeviceToHost));
            ck.decode16(h16.data(), got.data(), hn, bf ? 1 : 0);
        }
        std::printf("[check] %d of %d (b,h) heads of gpu 0 vs the CPU reference: rel_l2 = %.6e  max_abs = %.6e\n", heads, bh,
                    ck.rel_l2(got.data(), want.data(), hn), ck.max_abs(got.data(), want.data(), hn));
    }
    for (int i = 0; i < a.warmup; ++i)
        if (int rc = launch()) { std::fprintf(stderr, "fa_forward_ex -> %d\n", rc); return 1; }
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < a.iters; ++i)
        if (int rc = launch()) { std::fprintf(stderr, "fa_forward_ex -> %d\n", rc); return 1; }
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    s.ms = ms / a.iters;
    if (checksum || !a.dump.empty()) {
        std::vector<char> ho(obytes);
        HIP_OK(hipMemcpy(ho.data(), dout, obytes, hipMemcpyDeviceToHost));
        if (checksum && o32) {
            double c = 0;
            const float* f = reinterpret_cast<const float*>(ho.data());
            for (size_t i = 0; i < n; ++i) c += f[i];
            *checksum = c;
        }
        if (!a.dump.empty() && s.dev == 0) {
            if (FILE* fp = std::fopen(a.dump.c_str(), "wb")) {
                std::fwrite(ho.data(), 1, obytes, fp);
                std::fclose(fp);
            }
        }
    }
    (void)hipFree(dq); (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(dout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return 0;
}

int run_s16(const Args& a)
{
    // NUM_BATCH = B, SEQ_LEN = N, scale = 1/sqrt(16) (flashattn_streaming_16x16_mw.cu:322-329)
    const int B = a.B, L = a.N;
    const size_t nq = (size_t)B * 256, nk = (size_t)B * 16 * L;
    std::vector<uint16_t> hq(nq), hk(nk), hv(nk);
    for (size_t i = 0; i < nq; ++i) hq[i] = to_f16(normal_at(a.seed, i));
    for (size_t i = 0; i < nk; ++i) hk[i] = to_f16(normal_at(a.seed, nq + i));
    for (size_t i = 0; i < nk; ++i) hv[i] = to_f16(normal_at(a.seed, nq + nk + i));
    void *dq, *dk, *dv;
    float* dout;
    HIP_OK(hipMalloc(&dq, nq * 2));
    HIP_OK(hipMalloc(&dk, nk * 2));
    HIP_OK(hipMalloc(&dv, nk * 2));
    HIP_OK(hipMalloc((void**)&dout, nq * 4));
    HIP_OK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dv, hv.data(), nk * 2, hipMemcpyHostToDevice));
    if (a.check) {   // flashattn_streaming_16x16_mw.cu:352 (CPU reference), :371-402 (check launch, rel-L2)
        Checker ck;
        if (!ck.open(a.checker)) return 1;
        std::vector<float> fq(nq), fk(nk), fv(nk), want(nq), got(nq);
        ck.decode16(hq.data(), fq.data(), nq, 0);
        ck.decode16(hk.data(), fk.data(), nk, 0);
        ck.decode16(hv.data(), fv.data(), nk, 0);
        ck.streaming16(fq.data(), fk.data(), fv.data(), want.data(), B, L, 0.25f);
        if (int rc = flashattn_streaming_16x16_mw(dq, dk, dv, dout, B, L, 0.25f, nullptr)) return rc;
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(got.data(), dout, nq * 4, hipMemcpyDeviceToHost));
        std::printf("[check] all %d batch elements vs the CPU reference: rel_l2 = %.6e  max_abs = %.6e\n", B,
                    ck.rel_l2(got.data(), want.data(), nq), ck.max_abs(got.data(), want.data(), nq));
    }
    for (int i = 0; i < a.warmup; ++i)
        if (int rc = flashattn_streaming_16x16_mw(dq, dk, dv, dout, B, L, 0.25f, nullptr)) return rc;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < a.iters; ++i)
        if (int rc = flashattn_streaming_16x16_mw(dq, dk, dv, dout, B, L, 0.25f, nullptr)) return rc;
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= a.iters;
    // FLOP model of the v8+ drivers: (2*M*L*K + 2*M*L*Dv) * B  (flashattn_streaming_16x16_mw_v8.cu:446-448)
    const double flops = (2.0 * 16 * L * 16 + 2.0 * 16 * L * 16) * B;
    std::vector<float> ho(nq);
    HIP_OK(hipMemcpy(ho.data(), dout, nq * 4, hipMemcpyDeviceToHost));
    std::printf("GPU O[0, 0, 0..7]: ");
    for (int i = 0; i < 8; ++i) std::printf("%f ", ho[i]);
    std::printf("\n[16x16 streaming] NUM_BATCH=%d SEQ_LEN=%d  avg %.4f ms  %.4f TFLOPS\n", B, L, ms, flops / (ms * 1e-3) / 1e12);
    (void)hipFree(dq); (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(dout);
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    Args a;
    for (int i = 1; i < argc; ++i) {
        auto is = [&](const char* f) { return !std::strcmp(argv[i], f) && i + 1 < argc; };
        if (is("--B")) a.B = std::atoi(argv[++i]);
        else if (is("--H")) a.H = std::atoi(argv[++i]);
        else if (is("--N")) a.N = std::atoi(argv[++i]);
        else if (is("--d")) a.d = std::atoi(argv[++i]);
        else if (is("--iters")) a.iters = std::atoi(argv[++i]);
        else if (is("--warmup")) a.warmup = std::atoi(argv[++i]);
        else if (is("--gpus")) a.gpus = std::atoi(argv[++i]);
        else if (is("--algo")) a.algo = std::atoi(argv[++i]);
        else if (is("--seed")) a.seed = std::strtoull(argv[++i], nullptr, 10);
        else if (is("--dtype")) a.dtype = argv[++i];
        else if (is("--out")) a.out = argv[++i];
        else if (is("--family")) a.family = argv[++i];
        else if (is("--dump")) a.dump = argv[++i];
        else if (is("--checker")) a.checker = argv[++i];
        else if (is("--check-heads")) a.check_heads = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--check")) a.check = true;
        else if (!std::strcmp(argv[i], "--one-device")) a.one_device = true;
        else { std::fprintf(stderr, "unknown or incomplete flag %s\n", argv[i]); return 2; }
    }
    if (a.checker.empty()) a.checker = default_checker(argv[0]);
    std::printf("%s\n", fa_mi355_version());
    if (a.family == "s16") return run_s16(a);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "no GPU\n"); return 1; }
    const int G = std::max(1, a.one_device ? a.gpus : std::min(a.gpus, ndev));
    const int BH = a.B * a.H;
    std::vector<Shard> sh(G);
    for (int g = 0; g < G; ++g) {   // contiguous split of the flattened (b,h) axis, sizes differ by <= 1
        const int q = BH / G, r = BH % G;
        sh[g].dev = a.one_device ? 0 : g;
        sh[g].bh0 = g * q + std::min(g, r);
        sh[g].bh1 = sh[g].bh0 + q + (g < r ? 1 : 0);
    }
    double checksum = 0;
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
        th.emplace_back([&, g]() { sh[g].rc = run_general(a, sh[g], g == 0 ? &checksum : nullptr); });
    for (auto& t : th) t.join();
    double worst = 0;
    for (auto& s : sh) {
        if (s.rc) return s.rc;
        worst = std::max(worst, s.ms);
        const double fl = 4.0 * (s.bh1 - s.bh0) * (double)a.N * a.N * a.d;
        std::printf("  shard %d on gpu %d: (b,h) [%d,%d)  avg %.4f ms  %.2f TFLOPS\n", (int)(&s - &sh[0]), s.dev, s.bh0, s.bh1, s.ms, fl / (s.ms * 1e-3) / 1e12);
    }
    const double flops = 4.0 * BH * (double)a.N * a.N * a.d;                       // memprofile.cu:508
    const double bytes = 3.0 * BH * a.N * a.d * 2 + 1.0 * BH * a.N * a.d * (a.out == "f32" ? 4 : 2);  // memprofile.cu:518-520
    std::printf("[general] B=%d H=%d N=%d d=%d %s->%s gpus=%d  %.4f ms  %.2f TFLOPS aggregate  (%.1f GB/s algorithmic)  checksum(gpu0)=%.6f\n",
                a.B, a.H, a.N, a.d, a.dtype.c_str(), a.out.c_str(), G, worst, flops / (worst * 1e-3) / 1e12,
                bytes / (worst * 1e-3) / 1e9, checksum);
    return 0;
}
