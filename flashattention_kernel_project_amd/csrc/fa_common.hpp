// fa_common.hpp -- shared device helpers for the gfx950 (MI355X / CDNA4) attention-forward kernels.
//
// Written for gfx950 only: 64-lane wavefronts, v_mfma_f32_32x32x16_{f16,bf16},
// ds_read_b64_tr_b16, buffer_load/store through wave-uniform resource descriptors.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <tuple>
#include <utility>

namespace fa {

typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x2  __attribute__((ext_vector_type(2)));
typedef unsigned u32x4  __attribute__((ext_vector_type(4)));
typedef unsigned u32x2  __attribute__((ext_vector_type(2)));
typedef short    s16x4  __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4  __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2  __attribute__((ext_vector_type(2)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16   bf16x2 __attribute__((ext_vector_type(2)));

constexpr float kLog2e = 1.4426950408889634f;

// ---- element-type traits: fp16 / bf16 inputs, fp32 accumulation ---------------------------
struct F16 {
    static constexpr int id = 0;
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a),
                                                     __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        f16x2 v = {(_Float16)lo, (_Float16)hi};   // v_cvt_pk_f16_f32 (round-to-nearest-even)
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ float lo(unsigned w) {
        return (float)__builtin_bit_cast(f16x2, w)[0];
    }
    static __device__ __forceinline__ float hi(unsigned w) {
        return (float)__builtin_bit_cast(f16x2, w)[1];
    }
    static __device__ __forceinline__ float one(uint16_t b) {
        return (float)__builtin_bit_cast(_Float16, b);
    }
    static constexpr unsigned kOnes2 = 0x3C003C00u;   // two 1.0 halves
    // Row sums: fp32 v_add of the un-rounded p (rounding P to fp16 perturbs a weight by 2^-11: far
    // inside the tolerance, and v_add is cheaper than v_dot2c at the power cap).
#ifndef FA_F16_SUM_ROUNDED
#define FA_F16_SUM_ROUNDED 0
#endif
    static constexpr bool kSumRounded = FA_F16_SUM_ROUNDED != 0;
    // acc + lo(w) + hi(w) in fp32: one v_dot2c_f32_f16 against (1, 1)
    static __device__ __forceinline__ float sum2(unsigned w, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, w), __builtin_bit_cast(f16x2, kOnes2), acc, false);
    }
};

struct BF16 {
    static constexpr int id = 1;
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a),
                                                         __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        bf16x2 v = {(__bf16)lo, (__bf16)hi};      // v_cvt_pk_bf16_f32 (round-to-nearest-even)
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
    static __device__ __forceinline__ float one(uint16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
    static constexpr unsigned kOnes2 = 0x3F803F80u;   // two 1.0 bfloat16
    // Row sums over the ROUNDED p (one v_dot2c per packed pair): O = sum(p'v)/sum(p') keeps the
    // weights a convex combination, so a row dominated by one key returns that V row exactly instead
    // of V(1 + 2^-9); with un-rounded sums a peaked row can miss the 1e-2 bar at |V| ~ 4.
    static constexpr bool kSumRounded = true;
    static __device__ __forceinline__ float sum2(unsigned w, float acc) {   // v_dot2c_f32_bf16
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), __builtin_bit_cast(bf16x2, kOnes2), acc, false);
    }
};

// ---- buffer resources (wave-uniform base + byte count; out-of-range loads read 0, stores drop) --
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}
// per-lane offset + wave-uniform offset (SGPR): one address register serves many loads
__device__ __forceinline__ u32x4 buf_load16_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
// explicit cache policy (gfx94x/gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1)
template <int kAux>
__device__ __forceinline__ u32x4 buf_load16_cp(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, kAux);
}
// read-once streams (decode K/V): non-temporal hint (cache-policy bit 1 = nt on gfx94x/gfx950)
__device__ __forceinline__ u32x4 buf_load16_nt(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 2);
}
__device__ __forceinline__ u32x2 buf_load8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned voff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, unsigned voff, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}
__device__ __forceinline__ void buf_store16_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 2);
}
__device__ __forceinline__ void buf_store8_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 2);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, 0, 0);
}

// ---- LDS access ----------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ u32x4 lds_read16(const char* smem, unsigned off) {
    return *reinterpret_cast<const u32x4*>(smem + off);
}
__device__ __forceinline__ void lds_write16(char* smem, unsigned off, u32x4 v) {
    *reinterpret_cast<u32x4*>(smem + off) = v;
}
// the same by LDS byte ADDRESS (lds_addr(smem) + offset): callers that keep whole addresses in registers and want the rest of
// it in the instruction's immediate
__device__ __forceinline__ unsigned lds_addr(const char* smem) {
    return (unsigned)(size_t)(const lds_char*)smem;
}
__device__ __forceinline__ u32x4 lds_read16_at(unsigned addr) {
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    return *(const lds_u32x4*)(size_t)addr;
}
__device__ __forceinline__ void lds_write16_at(unsigned addr, u32x4 v) {
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    *(lds_u32x4*)(size_t)addr = v;
}
__device__ __forceinline__ unsigned lds_read4_at(unsigned addr) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    return *(const volatile lds_u32*)(size_t)addr;
}
__device__ __forceinline__ void lds_write4_at(unsigned addr, unsigned v) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    *(volatile lds_u32*)(size_t)addr = v;
}
__device__ __forceinline__ u32x2 lds_read_tr8_at(unsigned addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)addr);
    return __builtin_bit_cast(u32x2, v);
}
// ds_read_b64_tr_b16: per 16-lane group a [4 rows][16 cols] block of 16-bit elements comes back
// column-major; lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i.
__device__ __forceinline__ u32x2 lds_read_tr8(const char* smem, unsigned off) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_s16x4*)(__attribute__((address_space(3))) void*)(smem + off));
    return __builtin_bit_cast(u32x2, v);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// value of the other 32-lane half's copy of x (lane l <-> lane l^32)
__device__ __forceinline__ float swap_halves(float x) {
    return __shfl_xor(x, 32, 64);
}

// Dynamic LDS above the 64-KB default needs a per-function opt-in, and the attribute is per DEVICE: a
// process that drives several GPUs (bench/fa_bench --gpus G) must set it on each, so no static caching.
static inline hipError_t ensure_dyn_lds(const void* kernel, int bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// CU count of the CURRENT device, read per call (a one-process multi-GPU host drives several devices).
static inline int device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}
// Launch: hipLaunchKernel's own return value is this launch's status.  Nothing is cleared and nothing is read from the
// thread's sticky last-error slot, so an error the host application has pending (a failed launch of its own that it has
// not looked at yet) is neither hidden nor mistaken for ours.  launch_status() = the status of this thread's latest FA_LAUNCH.
inline thread_local hipError_t g_launch_status = hipSuccess;
static inline hipError_t launch_status() { return g_launch_status; }
template <typename... KArgs, typename... Args, size_t... I>
static inline void launch_k_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                 std::index_sequence<I...>, Args&&... args) {
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};   // the kernel's exact parameter types, in order
    void* ptrs[] = {static_cast<void*>(&std::get<I>(vals))...};
    g_launch_status = hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, stream);
}
template <typename... KArgs, typename... Args>
static inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count does not match the kernel's parameter list");
    launch_k_impl(kernel, grid, block, lds, stream, std::index_sequence_for<KArgs...>{}, static_cast<Args&&>(args)...);
}
#define FA_LAUNCH(...) ::fa::launch_k(__VA_ARGS__)

}  // namespace fa
