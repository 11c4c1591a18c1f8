// Shared host/device helpers for libivosw_hip.so (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ivosw.h"

namespace ivosw {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(ivosw_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define IVOSW_REQUIRE(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            ::ivosw::set_error("%s: %s", __func__, msg);           \
            return IVOSW_ERR_ARG;                                  \
        }                                                          \
    } while (0)

#define IVOSW_CHECK_LAUNCH()                                                        \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ::ivosw::set_error("%s: HIP error %s", __func__, hipGetErrorString(e__)); \
            return IVOSW_ERR_LAUNCH;                                                \
        }                                                                           \
    } while (0)

// Every entry point runs on the device that OWNS its buffers, whatever the caller's current device is (the reference
// builds torch.device(f'cuda:{gpu_id}') and never calls set_device, eval_agent_manet.py:63): the scope looks the device up
// from a device pointer, switches to it for the duration of the call and restores the caller's device afterwards.
struct DeviceScope {
    int prev = -1, dev = -1;
    bool ok = false;
    explicit DeviceScope(const void* device_ptr);
    ~DeviceScope();
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

#define IVOSW_ON_DEVICE_OF(ptr)                                                                    \
    ::ivosw::DeviceScope dscope__(ptr);                                                            \
    do {                                                                                           \
        if (!dscope__.ok) {                                                                        \
            ::ivosw::set_error("%s: %s is not a device pointer (no CPU fallback)", __func__, #ptr); \
            return IVOSW_ERR_ARG;                                                                  \
        }                                                                                          \
    } while (0)

// Ablation switches (skip MFMAs / stores / loads to time the rest; tools/ablate_bneck.sh, IVOSW_DEBUG_CONV, tunable BDBG) exist
// only in builds made with -DIVOSW_ABLATION=1 (IVOSW_ABLATION=1 python ivos-w_amd/build.py --force).  The default build
// compiles every test out; ivosw_ablation_build() reports which one is loaded and bench.py refuses to run on the other.
#ifndef IVOSW_ABLATION
#define IVOSW_ABLATION 0
#endif
#define ABL(dbg, bits) (IVOSW_ABLATION && ((dbg) & (bits)))

constexpr int WAVE = 64;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace.
struct Arena {
    char* base;
    size_t off;
    explicit Arena(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// ---------------------------------------------------------------- device helpers
typedef uint16_t bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair (lo in bits 0..15), round-to-nearest-even in hardware: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    union { bf16x2_t b; uint32_t u; } x;
    x.b = __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t);
    return x.u;
}

// ReLU AFTER the bf16 conversion, on the packed pair, as ONE integer instruction (v_pk_max_i16 against 0: a negative bf16 is a
// negative int16, -0 becomes +0): the same bits as pack2_bf16(fmaxf(lo, 0), fmaxf(hi, 0)) for every non-NaN input, with one VALU
// instruction per pair for the ReLU instead of two - the fused tower kernels are VALU-issue-bound in their epilogues (round 3: the
// res2 stage kernel took 32.6 k cycles per tile with every load and MFMA removed).
__device__ __forceinline__ uint32_t relu2_bf16(float lo, float hi) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    union { uint32_t u; s16x2_t v; } x;
    x.u = pack2_bf16(lo, hi);
    const s16x2_t z = {0, 0};
    x.v = __builtin_elementwise_max(x.v, z);
    return x.u;
}

// the same with the ReLU as a run-time, wave-uniform flag and no branch: max against INT16_MIN is the identity
__device__ __forceinline__ uint32_t act2_bf16(float lo, float hi, bool relu) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    union { uint32_t u; s16x2_t v; } x;
    x.u = pack2_bf16(lo, hi);
    const short f = relu ? (short)0 : (short)-32768;
    const s16x2_t z = {f, f};
    x.v = __builtin_elementwise_max(x.v, z);
    return x.u;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// The recurrences' activations on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each) instead of ocml's expf
// (range reduction) and an IEEE division (v_div_scale / Newton / v_div_fixup): ~30 of the ~300 instructions per row and step in
// kernels that are VALU-issue-bound.  Saturation is exact (exp2(+big) = inf -> rcp = 0; exp2(-big) = 0).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * fast_rcp(fast_exp(2.0f * x) + 1.0f); }
// branch-free tanh (ocml tanhf branches, which makes hipcc split the recurrence's fma chains across
// basic blocks and spill W_hh): saturates correctly, |abs err| ~1e-7
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f); }

// XCD-aware bijective remap of a linear block id: consecutive *logical* ids land on the same XCD
// (hardware round-robins physical ids over the 8 XCDs), so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    constexpr int NXCD = 8;
    const int q = nwg / NXCD, r = nwg % NXCD;
    const int xcd = bid % NXCD, slot = bid / NXCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

}  // namespace ivosw
