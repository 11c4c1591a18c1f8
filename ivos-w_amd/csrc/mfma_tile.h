// Device-side building blocks shared by the tower kernels (conv.hip, bottleneck.hip): the 128-B-row XOR-swizzled
// LDS tile image, LDS-DMA (global_load_lds_dwordx4) plumbing with counted vmcnt waits, and inline-asm LDS
// accesses that hipcc does not order against in-flight LDS-DMA.
#pragma once
#include "common.h"

namespace ivosw {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROWB = 128;  // bytes per tile row per K-tile (64 bf16 / 32 fp32 channels)

// byte offset of 16-B chunk `chunk` of row `row` in a swizzled [rows][128 B] image
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// runtime-selected counted wait (n is wave-uniform; folds to one s_waitcnt when n is a compile-time constant)
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        case 15: wait_vmcnt<15>(); break;
        case 16: wait_vmcnt<16>(); break;
        case 17: wait_vmcnt<17>(); break;
        case 18: wait_vmcnt<18>(); break;
        case 19: wait_vmcnt<19>(); break;
        case 20: wait_vmcnt<20>(); break;
        case 21: wait_vmcnt<21>(); break;
        case 22: wait_vmcnt<22>(); break;
        case 23: wait_vmcnt<23>(); break;
        case 24: wait_vmcnt<24>(); break;
        case 25: wait_vmcnt<25>(); break;
        case 26: wait_vmcnt<26>(); break;
        case 27: wait_vmcnt<27>(); break;
        case 28: wait_vmcnt<28>(); break;
        case 29: wait_vmcnt<29>(); break;
        case 30: wait_vmcnt<30>(); break;
        case 31: wait_vmcnt<31>(); break;
        case 32: wait_vmcnt<32>(); break;
        case 33: wait_vmcnt<33>(); break;
        case 34: wait_vmcnt<34>(); break;
        case 35: wait_vmcnt<35>(); break;
        case 36: wait_vmcnt<36>(); break;
        case 37: wait_vmcnt<37>(); break;
        case 38: wait_vmcnt<38>(); break;
        case 39: wait_vmcnt<39>(); break;
        case 40: wait_vmcnt<40>(); break;
        default: wait_vmcnt<0>(); break;
    }
}

// Fragment reads are inline asm: hipcc cannot prove a ds_read does not alias the in-flight LDS-DMA of another
// ring slot and would drain the whole pipeline (s_waitcnt vmcnt(0)) in front of every K-step.  Reads issued this
// way are not tracked by the compiler: lds_wait() (lgkmcnt(0) + scheduling fence) must precede their first use.
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// same with a compile-time byte offset (<= 65535): fragment rows that differ by a multiple of 8 rows share the swizzle
// key, so one address register per k-step + immediates replaces one register per (k-step, pixel tile) — which hipcc
// otherwise hoists out of the K loop and, in the register-heavy kernels, spills
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_o(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
__device__ __forceinline__ void lds_write_b64(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// 1 KiB per wave-instruction: lane l's 16 B land at lds_dst + 16*l (lds_dst wave-uniform)
__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    union { u32x4 u; bf16x8 v; } ua, ub;
    ua.u = a; ub.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, c, 0, 0, 0);
}

}  // namespace ivosw
