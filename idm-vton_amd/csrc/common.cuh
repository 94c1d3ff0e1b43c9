// common.cuh -- shared device/host helpers for the gfx950 kernels of libidmvton_hip.so.
// CDNA4 only: wave = 64 lanes, MFMA 32x32x16 (bf16/f16 in, f32 accumulate), LDS-DMA (buffer_load ... lds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/idmvton_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// Per-dtype traits: 8-wide / 4-wide vector types and the 32x32x16 MFMA.
template <typename T> struct VT;
template <> struct VT<bf16_t> {
    typedef bf16x8 v8; typedef bf16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct VT<f16_t> {
    typedef f16x8 v8; typedef f16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Wave-uniform value the compiler can prove uniform (needed for M0 / SGPR operands).
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// 16-byte LDS-DMA: every lane supplies a byte offset into the buffer `rs`; lane l's 16 bytes land at
// lds_base + 16*l (lds_base must be wave-uniform).  Out-of-range offsets (>= num_records) read as zero.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, void* lds_base, uint32_t voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_base), 16, voff, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
#define OOB_SENTINEL 0x80000000u

// Row-per-lane MFMA epilogue, 16-bit outputs: `a` = this lane's 4 columns of group 2gp (8(2gp) + 4u + j), `b` = of group 2gp+1.
// Lanes 32..63 of a trade places with lanes 0..31 of b (v_permlane32_swap), after which every lane owns 8 CONSECUTIVE columns
// 16gp + 8u + (0..7) of its row: one 16-byte store instead of two 8-byte ones (the tail is store-issue bound).  `dst` points at
// column 16gp + 8u of the lane's row and must be 16-byte aligned; both lanes of a pair (l, l+32) must be active.
template <typename V4> __device__ __forceinline__ void store_cols8(void* dst, V4 a, V4 b) {
    static_assert(sizeof(V4) == 8, "4 x 16-bit");
    uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
    const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
    *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
}

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7 over the reals): 1 v_rcp + 1 v_exp + 7 FMA-class instructions instead of
// libm erff's two-range polynomial (~35 instructions) -- the GEGLU epilogue evaluates it 64 times per thread with the matrix pipe idle.
// GELU(x) = x/2 (1 + erf(x/sqrt 2)) then carries an absolute error <= 0.75e-7 |x|, four orders below one fp16 / bf16 rounding.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Wave-wide sum on the DPP path (no LDS crossbar: __shfl_xor is ds_bpermute_b32, ~100 cycles a step): quad butterflies, row mirrors,
// then the four 16-lane row totals through readlane.  The result is wave-uniform and the summation ORDER is fixed (deterministic).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xf, 0xf, false));   // row_mirror: every lane = its row's total
    const float r0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Exchange between the two 32-lane halves of a wave without the LDS crossbar (v_permlane32_swap: lanes 32-63 of the first
// operand swap with lanes 0-31 of the second): max / sum of a value with its partner lane (lane ^ 32).
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b%8): gives each XCD a contiguous range of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- host side -------------------------------------------------------------------------------------------------
int idmvton_set_error(int code, const char* fmt, ...);
#define CHECK_ARG(cond, code, ...) do { if (!(cond)) return idmvton_set_error(code, __VA_ARGS__); } while (0)
#define CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) \
    return idmvton_set_error(IDMVTON_E_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// ---- e4m3 (OCP fp8) packing, shared by attention_f8.hip and the GEMM epilogue's IDMVTON_IO_OUT_F8 form ----
__device__ __forceinline__ float clamp448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
// four floats -> four e4m3 bytes (byte i = value i)
__device__ __forceinline__ int pack4_fp8(float a, float b, float c, float d) {
    int x = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, x, true);
}
