// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mi355_dit.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define FP8_E4M3_MAX 448.0f

// ---- host-side error plumbing -------------------------------------------
void mi355_set_error(const char* fmt, ...);
// MI355_ATTN_ASMV as read when the library was loaded (or at the last mi355_reload_env): nonzero
// = the attention kernel's LDS fragment reads as inline asm with hand-counted waits (default)
int mi355_env_attn_asmv();

#define MI355_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      mi355_set_error(__VA_ARGS__);       \
      return MI355_EINVAL;                \
    }                                     \
  } while (0)

#define MI355_CHECK_LAUNCH(name)                                              \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      mi355_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return MI355_EHIP;                                                      \
    }                                                                         \
  } while (0)

// ---- bf16 <-> f32 ----------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// Round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and
// keeps NaN a NaN.
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
// f32 -> bf16 -> f32 (an eager-mode bf16 rounding point)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// (a two-element vector conversion, so that both halves come out of ONE v_cvt_pk_bf16_f32
// wherever the call sits; two scalar casts are only merged when the compiler feels like it)
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// Device pointer (for the current device) of the two counters that record bf16 -> half
// conversions outside half's range: [0] blocks holding a value beyond +-65504 (stored
// saturated), [1] blocks whose largest magnitude is nonzero but below HALF_TINY_BLOCK, i.e.
// blocks that live entirely in half's subnormal range and lose relative precision.  Defined in
// elementwise.hip; returns nullptr (and sets the error string) when the symbol cannot be taken.
unsigned long long* mi355_f16_range_counters();

// 2^-13: with the largest value of a block at or above this, the subnormal spacing 2^-24 is
// below 2^-11 of it -- half's own mantissa resolution
#define HALF_TINY_BLOCK 1.220703125e-4f
#define HALF_MAX 65504.f

// Record one block's largest |value| (already reduced over the wave) in the range counters.
__device__ __forceinline__ void note_f16_block(unsigned long long* counters, float block_amax,
                                               int lane) {
  if (lane == 0) {
    if (block_amax > HALF_MAX) atomicAdd(counters, 1ull);
    else if (block_amax > 0.f && block_amax < HALF_TINY_BLOCK) atomicAdd(counters + 1, 1ull);
  }
}

// two floats -> bf16 (round to nearest even, the linear's result) -> the same values in IEEE
// half, packed lo | hi << 16.  Exact for bf16 values inside half's normal range; callers feed
// note_f16_block so that values outside it never pass silently.
__device__ __forceinline__ unsigned int pack_bf2_as_f16(float lo, float hi) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const float a = fminf(fmaxf(rbf(lo), -65504.f), 65504.f);
  const float b = fminf(fmaxf(rbf(hi), -65504.f), 65504.f);
  const h2 v = h2{(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned int, v);
}

// ---- f32 -> fp8 e4m3fn (OCP on gfx950), round-to-nearest-even ---------------
// v_cvt_pk_fp8_f32 converts two floats into two bytes of a dword.
__device__ __forceinline__ unsigned int cvt4_fp8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned int)w;
}
__device__ __forceinline__ float clamp_fp8(float v, float lo, float hi) {
  return fmaxf(fminf(v, hi), lo);
}

// ---- wave (64 lanes) reductions ---------------------------------------------
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ int64_t map_row(int64_t m, int64_t rpb, int64_t brows, int64_t off) {
  if (rpb == 0) return m;
  int64_t b = m / rpb;
  return b * brows + off + (m - b * rpb);
}

// 32-bit form for kernels whose row counts fit an int (cheap unsigned divide)
__device__ __forceinline__ int64_t map_row32(int m, int rpb, int64_t brows, int64_t off) {
  if (rpb == 0) return m;
  const unsigned b = (unsigned)m / (unsigned)rpb;
  return (int64_t)b * brows + off + (int64_t)((unsigned)m - b * (unsigned)rpb);
}

// tanh-GELU exactly as eager PyTorch evaluates it in fp32: 0.5*x*(1 + tanh(u)),
// u = sqrt(2/pi)*(x + 0.044715 x^3), INCLUDING the fp32 rounding of tanh(u) near +-1 (for
// x << 0 the sum 1 + tanh(u) cancels, and the eager result carries that quantisation, so a
// more accurate closed form would NOT match it).  tanh(u) = sign(u) * (1 - d),
// d = 2e/(1+e), e = exp(-2|u|) on the hardware exp2/rcp units; d is accurate to ~1e-7
// relative, so fl(1 - d) is the correctly rounded tanh up to rare double roundings.
// The operation sequence is spelled out (no contraction left to the compiler) so that the scalar
// form and the two-at-a-time form below give the same bits.
__device__ __forceinline__ float gelu_tanh_f(float x) {
#pragma clang fp contract(off)
  const float kBeta = 0.7978845608028654f;
  const float kKappa = 0.044715f;
  const float p = (kKappa * x) * x;
  const float u = kBeta * fmaf(p, x, x);
  const float e = __builtin_amdgcn_exp2f(fabsf(u) * -2.8853900817779268f);  // exp(-2|u|)
  // (1 - d) / 2 with d = 2e / (1 + e), and (1 + tanh) / 2: the halves are exact scalings of
  // the values in the formula above (the power of two commutes with every rounding), so
  // x * ((1 + tanh) / 2) has the bits of (0.5 x) * (1 + tanh) with four instructions less
  const float th_half = 0.5f + (-e) * __builtin_amdgcn_rcpf(1.0f + e);
  return x * (0.5f + copysignf(th_half, u));
}
// Two values at a time: the plain multiplies / adds / fmas become packed fp32 instructions
// (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32: two results per issue slot); exp2 and rcp stay
// one per value.  A GEMM epilogue with this activation is VALU-bound and not overlapped with
// MFMAs, so the instruction count is its time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_tanh_f2(f32x2 x) {
#pragma clang fp contract(off)
  const f32x2 kBeta = f32x2{0.7978845608028654f, 0.7978845608028654f};
  const f32x2 kKappa = f32x2{0.044715f, 0.044715f};
  const f32x2 half = f32x2{0.5f, 0.5f};
  const f32x2 p = (kKappa * x) * x;
  const f32x2 u = kBeta * __builtin_elementwise_fma(p, x, x);
  // |u| as a source modifier of a plain multiply (the packed form has none)
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(fabsf(u[0]) * -2.8853900817779268f),
                        __builtin_amdgcn_exp2f(fabsf(u[1]) * -2.8853900817779268f)};
  const f32x2 den = f32x2{1.0f, 1.0f} + e;
  const f32x2 rc = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 th_half = half + (-e) * rc;
  const f32x2 ths = f32x2{copysignf(th_half[0], u[0]), copysignf(th_half[1], u[1])};
  return x * (half + ths);
}
__device__ __forceinline__ float silu_f(float x) {
  return x / (1.0f + __expf(-x));
}

static inline mi355_rowmap rowmap_or_identity(const mi355_rowmap* p) {
  mi355_rowmap r = {0, 0, 0};
  if (p) r = *p;
  return r;
}
