// Split-precision matrix-core arithmetic shared by the gfx950 GEMM-shaped kernels:
// an fp32 operand is written as x = x1 + x2 + x3 with bf16 terms (round-to-nearest-even), and a
// product A*B is accumulated in fp32 from the six term products of order <= 2
//     a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2)
// on v_mfma_f32_32x32x16_bf16: fp32-rounding-class error at 6/16 of the fp32 matrix-pipe time.
#pragma once
#include "snet_common.h"

namespace snet {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// native clang vectors: HIP's float4/uint4 are structs wrapping unions, and arrays of them are not
// scalarised by SROA (they end up in scratch memory)
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct Split3 {
  bf16x8 t[3];
};

using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

__device__ __forceinline__ bf16x8 cat4(bf16x2 a, bf16x2 b, bf16x2 c, bf16x2 d) {
  const bf16x4 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3);
  const bf16x4 cd = __builtin_shufflevector(c, d, 0, 1, 2, 3);
  return __builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7);
}

// x = x1 + x2 + x3 (bf16 each, round-to-nearest-even via v_cvt_pk_bf16_f32); pure register code
__device__ __forceinline__ Split3 split8(const float (&v)[8]) {
  bf16x2 h[4], m[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f32x2 x = {v[2 * p], v[2 * p + 1]};
    h[p] = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h[p], f32x2);
    m[p] = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m[p], f32x2);
    l[p] = __builtin_convertvector(r2, bf16x2);
  }
  Split3 s;
  s.t[0] = cat4(h[0], h[1], h[2], h[3]);
  s.t[1] = cat4(m[0], m[1], m[2], m[3]);
  s.t[2] = cat4(l[0], l[1], l[2], l[3]);
  return s;
}

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 u) { return __builtin_bit_cast(bf16x8, u); }

// acc += A * B with both operands given as 3-term splits (A from packed fragments a[0..2])
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], const Split3 &b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.t[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b.t[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.t[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.t[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.t[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.t[0], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 mfma6(const Split3 &a, const bf16x8 (&b)[3], f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b[0], acc, 0, 0, 0);
  return acc;
}


__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// ---- N-term variants for the 16x16x32 tiles of the fused tensor-product kernels --------------------
// bf16 terms (F16 = false): NT = 3: six products of order <= 2 (fp32-rounding class); NT = 2: a0b0 + a0b1 + a1b0
// (about 2^-17 relative per product, "bf16x3"); NT = 1: plain bf16 operands.
// fp16 terms (F16 = true, NT = 2, "f16x3"): x = hi + lo with two fp16 terms (round-to-nearest: 22 significand
// bits), three products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 -- the dropped lo*lo term is 2^-22
// relative, so the result is in the fp32-rounding class (measured 8e-8 rms relative on a K = 64 product, plain
// fp32 FMA chains give 1.4e-7) at the matrix-core cost of bf16x3.  fp16 has 5 exponent bits: the CALLER scales
// each operand by a power of two so that its largest magnitude is below 2^14 (scale_exp below; weights on the
// host, activations / gradients per tile in the kernel); entries more than 17 binades below the largest lose
// relative precision gradually (absolute error <= 2^-25 of the scaled unit), entries can never overflow.
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int NT>
struct SplitN {
  bf16x8 t[NT];  // 128-bit operand registers (fp16 terms are carried in the same container)
};

__device__ __forceinline__ bf16x8 cat4h(f16x2 a, f16x2 b, f16x2 c, f16x2 d) {
  const f16x4 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3);
  const f16x4 cd = __builtin_shufflevector(c, d, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8, (f16x8)__builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NT, bool F16 = false>
__device__ __forceinline__ SplitN<NT> splitn8(const float (&v)[8]) {
  SplitN<NT> s;
  if constexpr (F16) {
    static_assert(NT == 2, "fp16 operands are split into two terms");
    f16x2 h[4];
    unsigned lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f32x2 x = {v[2 * p], v[2 * p + 1]};
      h[p] = __builtin_convertvector(x, f16x2);   // v_cvt_pk_f16_f32, round-to-nearest-even
      // lo = fp16(x - hi): one mixed-precision FMA per value (fp16 hi half x -1.0 + fp32 x, result rounded to fp16 into one half of
      // the destination) instead of two conversions back to fp32, a packed subtraction and a packed conversion per PAIR -- 12
      // instead of 20 vector instructions per 8-value operand (round 6; hipcc does not select v_fma_mix* for the C expression).
      // x - hi is exact in fp32, so the single rounding to fp16 is the one the four-instruction form performed.
      const unsigned hp = __builtin_bit_cast(unsigned, h[p]);
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lw[p]) : "v"(hp), "v"(x[0]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw[p]) : "v"(hp), "v"(x[1]));
    }
    // A matrix instruction must not read a register within two wait states of the vector instruction that wrote it; hipcc pads
    // that for instructions it knows, not for the contents of an asm statement (measured: the last-layer lmax-3 kernels returned
    // g_h2 3e-3 off -- stale lo terms -- until this was here).  One s_nop for the whole operand, tied to all four words so that it
    // stays between the last of the eight FMAs and the first matrix instruction that reads them.
    asm("s_nop 1" : "+v"(lw[0]), "+v"(lw[1]), "+v"(lw[2]), "+v"(lw[3]));
    f16x2 l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) l[p] = __builtin_bit_cast(f16x2, lw[p]);
    s.t[0] = cat4h(h[0], h[1], h[2], h[3]);
    s.t[1] = cat4h(l[0], l[1], l[2], l[3]);
  } else {
    bf16x2 q[NT][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x2 r = {v[2 * p], v[2 * p + 1]};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        q[t][p] = __builtin_convertvector(r, bf16x2);
        if (t + 1 < NT) r = r - __builtin_convertvector(q[t][p], f32x2);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) s.t[t] = cat4(q[t][0], q[t][1], q[t][2], q[t][3]);
  }
  return s;
}

__device__ __forceinline__ f32x4 mfma16_f16(bf16x8 a, bf16x8 b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

// acc += A * B on v_mfma_f32_16x16x32_{bf16,f16}, A given as NT packed fragments, B as an NT-term split;
// smallest products first
template <int NT, bool F16 = false>
__device__ __forceinline__ f32x4 mfma16_split(const bf16x8 (&a)[NT], const SplitN<NT> &b, f32x4 acc) {
  if constexpr (F16) {
    acc = mfma16_f16(a[0], b.t[1], acc);
    acc = mfma16_f16(a[1], b.t[0], acc);
    acc = mfma16_f16(a[0], b.t[0], acc);
    return acc;
  } else {
    if constexpr (NT == 3) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b.t[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b.t[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b.t[1], acc, 0, 0, 0);
    }
    if constexpr (NT >= 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b.t[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b.t[0], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b.t[0], acc, 0, 0, 0);
    return acc;
  }
}

// the same with A as the device-side split and B as packed fragments
template <int NT, bool F16 = false>
__device__ __forceinline__ f32x4 mfma16_split(const SplitN<NT> &a, const bf16x8 (&b)[NT], f32x4 acc) {
  if constexpr (F16) {
    acc = mfma16_f16(a.t[0], b[1], acc);
    acc = mfma16_f16(a.t[1], b[0], acc);
    acc = mfma16_f16(a.t[0], b[0], acc);
    return acc;
  } else {
    if constexpr (NT == 3) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[1], b[1], acc, 0, 0, 0);
    }
    if constexpr (NT >= 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[1], b[0], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[0], b[0], acc, 0, 0, 0);
    return acc;
  }
}

// ---- power-of-two operand scaling for the fp16 terms (all wave-uniform: the factors live in SGPRs) ------------
// largest value of a non-negative per-lane quantity over the wavefront, returned in every lane
__device__ __forceinline__ float wave_max(float m) {
  m = fmaxf(m, lane_xor<1>(m));
  m = fmaxf(m, lane_xor<2>(m));
  m = fmaxf(m, lane_xor<4>(m));
  m = fmaxf(m, lane_xor<8>(m));
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(m)));
}
// e with m < 2^e (m finite, >= 0; 0 for m == 0), clamped so that 2^(+-(14 - e)) stays a normal fp32 number
__device__ __forceinline__ int bound_exp(float m) {
  const int e = __builtin_amdgcn_frexp_expf(m);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
// 2^k as fp32, k clamped to the normal range (callers add up to four exponents)
__device__ __forceinline__ float pow2f(int k) { return __builtin_ldexpf(1.0f, k < -126 ? -126 : (k > 126 ? 126 : k)); }
constexpr int F16_TOP = 14;  // scaled operands stay below 2^14 (fp16 overflows at 2^16)
__device__ __forceinline__ float max8(const float (&v)[8]) {
  return fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
               fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
}

}  // namespace snet
