// Device-side activation I/O shared by the 16-bit kernels: activations live in HBM as fp32 (f32 / f16x3
// modes) or as bf16 (bf16 mode: half the traffic); everything in registers is fp32.
#pragma once
#include "fdsr_kernels.h"

namespace fdsr {

typedef float f32x4_io __attribute__((ext_vector_type(4)));

// activations in HBM: fp32 (f32 / f16x3 modes) or bf16 (bf16 mode)
__device__ __forceinline__ float bf16_bits_to_f32(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// f16x3 range guard: the hi/lo split clamps every staged value to the f16 range.  GroupNorm'ed inputs are re-scaled before the
// split, but a RAW conv input (res_conv, down/upsample convs, the rider's chunks) beyond +-65504 would be clamped SILENTLY:
// the staging paths of raw inputs raise a sticky device flag instead (fdsr_check_saturation reads and clears it).
__device__ __forceinline__ void sat_check(int* flag, f32x4_io v, float lim) {
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  if (m > lim) *flag = 1;
}
// the XCD this wave runs on (HW_REG_XCC_ID[3:0]): the shard of the consumer-side GroupNorm tables its workgroup adds to
__device__ __forceinline__ int xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return (int)(v & (GSUM_SHARDS - 1));
}
// one channel pair's (sum, sumsq) of a producer tile -> this XCD's shard of the tensor's fixed-point table (ConvParams::gsum_out).
// Workgroup-scope atomics carry no sc1: the XCD's L2 executes them (all adders of the shard share that L2), ~10 ns each instead of a
// memory-side read-modify-write per adder; the end of the kernel writes the lines back for the consumer launch.
__device__ __forceinline__ void gsum_add(unsigned long long* tab, int n, int npairs, int pair, int shard, float s1, float s2) {
  unsigned long long* g = tab + (((size_t)n * GSUM_SHARDS + shard) * npairs + pair) * 2;
  const long long a = __float2ll_rn(s1 * (float)(1 << GSUM_BITS1)), b = __float2ll_rn(s2 * (float)(1 << GSUM_BITS2));
  __hip_atomic_fetch_add(g, (unsigned long long)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(g + 1, (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)v); }
// f16 stores saturate (an fp32 value beyond +-65504 would become inf and poison every later layer; the f16x3 mode clamps the same
// way when it splits): v_med3_f32 + v_cvt_f16_f32
__device__ __forceinline__ unsigned short f32_to_f16_bits(float v) {
  return __builtin_bit_cast(unsigned short, (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f));
}
template <int PREC> struct ActIO;
template <> struct ActIO<PREC_F16X3> {
  typedef f32x4_io Quad;   // four consecutive channels as loaded
  static constexpr int ESZ = 4;
  static __device__ __forceinline__ Quad load4(const float* base, size_t idx) { return *reinterpret_cast<const f32x4_io*>(base + idx); }
  static __device__ __forceinline__ f32x4_io widen(Quad q) { return q; }
  static __device__ __forceinline__ float load1(const float* base, size_t idx) { return base[idx]; }
  static __device__ __forceinline__ void store1(float* base, size_t idx, float v) { base[idx] = v; }
};
template <> struct ActIO<PREC_BF16> {
  typedef uint2 Quad;
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ Quad load4(const float* base, size_t idx) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
  }
  static __device__ __forceinline__ f32x4_io widen(Quad q) {
    f32x4_io r = {__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xffff0000u),
               __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xffff0000u)};
    return r;
  }
  static __device__ __forceinline__ float load1(const float* base, size_t idx) {
    return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(base)[idx]);
  }
  static __device__ __forceinline__ void store1(float* base, size_t idx, float v) {
    reinterpret_cast<unsigned short*>(base)[idx] = f32_to_bf16_bits(v);
  }
};
template <> struct ActIO<PREC_F16> {
  typedef uint2 Quad;
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ Quad load4(const float* base, size_t idx) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
  }
  static __device__ __forceinline__ f32x4_io widen(Quad q) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, q.x), b = __builtin_bit_cast(h2, q.y);
    f32x4_io r = {(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
    return r;
  }
  static __device__ __forceinline__ float load1(const float* base, size_t idx) {
    return f16_bits_to_f32(reinterpret_cast<const unsigned short*>(base)[idx]);
  }
  static __device__ __forceinline__ void store1(float* base, size_t idx, float v) {
    reinterpret_cast<unsigned short*>(base)[idx] = f32_to_f16_bits(v);
  }
};

// four fp32 values -> four 16-bit values of the mode's storage format, packed (PREC_BF16 / PREC_F16)
template <int PREC>
__device__ __forceinline__ uint2 pack4_16(f32x4_io v) {
  uint2 pk;
  if (PREC == PREC_F16) {
    pk.x = (unsigned)f32_to_f16_bits(v[0]) | ((unsigned)f32_to_f16_bits(v[1]) << 16);
    pk.y = (unsigned)f32_to_f16_bits(v[2]) | ((unsigned)f32_to_f16_bits(v[3]) << 16);
  } else {
    pk.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
    pk.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
  }
  return pk;
}
// ... by the run-time code of ConvParams::out_bf16 (1 bf16, 2 f16)
__device__ __forceinline__ unsigned short f32_to_16_bits(float v, int code) { return code == 2 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v); }
// the MFMA operand form of a staged quad: no range clamp (GroupNorm'ed / Swish'ed values and bf16 / f16 activations read back are in
// range; a RAW fp32 input is clamped by the caller where the f16x3 mode clamps it)
template <int PREC>
__device__ __forceinline__ uint2 stage4_16(f32x4_io v) {
  if (PREC == PREC_F16) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 hb = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    return __builtin_bit_cast(uint2, hb);
  }
  typedef __bf16 b4_ __attribute__((ext_vector_type(4)));
  const b4_ hb = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  return __builtin_bit_cast(uint2, hb);
}
// one 16-bit MFMA of the mode: v_mfma_f32_16x16x32_{bf16,f16} / v_mfma_f32_32x32x16_{bf16,f16} on 16-byte operand registers
typedef float io_f32x4 __attribute__((ext_vector_type(4)));
typedef float io_f32x16 __attribute__((ext_vector_type(16)));
template <int PREC>
__device__ __forceinline__ io_f32x4 mfma16_k32(uint4 a, uint4 b, io_f32x4 c) {
  typedef __bf16 b8_ __attribute__((ext_vector_type(8)));
  typedef _Float16 h8_ __attribute__((ext_vector_type(8)));
  if (PREC == PREC_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_, a), __builtin_bit_cast(b8_, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_, a), __builtin_bit_cast(h8_, b), c, 0, 0, 0);
}
template <int PREC>
__device__ __forceinline__ io_f32x16 mfma32_k16(uint4 a, uint4 b, io_f32x16 c) {
  typedef __bf16 b8_ __attribute__((ext_vector_type(8)));
  typedef _Float16 h8_ __attribute__((ext_vector_type(8)));
  if (PREC == PREC_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_, a), __builtin_bit_cast(b8_, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_, a), __builtin_bit_cast(h8_, b), c, 0, 0, 0);
}

}  // namespace fdsr
