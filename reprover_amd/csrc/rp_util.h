// Shared host/device helpers for libreprover_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/reprover_hip.h"

namespace rp {

// ---- error plumbing --------------------------------------------------------------------------
extern thread_local std::string g_last_error;

inline RpStatus fail(RpStatus code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline RpStatus fail(RpStatus code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define RP_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return rp::fail(RP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                      __LINE__);                                                                  \
  } while (0)

#define RP_CHECK_LAUNCH() RP_HIP(hipGetLastError())

#define RP_REQUIRE(cond, ...)                                                                     \
  do {                                                                                            \
    if (!(cond)) return rp::fail(RP_E_INVALID, __VA_ARGS__);                                      \
  } while (0)

// ---- optional per-kernel event timing (rp_profile_*) ------------------------------------------
void prof_begin(hipStream_t stream, int kernel_class);
void prof_end(hipStream_t stream);
struct ProfScope {
  hipStream_t s;
  ProfScope(hipStream_t stream, int kernel_class) : s(stream) { prof_begin(stream, kernel_class); }
  ~ProfScope() { prof_end(s); }
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: every launcher keeps one
// flag per device ordinal (static, per template instantiation) and sets the attribute on first use there.
constexpr int RP_MAX_DEVICES = 64;
struct LdsAttrOnce {
  bool done[RP_MAX_DEVICES] = {};
  hipError_t ensure(const void* kern, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < RP_MAX_DEVICES && done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < RP_MAX_DEVICES) done[dev] = true;
    return e;
  }
};

// ---- device types ----------------------------------------------------------------------------
typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B operand: 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;  // MX MFMA A/B operand: 32 fp8 = 8 VGPRs

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf2f(v); }

// float -> bfloat16, round-to-nearest-even (as torch's conversion).  Going through the native
// __bf16 type lets hipcc emit the gfx950 hardware conversion (v_cvt_pk_bf16_f32) instead of the
// five-instruction integer sequence.
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  f2_t v = {lo, hi};
  bf2_t h = __builtin_convertvector(v, bf2_t);
  return __builtin_bit_cast(uint32_t, h);
}

// gelu_new (tanh form), transformers/activations.py NewGELUActivation:
//   0.5 u (1 + tanh(z)),  z = sqrt(2/pi) (u + 0.044715 u^3)   ==   u * sigmoid(2 z)   ==   u / (1 + 2^(-2 z log2 e))
// evaluated with one v_exp_f32 and one v_rcp_f32 (both ~1 ulp; the result is rounded to bf16 by the
// caller).  u -> -inf gives 2^(+big) = inf, rcp = 0, result -0; u -> +inf gives u.
__device__ __forceinline__ float gelu_new(float u) {
  constexpr float k1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  constexpr float k2 = k1 * 0.044715f;
  const float a = u * __builtin_fmaf(k2, u * u, k1);
  return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}

// Two outputs of the row-scaled gated GELU  y = gelu_new(sc ag) (sc au)  (ag / au = the gate / up accumulators, sc = the
// token's RMSNorm factor, constant per lane in the GEMM epilogues) with the scale folded into per-lane constants (round 6):
//     y = ag au sc^2 / (1 + 2^(ag (K2 ag^2 + K1))),   K1 = k1 sc, K2 = k2 sc^3,   sc^2 / (1 + e) = 1 / (e / sc^2 + 1 / sc^2)
// = 4 multiplies + 2 fmas + exp2 + rcp per output, all but the two transcendentals as PACKED fp32 instructions (v_pk_mul_f32
// / v_pk_fma_f32: two outputs each) - 5.5 instructions per output where the unfolded form took 8 (the gated-GELU epilogue is
// issue-bound: ~5 us of VALU work per 256 x 256 tile).  Same limits as gelu_new: ag -> -inf gives e = inf, rcp = 0, y = -0.
typedef __attribute__((ext_vector_type(2))) float f32x2;
struct GegluConsts {
  f32x2 K1, K2, inv_s2;
  __device__ __forceinline__ explicit GegluConsts(float sc) {
    constexpr float k1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    constexpr float k2 = k1 * 0.044715f;
    const float s2 = sc * sc, i2 = __builtin_amdgcn_rcpf(s2);
    K1 = f32x2{k1 * sc, k1 * sc};
    K2 = f32x2{k2 * s2 * sc, k2 * s2 * sc};
    inv_s2 = f32x2{i2, i2};
  }
};
__device__ __forceinline__ f32x2 geglu2(f32x2 ag, f32x2 au, const GegluConsts& c) {
  const f32x2 a = ag * __builtin_elementwise_fma(c.K2, ag * ag, c.K1);
  const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const f32x2 d = __builtin_elementwise_fma(e, c.inv_s2, c.inv_s2);
  const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return (ag * au) * r;
}

// v of the lane a DPP control word names (quad_perm / row_half_mirror ...): ONE VALU instruction where __shfl_xor goes through
// the LDS crossbar (xor, shift, ds_bpermute_b32, compare, select: five).  0xB1 = quad_perm(1,0,3,2) = lane ^ 1,
// 0x4E = quad_perm(2,3,0,1) = lane ^ 2, 0x141 = row_half_mirror = lane 7 - i of each group of 8.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// monotone float <-> uint32 map (larger float = larger uint)
__host__ __device__ __forceinline__ uint32_t f2ord(float f) {
#ifdef __HIP_DEVICE_COMPILE__
  uint32_t u = __float_as_uint(f);
#else
  uint32_t u;
  memcpy(&u, &f, 4);
#endif
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// CUs of the current device: an attribute query (not the slow property struct), cached per device.  ONE source for the
// planners below and for encode_pass's main_rows(): plans computed in two places must agree on a part that does not have
// 256 CUs.
inline int device_cu_count() {
  static int cached_dev = -1, cached = 256;
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev == cached_dev) return cached;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) {
    cached_dev = dev;
    cached = v;
    return v;
  }
  return 256;
}

}  // namespace rp
