// Shared device helpers for the gfx950 kernels.  wave = 64 lanes; all block sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gan_amd.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define GAN_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

// Every kernel launch of the library goes through this macro: the diagnostic launch log (host_util.cpp, option diag.launch_log)
// notes the kernel's host stub, then the launch proceeds as hipLaunchKernelGGL.
void gan_launch_note(const void* kernel_host_fn);
size_t gan_launch_log_ptrs(const void** out, size_t cap);
#define GAN_LAUNCH(kern, ...)                       \
  do {                                              \
    gan_launch_note((const void*)(kern));           \
    hipLaunchKernelGGL(kern, __VA_ARGS__);          \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ float ld_f(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void st_f(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_f(bf16_t* p, float v) { *p = (bf16_t)v; }
__device__ __forceinline__ void st_f(f16_t* p, float v) { *p = (f16_t)v; }

template <typename T> struct VecOf;  // elements per 16-byte vector
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
template <> struct VecOf<f16_t> { static constexpr int N = 8; };

// unpack a 16-byte vector of T into floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y);
  out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  out[4] = __uint_as_float(v.z << 16); out[5] = __uint_as_float(v.z & 0xffff0000u);
  out[6] = __uint_as_float(v.w << 16); out[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ float h2f(uint32_t bits16) { const uint16_t u = (uint16_t)bits16; return (float)*(const f16_t*)&u; }
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& v, float* out) {
  out[0] = h2f(v.x); out[1] = h2f(v.x >> 16); out[2] = h2f(v.y); out[3] = h2f(v.y >> 16);
  out[4] = h2f(v.z); out[5] = h2f(v.z >> 16); out[6] = h2f(v.w); out[7] = h2f(v.w >> 16);
}
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  f16_t a = (f16_t)lo, b = (f16_t)hi;
  return (uint32_t)(*(uint16_t*)&a) | ((uint32_t)(*(uint16_t*)&b) << 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16_t a = (bf16_t)lo, b = (bf16_t)hi;
  return (uint32_t)(*(uint16_t*)&a) | ((uint32_t)(*(uint16_t*)&b) << 16);
}
// two floats -> one 32-bit word of the 16-bit storage type T
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_h2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<float>(float lo, float hi) { return 0; }   // (never used: fp32 stores floats)
template <typename T> __device__ __forceinline__ uint4 pack16(const float* in);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* in) {
  return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
}
template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* in) {
  return make_uint4(pack_h2(in[0], in[1]), pack_h2(in[2], in[3]), pack_h2(in[4], in[5]), pack_h2(in[6], in[7]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* in) {
  return make_uint4(pack_bf2(in[0], in[1]), pack_bf2(in[2], in[3]), pack_bf2(in[4], in[5]), pack_bf2(in[6], in[7]));
}

// 16x16x32 MFMA on 16-bit operands held as 4 dwords (8 elements) per lane, fp32 accumulate
template <typename T> __device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<bf16_t>(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<f16_t>(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8*)&a, *(const f16x8*)&b, c, 0, 0, 0);
}
int gan_opt(const char* key);      // planner options (host_util.cpp; changed only by gan_set_option)
// One element of the TF-form Adam step (Keras optimizer_v2/adam.py): the ONE definition used by every kernel that applies it, so
// that the update fused into a wgrad epilogue is bit-identical to the stand-alone kernels.
__device__ __forceinline__ void gan_adam1(float& p, float& m, float& v, float g, float gscale, float omb1, float omb2, float lr, float eps) {
  const float gr = g * gscale;
  m += (gr - m) * omb1;
  v += (gr * gr - v) * omb2;
  p -= (m * lr) / (sqrtf(v) + eps);
}

static inline bool gan_dtype_ok(int dtype) { return dtype == GAN_F32 || dtype == GAN_BF16 || dtype == GAN_F16; }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == GAN_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == GAN_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GAN_ACT_TANH) return tanhf(v);
  return v;
}

// compile-time activation (the run-time form costs a chain of selects per element in streaming epilogues)
template <int ACT> __device__ __forceinline__ float act_c(float z, float slope) {
  if (ACT == GAN_ACT_LRELU) return z > 0.f ? z : z * slope;
  if (ACT == GAN_ACT_RELU) return z > 0.f ? z : 0.f;
  if (ACT == GAN_ACT_TANH) return tanhf(z);
  return z;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ilog2_exact(int v) {  // -1 if not a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Division by a launch-invariant divisor (Granlund-Montgomery): q = (t + ((n - t) >> s1)) >> s2, t = mulhi(mp, n)
struct FastDiv {
  uint32_t mp, s1, s2, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.mp = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  uint32_t t = __umulhi(f.mp, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
