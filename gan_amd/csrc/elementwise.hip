// HBM-bound kernels of the training step other than normalisation (norm.hip): BCE-from-logits / L1 losses
// with fused gradients, TF-form Adam, weight layout preparation, dropout-mask generation and
// fp32<->typed packing.  Reductions are two-stage and deterministic, never atomics.
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bce_kernel(const float* x, long long count, float target, float grad_scale, T* dx,
                                                  int dx_pitch, float* partial, const float* ls) {
  if (ls) grad_scale *= ls[0];          // dynamic loss scale (fp16 path): gradients only, the reported loss is unscaled
  // one logit per thread and step (a single 1024-thread block spent 16 us in libm on one CU); block sums go to
  // `partial`, l1_finalize_kernel adds them in a fixed order
  __shared__ float red[4];
  float s = 0.f;
  const float inv = 1.0f / (float)count;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    const float e = expf(-fabsf(v));
    s += fmaxf(v, 0.f) - v * target + log1pf(e);
    if (dx) {
      const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      st_f(dx + i * dx_pitch, grad_scale * (sig - target) * inv);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// The three BinaryCrossentropy terms of one Pix2Pix step in ONE pass over the two logit maps (pix2pix.py:167-188,
// base_gan.py:227-245): gan_loss = BCE(1, D(fake)); disc_loss = 0.5 * (BCE(1, D(real)) + BCE(0, D(fake))), with the
// three logit gradients.  Block partial sums [blocks][3] -> patchgan_finalize_kernel (fixed order), which also forms
// gen_total_loss = gan_loss + lambda * l1 from the L1 term already in the loss vector.
template <typename T>
__global__ __launch_bounds__(256) void patchgan_bce_kernel(const float* real, const float* fake, long long count,
                                                           T* g_dfake, T* d_dreal, T* d_dfake, int pitch, float* partial,
                                                           const float* ls) {
  __shared__ float red[4][3];
  float s[3] = {0.f, 0.f, 0.f};
  const float inv = (ls ? ls[0] : 1.0f) / (float)count;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float r = real[i], f = fake[i];
    const float er = expf(-fabsf(r)), ef = expf(-fabsf(f));
    const float lr = log1pf(er), lf = log1pf(ef);
    s[0] += fmaxf(f, 0.f) - f + lf;                 // BCE(1, fake)
    s[1] += fmaxf(r, 0.f) - r + lr;                 // BCE(1, real)
    s[2] += fmaxf(f, 0.f) + lf;                     // BCE(0, fake)
    const float sr = r >= 0.f ? 1.f / (1.f + er) : er / (1.f + er);
    const float sf = f >= 0.f ? 1.f / (1.f + ef) : ef / (1.f + ef);
    if (g_dfake) st_f(g_dfake + i * pitch, (sf - 1.f) * inv);
    if (d_dreal) st_f(d_dreal + i * pitch, 0.5f * (sr - 1.f) * inv);
    if (d_dfake) st_f(d_dfake + i * pitch, 0.5f * sf * inv);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float w = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = w;
  }
  __syncthreads();
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(64) void patchgan_finalize_kernel(const float* partial, int n, double inv_count, float lambda,
                                                               const float* l1, float* gen_total, float* gan_loss, float* disc_loss) {
  double s = 0;
  const int k = threadIdx.x;                          // lanes 0..2: one BCE term each, summed in block order
  if (k < 3)
    for (int i = 0; i < n; ++i) s += partial[i * 3 + k];
  const float v = (float)(s * inv_count);
  const float g = __shfl(v, 0, 64), r = __shfl(v, 1, 64), f = __shfl(v, 2, 64);
  if (k == 0) {
    *gan_loss = g;
    *disc_loss = 0.5f * r + 0.5f * f;
    if (gen_total) *gen_total = g + lambda * (l1 ? *l1 : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void l1_kernel(const T* a, int apitch, const T* b, int bpitch, int C, long long pixels,
                                                 float gscale, T* da, int dapitch, float* partial, const float* ls) {
  __shared__ float red[4];
  float s = 0.f;
  if (ls) gscale *= ls[0];
  const long long total = pixels * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long px = i / C;
    int c = (int)(i % C);
    float d = ld_f(a + px * apitch + c) - ld_f(b + px * bpitch + c);
    s += fabsf(d);
    if (da) st_f(da + px * dapitch + c, d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void l1_finalize_kernel(const float* partial, int n, double inv_count, float loss_scale,
                                                          int acc, float* loss_out) {
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float t = (float)(red[0] * inv_count) * loss_scale;
    loss_out[0] = acc ? loss_out[0] + t : t;
  }
}

// ------------------------------------------------------------------------------------------------
// Adam (TF form), weight prep, dropout, pack
// ------------------------------------------------------------------------------------------------
// Dynamic loss scale state (fp16 path), 4 floats on the device: [0] scale, [1] 1/scale, [2] consecutive finite steps,
// [3] != 0: this step's gradients hold an inf/nan -> every Adam kernel (and the step counter) skips the step.
__global__ __launch_bounds__(256) void grads_check_kernel(const uint4* __restrict__ g, long long nvec, float* ls) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const uint4 v = g[i];
    bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
           ((v.w & 0x7f800000u) == 0x7f800000u);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) ls[3] = 1.0f;       // every writer stores the same value
}
__global__ void loss_scale_update_kernel(float* ls, int growth_interval, float max_scale) {
  float scale = ls[0], good = ls[2];
  if (ls[3] != 0.f) { scale = fmaxf(scale * 0.5f, 1.0f); good = 0.f; }
  else if (++good >= (float)growth_interval) { scale = fminf(scale * 2.0f, max_scale); good = 0.f; }
  ls[0] = scale; ls[1] = 1.0f / scale; ls[2] = good; ls[3] = 0.f;
}
__global__ void adam_begin_kernel(int32_t* step, float* lr_t, float lr, float b1, float b2, const float* ls) {
  if (ls && ls[3] != 0.f) return;        // skipped step: iterations and lr_t stay
  int t = *step + 1;
  *step = t;
  double v = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
  *lr_t = (float)v;
}
// GW: the gradient is read from the data-parallel wire buffer (bf16, summed over ranks) instead of the fp32 gradient buffer
__device__ __forceinline__ float4 ld_grad4(const void* g, size_t i4, bool wire) {
  if (!wire) return ((const float4*)g)[i4];
  const uint2 q = ((const uint2*)g)[i4];
  return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
}
template <bool GW>
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v,
                                                   const void* __restrict__ g, long long nvec, const float* lr_t,
                                                   float omb1, float omb2, float eps, float gscale, const float* ls) {
  const float lr = *lr_t;
  if (ls) { if (ls[3] != 0.f) return; gscale *= ls[1]; }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    float4 pp = p[i], mm = m[i], vv = v[i];
    const float4 gg = ld_grad4(g, (size_t)i, GW);
#define ADAM1(f)                                     \
  {                                                  \
    float gr = gg.f * gscale;                        \
    mm.f += (gr - mm.f) * omb1;                      \
    vv.f += (gr * gr - vv.f) * omb2;                 \
    pp.f -= (mm.f * lr) / (sqrtf(vv.f) + eps);       \
  }
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

// master [16][A][B] fp32 -> nat [16][A][B8] and tr [16][B][A8] (typed).  64x64 tiles through LDS so that the
// master read, the nat write (both along B) and the tr write (along A) are all coalesced.
template <typename T>
__global__ __launch_bounds__(256) void wprep_kernel(const float* master, int A, int B, T* nat, T* tr) {
  __shared__ float tile[64][65];
  const int B8 = (B + 7) & ~7, A8 = (A + 7) & ~7;
  const int b0 = blockIdx.x * 64, a0 = blockIdx.y * 64, tap = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int a = a0 + ty + 4 * i, b = b0 + tx;
    float v = (a < A && b < B) ? master[((size_t)tap * A + a) * B + b] : 0.f;
    tile[ty + 4 * i][tx] = v;
    if (nat && a < A && b < B8) st_f(nat + ((size_t)tap * A + a) * B8 + b, v);
  }
  __syncthreads();
  if (tr) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      int b = b0 + ty + 4 * i, a = a0 + tx;
      if (b < B && a < A8) st_f(tr + ((size_t)tap * B + b) * A8 + a, tile[tx][ty + 4 * i]);
    }
  }
}

// all kernels of a network in ONE launch: blockIdx.x -> (tensor, tap, 64x64 tile) through a small device table
struct PrepEntry { const float* master; void* nat; void* tr; int32_t A, B, tile_start, tiles_b; };
template <typename T>
__global__ __launch_bounds__(256) void wprep_multi_kernel(const PrepEntry* ents, int n) {
  __shared__ float tile[64][65];
  int e = 0;
  while (e + 1 < n && (int)blockIdx.x >= ents[e + 1].tile_start) ++e;
  const PrepEntry en = ents[e];
  const int A = en.A, B = en.B;
  const int B8 = (B + 7) & ~7, A8 = (A + 7) & ~7;
  const int tiles_a = (A8 + 63) / 64;
  int t = blockIdx.x - en.tile_start;
  const int tb = t % en.tiles_b; t /= en.tiles_b;
  const int ta = t % tiles_a, tap = t / tiles_a;
  const int b0 = tb * 64, a0 = ta * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  T* nat = (T*)en.nat;
  T* tr = (T*)en.tr;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int a = a0 + ty + 4 * i, b = b0 + tx;
    float v = (a < A && b < B) ? en.master[((size_t)tap * A + a) * B + b] : 0.f;
    tile[ty + 4 * i][tx] = v;
    if (nat && a < A && b < B8) st_f(nat + ((size_t)tap * A + a) * B8 + b, v);
  }
  __syncthreads();
  if (tr) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      int b = b0 + ty + 4 * i, a = a0 + tx;
      if (b < B && a < A8) st_f(tr + ((size_t)tap * B + b) * A8 + a, tile[tx][ty + 4 * i]);
    }
  }
}

// TF-form Adam fused with the NK weight prep: one 64x64 tile of one kernel tensor per block.  Each weight is read
// once with its gradient and moments, updated, and leaves as fp32 master + moments and as the two typed copies the
// GEMMs consume (the transposed one through the LDS tile), instead of Adam (28 B/param) followed by a prep pass that
// re-reads the master (8 B/param more).  16-byte accesses along the contiguous axis when the tensor allows.
struct AdamBases { float* master; float* m; float* v; const void* grad; };
template <typename T, bool GW>
__global__ __launch_bounds__(256) void adam_prep_multi_kernel(const PrepEntry* ents, int n, AdamBases ab, const float* lr_t,
                                                              float omb1, float omb2, float eps, float gscale, const float* ls) {
  // the tile only feeds the TYPED transposed copy: it holds the updated weights already rounded to T (16-bit paths: 8.4 KB instead of
  // 16.6 KB - a workgroup of this HBM-bound pass now fits into the 16 KB of LDS a 256x128 ping-pong GEMM block leaves on its CU)
  typedef typename std::conditional<sizeof(T) == 2, uint16_t, float>::type TileE;
  __shared__ TileE tile[64][sizeof(T) == 2 ? 66 : 65];
  auto enc = [](float w) -> TileE {
    if constexpr (sizeof(T) == 2) return (TileE)(pack2<T>(w, 0.f) & 0xffffu);
    else return w;
  };
  if (ls) { if (ls[3] != 0.f) return; gscale *= ls[1]; }      // skipped step (weights and their NK copies stay) / unscale
  int e = 0;
  while (e + 1 < n && (int)blockIdx.x >= ents[e + 1].tile_start) ++e;
  const PrepEntry en = ents[e];
  const int A = en.A, B = en.B;
  const int B8 = (B + 7) & ~7, A8 = (A + 7) & ~7;
  const int tiles_a = (A8 + 63) / 64;
  int t = blockIdx.x - en.tile_start;
  const int tb = t % en.tiles_b; t /= en.tiles_b;
  const int ta = t % tiles_a, tap = t / tiles_a;
  const int b0 = tb * 64, a0 = ta * 64;
  const size_t base = (size_t)(en.master - ab.master);        // this tensor's offset in the flat buffers
  const float lr = *lr_t;
  T* nat = (T*)en.nat;
  T* tr = (T*)en.tr;
  auto adam1 = [&](float& p, float& m, float& v, float g) { gan_adam1(p, m, v, g, gscale, omb1, omb2, lr, eps); };
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 4 consecutive b per thread, 16 rows per pass
  const bool vecB = (B & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int a = a0 + ty + 16 * i, b = b0 + tx * 4;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (a < A && b < B) {
      const size_t o = base + ((size_t)tap * A + a) * B + b;
      if (vecB) {
        float4 pp = *(const float4*)(ab.master + o), mm = *(const float4*)(ab.m + o), vv = *(const float4*)(ab.v + o);
        const float4 gg = ld_grad4(ab.grad, o >> 2, GW);          // o % 4 == 0 here (B % 4 == 0, 64-element aligned tensors)
        adam1(pp.x, mm.x, vv.x, gg.x); adam1(pp.y, mm.y, vv.y, gg.y); adam1(pp.z, mm.z, vv.z, gg.z); adam1(pp.w, mm.w, vv.w, gg.w);
        *(float4*)(ab.master + o) = pp; *(float4*)(ab.m + o) = mm; *(float4*)(ab.v + o) = vv;
        w[0] = pp.x; w[1] = pp.y; w[2] = pp.z; w[3] = pp.w;
      } else {
        for (int k = 0; k < 4 && b + k < B; ++k) {
          float pp = ab.master[o + k], mm = ab.m[o + k], vv = ab.v[o + k];
          adam1(pp, mm, vv, GW ? __uint_as_float((uint32_t)((const uint16_t*)ab.grad)[o + k] << 16) : ((const float*)ab.grad)[o + k]);
          ab.master[o + k] = pp; ab.m[o + k] = mm; ab.v[o + k] = vv;
          w[k] = pp;
        }
      }
      if (nat) {
        T* dst = nat + ((size_t)tap * A + a) * B8 + b;           // B8 % 8 == 0, b % 4 == 0: 8-byte (bf16) / 16-byte aligned
        if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2(pack2<T>(w[0], w[1]), pack2<T>(w[2], w[3]));
        else *(float4*)dst = make_float4(w[0], w[1], w[2], w[3]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 16 * i][tx * 4 + k] = enc(w[k]);
  }
  __syncthreads();
  if (tr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = b0 + ty + 16 * i, a = a0 + tx * 4;         // 4 consecutive a per thread
      if (b < B && a < A8) {
        T* dst = tr + ((size_t)tap * B + b) * A8 + a;
        const TileE w0 = tile[tx * 4][ty + 16 * i], w1 = tile[tx * 4 + 1][ty + 16 * i];
        const TileE w2 = tile[tx * 4 + 2][ty + 16 * i], w3 = tile[tx * 4 + 3][ty + 16 * i];
        if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2((uint32_t)w0 | ((uint32_t)w1 << 16), (uint32_t)w2 | ((uint32_t)w3 << 16));
        else *(float4*)dst = make_float4(w0, w1, w2, w3);
      }
    }
  }
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void dropout_kernel(uint8_t* mask, long long count, uint64_t seed, const int32_t* step,
                                                      uint32_t sid) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  long long w = i * 8;
  if (w >= count) return;
  uint64_t key = mix64(seed ^ ((uint64_t)(uint32_t)(*step) << 32) ^ sid);
  uint64_t h = mix64(key ^ (uint64_t)i);
  for (int e = 0; e < 8 && w + e < count; ++e) mask[w + e] = (uint8_t)((h >> (e * 8 + 7)) & 1);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* src, T* dst, int C, int pitch, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  st_f(dst + (i / C) * pitch + (i % C), src[i]);
}
// up to 4 (source, destination view) pairs of the same shape in one launch (blockIdx.y = pair): the input pipeline of
// a step packs the same two images into the generator's and the discriminator's typed input buffers
struct PackMulti { const float* src[4]; void* dst[4]; int pitch[4]; };
// Consecutive lanes on consecutive elements (4-byte loads coalesce; the 2-byte stores of a wave land in one 1-KB span of the
// channel-padded destination), 8 elements per thread a workgroup-stride apart, 32-bit index arithmetic when the tensor allows.
// (Round 4 tried 8 CONSECUTIVE elements per thread - vector loads, but every store instruction then touched 64 different lines:
// 18-24 -> 26 us.)
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackMulti pm, int C, long long total) {
  const int k = blockIdx.y;
  const float* __restrict__ src = pm.src[k];
  T* __restrict__ dst = (T*)pm.dst[k];
  const int pitch = pm.pitch[k];
  const long long base = (long long)blockIdx.x * 2048 + threadIdx.x;
  if (total < 0x7fffffffLL) {
    const unsigned tot = (unsigned)total, b0 = (unsigned)base;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned i = b0 + e * 256u;
      if (i < tot) {
        const unsigned pix = C == 1 ? i : i / (unsigned)C, c = C == 1 ? 0u : i - pix * (unsigned)C;
        st_f(dst + (size_t)pix * pitch + c, src[i]);
      }
    }
  } else {
    for (int e = 0; e < 8; ++e) {
      const long long i = base + e * 256;
      if (i < total) st_f(dst + (i / C) * pitch + (i % C), src[i]);
    }
  }
}
struct DropMulti { uint8_t* mask[4]; long long count[4]; uint32_t sid[4]; };
// `draws` (optional): {number of launches so far, block ticket}: every block reads the count before it takes a ticket, the
// last ticket holder advances it - successive calls draw new masks even while *step stands still (validation passes).
__global__ __launch_bounds__(256) void dropout_multi_kernel(const DropMulti dm, uint64_t seed, const int32_t* step, int32_t* draws) {
  // 8 hash words (64 mask bytes) per thread: the launch ends in one ticket per workgroup on ONE address (~15 ns each, serialised) -
  // 768 workgroups of one word per thread spent 11 us there for 0.7 MB of masks
  const int k = blockIdx.y;
  const uint32_t draw = draws ? (uint32_t)draws[0] : 0u;
  uint64_t key = mix64(seed ^ ((uint64_t)(uint32_t)(*step) << 32) ^ dm.sid[k]);
  if (draw) key = mix64(key + draw);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8 + u;      // hash word index: 8 mask bytes
    const long long w = i * 8;
    if (w < dm.count[k]) {
      const uint64_t h = mix64(key ^ (uint64_t)i);
      if (w + 8 <= dm.count[k] && (((uintptr_t)(dm.mask[k] + w)) & 7) == 0)      // one 8-byte store: bit 7 of every byte of h -> 0/1 bytes
        *(uint64_t*)(dm.mask[k] + w) = (h >> 7) & 0x0101010101010101ull;
      else
        for (int e = 0; e < 8 && w + e < dm.count[k]; ++e) dm.mask[k][w + e] = (uint8_t)((h >> (e * 8 + 7)) & 1);
    }
  }
  if (draws) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = atomicAdd(&draws[1], 1);
      if (t == (int)(gridDim.x * gridDim.y) - 1) { draws[1] = 0; draws[0] = (int32_t)(draw + 1u); }
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void unpack_kernel(const T* src, float* dst, int C, int pitch, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  dst[i] = ld_f(src + (i / C) * pitch + (i % C));
}

template <typename T>
__global__ __launch_bounds__(256) void copy_view_kernel(const T* src, int spitch, T* dst, int dpitch, int C, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  dst[(i / C) * dpitch + (i % C)] = src[(i / C) * spitch + (i % C)];
}

// run f with a null pointer of the storage type as its tag
template <typename F> static inline int with_dtype(int dtype, F&& f) {
  if (dtype == GAN_F32) return f((float*)nullptr);
  if (dtype == GAN_F16) return f((f16_t*)nullptr);
  if (dtype == GAN_BF16) return f((bf16_t*)nullptr);
  return GAN_E_ARG;
}
#define GAN_TAG_T(tag) typename std::remove_pointer<decltype(tag)>::type

// out[k] = a[k] + b[k] + c[k] for n <= 64 loss scalars (CycleGAN's total generator losses, cycle_gan.py:243-244)
__global__ void sum3_kernel(const float* a, const float* b, const float* c, float* out, int n) {
  const int k = threadIdx.x;
  if (k < n) out[k] = a[k] + b[k] + c[k];
}

// gradient wire format of the data-parallel exchange (gan_amd/ddp.py): fp32 <-> bf16, 8 elements per thread
__global__ __launch_bounds__(256) void grad_pack_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const float4 a = src[2 * i], b = src[2 * i + 1];
    dst[i] = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
}
__global__ __launch_bounds__(256) void grad_unpack_kernel(const uint4* __restrict__ src, float4* __restrict__ dst, long long n8, float scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8];
    unpack16<bf16_t>(src[i], v);
    dst[2 * i] = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
    dst[2 * i + 1] = make_float4(v[4] * scale, v[5] * scale, v[6] * scale, v[7] * scale);
  }
}

extern "C" {

int gan_bce_logits(const float* x, int64_t count, float target, float loss_scale, int32_t loss_accumulate,
                   float* loss_out, float grad_scale, int32_t dtype, void* dx, int32_t dx_pitch, float* workspace,
                   const float* scale_state, gan_stream_t stream) {
  if (!x || !loss_out || !workspace || count <= 0) return GAN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((count + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  int rc = with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(bce_kernel<T>, dim3(blocks), dim3(256), 0, st, x, (long long)count, target, grad_scale, (T*)dx, dx_pitch,
                       workspace, scale_state);
    GAN_CHECK_LAUNCH();
    return 0;
  });
  if (rc) return rc;
  GAN_LAUNCH(l1_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, blocks, 1.0 / (double)count,
                     loss_scale, loss_accumulate, loss_out);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_patchgan_losses(const float* real_logits, const float* fake_logits, int64_t count, int32_t dtype, void* g_dfake,
                        void* d_dreal, void* d_dfake, int32_t pitch, float lambda, const float* l1, float* gen_total,
                        float* gan_loss, float* disc_loss, float* workspace, const float* scale_state, gan_stream_t stream) {
  if (!real_logits || !fake_logits || count <= 0 || !gan_loss || !disc_loss || !workspace) return GAN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((count + 255) / 256);
  if (blocks > 256) blocks = 256;
  int rc = with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(patchgan_bce_kernel<T>, dim3(blocks), dim3(256), 0, st, real_logits, fake_logits, (long long)count,
                       (T*)g_dfake, (T*)d_dreal, (T*)d_dfake, pitch, workspace, scale_state);
    GAN_CHECK_LAUNCH();
    return 0;
  });
  if (rc) return rc;
  GAN_LAUNCH(patchgan_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, blocks, 1.0 / (double)count, lambda,
                     l1, gen_total, gan_loss, disc_loss);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_l1(int32_t dtype, const GanTensor* a, const GanTensor* b, float loss_scale, int32_t loss_accumulate,
           float* loss_out, float grad_scale, const GanTensor* da, float* workspace, const float* scale_state,
           gan_stream_t stream) {
  if (!a || !b || !a->ptr || !b->ptr || !loss_out || !workspace) return GAN_E_ARG;
  if (a->n != b->n || a->h != b->h || a->w != b->w || a->c != b->c) return GAN_E_SHAPE;
  long long pixels = (long long)a->n * a->h * a->w;
  long long total = pixels * a->c;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  float gs = grad_scale / (float)total;
  hipStream_t st = (hipStream_t)stream;
  int rc = with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(l1_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)a->ptr, a->pitch, (const T*)b->ptr, b->pitch, a->c,
                       pixels, gs, da ? (T*)da->ptr : nullptr, da ? da->pitch : 0, workspace, scale_state);
    GAN_CHECK_LAUNCH();
    return 0;
  });
  if (rc) return rc;
  GAN_LAUNCH(l1_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, blocks, 1.0 / (double)total,
                     loss_scale, loss_accumulate, loss_out);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_adam_begin(int32_t* step, float* lr_t, float lr, float beta1, float beta2, const float* scale_state, gan_stream_t stream) {
  if (!step || !lr_t) return GAN_E_ARG;
  GAN_LAUNCH(adam_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, lr_t, lr, beta1, beta2, scale_state);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_adam_tf(float* param, float* m, float* v, const void* grad, int64_t count, const float* lr_t, float beta1,
                float beta2, float eps, float grad_scale, const float* scale_state, int32_t grad_bf16, gan_stream_t stream) {
  if (!param || !m || !v || !grad || !lr_t || count <= 0 || count % 4) return GAN_E_ARG;
  if (((uintptr_t)grad & (grad_bf16 ? 7 : 15))) return GAN_E_ARG;
  long long nvec = count / 4;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (grad_bf16)
    GAN_LAUNCH(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param, (float4*)m,
                       (float4*)v, grad, nvec, lr_t, 1.f - beta1, 1.f - beta2, eps, grad_scale, scale_state);
  else
    GAN_LAUNCH(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param, (float4*)m,
                       (float4*)v, grad, nvec, lr_t, 1.f - beta1, 1.f - beta2, eps, grad_scale, scale_state);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_grads_check(const float* grad, int64_t count, float* scale_state, gan_stream_t stream) {
  if (!grad || !scale_state || count <= 0 || count % 4 || ((uintptr_t)grad & 15)) return GAN_E_ARG;
  const long long nvec = count / 4;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GAN_LAUNCH(grads_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)grad, nvec, scale_state);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_loss_scale_update(float* scale_state, int32_t growth_interval, float max_scale, gan_stream_t stream) {
  if (!scale_state || growth_interval <= 0 || !(max_scale >= 1.f)) return GAN_E_ARG;
  GAN_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale_state, growth_interval, max_scale);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_weights_prepare(const float* master, int32_t A, int32_t B, int32_t dtype, void* nk_native, void* nk_transposed,
                        gan_stream_t stream) {
  if (!master || A <= 0 || B <= 0 || (!nk_native && !nk_transposed)) return GAN_E_ARG;
  const int B8 = (B + 7) & ~7, A8 = (A + 7) & ~7;
  dim3 grid((unsigned)((B8 + 63) / 64), (unsigned)((A8 + 63) / 64), 16);
  hipStream_t st = (hipStream_t)stream;
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(wprep_kernel<T>, grid, dim3(256), 0, st, master, A, B, (T*)nk_native, (T*)nk_transposed);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_weights_prepare_multi(const void* entries_dev, int32_t n, int32_t total_tiles, int32_t dtype, gan_stream_t stream) {
  if (!entries_dev || n <= 0 || total_tiles <= 0) return GAN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(wprep_multi_kernel<T>, dim3((unsigned)total_tiles), dim3(256), 0, st, (const PrepEntry*)entries_dev, n);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_adam_prepare_multi(const void* entries_dev, int32_t n, int32_t total_tiles, int32_t dtype, float* master, float* m,
                           float* v, const void* grad, const float* lr_t, float beta1, float beta2, float eps,
                           float grad_scale, const float* scale_state, int32_t grad_bf16, gan_stream_t stream) {
  if (!entries_dev || n <= 0 || total_tiles <= 0 || !master || !m || !v || !grad || !lr_t) return GAN_E_ARG;
  if (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) & 15 || ((uintptr_t)grad & (grad_bf16 ? 7 : 15))) return GAN_E_ARG;
  const AdamBases ab = {master, m, v, grad};
  hipStream_t st = (hipStream_t)stream;
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    if (grad_bf16)
      GAN_LAUNCH((adam_prep_multi_kernel<T, true>), dim3((unsigned)total_tiles), dim3(256), 0, st, (const PrepEntry*)entries_dev, n, ab,
                         lr_t, 1.f - beta1, 1.f - beta2, eps, grad_scale, scale_state);
    else
      GAN_LAUNCH((adam_prep_multi_kernel<T, false>), dim3((unsigned)total_tiles), dim3(256), 0, st, (const PrepEntry*)entries_dev, n, ab,
                         lr_t, 1.f - beta1, 1.f - beta2, eps, grad_scale, scale_state);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_dropout_mask(uint8_t* mask, int64_t count, uint64_t seed, const int32_t* step, uint32_t stream_id, gan_stream_t stream) {
  if (!mask || count <= 0 || !step) return GAN_E_ARG;
  long long words = (count + 7) / 8;
  GAN_LAUNCH(dropout_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask,
                     (long long)count, seed, step, stream_id);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_pack(int32_t dtype, const float* src, const GanTensor* dst, gan_stream_t stream) {
  if (!src || !dst || !dst->ptr) return GAN_E_ARG;
  long long total = (long long)dst->n * dst->h * dst->w * dst->c;
  dim3 grid((unsigned)((total + 255) / 256));
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(pack_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, src, (T*)dst->ptr, dst->c, dst->pitch, total);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_pack_multi(int32_t dtype, int32_t n, const float* const* srcs, const GanTensor* dsts, gan_stream_t stream) {
  if (!srcs || !dsts || n <= 0 || n > 4) return GAN_E_ARG;
  if (!gan_dtype_ok(dtype)) return GAN_E_ARG;
  PackMulti pm;
  for (int k = 0; k < n; ++k) {
    if (!srcs[k] || !dsts[k].ptr) return GAN_E_ARG;
    if (dsts[k].n != dsts[0].n || dsts[k].h != dsts[0].h || dsts[k].w != dsts[0].w || dsts[k].c != dsts[0].c) return GAN_E_SHAPE;
    pm.src[k] = srcs[k]; pm.dst[k] = dsts[k].ptr; pm.pitch[k] = dsts[k].pitch;
  }
  const long long total = (long long)dsts[0].n * dsts[0].h * dsts[0].w * dsts[0].c;
  if (total <= 0) return GAN_E_SHAPE;
  dim3 grid((unsigned)((total + 2047) / 2048), (unsigned)n);
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(pack_multi_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, pm, dsts[0].c, total);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_dropout_mask_multi(int32_t n, uint8_t* const* masks, const int64_t* counts, uint64_t seed, const int32_t* step,
                           const uint32_t* stream_ids, int32_t* draws, gan_stream_t stream) {
  if (!masks || !counts || !stream_ids || !step || n <= 0 || n > 4) return GAN_E_ARG;
  DropMulti dm;
  long long maxw = 0;
  for (int k = 0; k < n; ++k) {
    if (!masks[k] || counts[k] <= 0) return GAN_E_ARG;
    dm.mask[k] = masks[k]; dm.count[k] = counts[k]; dm.sid[k] = stream_ids[k];
    const long long w = (counts[k] + 7) / 8;
    if (w > maxw) maxw = w;
  }
  GAN_LAUNCH(dropout_multi_kernel, dim3((unsigned)((maxw + 2047) / 2048), (unsigned)n), dim3(256), 0, (hipStream_t)stream, dm,
                     seed, step, draws);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_unpack(int32_t dtype, const GanTensor* src, float* dst, gan_stream_t stream) {
  if (!src || !dst || !src->ptr) return GAN_E_ARG;
  long long total = (long long)src->n * src->h * src->w * src->c;
  dim3 grid((unsigned)((total + 255) / 256));
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(unpack_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src->ptr, dst, src->c, src->pitch, total);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_copy_view(int32_t dtype, const GanTensor* src, const GanTensor* dst, gan_stream_t stream) {
  if (!src || !dst || !src->ptr || !dst->ptr) return GAN_E_ARG;
  if (src->n != dst->n || src->h != dst->h || src->w != dst->w || src->c != dst->c) return GAN_E_SHAPE;
  long long total = (long long)src->n * src->h * src->w * src->c;
  dim3 grid((unsigned)((total + 255) / 256));
  return with_dtype(dtype, [&](auto* tag) {
    typedef GAN_TAG_T(tag) T;
    GAN_LAUNCH(copy_view_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src->ptr, src->pitch, (T*)dst->ptr,
                       dst->pitch, src->c, total);
    GAN_CHECK_LAUNCH();
    return 0;
  });
}

int gan_sum3(const float* a, const float* b, const float* c, float* out, int32_t n, gan_stream_t stream) {
  if (!a || !b || !c || !out || n <= 0 || n > 64) return GAN_E_ARG;
  GAN_LAUNCH(sum3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, out, n);
  GAN_CHECK_LAUNCH();
  return 0;
}
int gan_grad_pack(const float* src, void* dst_bf16, int64_t count, gan_stream_t stream) {
  if (!src || !dst_bf16 || count <= 0 || count % 8 || (((uintptr_t)src | (uintptr_t)dst_bf16) & 15)) return GAN_E_ARG;
  const long long n8 = count / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  GAN_LAUNCH(grad_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (uint4*)dst_bf16, n8);
  GAN_CHECK_LAUNCH();
  return 0;
}
int gan_grad_unpack(const void* src_bf16, float* dst, int64_t count, float scale, gan_stream_t stream) {
  if (!src_bf16 || !dst || count <= 0 || count % 8 || (((uintptr_t)src_bf16 | (uintptr_t)dst) & 15)) return GAN_E_ARG;
  const long long n8 = count / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  GAN_LAUNCH(grad_unpack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_bf16, (float4*)dst, n8, scale);
  GAN_CHECK_LAUNCH();
  return 0;
}
const char* gan_version(void) { return "gan_amd 0.2 (gfx950)"; }
}

// ---- diagnostic launch log: names of the recorded launches (include/gan_amd.h) ---------------------------------------------
#include <cstring>
#include <string>
#include <vector>
extern "C" size_t gan_launch_log(char* buf, size_t cap) {
  std::vector<const void*> ptrs(gan_launch_log_ptrs(nullptr, 0));
  gan_launch_log_ptrs(ptrs.data(), ptrs.size());
  std::string all;
  for (const void* f : ptrs) {
    const char* nm = hipKernelNameRefByPtr(f, nullptr);
    all += nm ? nm : "?";
    all += '\n';
  }
  if (buf && cap) {
    const size_t n = all.size() < cap - 1 ? all.size() : cap - 1;
    std::memcpy(buf, all.data(), n);
    buf[n] = 0;
  }
  return all.size() + 1;
}
