// Streaming convolution kernels for the layers that have <= 8 channels on one side (16-bit paths: bf16 and fp16).
//
// The U-Net's first/last layers and the PatchGAN's first/last layers (base_gan.py:141-166, :176-204) move
// 50-100 MB of activations for a few GFLOP: they are HBM-bound, and the LDS-tiled implicit GEMM spends its
// time staging each thick pixel 16 times.  Two formulations read/write the thick tensor exactly once:
//
//  * thin-K (8-channel input -> thick output; Conv2D of the image, dgrad of the tanh head, dgrad of the
//    logits layer): the 4x4x8 im2col patch of a pixel is gathered straight into MFMA operand registers
//    (one 16-byte load per tap), the weights live in registers for the whole kernel, and the accumulators
//    are stored as whole 16-byte channel vectors.  No LDS at all.
//  * thin-N (thick input -> <= 6 output channels; tanh head, logits layer, dgrad into the image): the
//    convolution is split into Z[pixel][c][tap] = x[pixel,:] . W[tap][c][:]  (a plain streaming GEMM, x read
//    once, 16 taps = one MFMA tile) followed by a col2im gather of Z (a few MB, cache resident) with bias
//    and activation fused.
//
// Both keep the tap / padding / parity conventions of conv_gemm.hip (GemmParams), so they are drop-in
// replacements selected by thin_family(); the fp32 parity path always uses the tiled kernel.
#include "common.h"
#include "conv_params.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short s16x8;
#ifndef THIN_ZS
#define THIN_ZS 20          // floats per Z record of conv_thin_n_fused_kernel (16 taps + padding against LDS bank conflicts)
#endif


// ------------------------------------------------------------------------------------------------
// thin-K: Y[pixel, co] = act(bias + sum_{tap, c<8} X[src(pixel, tap), c] * W[tap][co][c])
// MFMA roles: A = weights (16 output channels x K), B = im2col patch (K x 16 pixels), K = 16 taps x 8.
// A wave owns 64 output channels (blockIdx.y) and walks 16-pixel tiles.  The 16 A rows of an MFMA pair are
// mapped to channels so that a lane ends up with 8 consecutive channels of its pixel: one 16-byte store.
struct ThinKParams {
  const void* x; const void* w; void* y; const float* bias;
  int Hs, Ws, xpitch, Hg, Wg, M, S, dy0, dstep, Wrows, ypitch, act, tiles;
  float slope;
  FastDiv divWg, divHg;
};

// ACT: the activation at compile time (GAN_ACT_NONE / GAN_ACT_LRELU: what the networks use on these layers), or -1 = p.act at run time -
// the run-time form compiled to three scalar branches per output value (165 branches in the kernel, 48 taken-or-not per 16-pixel tile)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void conv_thin_k_kernel(const ThinKParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int cbase = blockIdx.y * 64;
  uint4 wf[4][4];          // [channel tile][k step = tap row]
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int ch = cbase + (ct >> 1) * 32 + (n >> 2) * 8 + (ct & 1) * 4 + (n & 3);
#pragma unroll
    for (int s = 0; s < 4; ++s) wf[ct][s] = *(const uint4*)((const T*)p.w + ((size_t)(s * 4 + q) * p.Wrows + ch) * 8);
  }
  float bv[2][8];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[pr][i] = p.bias ? p.bias[cbase + pr * 32 + q * 8 + i] : 0.f;

  for (int tile = blockIdx.x * 4 + wave; tile < p.tiles; tile += gridDim.x * 4) {
    const int m = tile * 16 + n;
    const unsigned t = fdiv((unsigned)m, p.divWg);
    const int gx = m - (int)t * p.Wg;
    const unsigned img = fdiv(t, p.divHg);
    const int gy = (int)t - (int)img * p.Hg;
    const bool rowok = m < p.M;
    const int sx = gx * p.S + p.dy0 + q * p.dstep;       // dx0 == dy0 for every supported op
    uint4 xf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int sy = gy * p.S + p.dy0 + s * p.dstep;
      xf[s] = make_uint4(0, 0, 0, 0);
      if (rowok && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
        xf[s] = *(const uint4*)((const T*)p.x + ((size_t)(img * p.Hs + sy) * p.Ws + sx) * p.xpitch);
    }
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[ct] = mma16<T>(wf[ct][s], xf[s], acc[ct]);
    }
    if (rowok) {
      T* yp = (T*)p.y + ((size_t)(img * p.Hg + gy) * p.Wg + gx) * p.ypitch + cbase + q * 8;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (ACT >= 0) {
            v[i] = act_c<ACT>(acc[2 * pr][i] + bv[pr][i], p.slope);
            v[4 + i] = act_c<ACT>(acc[2 * pr + 1][i] + bv[pr][4 + i], p.slope);
          } else {
            v[i] = apply_act(acc[2 * pr][i] + bv[pr][i], p.act, p.slope);
            v[4 + i] = apply_act(acc[2 * pr + 1][i] + bv[pr][4 + i], p.act, p.slope);
          }
        }
        *(uint4*)(yp + pr * 32) = pack16<T>(v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// thin-N step 1: Z[pixel][c][tap] = sum_ci X[pixel, ci] * W[tap][c][ci]   (fp32 Z, taps = the MFMA's 16 rows)
// Weight fragments sit in LDS in operand order (CO * KS KiB); x is read once with 16-byte loads.
struct ThinNParams {
  const void* x; const void* w; float* z;
  int Mx, xpitch, Cin, Wrows, CO, tiles;
};

template <typename T, int KS>      // Cin = 32 * KS
__global__ __launch_bounds__(256) void conv_thin_n_kernel(const ThinNParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* wl = (uint4*)smem;                                // [c][s][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  for (int e = threadIdx.x; e < p.CO * KS * 64; e += 256) {
    const int l = e & 63, s = (e >> 6) % KS, c = (e >> 6) / KS;
    wl[e] = *(const uint4*)((const T*)p.w + ((size_t)(l & 15) * p.Wrows + c) * p.Cin + s * 32 + (l >> 4) * 8);
  }
  __syncthreads();
  for (int tile = blockIdx.x * 4 + wave; tile < p.tiles; tile += gridDim.x * 4) {
    const int pix = tile * 16 + n;
    const bool ok = pix < p.Mx;
    uint4 xf[KS];
    const T* xp = (const T*)p.x + (size_t)pix * p.xpitch + q * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = ok ? *(const uint4*)(xp + s * 32) : make_uint4(0, 0, 0, 0);
    for (int c = 0; c < p.CO; ++c) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mma16<T>(wl[(c * KS + s) * 64 + lane], xf[s], acc);
      if (ok) *(f32x4*)(p.z + ((size_t)pix * p.CO + c) * 16 + q * 4) = acc;
    }
  }
}

// thin-N step 2: Y[out pixel, c] = act(bias[c] + sum over the valid taps of Z[src pixel][c][tap])
struct Col2imParams {
  const float* z; void* y; const float* bias;
  int Hs, Ws, Ho, Wo, CO, ypitch, out_f32, act, parity, S, dy0, dstep, f16;
  float slope;
  long long total;
  FastDiv divWo, divHo;
};

__global__ __launch_bounds__(256) void conv_thin_col2im_kernel(const Col2imParams p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const unsigned t = fdiv((unsigned)idx, p.divWo);
  const int X = (int)idx - (int)t * p.Wo;
  const unsigned img = fdiv(t, p.divHo);
  const int Y = (int)t - (int)img * p.Ho;
  int iy[4], ky[4], ix[4], kx[4], ny, nx;
  if (p.parity) {          // y[2g+py] = sum_ty x[g + py - ty] * w[1 - py + 2 ty]   (conv_gemm.hip parity taps)
    ny = nx = 2;
    const int py = Y & 1, gy = Y >> 1, px = X & 1, gx = X >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      iy[a] = gy + py - a; ky[a] = 1 - py + 2 * a;
      ix[a] = gx + px - a; kx[a] = 1 - px + 2 * a;
    }
  } else {                 // y[g] = sum_t x[g*S + dy0 + t*dstep] * w[t]
    ny = nx = 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      iy[a] = Y * p.S + p.dy0 + a * p.dstep; ky[a] = a;
      ix[a] = X * p.S + p.dy0 + a * p.dstep; kx[a] = a;
    }
  }
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < ny; ++a) {
    if ((unsigned)iy[a] >= (unsigned)p.Hs) continue;
    for (int b = 0; b < nx; ++b) {
      if ((unsigned)ix[b] >= (unsigned)p.Ws) continue;
      const float* zp = p.z + ((size_t)(img * p.Hs + iy[a]) * p.Ws + ix[b]) * p.CO * 16 + ky[a] * 4 + kx[b];
#pragma unroll
      for (int c = 0; c < 6; ++c)
        if (c < p.CO) acc[c] += zp[c * 16];
    }
  }
  const size_t o = ((size_t)(img * p.Ho + Y) * p.Wo + X) * p.ypitch;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    if (c >= p.CO) break;
    float v = acc[c] + (p.bias ? p.bias[c] : 0.f);
    v = apply_act(v, p.act, p.slope);
    if (p.out_f32) ((float*)p.y)[o + c] = v;
    else if (p.f16) ((f16_t*)p.y)[o + c] = (f16_t)v;
    else ((bf16_t*)p.y)[o + c] = (bf16_t)v;
  }
}

// thin-N in ONE kernel for <= 2 output channels (the tanh head, the logits layer and the dgrad into the image at C = 1): a block
// owns a TH x TW tile of the GEMM grid, computes Z for the tile plus its halo into LDS (same MFMA formulation as step 1 above)
// and gathers the outputs from there - Z never goes through memory (33.5 MB written + read per launch before) and one launch
// instead of two.  The halo pixels are computed twice (tile 16 x 32: 1.2x the input reads, mostly L2 hits).
//   parity (stride-2 transposed conv / conv dgrad): halo 1 pixel each side; a thread produces the 2 x 2 output quad of a grid pixel
//   stride 1, pad 1 (logits layer): halo 1 before, 2 after; a thread produces one output pixel
struct ThinNFusedParams {
  const void* x; const void* w; void* y; const float* bias;
  int Hs, Ws, xpitch, Cin, Wrows, CO;
  int Ho, Wo, ypitch, out_f32, act, f16;
  float slope;
  int tilesY, tilesX;
};

template <typename T, int KS, bool PARITY, int TH, int TW>
__global__ __launch_bounds__(256) void conv_thin_n_fused_kernel(const ThinNFusedParams p) {
  constexpr int HB = 1, HA = PARITY ? 1 : 2;                 // halo before / after
  constexpr int SH = TH + HB + HA, SW = TW + HB + HA, NPX = SH * SW, NT = (NPX + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* wl = (uint4*)smem;                                  // [c][s][lane] weight fragments
  // Z records: [c][pixel][ZS floats], 16 taps + 4 floats of padding - the gather below reads one tap of 64 consecutive pixels per
  // instruction: at a pixel stride of 16 floats that was a 16-way bank conflict (80 % of the kernel's LDS cycles in the SQ counters),
  // at 20 floats it is 4-way and the 16-byte stores stay aligned
  constexpr int ZS = THIN_ZS;
  float* zl = (float*)(smem + (size_t)p.CO * KS * 1024);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % p.tilesX; b /= p.tilesX;
  const int ty = b % p.tilesY; const int img = b / p.tilesY;
  const int y0 = ty * TH - HB, x0 = tx * TW - HB;            // source origin of the halo region
  for (int e = threadIdx.x; e < p.CO * KS * 64; e += 256) {
    const int l = e & 63, s = (e >> 6) % KS, c = (e >> 6) / KS;
    wl[e] = *(const uint4*)((const T*)p.w + ((size_t)(l & 15) * p.Wrows + c) * p.Cin + s * 32 + (l >> 4) * 8);
  }
  __syncthreads();
  for (int tile = wave; tile < NT; tile += 4) {
    const int e = tile * 16 + n;
    const int hy = e / SW, hx = e - hy * SW;
    const int sy = y0 + hy, sx = x0 + hx;
    const bool ok = e < NPX && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
    uint4 xf[KS];
    const T* xp = (const T*)p.x + ((size_t)(img * p.Hs + sy) * p.Ws + sx) * p.xpitch + q * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = ok ? *(const uint4*)(xp + s * 32) : make_uint4(0, 0, 0, 0);
    for (int c = 0; c < p.CO; ++c) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mma16<T>(wl[(c * KS + s) * 64 + lane], xf[s], acc);
      if (e < NPX) *(f32x4*)(zl + ((size_t)c * NPX + e) * ZS + q * 4) = acc;      // (pixels outside the map: x = 0, so Z = 0)
    }
  }
  __syncthreads();
  auto store = [&](int Y, int X, int c, float v) {
    v += p.bias ? p.bias[c] : 0.f;
    v = apply_act(v, p.act, p.slope);
    const size_t o = ((size_t)(img * p.Ho + Y) * p.Wo + X) * p.ypitch + c;
    if (p.out_f32) ((float*)p.y)[o] = v;
    else if (p.f16) ((f16_t*)p.y)[o] = (f16_t)v;
    else ((bf16_t*)p.y)[o] = (bf16_t)v;
  };
  for (int t = threadIdx.x; t < TH * TW; t += 256) {
    const int ly = t / TW, lx = t - ly * TW;
    const int gy = ty * TH + ly, gx = tx * TW + lx;          // grid position (parity: source pixel; stride 1: output pixel)
    if (PARITY) {
      if (gy >= p.Hs || gx >= p.Ws) continue;
      for (int c = 0; c < p.CO; ++c) {
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            float v = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb)     // y[2g+py] = sum_a x[g + py - a] * w[1 - py + 2a]
                v += zl[((size_t)c * NPX + (ly + HB + py - a) * SW + (lx + HB + px - bb)) * ZS + (1 - py + 2 * a) * 4 + (1 - px + 2 * bb)];
            store(2 * gy + py, 2 * gx + px, c, v);
          }
      }
    } else {
      if (gy >= p.Ho || gx >= p.Wo) continue;
      for (int c = 0; c < p.CO; ++c) {
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int bb = 0; bb < 4; ++bb)         // y[g] = sum_a x[g - 1 + a] * w[a]
            v += zl[((size_t)c * NPX + (ly + a) * SW + (lx + bb)) * ZS + a * 4 + bb];
        store(gy, gx, c, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
int thin_family(const GanConvDesc* d, int op, const GemmParams& p) {
  const int on = gan_opt("conv.thin"), off = (on & 1) ? (~on & 6) : 1;      // off: 1 = none, bit 1 = no thin-N, bit 2 = no thin-K
  if (off == 1 || d->dtype == GAN_F32) return 0;
  const GanTensor &x = d->x, &y = d->y;
  // thin-N: few output channels.  Z needs one MFMA tile per output channel; weights CO*KS KiB of LDS.
  if (!(off & 2) && y.c <= 6 && (x.c == 64 || x.c == 128 || x.c == 512) && (x.c / 32) * y.c <= 48 &&
      x.pitch % 8 == 0 && ((uintptr_t)x.ptr & 15) == 0 && (p.parity || (p.dstep == 1 && p.dy0 == -1)))
    return 1;
  // thin-K: one 16-byte vector per input pixel, output channels in 64-wide groups, 16-byte stores
  if (!(off & 4) && !p.parity && x.c == 8 && x.pitch % 8 == 0 && y.c % 64 == 0 && !d->y_f32 && p.vec_store &&
      d->w_rows >= y.c && p.dx0 == p.dy0)
    return 2;
  return 0;
}

// thin-N in one kernel (conv_thin_n_fused_kernel): <= 2 output channels, parity form or stride 1 with pad 1
static bool thin_n_fused_ok(const GanConvDesc* d, const GemmParams& p) {
  return gan_opt("conv.thin_fused") && d->y.c <= 2 && (p.parity || (p.S == 1 && p.dstep == 1 && p.dy0 == -1 && p.dx0 == -1));
}

size_t thin_workspace_bytes(int family, const GanConvDesc* d, const GemmParams& p) {
  if (family != 1 || thin_n_fused_ok(d, p)) return 0;
  return (size_t)d->x.n * d->x.h * d->x.w * d->y.c * 16 * sizeof(float);
}

int thin_launch(int family, const GanConvDesc* d, const GemmParams& p, hipStream_t st) {
  const GanTensor &x = d->x, &y = d->y;
  if (family == 2) {
    ThinKParams k;
    k.x = p.x; k.w = p.w; k.y = p.y; k.bias = p.bias;
    k.Hs = p.Hs; k.Ws = p.Ws; k.xpitch = p.xpitch; k.Hg = p.Hg; k.Wg = p.Wg; k.M = p.M; k.S = p.S;
    k.dy0 = p.dy0; k.dstep = p.dstep; k.Wrows = p.Wrows; k.ypitch = p.ypitch; k.act = p.act; k.slope = p.slope;
    k.tiles = (p.M + 15) / 16;
    k.divWg = p.divWg; k.divHg = p.divHg;
    const int groups = y.c / 64;
    int gx = (k.tiles + 3) / 4;
    const int capw = gan_opt("conv.thin_k_blocks");            // workgroups of 4 waves per 64-channel group (a wave walks tiles/4/blocks tiles)
    const int cap = capw / groups > 256 ? capw / groups : 256;
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)groups);
#define THIN_K(TT)                                                                                              \
    do {                                                                                                        \
      if (k.act == GAN_ACT_NONE) GAN_LAUNCH((conv_thin_k_kernel<TT, GAN_ACT_NONE>), grid, dim3(256), 0, st, k);         \
      else if (k.act == GAN_ACT_LRELU) GAN_LAUNCH((conv_thin_k_kernel<TT, GAN_ACT_LRELU>), grid, dim3(256), 0, st, k);  \
      else GAN_LAUNCH((conv_thin_k_kernel<TT, -1>), grid, dim3(256), 0, st, k);                                 \
    } while (0)
    if (d->dtype == GAN_F16) THIN_K(f16_t); else THIN_K(bf16_t);
#undef THIN_K
    GAN_CHECK_LAUNCH();
    return 0;
  }
  if (family != 1) return GAN_E_ARG;
  if (thin_n_fused_ok(d, p)) {
    ThinNFusedParams f;
    f.x = p.x; f.w = p.w; f.y = p.y; f.bias = p.bias;
    f.Hs = p.Hs; f.Ws = p.Ws; f.xpitch = p.xpitch; f.Cin = x.c; f.Wrows = p.Wrows; f.CO = y.c;
    f.Ho = p.Ho; f.Wo = p.Wo; f.ypitch = p.ypitch; f.out_f32 = p.out_f32; f.act = p.act; f.f16 = d->dtype == GAN_F16; f.slope = p.slope;
    const int KS = x.c / 32;
    auto launch_f = [&](auto* tag, auto parc) -> int {
      typedef typename std::remove_pointer<decltype(tag)>::type T;
      constexpr bool PAR = decltype(parc)::value;
      constexpr int TH = PAR ? 16 : 10, TW = 32;
      constexpr int NPX = (TH + (PAR ? 2 : 3)) * (TW + (PAR ? 2 : 3));
      f.tilesY = ((PAR ? p.Hs : p.Ho) + TH - 1) / TH; f.tilesX = ((PAR ? p.Ws : p.Wo) + TW - 1) / TW;
      const size_t smem = (size_t)f.CO * KS * 1024 + (size_t)NPX * f.CO * THIN_ZS * sizeof(float);
      const dim3 grid((unsigned)(x.n * f.tilesY * f.tilesX));
#define THIN_F(KSV)                                                                                                                  \
      {                                                                                                                               \
        auto kern = conv_thin_n_fused_kernel<T, KSV, PAR, TH, TW>;                                                                    \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
        if (e != hipSuccess) return (int)e;                                                                                           \
        GAN_LAUNCH(kern, grid, dim3(256), smem, st, f);                                                                       \
      }
      if (KS == 2) THIN_F(2) else if (KS == 4) THIN_F(4) else THIN_F(16)
#undef THIN_F
      GAN_CHECK_LAUNCH();
      return 0;
    };
    if (d->dtype == GAN_F16) return p.parity ? launch_f((f16_t*)nullptr, std::true_type{}) : launch_f((f16_t*)nullptr, std::false_type{});
    return p.parity ? launch_f((bf16_t*)nullptr, std::true_type{}) : launch_f((bf16_t*)nullptr, std::false_type{});
  }
  const size_t zbytes = thin_workspace_bytes(1, d, p);
  if (!d->workspace || d->workspace_bytes < zbytes) return GAN_E_WORKSPACE;
  ThinNParams n;
  n.x = p.x; n.w = p.w; n.z = (float*)d->workspace;
  n.Mx = x.n * x.h * x.w; n.xpitch = p.xpitch; n.Cin = x.c; n.Wrows = p.Wrows; n.CO = y.c;
  n.tiles = (n.Mx + 15) / 16;
  const int KS = x.c / 32;
  int gx = (n.tiles + 3) / 4;
  if (gx > 2048) gx = 2048;
  const size_t smem = (size_t)n.CO * KS * 1024;
  auto launch_n = [&](auto* tag) {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    if (KS == 2) GAN_LAUNCH((conv_thin_n_kernel<T, 2>), dim3((unsigned)gx), dim3(256), smem, st, n);
    else if (KS == 4) GAN_LAUNCH((conv_thin_n_kernel<T, 4>), dim3((unsigned)gx), dim3(256), smem, st, n);
    else GAN_LAUNCH((conv_thin_n_kernel<T, 16>), dim3((unsigned)gx), dim3(256), smem, st, n);
  };
  if (d->dtype == GAN_F16) launch_n((f16_t*)nullptr); else launch_n((bf16_t*)nullptr);
  GAN_CHECK_LAUNCH();
  Col2imParams c;
  c.z = n.z; c.y = p.y; c.bias = p.bias;
  c.Hs = p.Hs; c.Ws = p.Ws; c.Ho = p.Ho; c.Wo = p.Wo; c.CO = y.c; c.ypitch = p.ypitch; c.out_f32 = p.out_f32;
  c.f16 = d->dtype == GAN_F16;
  c.act = p.act; c.parity = p.parity; c.S = p.S; c.dy0 = p.dy0; c.dstep = p.dstep; c.slope = p.slope;
  c.total = (long long)x.n * p.Ho * p.Wo;
  c.divWo = make_fastdiv((uint32_t)p.Wo); c.divHo = make_fastdiv((uint32_t)p.Ho);
  GAN_LAUNCH(conv_thin_col2im_kernel, dim3((unsigned)((c.total + 255) / 256)), dim3(256), 0, st, c);
  GAN_CHECK_LAUNCH();
  return 0;
}
