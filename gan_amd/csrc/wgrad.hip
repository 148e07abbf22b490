// Kernel-gradient (wgrad) of Conv2D / Conv2DTranspose k4 for gfx950:
//
//   dW[tap][ca][cb] = sum_m  BIG[src(m, tap), ca] * SMALL[m, cb]        m = (image, gy, gx) on the coarse grid
//   src(m, tap=(kh,kw)) = (gy*S + kh - 1, gx*S + kw - 1), zero outside the fine map.
//
// Conv2D: BIG = layer input, SMALL = dy  -> dW is HWIO.   Conv2DTranspose: BIG = dy, SMALL = layer input
// -> dW is (kh,kw,cout,cin).  Both are the Keras master layouts, so gradients land byte-for-byte where
// Adam reads them.
//
// The reduction index m is the NHWC *row* index, i.e. the MFMA K dimension is strided in memory for both
// operands.  Tiles are staged row-major [m][channel] in LDS (coalesced 16-B global loads, 16-B LDS stores)
// and the K-major fragments are produced by gfx950's transposing LDS read ds_read_b64_tr_b16 (bf16) or by
// plain ds_read_b32 (fp32, one element per lane per v_mfma_f32_16x16x4_f32).  "fold" mode handles 8-channel
// (zero-padded C=1..6) BIG tensors by folding the 16 taps into the tile's channel axis, so the SMALL
// tensor is read once instead of 16 times.  Large reductions split M across blockIdx.z into fp32 slabs.
#include "common.h"
#include <type_traits>

#ifdef GAN_DIAG   // diagnostic build only (tools/diag_build.sh): in-kernel stamps of the ping-pong kernel, as in conv_gemm.hip
static unsigned long long* g_wdiag = nullptr;
extern "C" void gan_wdiag_set(void* ptr) { g_wdiag = (unsigned long long*)ptr; }
#define WDIAG_STAMP(i) do { if (p.diag && tid == 0) p.diag[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WSEG_DECL unsigned long long seg_t = 0, seg_sum[6] = {0, 0, 0, 0, 0, 0}
#define WSEG_T0 do { if (p.diag) { __builtin_amdgcn_sched_barrier(0); seg_t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define WSEG_ADD(k) do { if (p.diag) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); seg_sum[k] += n_ - seg_t; seg_t = n_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define WSEG_STORE do { if (p.diag && tid == 0) for (int k_ = 0; k_ < 6; ++k_) p.diag[(size_t)blockIdx.x * 16 + 8 + k_] = seg_sum[k_]; } while (0)
#else
#define WDIAG_STAMP(i) do {} while (0)
#define WSEG_DECL
#define WSEG_T0 do {} while (0)
#define WSEG_ADD(k) do {} while (0)
#define WSEG_STORE do {} while (0)
#endif

struct WgradParams {
  const void* big; const void* small; float* dw; float* slab;
  int Hb, Wb, bpitch, Ca;
  int spitch, Cb;
  int S, M;
  FastDiv divW, divH;        // coarse-grid width / height
  int CaReal, CbReal;
  int tilesB;
  int splits, kchunks;
  int accumulate, fold;
  int swap, shift;  // swap: roles exchanged (thin SMALL tensor folded, thick BIG tensor streamed), taps flipped, rows shifted
  // GanAdamFuse (adam != 0: un-split launch, !fold, !swap, !accumulate, CaReal % 8 == CbReal % 8 == 0): the epilogue applies Adam
  int adam;
  // adam == 1 only: taps whose source position lies outside the map for EVERY grid position (a 2x2 -> 1x1 layer uses 4 of its 16
  // taps): their gradient is exactly zero, so with zero moments the step changes nothing - their blocks check m == v == 0 and leave
  // (bit-identical to the update); ahalves == 2: the grid's z index names which 64-row half of the tile a block updates (the GEMM of
  // such a layer is a few MFMAs; twice the blocks stream the optimiser state of the live taps)
  int dead_taps, ahalves;
  float* aw; float* am; float* av; void* anat; void* atr; const float* alr;
  float omb1, omb2, aeps;
  void* wire;                // GanWgradDesc.dw_wire honoured: the result goes out as bfloat16 into the exchange's wire buffer, dw untouched
#ifdef GAN_DIAG
  unsigned long long* diag;
#endif
};

template <typename T, int TA, int TB, int WAVES_A, int WAVES_B, bool TR>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int ES = sizeof(T);
  constexpr int BKM = 128 / ES;                 // rows (m) per K chunk: 64 bf16 / 32 fp32
  constexpr int RSA = TA * ES + 16, RSB = TB * ES + 16;   // LDS row strides in bytes
  constexpr int WTA = TA / WAVES_A, WTB = TB / WAVES_B, MT = WTA / 16, NT = WTB / 16;
  constexpr int VPA = TA / VEC, VPB = TB / VEC;  // vectors per row
  constexpr int AI = (BKM * VPA + 255) / 256, BI = (BKM * VPB + 255) / 256;
  static_assert(WAVES_A * WAVES_B == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                      // [2][BKM][RSA]
  unsigned char* Bs = smem + 2 * BKM * RSA;      // [2][BKM][RSB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WAVES_B, wb = wave % WAVES_B;
  const int r = lane & 15, q = lane >> 4;
  const int ta = blockIdx.x / p.tilesB, tb = blockIdx.x % p.tilesB;
  const int ca0 = ta * TA, cb0 = tb * TB;
  const int tap = blockIdx.y, split = blockIdx.z;
  const int kh = tap >> 2, kw = tap & 3;
  const T* bg = (const T*)p.big;
  const T* sg = (const T*)p.small;

  uint4 ra[AI], rb[BI];
  auto gload = [&](int kc) {
    const int mbase = kc * BKM;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int idx = tid + 256 * i;
      int row = idx / VPA, cvi = idx % VPA;
      uint4 v = make_uint4(0, 0, 0, 0);
      unsigned m = mbase + row;
      if (idx < BKM * VPA && m < (unsigned)p.M) {
        unsigned t = fdiv(m, p.divW);
        int gx = m - t * p.divW.d;
        unsigned img = fdiv(t, p.divH);
        int gy = t - img * p.divH.d;
        int akh = kh, akw = kw, c = ca0 + cvi * VEC;
        if (p.fold) {                       // channel axis = (tap, 8 padded channels)
          int ft = (cvi * VEC) >> 3;
          akh = ft >> 2; akw = ft & 3; c = (cvi * VEC) & 7;
        }
        int sy = gy * p.S + akh - 1, sx = gx * p.S + akw - 1;
        if ((unsigned)sy < (unsigned)p.Hb && (unsigned)sx < (unsigned)p.Wb && c < p.Ca)
          v = *(const uint4*)(bg + ((size_t)(img * p.Hb + sy) * p.Wb + sx) * (size_t)p.bpitch + c);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int idx = tid + 256 * i;
      int row = idx / VPB, cvi = idx % VPB;
      uint4 v = make_uint4(0, 0, 0, 0);
      unsigned m = mbase + row;
      int c = cb0 + cvi * VEC;
      if (idx < BKM * VPB && m < (unsigned)p.M && c < p.Cb)
        v = *(const uint4*)(sg + (size_t)m * p.spitch + c);
      rb[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int idx = tid + 256 * i;
      int row = idx / VPA, cvi = idx % VPA;
      if (idx < BKM * VPA) *(uint4*)(As + (buf * BKM + row) * RSA + cvi * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int idx = tid + 256 * i;
      int row = idx / VPB, cvi = idx % VPB;
      if (idx < BKM * VPB) *(uint4*)(Bs + (buf * BKM + row) * RSB + cvi * 16) = rb[i];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kc_begin = (int)((long long)p.kchunks * split / p.splits);
  const int kc_end = (int)((long long)p.kchunks * (split + 1) / p.splits);
  int buf = 0;
  if (kc_begin < kc_end) { gload(kc_begin); lstore(0); }
  __syncthreads();
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const bool more = kc + 1 < kc_end;
    if (more) gload(kc + 1);
    {
    const unsigned char* Ab = As + buf * BKM * RSA + (wa * WTA) * ES;
    const unsigned char* Bb = Bs + buf * BKM * RSB + (wb * WTB) * ES;
    if constexpr (sizeof(T) == 4) {
#pragma unroll 2
      for (int kk = 0; kk < BKM / 4; ++kk) {
        float af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const float*)(Ab + (kk * 4 + q) * RSA + (i * 16 + r) * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *(const float*)(Bb + (kk * 4 + q) * RSB + (j * 16 + r) * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKM / 32; ++ks) {
        bf16x8 af[MT], bf[NT];
        if constexpr (TR) {
          // 16-lane group q reads rows ks*32+8q+{0..3} then {4..7}; lane i of the group supplies the address
          // of row (i>>2), columns 4*(i&3).. and receives column i of the 4 rows.
          const int rr = ks * 32 + 8 * q + (r >> 2), cc = 4 * (r & 3);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const unsigned char* a0 = Ab + rr * RSA + (i * 16 + cc) * 2;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * RSA));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            af[i] = *(bf16x8*)&v;
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const unsigned char* b0 = Bb + rr * RSB + (j * 16 + cc) * 2;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0 + 4 * RSB));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            bf[j] = *(bf16x8*)&v;
          }
        } else {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              ((uint16_t*)&af[i])[e] = *(const uint16_t*)(Ab + (ks * 32 + 8 * q + e) * RSA + (i * 16 + r) * 2);
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              ((uint16_t*)&bf[j])[e] = *(const uint16_t*)(Bb + (ks * 32 + 8 * q + e) * RSB + (j * 16 + r) * 2);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = mma16<T>(*(const uint4*)&af[i], *(const uint4*)&bf[j], acc[i][j]);
      }
    }
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // D[row = ca][col = cb]; row = q*4 + e, col = r
  const size_t per_split = (size_t)16 * p.CaReal * p.CbReal;
  float* out = p.splits > 1 ? p.slab + (size_t)split * per_split : p.dw;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int a = ca0 + wa * WTA + i * 16 + q * 4 + e;
      int otap = tap, oc = a;
      if (p.fold) { otap = a >> 3; oc = a & 7; if (a >= 128) oc = p.CaReal; }
      if (oc < p.CaReal) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          int cb = cb0 + wb * WTB + j * 16 + r;
          if (cb < p.CbReal) {
            size_t o = ((size_t)otap * p.CaReal + oc) * p.CbReal + cb;
            if (p.splits == 1 && p.accumulate) out[o] += acc[i][j][e];
            else out[o] = acc[i][j][e];
          }
        }
      }
    }
}

// Sum of the split slabs: a block is EV float4 element-vectors x SG split-groups (EV * SG = 256, consecutive
// threads on consecutive vectors: coalesced 16-byte reads).  A thread adds its splits 4 at a time (independent
// loads), the SG partial sums meet in LDS and are added in a fixed order: deterministic.  SG is chosen on the
// host so that small tensors with hundreds of slabs still spread over >= 64K threads.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slab, float* dw, long long count4, int splits,
                                                           int accumulate, int log2sg, void* wire) {
  __shared__ f32x4 red[256];
  const int EV = 256 >> log2sg, SG = 1 << log2sg;
  const int ev = threadIdx.x & (EV - 1), sg = threadIdx.x >> (8 - log2sg);
  const long long e = (long long)blockIdx.x * EV + ev;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < count4) {
    const f32x4* src = (const f32x4*)slab + e;
    int k = sg;
    for (; k + 3 * SG < splits; k += 4 * SG) {
      const f32x4 a = src[(size_t)k * count4], b = src[(size_t)(k + SG) * count4];
      const f32x4 c = src[(size_t)(k + 2 * SG) * count4], d = src[(size_t)(k + 3 * SG) * count4];
      s += a; s += b; s += c; s += d;
    }
    for (; k < splits; k += SG) s += src[(size_t)k * count4];
  }
  if (SG > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (sg != 0) return;
    for (int j = 1; j < SG; ++j) s += red[ev + EV * j];
  }
  if (e < count4) {
    if (wire) { ((uint2*)wire)[e] = make_uint2(pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])); return; }      // bf16 wire format (gan_grad_pack's rounding)
    f32x4* o = (f32x4*)dw + e;
    *o = accumulate ? *o + s : s;
  }
}

// The same slab sum that ENDS in the optimiser step (GanAdamFuse on a split launch): a block owns a 64 x 64 tile of one tap of
// dW[tap][a][b], sums the splits of its elements in exactly wgrad_reduce_kernel's order (SG interleaved partial sums, each taken
// sequentially, added in order), applies TF-form Adam (gan_adam1: bit-identical to reduce -> fp32 gradient -> gan_adam_prepare_multi)
// and refreshes both typed NK copies - adam_prep_multi_kernel's access pattern: whole 256-byte row segments of master / m / v per
// 16 lanes, the transposed copy through LDS.  The fp32 gradient is never written.  A % 8 == 0, B % 8 == 0.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_adam_kernel(const float* __restrict__ slab, size_t per_split, int splits, int log2sg, int A, int B,
                                                                float* __restrict__ aw, float* __restrict__ am, float* __restrict__ av,
                                                                T* __restrict__ nat, T* __restrict__ tr, const float* __restrict__ lr_t,
                                                                float omb1, float omb2, float eps) {
  __shared__ float tile[64][65];
  const int tiles_b = (B + 63) / 64, tiles_a = (A + 63) / 64;
  int t = blockIdx.x;
  const int tb = t % tiles_b; t /= tiles_b;
  const int ta = t % tiles_a, tap = t / tiles_a;
  const int b0 = tb * 64, a0 = ta * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const float lr = *lr_t;
  const int SG = 1 << log2sg;
  const int b = b0 + tx * 4;
  // phase 1: every load of the thread's 4 rows (slab sums, master, m, v) before the first store: 4 rows' worth in flight
  f32x4 g[4];
  float4 pp[4], mm[4], vv[4];
  size_t o[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int a = a0 + ty + 16 * i;
    ok[i] = a < A && b < B;
    o[i] = ok[i] ? ((size_t)tap * A + a) * B + b : 0;
    pp[i] = *(const float4*)(aw + o[i]); mm[i] = *(const float4*)(am + o[i]); vv[i] = *(const float4*)(av + o[i]);
    g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (SG == 1) {                                                 // the common case: one sequential sum per element, 4 rows interleaved
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      f32x4 x[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[u][i] = *(const f32x4*)(slab + (size_t)(k + u) * per_split + o[i]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] += x[u][i];
    }
    for (; k < splits; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] += *(const f32x4*)(slab + (size_t)k * per_split + o[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* src = slab + o[i];
      for (int j = 0; j < SG; ++j) {                             // split group j of wgrad_reduce_kernel: k = j, j + SG, ...
        f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
        int k = j;
        for (; k + 3 * SG < splits; k += 4 * SG) {
          const f32x4 x0 = *(const f32x4*)(src + (size_t)k * per_split), x1 = *(const f32x4*)(src + (size_t)(k + SG) * per_split);
          const f32x4 x2 = *(const f32x4*)(src + (size_t)(k + 2 * SG) * per_split), x3 = *(const f32x4*)(src + (size_t)(k + 3 * SG) * per_split);
          sacc += x0; sacc += x1; sacc += x2; sacc += x3;
        }
        for (; k < splits; k += SG) sacc += *(const f32x4*)(src + (size_t)k * per_split);
        if (j == 0) g[i] = sacc; else g[i] += sacc;
      }
    }
  }
  // phase 2: the optimiser step, the native NK rows, the updated weights into LDS for the transposed copy
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok[i]) {
      gan_adam1(pp[i].x, mm[i].x, vv[i].x, g[i][0], 1.f, omb1, omb2, lr, eps);
      gan_adam1(pp[i].y, mm[i].y, vv[i].y, g[i][1], 1.f, omb1, omb2, lr, eps);
      gan_adam1(pp[i].z, mm[i].z, vv[i].z, g[i][2], 1.f, omb1, omb2, lr, eps);
      gan_adam1(pp[i].w, mm[i].w, vv[i].w, g[i][3], 1.f, omb1, omb2, lr, eps);
      *(float4*)(aw + o[i]) = pp[i]; *(float4*)(am + o[i]) = mm[i]; *(float4*)(av + o[i]) = vv[i];
      w[0] = pp[i].x; w[1] = pp[i].y; w[2] = pp[i].z; w[3] = pp[i].w;
      if (nat) *(uint2*)(nat + o[i]) = make_uint2(pack2<T>(w[0], w[1]), pack2<T>(w[2], w[3]));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 16 * i][tx * 4 + k] = w[k];
  }
  __syncthreads();
  if (tr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int bb = b0 + ty + 16 * i, a = a0 + tx * 4;          // 4 consecutive a per thread
      if (bb < B && a < A)
        *(uint2*)(tr + ((size_t)tap * B + bb) * A + a) = make_uint2(pack2<T>(tile[tx * 4][ty + 16 * i], tile[tx * 4 + 1][ty + 16 * i]),
                                                                    pack2<T>(tile[tx * 4 + 2][ty + 16 * i], tile[tx * 4 + 3][ty + 16 * i]));
    }
  }
}

static int wgrad_reduce_log2sg(long long count4, int splits) {
  int log2sg = 0;
  while (log2sg < 6 && (count4 << log2sg) < 65536 && (8 << log2sg) <= splits) ++log2sg;
  return log2sg;
}

struct WgradPlan { WgradParams p; int TA, TB; dim3 grid; size_t slab_bytes; bool pp; };

__device__ __forceinline__ uint2 wire_bf16x4(const f32x4& v) { return make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); }

// GanWgradDesc.dw_wire (data-parallel steps): the launch's last kernel - slab reduce, or the epilogue of an un-split LDS-DMA / ping-pong
// launch - writes the gradient as bfloat16 straight into the exchange's wire buffer (same element offsets as dw): no fp32 gradient,
// no gan_grad_pack pass.  `lds_dma`: the launch runs on the LDS-DMA kernels (the register-staged fallback has no such epilogue).
static void plan_wire(const GanWgradDesc* d, WgradParams& p, bool lds_dma) {
  if (!d->dw_wire || d->adam_fuse || d->accumulate || p.fold || p.swap || p.CbReal % 4 || ((uintptr_t)d->dw_wire & 7)) return;
  if (p.splits > 1 || lds_dma) p.wire = d->dw_wire;
}

// GanAdamFuse on a SPLIT launch: the slab reduce ends in the optimiser step (wgrad_reduce_adam_kernel), p.adam = 2
static void plan_reduce_adam(const GanWgradDesc* d, WgradParams& p) {
  const GanAdamFuse* af = d->adam_fuse;
  // (tensors under 2^20 parameters stay on the flat reduce + the caller's multi-tensor pass: 64 x 64 tiles of a small kernel are too few
  // blocks to stream at HBM rate - measured 4.2 TB/s over all split layers of the generator against 5.4 TB/s for the separate passes)
  const int min_params = gan_opt("wgrad.reduce_adam_min_params");
  if (!af || !gan_opt("wgrad.reduce_adam") || p.splits <= 1 || p.fold || p.swap || d->accumulate || d->dtype == GAN_F32 || p.CaReal % 8 || p.CbReal % 8 ||
      (long long)16 * p.CaReal * p.CbReal < (long long)min_params ||
      !af->master || !af->m || !af->v || !af->lr_t ||
      (((uintptr_t)af->master | (uintptr_t)af->m | (uintptr_t)af->v | (uintptr_t)af->nk_native | (uintptr_t)af->nk_transposed) & 15))
    return;
  p.adam = 2;
  p.aw = af->master; p.am = af->m; p.av = af->v; p.anat = af->nk_native; p.atr = af->nk_transposed; p.alr = af->lr_t;
  p.omb1 = 1.f - af->beta1; p.omb2 = 1.f - af->beta2; p.aeps = af->eps;
}

static int plan_wgrad(const GanWgradDesc* d, WgradPlan* pl, bool allow_swap = true) {
  if (!d || d->struct_size != sizeof(GanWgradDesc) || !d->big.ptr || !d->small.ptr || !d->dw) return GAN_E_ARG;
  if (!gan_dtype_ok(d->dtype)) return GAN_E_ARG;
  if (((uintptr_t)d->dw | (uintptr_t)d->workspace) & 15) return GAN_E_ARG;      // float4 slab reduction
  const GanTensor &b = d->big, &s = d->small;
  if (b.c % 8 || s.c % 8 || b.pitch % 8 || s.pitch % 8 || b.pitch < b.c || s.pitch < s.c) return GAN_E_SHAPE;
  if (d->big_c > b.c || d->small_c > s.c || d->big_c <= 0 || d->small_c <= 0 || b.n != s.n) return GAN_E_SHAPE;
  if (d->stride != 1 && d->stride != 2) return GAN_E_SHAPE;
  if (s.h != (b.h + 2 - 4) / d->stride + 1 || s.w != (b.w + 2 - 4) / d->stride + 1) return GAN_E_SHAPE;
  long long M = (long long)s.n * s.h * s.w;
  if (M <= 0 || M > 0x7fffffffLL) return GAN_E_SHAPE;
  WgradParams& p = pl->p;
  p.big = b.ptr; p.small = s.ptr; p.dw = d->dw; p.slab = (float*)d->workspace;
  p.Hb = b.h; p.Wb = b.w; p.bpitch = b.pitch; p.Ca = b.c; p.spitch = s.pitch; p.Cb = s.c;
  p.S = d->stride; p.M = (int)M; p.divW = make_fastdiv(s.w); p.divH = make_fastdiv(s.h);
  p.CaReal = d->big_c; p.CbReal = d->small_c; p.accumulate = d->accumulate;
  p.fold = (b.c == 8) ? 1 : 0;
  p.swap = 0; p.shift = 0; p.adam = 0; p.wire = nullptr; p.dead_taps = 0; p.ahalves = 1;
#ifdef GAN_DIAG
  p.diag = g_wdiag;
#endif
  int TA, TB, tilesA, taps;
  if (allow_swap && !p.fold && d->stride == 1 && s.c == 8 && b.c >= 64) {
    // Thin SMALL tensor (the logits layer's dy): iterate over the BIG grid instead and fold the 16 taps of the thin
    // tensor into the tile's channel axis, so the thick tensor is streamed once instead of once per tap.
    // dW[kh][kw] = sum_P big[P] * small[P + 1 - k]: with the folded tap index f = 3 - k that is the usual
    // gather P + f - 1, shifted by -1; the epilogue un-flips the taps and writes [tap][thick][thin].
    p.swap = 1; p.fold = 1; p.shift = -1;
    p.big = s.ptr; p.Hb = s.h; p.Wb = s.w; p.bpitch = s.pitch; p.Ca = s.c;
    p.small = b.ptr; p.spitch = b.pitch; p.Cb = b.c;
    M = (long long)b.n * b.h * b.w;
    p.M = (int)M; p.divW = make_fastdiv(b.w); p.divH = make_fastdiv(b.h);
    p.CaReal = d->small_c; p.CbReal = d->big_c;
    TA = 128; tilesA = 1; taps = 1; TB = b.c >= 128 ? 128 : 64;
  } else if (p.fold) { TA = 128; tilesA = 1; taps = 1; TB = s.c >= 128 ? 128 : (s.c >= 64 ? 64 : 16); if (TB == 16) return GAN_E_SHAPE; }
  else {
    taps = 16;
    TB = s.c >= 128 ? 128 : (s.c >= 64 ? 64 : 16);
    TA = b.c >= 128 ? 128 : (b.c >= 64 ? 64 : 16);
    if (gan_opt("wgrad.tile256") && b.c >= 256 && s.c >= 128 && d->dtype != GAN_F32) TA = 256;
    if (TB == 16) TA = TA == 16 ? 64 : TA;   // supported: (128|64, 16)
    if (TA == 64 && TB == 16) {}
    if (TA == 16 && TB == 16) return GAN_E_SHAPE;
    tilesA = (b.c + TA - 1) / TA;
  }
  pl->pp = false;
  {
    // big reductions on 16-bit storage: the 256 x 256 ping-pong kernel (rows = (tap, channel) pairs in units of 64)
    const int use_pp = gan_opt("wgrad.pingpong");
    // reduction rows per split at least: 1024 alone on the chip; 2048 when the launch shares it with other lanes (~128 blocks with
    // longer K loops and half the slab traffic: slower alone, +1.3 % in the captured Pix2Pix step; option wgrad.pingpong_min_rows overrides)
    const int min_rows_env = gan_opt("wgrad.pingpong_min_rows");
    const int min_rows = min_rows_env > 0 ? min_rows_env : (d->concurrent ? 2048 : 1024);
    const int pp128 = gan_opt("wgrad.pingpong_128"), mingf = gan_opt("wgrad.pingpong_min_gflop");
    // (option wgrad.pingpong_128: a 128-channel SMALL tensor on the 256-column tile - half the columns are computed and dropped)
    if (use_pp && allow_swap && !p.fold && !p.swap && d->dtype != GAN_F32 && (b.c == 64 || b.c == 128 || b.c % 256 == 0) &&
        (s.c % 256 == 0 || (pp128 && s.c == 128)) && M >= 2 * min_rows &&
        2.0 * (double)M * 16.0 * b.c * s.c >= 1.0e9 * mingf) {      // every block writes a 256 KB fp32 slab tile (64 MB per launch with the
                                                            // reduce pass): pays from ~30 GFLOP up (measured against the 128x128 kernel)
      const long long tiles = (long long)(16 * b.c / 256) * ((s.c + 255) / 256);
      long long sp = (256 + tiles - 1) / tiles;
      if (sp > M / min_rows) sp = M / min_rows;
      if (sp < 1) sp = 1;
      if (sp > 1024) sp = 1024;
      p.tilesB = (s.c + 255) / 256;
      p.kchunks = (int)((M + 63) / 64);
      if (sp >= 8) sp &= ~7LL;
      p.splits = (int)sp;
      pl->pp = true; pl->TA = 256; pl->TB = 256;
      if (sp >= 8) sp &= ~7LL;                     // whole groups of 8 splits: one per XCD (wgrad_pp_kernel's block mapping)
      pl->grid = dim3((unsigned)(tiles * sp), 1, 1);
      pl->slab_bytes = sp > 1 ? (size_t)sp * 16 * p.CaReal * p.CbReal * sizeof(float) : 0;
      plan_reduce_adam(d, p);
      plan_wire(d, p, true);
      return 0;
    }
  }
  int tilesB = (p.Cb + TB - 1) / TB;
  p.tilesB = tilesB;
  const int bkm = d->dtype == GAN_F32 ? 32 : 64;
  p.kchunks = (int)((M + bkm - 1) / bkm);
  long long blocks = (long long)tilesA * tilesB * taps;
  int splits = 1;
  const int wtarget = gan_opt("wgrad.split_target");
  // fold mode streams the SMALL tensor once: HBM-bound, wants many blocks; beside a mirror chain (concurrent == 2) half the chip is the target
  const long long target = p.fold ? gan_opt("wgrad.fold_split_target") : (d->concurrent >= 2 ? (wtarget + 1) / 2 : wtarget);
  if (blocks < target) {
    splits = (int)((target + blocks - 1) / blocks);
    int maxs = p.kchunks / 4; if (maxs < 1) maxs = 1;
    if (splits > maxs) splits = maxs;
    if (splits > 1024) splits = 1024;
  }
  p.splits = splits;
  plan_reduce_adam(d, p);
  plan_wire(d, p, allow_swap);
  if (const GanAdamFuse* af = d->adam_fuse; af && allow_swap && splits == 1 && !p.fold && !p.swap && !d->accumulate && d->dtype != GAN_F32 &&
      taps == 16 && TA == 128 && TB == 128 && p.CaReal % 8 == 0 && p.CbReal % 8 == 0 && af->master && af->m && af->v && af->lr_t &&
      !(((uintptr_t)af->master | (uintptr_t)af->m | (uintptr_t)af->v | (uintptr_t)af->nk_native | (uintptr_t)af->nk_transposed) & 15)) {
    p.adam = 1;
    p.aw = af->master; p.am = af->m; p.av = af->v; p.anat = af->nk_native; p.atr = af->nk_transposed; p.alr = af->lr_t;
    p.omb1 = 1.f - af->beta1; p.omb2 = 1.f - af->beta2; p.aeps = af->eps;
    if (gan_opt("wgrad.dead_taps")) {
      for (int t = 0; t < 16; ++t) {
        bool ly = false, lx = false;
        for (int g = 0; g < s.h && !ly; ++g) ly = (unsigned)(g * p.S + (t >> 2) - 1) < (unsigned)p.Hb;
        for (int g = 0; g < s.w && !lx; ++g) lx = (unsigned)(g * p.S + (t & 3) - 1) < (unsigned)p.Wb;
        if (!(ly && lx)) p.dead_taps |= 1 << t;
      }
      if (p.dead_taps && M <= 64) p.ahalves = 2;
    }
    // a launch with a short reduction is the optimiser step with a few MFMAs in front: one 128 x 128 tile per workgroup is 256 workgroups
    // for a 4.2 M-parameter kernel - one per CU, every phase of the epilogue exposed; with the tile's two 64-row halves on two workgroups
    // (each repeats the small GEMM) two of them share a CU and overlap
    if (M <= gan_opt("wgrad.adam_halves_max_rows")) p.ahalves = 2;
  }
  pl->TA = TA; pl->TB = TB;
  pl->grid = dim3((unsigned)(tilesA * tilesB), (unsigned)taps, (unsigned)(p.ahalves == 2 ? 2 : splits));
  pl->slab_bytes = splits > 1 ? (size_t)splits * 16 * p.CaReal * p.CbReal * sizeof(float) : 0;
  return 0;
}

template <typename T, int TA, int TB, int WA, int WB, bool TR>
static int launch_wcfg(const WgradPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr int ES = sizeof(T);
  constexpr size_t smem = 2 * (128 / ES) * ((TA * ES + 16) + (TB * ES + 16));
  auto kern = wgrad_kernel<T, TA, TB, WA, WB, TR>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  GAN_LAUNCH(kern, pl.grid, dim3(256), smem, st, pl.p);
  GAN_CHECK_LAUNCH();
  return 0;
}

template <typename T, bool TR>
static int launch_wgrad(const WgradPlan& pl, hipStream_t st) {
  const int key = pl.TA * 1000 + pl.TB;
  switch (key) {
    case 128128: return launch_wcfg<T, 128, 128, 2, 2, TR>(pl, st);
    case 128064: return launch_wcfg<T, 128, 64, 2, 2, TR>(pl, st);
    case 64128: return launch_wcfg<T, 64, 128, 2, 2, TR>(pl, st);
    case 64064: return launch_wcfg<T, 64, 64, 2, 2, TR>(pl, st);
    case 128016: return launch_wcfg<T, 128, 16, 4, 1, TR>(pl, st);
    case 64016: return launch_wcfg<T, 64, 16, 4, 1, TR>(pl, st);
    default: return GAN_E_SHAPE;
  }
}

// ------------------------------------------------------------------------------------------------
// v2: LDS-DMA pipeline.  Both operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds; zero
// padding / ragged rows through the descriptor's range check), row-major [m][channel] rows, 16-byte slots
// XOR-swizzled by SWZ(row) through the SOURCE address so the transposing fragment reads of a 32-lane half
// cover all 64 banks; NS stages with counted vmcnt + raw s_barrier; fragment reads in inline asm (hipcc would
// otherwise drain every in-flight LDS-DMA before a compiler-visible LDS read).
template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int TA, int TB, int WAVES_A, int WAVES_B, int NS>
__global__ __launch_bounds__(64 * WAVES_A * WAVES_B) void wgrad_dma_kernel(const WgradParams p, unsigned bigbytes, unsigned smallbytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int VEC = VecOf<T>::N;
  constexpr int ES = sizeof(T);
  constexpr int BKM = 128 / ES;                       // rows (m) per stage: 64 bf16 / 32 fp32
  constexpr int RSA = TA * ES, RSB = TB * ES;         // unpadded LDS row strides
  constexpr int LPA = RSA / 16, LPB = RSB / 16;       // lanes (16-B slots) per row
  constexpr int RPA = 64 / LPA, RPB = 64 / LPB;       // rows per 1-KiB piece
  constexpr int PA = BKM / RPA, PB = BKM / RPB;       // pieces per stage
  constexpr int NW = WAVES_A * WAVES_B;
  constexpr int AI = (PA + NW - 1) / NW, BI = (PB + NW - 1) / NW; // per wave
  constexpr int STAGE = BKM * (RSA + RSB);
  constexpr int WTA = TA / WAVES_A, WTB = TB / WAVES_B, MT = WTA / 16, NT = WTB / 16;
  static_assert(NS == 2 || (PA % NW == 0 && PB % NW == 0), "counted vmcnt needs equal pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // slot swizzle: rows {0..3} and {8..11} (and {4..7}, {12..15}) of a k-step must land on 8 different slot pairs
  auto swz = [](int row, int lanes_per_row) { return (((row & 3) | (((row >> 3) & 1) << 2)) << 1) & (lanes_per_row - 1) & ~1; };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave / WAVES_B, wb = wave % WAVES_B;
  const int r = lane & 15, q = lane >> 4;
  const int ta = blockIdx.x / p.tilesB, tb = blockIdx.x % p.tilesB;
  const int ca0 = ta * TA, cb0 = tb * TB;
  const int tap = blockIdx.y;
  const int zhalf = p.ahalves == 2 ? (int)blockIdx.z : -1, split = p.ahalves == 2 ? 0 : (int)blockIdx.z;
  const int kh = tap >> 2, kw = tap & 3;
  // GanAdamFuse: a tile whose gradient is exactly zero AND whose moments are zero is left as it is by the step (m, v stay 0,
  // the weight moves by 0 / (0 + eps)): the block reads m and v (8 of the 28 bytes per parameter) and leaves.  Two cases: a tap that
  // never meets the map at this shape (known before the GEMM: dead_taps) and an all-zero accumulator tile after it (e.g. everything
  // above an InstanceNorm over a single position).  A block that finds a non-zero moment runs the full path: always bit-identical.
  auto moments_zero = [&]() -> bool {
    unsigned nz = 0;
    for (int it = 0; it < TA * TB / 1024; ++it) {
      const int idx = tid + it * 256, row = idx / (TB / 4), c4 = (idx % (TB / 4)) * 4;
      const int a = ca0 + row, cb = cb0 + c4;
      if ((zhalf < 0 || (row >> 6) == zhalf) && a < p.CaReal && cb < p.CbReal) {
        const size_t o = ((size_t)tap * p.CaReal + a) * p.CbReal + cb;
        const uint4 m4 = *(const uint4*)(p.am + o), v4 = *(const uint4*)(p.av + o);
        nz |= m4.x | m4.y | m4.z | m4.w | v4.x | v4.y | v4.z | v4.w;
      }
    }
    return !__syncthreads_or(nz != 0);
  };
  if (p.adam == 1 && ((p.dead_taps >> tap) & 1) && moments_zero()) return;
  const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)p.big, 0, bigbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsmall = __builtin_amdgcn_make_buffer_rsrc((void*)p.small, 0, smallbytes, 0x00020000);

  // per-lane constants of the pieces this wave issues
  const int arow_l = lane / LPA, aslot = lane % LPA, brow_l = lane / LPB, bslot = lane % LPB;
  auto issue = [&](int kc, int stage) {
    unsigned char* As = smem + stage * STAGE;
    unsigned char* Bs = As + BKM * RSA;
    const unsigned mbase = (unsigned)kc * BKM;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int pc = wave + NW * i;
      if (pc < PA) {
        const int row = pc * RPA + arow_l;
        const unsigned m = mbase + row;
        int off = (int)0x80000000;
        if (m < (unsigned)p.M) {
          const unsigned t = fdiv(m, p.divW);
          const int gx = m - t * p.divW.d;
          const unsigned img = fdiv(t, p.divH);
          const int gy = t - img * p.divH.d;
          const int ce = (aslot ^ swz(row, LPA)) * VEC;          // element index inside the tile row
          int akh = kh, akw = kw, c = ca0 + ce;
          if (p.fold) { const int ft = ce >> 3; akh = ft >> 2; akw = ft & 3; c = ce & 7; }
          const int sy = gy * p.S + akh - 1 + p.shift, sx = gx * p.S + akw - 1 + p.shift;
          if ((unsigned)sy < (unsigned)p.Hb && (unsigned)sx < (unsigned)p.Wb && c < p.Ca)
            off = (int)((((size_t)(img * p.Hb + sy) * p.Wb + sx) * (size_t)p.bpitch + c) * ES);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbig, (__attribute__((address_space(3))) void*)(As + pc * 1024), 16, off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int pc = wave + NW * i;
      if (pc < PB) {
        const int row = pc * RPB + brow_l;
        const unsigned m = mbase + row;
        const int c = cb0 + (bslot ^ swz(row, LPB)) * VEC;
        const int off = (m < (unsigned)p.M && c < p.Cb) ? (int)(((size_t)m * p.spitch + c) * ES) : (int)0x80000000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsmall, (__attribute__((address_space(3))) void*)(Bs + pc * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto compute = [&](int stage) {
    const unsigned abase = lds_base + stage * STAGE, bbase = abase + BKM * RSA;
    if constexpr (sizeof(T) == 4) {
      // fp32: A[row = ca][k = m] one element per lane per v_mfma_f32_16x16x4_f32: element (m = kk*4+q, c = tile col + r)
#pragma unroll 2
      for (int kk = 0; kk < BKM / 4; ++kk) {
        const int row = kk * 4 + q;
        float af[MT], bfv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int col = wa * WTA + i * 16 + r;
          asm volatile("ds_read_b32 %0, %1" : "=v"(af[i]) : "v"(abase + row * RSA + (((col >> 2) ^ swz(row, LPA)) << 4) + (col & 3) * 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = wb * WTB + j * 16 + r;
          asm volatile("ds_read_b32 %0, %1" : "=v"(bfv[j]) : "v"(bbase + row * RSB + (((col >> 2) ^ swz(row, LPB)) << 4) + (col & 3) * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfv[j], af[i], acc[i][j], 0, 0, 0);   // transposed: lane = row a, 4 cols
      }
    } else {
      // bf16: 16-lane group q reads rows ks*32+8q+{0..3} then {4..7} with ds_read_b64_tr_b16; lane i of the group
      // supplies row (i>>2), elements 4*(i&3).. of the 16-column block and receives column i of the 4 rows
      typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
      for (int ks = 0; ks < BKM / 32; ++ks) {
        s16x4 alo[MT], ahi[MT], blo[NT], bhi[NT];
        const int row0 = ks * 32 + 8 * q + (r >> 2), row1 = row0 + 4;
        const int sub = ((r & 3) >> 1), half = (r & 1) * 8;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int slot = (wa * WTA + i * 16) / 8 + sub;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(alo[i]) : "v"(abase + row0 * RSA + ((slot ^ swz(row0, LPA)) << 4) + half));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ahi[i]) : "v"(abase + row1 * RSA + ((slot ^ swz(row1, LPA)) << 4) + half));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int slot = (wb * WTB + j * 16) / 8 + sub;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[j]) : "v"(bbase + row0 * RSB + ((slot ^ swz(row0, LPB)) << 4) + half));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bhi[j]) : "v"(bbase + row1 * RSB + ((slot ^ swz(row1, LPB)) << 4) + half));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          s16x8 av = __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            s16x8 bv = __builtin_shufflevector(blo[j], bhi[j], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[i][j] = mma16<T>(*(const uint4*)&bv, *(const uint4*)&av, acc[i][j]);
          }
        }
      }
    }
  };

  const int kc_begin = (int)((long long)p.kchunks * split / p.splits);
  const int kc_end = (int)((long long)p.kchunks * (split + 1) / p.splits);
  const int nk = kc_end - kc_begin;
  constexpr int PT = AI + BI;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(kc_begin + s, s);
  int st_c = 0, st_i = NS - 1;
  for (int i = 0; i < nk; ++i) {
    const int pending = nk - 1 - i < NS - 2 ? nk - 1 - i : NS - 2;
    if (NS >= 4 && pending == 2) wg_wait_vmcnt<2 * PT>();
    else if (NS >= 3 && pending == 1) wg_wait_vmcnt<PT>();
    else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (i + NS - 1 < nk) issue(kc_begin + i + NS - 1, st_i);
    compute(st_c);
    st_c = st_c + 1 == NS ? 0 : st_c + 1;
    st_i = st_i + 1 == NS ? 0 : st_i + 1;
  }

  const size_t per_split = (size_t)16 * p.CaReal * p.CbReal;
  float* out = p.splits > 1 ? p.slab + (size_t)split * per_split : p.dw;
  // The MFMAs ran with the SMALL-tensor fragment as the "A" operand, so acc[i][j][e] = dW[a = i*16 + (lane & 15)]
  // [cb = j*16 + (lane >> 4)*4 + e]: four consecutive cb per lane -> one 16-byte store where the layout allows.
  const bool vec4 = !p.swap && p.CbReal % 4 == 0;
  if (p.adam) {
    // GanAdamFuse (128 x 128 tiles, 4 waves as 2 x 2; the planner asks for nothing else): the gradient tile goes through LDS,
    // 64 rows at a time, so that the optimiser step touches master / m / v in whole 512-byte rows (32 lanes x 16 bytes, as
    // adam_prep_multi_kernel does: straight from the accumulator layout it was 64-byte pieces of 16 rows per instruction and
    // the launch took twice as long as the stand-alone pass), writes the native NK rows, leaves the UPDATED weights in LDS and
    // transposes them into the other NK copy.  Same per-element arithmetic as the stand-alone kernels (gan_adam1); dw is not written.
    if constexpr (!(sizeof(T) == 2 && TA == 128 && TB == 128 && WAVES_A == 2 && WAVES_B == 2)) {
      __builtin_trap();                                      // launch_wdma refuses p.adam for every other instantiation (GAN_E_SHAPE)
    } else {
      constexpr int LP = 129;                                // padded row of the 64 x 128 fp32 half tile
      float* tile = (float*)smem;
      const float lr = *p.alr;
      {
        unsigned gnz = 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) gnz |= __float_as_uint(acc[i][j][e]) << 1;       // (+0 and -0 alike)
        if (!__syncthreads_or(gnz != 0) && moments_zero()) return;
      }
      __syncthreads();                                       // every wave has left the last stage
      for (int h = 0; h < 2; ++h) {
        if (zhalf >= 0 && h != zhalf) continue;             // (block-uniform: this block updates one half only)
        if (wa == h) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) tile[(i * 16 + r) * LP + wb * WTB + j * 16 + q * 4 + e] = acc[i][j][e];
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
          const int idx = tid + it * 256, row = idx >> 5, c4 = (idx & 31) * 4;
          const int a = ca0 + h * 64 + row, cb = cb0 + c4;
          if (a < p.CaReal && cb < p.CbReal) {
            const size_t o = ((size_t)tap * p.CaReal + a) * p.CbReal + cb;
            float* t4 = tile + row * LP + c4;
            float4 pp = *(const float4*)(p.aw + o), mm = *(const float4*)(p.am + o), vv = *(const float4*)(p.av + o);
            gan_adam1(pp.x, mm.x, vv.x, t4[0], 1.f, p.omb1, p.omb2, lr, p.aeps);
            gan_adam1(pp.y, mm.y, vv.y, t4[1], 1.f, p.omb1, p.omb2, lr, p.aeps);
            gan_adam1(pp.z, mm.z, vv.z, t4[2], 1.f, p.omb1, p.omb2, lr, p.aeps);
            gan_adam1(pp.w, mm.w, vv.w, t4[3], 1.f, p.omb1, p.omb2, lr, p.aeps);
            *(float4*)(p.aw + o) = pp; *(float4*)(p.am + o) = mm; *(float4*)(p.av + o) = vv;
            if (p.anat) *(uint2*)((T*)p.anat + o) = make_uint2(pack2<T>(pp.x, pp.y), pack2<T>(pp.z, pp.w));
            t4[0] = pp.x; t4[1] = pp.y; t4[2] = pp.z; t4[3] = pp.w;
          }
        }
        __syncthreads();
        if (p.atr) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256, bl = idx >> 3, a8 = (idx & 7) * 8;
            const int b = cb0 + bl, a = ca0 + h * 64 + a8;
            if (b < p.CbReal && a < p.CaReal) {
              const float* c = tile + a8 * LP + bl;
              const uint4 w8 = make_uint4(pack2<T>(c[0], c[LP]), pack2<T>(c[2 * LP], c[3 * LP]), pack2<T>(c[4 * LP], c[5 * LP]),
                                          pack2<T>(c[6 * LP], c[7 * LP]));
              *(uint4*)((T*)p.atr + ((size_t)tap * p.CbReal + b) * p.CaReal + a) = w8;
            }
          }
        }
        __syncthreads();
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int a = ca0 + wa * WTA + i * 16 + r;
    int otap = tap, oc = a;
    if (p.fold) { otap = a >> 3; oc = a & 7; }
    if (oc >= p.CaReal) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int cb = cb0 + wb * WTB + j * 16 + q * 4;
      if (vec4) {
        if (cb < p.CbReal) {
          const size_t oo = ((size_t)otap * p.CaReal + oc) * p.CbReal + cb;
          if (p.wire && p.splits == 1) { *(uint2*)((unsigned short*)p.wire + oo) = wire_bf16x4(acc[i][j]); continue; }
          f32x4* o = (f32x4*)(out + oo);
          *o = (p.splits == 1 && p.accumulate) ? *o + acc[i][j] : acc[i][j];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + e < p.CbReal) {
            const size_t o = p.swap ? ((size_t)(15 - otap) * p.CbReal + cb + e) * p.CaReal + oc
                                    : ((size_t)otap * p.CaReal + oc) * p.CbReal + cb + e;
            if (p.splits == 1 && p.accumulate) out[o] += acc[i][j][e];
            else out[o] = acc[i][j][e];
          }
      }
    }
  }
#endif
}

template <typename T, int TA, int TB, int WA, int WB, int NS>
static int launch_wdma(const WgradPlan& pl, unsigned bigbytes, unsigned smallbytes, hipStream_t st) {
  static bool attr_set = false;
  constexpr int ES = sizeof(T);
  constexpr size_t smem = (size_t)NS * (128 / ES) * (TA + TB) * ES;
  auto kern = wgrad_dma_kernel<T, TA, TB, WA, WB, NS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (pl.p.adam && !(sizeof(T) == 2 && TA == 128 && TB == 128 && WA == 2 && WB == 2)) return GAN_E_SHAPE;   // only that epilogue carries GanAdamFuse
  GAN_LAUNCH(kern, pl.grid, dim3(64 * WA * WB), smem, st, pl.p, bigbytes, smallbytes);
  GAN_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int launch_wgrad_dma(const WgradPlan& pl, unsigned bigbytes, unsigned smallbytes, hipStream_t st) {
  const int key = pl.TA * 1000 + pl.TB;
  switch (key) {
    case 256128: return launch_wdma<T, 256, 128, 4, 2, 2>(pl, bigbytes, smallbytes, st);
    case 128128: return launch_wdma<T, 128, 128, 2, 2, 2>(pl, bigbytes, smallbytes, st);
    case 128064: return launch_wdma<T, 128, 64, 2, 2, 3>(pl, bigbytes, smallbytes, st);
    case 64128: return launch_wdma<T, 64, 128, 2, 2, 3>(pl, bigbytes, smallbytes, st);
    case 64064: return launch_wdma<T, 64, 64, 2, 2, 3>(pl, bigbytes, smallbytes, st);
    case 128016: return launch_wdma<T, 128, 16, 4, 1, 2>(pl, bigbytes, smallbytes, st);
    case 64016: return launch_wdma<T, 64, 16, 4, 1, 2>(pl, bigbytes, smallbytes, st);
    default: return GAN_E_SHAPE;
  }
}

// ------------------------------------------------------------------------------------------------
// v3: "ping-pong" kernel, 256 x 256 output tile (rows = (tap, BIG channel), columns = SMALL channel), 8 waves as
// 2 x 4 with 128 x 64 wave tiles, the structure of conv_gemm_pp_kernel (conv_gemm.hip has the ordering rules): waves
// 4..7 run one barrier behind waves 0..3, so that on every SIMD one wave multiplies while its partner stages and reads;
// LDS-DMA pieces stay in flight across barriers behind a counted vmcnt.  A 128 x 128 tile needs twice the staged bytes
// per FLOP and is bound by the CU's LDS-DMA rate at ~45 % of the MFMA peak; this one halves them.
//
// A K tile = 64 reduction rows m.  Both operands are staged as 64-row x 64-channel PANELS (128-byte rows, 8 KiB, 8
// pieces): 4 BIG panels (wave row x row half; a panel is one tap's 64-channel slice) and 4 SMALL panels (one per wave
// column).  A wave issues piece `wave` of every panel, so a lane serves ONE reduction row per K tile: the row is decoded
// once per K tile (window origin + 16-bit tap mask, as in the convolution kernel).  16-byte chunks are XOR-swizzled
// through the source address by ((row >> 1) & 1 | (row >> 3) & 1 << 1) << 1, which makes the transposing fragment reads
// (ds_read_b64_tr_b16: rows 8q..8q+3 of a 32-byte column segment per 16-lane group) bank-conflict free and is the same
// for every read a lane issues.
// Slots of a K tile per wave, in staging order: A0 A0 | B B | B B | A1 A1 (A0/A1 = first/second 64 rows of each wave
// row); first read in phase 0 0 0 0 0 0 2 2, last read 0 0 1 1 1 1 2 2 -> staged 6 phases ahead, at most 6 pieces
// outstanding at the end of a load segment, 2 stages.
template <int IMM> __device__ __forceinline__ void lds_read_tr(s16x4& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int N, typename F> __device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (N > 0) { wg_static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// NTAP = 0: the reduction row of a K tile is decoded in the loop (45 vector instructions per K tile in front of the pieces + 4 per BIG
// piece).  NTAP = 1 | 2 | 4 (round 4; the number of distinct taps among the tile's four BIG panels): the block decodes ITS rows once, into
// a table in LDS behind the stages - per row and tap `window origin + tap offset`, or 0x80000000 where the tap leaves the map -; in the
// loop a lane reads its row's entries one K tile ahead (ds_read, consumed behind the MATH segment's lgkmcnt(0)), adds its swizzled chunk
// (NTAP vector instructions per K tile) and the pieces name the results + the panel's channel offset as scalar offset; the SMALL panels
// take a loop-invariant row register + a scalar (K tile, panel) offset, and K tiles past the block's range a zero-record descriptor.
// A knock-out of the decode alone measured -12 % on the class (+1.5 % on the step).
template <typename T, int NTAP>
__global__ __launch_bounds__(512) void wgrad_pp_kernel(const WgradParams p, unsigned bigbytes, unsigned smallbytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(sizeof(T) == 2, "16-bit storage types");
#ifndef WPP_DP
#define WPP_DP 5      // (6 = the deepest the two stages allow: 1 % slower; 4: +12 %)
#endif
  constexpr int PANEL = 64 * 128, STAGE = 8 * PANEL, NB = 2, PH = 4, PP = 2, DP = WPP_DP, VMW = 2 * DP - 6, QS = 8;
  static_assert(DP >= 4 && DP <= 6, "staged 4..6 phases ahead (WAR: <= 6 with two stages; RAW: the last B pair is read DP - 2 phases after its issue)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  WDIAG_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  // Block -> (tile, split).  Every tile of one split reads the same reduction rows of both tensors, so a split's tiles
  // are kept on ONE XCD (blocks b and b + 8 share an XCD and its L2): XCD x takes the splits x, x + 8, ... and walks
  // their tiles; otherwise each of the 8 L2s fetches every row once per tile.  (Speed only: any mapping is correct.)
  const int tiles = (int)gridDim.x / p.splits;
  int tile, split;
  if ((p.splits & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles; split = xcd + 8 * (j / tiles);
  } else {
    tile = blockIdx.x % tiles; split = blockIdx.x / tiles;
  }
  const int ta = tile / p.tilesB, tb = tile % p.tilesB;
  const int tr0 = ta * 256, cb0 = tb * 256;
  const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)p.big, 0, bigbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsmall = __builtin_amdgcn_make_buffer_rsrc((void*)p.small, 0, smallbytes, 0x00020000);

  // staging: this lane's row inside every panel, its swizzled source chunk; the 4 BIG panels' tap / channel offsets
  const int lrow = lane >> 3, prow = wave * 8 + lrow;
  const int chunk16 = (((lane & 7) ^ ((((prow >> 1) & 1) | (((prow >> 3) & 1) << 1)) << 1)) << 4);
  int s_aoff[4];
  unsigned s_bit[4];
#pragma unroll
  for (int pn = 0; pn < 4; ++pn) {                       // panel pn = wave-row * 2 + half
    const int row0 = tr0 + (pn >> 1) * 128 + (pn & 1) * 64;
    const int tap = row0 / p.Ca, ch = row0 % p.Ca;
    s_aoff[pn] = (((tap >> 2) * p.Wb + (tap & 3)) * p.bpitch + ch) * 2;
    s_bit[pn] = 1u << tap;
  }
  const int kc_begin = (int)((long long)p.kchunks * split / p.splits);
  const int kc_end = (int)((long long)p.kchunks * (split + 1) / p.splits);
  const int nk = kc_end - kc_begin;

  int a_org = 0, b_off = (int)0x80000000;
  unsigned a_msk = 0;
  auto rowinfo = [&](int ts) {                           // decode this lane's reduction row of K tile ts  (NTAP = 0)
    const unsigned m = (unsigned)(kc_begin + ts) * 64u + (unsigned)prow;
    a_msk = 0; a_org = 0; b_off = (int)0x80000000;
    if (ts < nk && m < (unsigned)p.M) {
      const unsigned t = fdiv(m, p.divW);
      const int gx = (int)(m - t * p.divW.d);
      const unsigned img = fdiv(t, p.divH);
      const int gy = (int)(t - img * p.divH.d);
      const int sy0 = gy * p.S - 1, sx0 = gx * p.S - 1;
      unsigned vx = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) vx |= ((unsigned)(sx0 + k) < (unsigned)p.Wb ? 1u : 0u) << k;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((unsigned)(sy0 + k) < (unsigned)p.Hb) a_msk |= vx << (4 * k);
      a_org = (int)((((long long)((int)img * p.Hb + sy0) * p.Wb + sx0) * (long long)p.bpitch) * 2) + chunk16;
      b_off = (int)(((size_t)m * p.spitch + cb0) * 2) + chunk16;
    }
  };
  // ---- NTAP > 0: row table ----
  constexpr int NT_ = NTAP > 0 ? NTAP : 1;
  int s_ch[4], s_tapoff[NT_];                              // panel channel offsets (scalar offsets of the pieces); tap offsets of the table's columns
  int voff[NT_] = {}, ent[NT_] = {};                       // this lane's BIG piece offsets of the K tile being issued; the next tile's table entries
  const __amdgpu_buffer_rsrc_t rsmall0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.small, 0, 0u, 0x00020000);
  const int b_off0 = (int)((((size_t)kc_begin * 64 + prow) * p.spitch + cb0) * 2) + chunk16;
  const int tilebytes = 64 * p.spitch * 2;
  const unsigned tbl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + 2 * STAGE);
  if constexpr (NTAP > 0) {
#pragma unroll
    for (int pn = 0; pn < 4; ++pn) s_ch[pn] = ((tr0 + (pn >> 1) * 128 + (pn & 1) * 64) % p.Ca) * 2;
#pragma unroll
    for (int k = 0; k < NTAP; ++k) {
      const int pn = k * 4 / NTAP;                         // first panel of the k-th distinct tap (1: all panels, 2: wave rows, 4: every panel)
      const int tap = (tr0 + (pn >> 1) * 128 + (pn & 1) * 64) / p.Ca;
      s_tapoff[k] = (((tap >> 2) * p.Wb + (tap & 3)) * p.bpitch) * 2;
    }
    const int nrows = (nk + 2) * 64;                       // two K tiles past the end are issued (out of range)
    int* tbl = (int*)(smem + 2 * STAGE);
    for (int row = tid; row < nrows; row += 512) {
      const unsigned m = (unsigned)kc_begin * 64u + (unsigned)row;
      int e[NTAP];
#pragma unroll
      for (int k = 0; k < NTAP; ++k) e[k] = (int)0x80000000;
      if (row < nk * 64 && m < (unsigned)p.M) {
        const unsigned t = fdiv(m, p.divW);
        const int gx = (int)(m - t * p.divW.d);
        const unsigned img = fdiv(t, p.divH);
        const int gy = (int)(t - img * p.divH.d);
        const int sy0 = gy * p.S - 1, sx0 = gx * p.S - 1;
        const int org = (int)((((long long)((int)img * p.Hb + sy0) * p.Wb + sx0) * (long long)p.bpitch) * 2);
#pragma unroll
        for (int k = 0; k < NTAP; ++k) {
          const int pn = k * 4 / NTAP;
          const int tap = (tr0 + (pn >> 1) * 128 + (pn & 1) * 64) / p.Ca;
          if ((unsigned)(sy0 + (tap >> 2)) < (unsigned)p.Hb && (unsigned)(sx0 + (tap & 3)) < (unsigned)p.Wb) e[k] = org + s_tapoff[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NTAP; ++k) tbl[row * NTAP + k] = e[k];
    }
    __syncthreads();
  }
  auto read_table = [&](int ts) {                          // this lane's entries of K tile ts -> ent[] (valid after the next lgkmcnt(0))
    const unsigned a = tbl_lds + (unsigned)((ts * 64 + prow) * 4 * NTAP);
    if constexpr (NTAP == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(ent[0]) : "v"(a));
    if constexpr (NTAP == 2) { typedef __attribute__((ext_vector_type(2))) int i2; i2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); ent[0] = v[0]; ent[1] = v[1]; }
    if constexpr (NTAP == 4) { typedef __attribute__((ext_vector_type(4))) int i4; i4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); ent[0] = v[0]; ent[1] = v[1]; ent[2] = v[2]; ent[3] = v[3]; }
  };
  auto make_voff = [&]() {
#pragma unroll
    for (int k = 0; k < NT_; ++k) voff[k] = ent[k] + chunk16;
  };
  auto issue_slots = [&](auto I0c, auto CNTc, int ts) {
    constexpr int I0 = decltype(I0c)::value, CNT = decltype(CNTc)::value;
    unsigned char* st = smem + (ts & 1) * STAGE + wave * 1024;
    wg_static_for<CNT>([&](auto Ic) {
      constexpr int i = I0 + decltype(Ic)::value;
      if constexpr (i < 2 || i >= 6) {
        constexpr int pn = (i & 1) * 2 + (i >= 6 ? 1 : 0);
        if constexpr (NTAP > 0) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rbig, (__attribute__((address_space(3))) void*)(st + pn * PANEL), 16, voff[pn * NTAP / 4], s_ch[pn], 0, 0);
        } else {
          const int off = (a_msk & s_bit[pn]) ? a_org + s_aoff[pn] : (int)0x80000000;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rbig, (__attribute__((address_space(3))) void*)(st + pn * PANEL), 16, off, 0, 0, 0);
        }
      } else {
        constexpr int k = i - 2;
        if constexpr (NTAP > 0) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ts < nk ? rsmall : rsmall0, (__attribute__((address_space(3))) void*)(st + (4 + k) * PANEL), 16, b_off0,
                                                   ts * tilebytes + k * 128, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsmall, (__attribute__((address_space(3))) void*)(st + (4 + k) * PANEL), 16,
                                                   b_off + k * 128, 0, 0, 0);
        }
      }
    });
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: element column tile*16 + 4*(r&3) of rows s*32 + 8q + (r>>2) (+4): byte (tile ^ swz) << 5 | ((r>>1)&1) << 4 | 8*(r&1)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned swl = (unsigned)(((r >> 3) & 1) | ((q & 1) << 1));
  const unsigned lrow_off = (unsigned)((8 * q + (r >> 2)) * 128 + (((r >> 1) & 1) << 4) + 8 * (r & 1));

  WDIAG_STAMP(1);
  // prologue: slots of phases -6 .. -1 = all of K tile 0 and A0, B, B of K tile 1
  auto row_state = [&](int ts) {                           // the row state of K tile ts, now (prologue)
    if constexpr (NTAP > 0) { read_table(ts); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); make_voff(); }
    else rowinfo(ts);
  };
  row_state(0);
  issue_slots(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{}, 0);
  if constexpr (DP > 4 || NTAP > 0) row_state(1);          // (DP = 4 with the table: the loop's first phase issues tile 1's first pieces)
  if constexpr (DP > 4) issue_slots(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (DP - 4)>{}, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
  __builtin_amdgcn_s_barrier();
  WDIAG_STAMP(2);
  if (wr == 1) __builtin_amdgcn_s_barrier();

  s16x4 alo[2][4], ahi[2][4], b0lo[2][2], b0hi[2][2], b1lo[2][2], b1hi[2][2];
  WSEG_DECL;
  WSEG_T0;
  // fragment addresses of both stages: loop invariants (the K loop is unrolled by the stage so that a LOAD segment adds nothing to them)
  unsigned btA[2][4], btB[2][4];
#pragma unroll
  for (int st_ = 0; st_ < 2; ++st_)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned a = lds_base + (unsigned)(st_ * STAGE) + lrow_off + ((k ^ swl) << 5);
      btA[st_][k] = a + wr * 2 * PANEL; btB[st_][k] = a + wc * PANEL;
    }
  auto ktile = [&](auto STc, int t) {
    constexpr int ST = decltype(STc)::value;
    wg_static_for<PH>([&](auto Pc) {
      constexpr int ph = decltype(Pc)::value;
      constexpr int mh = ph < 2 ? 0 : 1, nh = (ph == 1 || ph == 2) ? 1 : 0;
      // ---- LOAD segment ----
      constexpr int x0 = PP * (ph + DP);                   // 12, 14, 16, 18 -> (tile t+1: slots 4,5 | 6,7), (tile t+2: 0,1 | 2,3)
      if constexpr (NTAP == 0 && (ph + DP) % 4 == 0) rowinfo(t + (ph + DP) / 4);
      issue_slots(std::integral_constant<int, x0 % QS>{}, std::integral_constant<int, PP>{}, t + x0 / QS);
      if constexpr (NTAP > 0 && (ph + 1 + DP) % 4 == 0) read_table(t + (ph + 1 + DP) / 4);    // (the NEXT phase issues that tile's first pieces)
      WSEG_ADD(5);                                         // row decode + LDS-DMA issue
      if constexpr (ph == 0 || ph == 1) {
        wg_static_for<2>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          constexpr int imm = 4 * PANEL;                   // SMALL panels follow the 4 BIG panels
          wg_static_for<2>([&](auto Sc) {
            constexpr int s2 = decltype(Sc)::value;
            if constexpr (nh == 0) { lds_read_tr<imm + s2 * 4096>(b0lo[s2][j], btB[ST][j]); lds_read_tr<imm + s2 * 4096 + 512>(b0hi[s2][j], btB[ST][j]); }
            else { lds_read_tr<imm + s2 * 4096>(b1lo[s2][j], btB[ST][2 + j]); lds_read_tr<imm + s2 * 4096 + 512>(b1hi[s2][j], btB[ST][2 + j]); }
          });
        });
      }
      if constexpr (ph == 0 || ph == 2) {
        wg_static_for<4>([&](auto Ic) {
          constexpr int i = decltype(Ic)::value;
          wg_static_for<2>([&](auto Sc) {
            constexpr int s2 = decltype(Sc)::value;
            constexpr int imm = mh * PANEL + s2 * 4096;
            lds_read_tr<imm>(alo[s2][i], btA[ST][i]); lds_read_tr<imm + 512>(ahi[s2][i], btA[ST][i]);
          });
        });
      }
      WSEG_ADD(0);                                         // fragment read issue
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
      WSEG_ADD(1);                                         // waiting for older pieces
      __builtin_amdgcn_s_barrier();
      WSEG_ADD(2);                                         // waiting for the partner group's MATH segment
      // ---- MATH segment ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (NTAP > 0 && (ph + 1 + DP) % 4 == 0) make_voff();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 av = __builtin_shufflevector(alo[s2][i], ahi[s2][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const s16x8 bv = nh == 0 ? __builtin_shufflevector(b0lo[s2][j], b0hi[s2][j], 0, 1, 2, 3, 4, 5, 6, 7)
                                     : __builtin_shufflevector(b1lo[s2][j], b1hi[s2][j], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[mh * 4 + i][nh * 2 + j] = mma16<T>(*(const uint4*)&bv, *(const uint4*)&av, acc[mh * 4 + i][nh * 2 + j]);
          }
        }
      __builtin_amdgcn_s_setprio(0);
      WSEG_ADD(3);                                         // fragment wait + MFMAs
      __builtin_amdgcn_s_barrier();
      WSEG_ADD(4);                                         // waiting for the partner group's LOAD segment
    });
  };
  for (int t = 0; t < nk; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) ktile(std::integral_constant<int, 1>{}, t + 1);
  }
  WSEG_STORE;
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WDIAG_STAMP(3);

  // acc[i][j][e] = dW[tile row wr*128 + i*16 + r][column wc*64 + j*16 + q*4 + e]
  const size_t per_split = (size_t)16 * p.CaReal * p.CbReal;
  float* out = p.splits > 1 ? p.slab + (size_t)split * per_split : p.dw;
  const bool vec4 = p.CbReal % 4 == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int R = tr0 + wr * 128 + i * 16 + r;
    const int tap = R / p.Ca, ca = R % p.Ca;
    if (tap >= 16 || ca >= p.CaReal) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cb = cb0 + wc * 64 + j * 16 + q * 4;
      float* o = out + ((size_t)tap * p.CaReal + ca) * p.CbReal + cb;
      if (vec4) {
        if (cb < p.CbReal) {
          if (p.wire && p.splits == 1) *(uint2*)((unsigned short*)p.wire + (o - out)) = wire_bf16x4(acc[i][j]);
          else *(f32x4*)o = (p.splits == 1 && p.accumulate) ? *(f32x4*)o + acc[i][j] : acc[i][j];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + e < p.CbReal) o[e] = (p.splits == 1 && p.accumulate) ? o[e] + acc[i][j][e] : acc[i][j][e];
      }
    }
  }
#ifdef GAN_DIAG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the slab stores have been acknowledged: the stamp closes the epilogue)
#endif
  WDIAG_STAMP(4);
#endif
}

template <typename T, int NTAP>
static int launch_wpp_v(const WgradPlan& pl, unsigned bigbytes, unsigned smallbytes, hipStream_t st) {
  static bool attr_set = false;
  constexpr size_t smem = 2 * 8 * 64 * 128 + (NTAP > 0 ? 32768 : 0);      // stages + row table
  auto kern = wgrad_pp_kernel<T, NTAP>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  GAN_LAUNCH(kern, pl.grid, dim3(512), smem, st, pl.p, bigbytes, smallbytes);
  GAN_CHECK_LAUNCH();
  return 0;
}
template <typename T>
static int launch_wpp(const WgradPlan& pl, unsigned bigbytes, unsigned smallbytes, hipStream_t st) {
  // row table (wgrad_pp_kernel<T, NTAP>): one column per distinct tap among a tile's four 64-channel panels; (K tiles per block + 2) x 64 rows
  const int ntap = pl.p.Ca % 256 == 0 ? 1 : (pl.p.Ca % 128 == 0 ? 2 : 4);
  const long long nkmax = (pl.p.kchunks + pl.p.splits - 1) / pl.p.splits + 1;
  if (gan_opt("wgrad.row_table") && pl.p.Ca % 64 == 0 && (nkmax + 2) * 64 * 4 * ntap <= 32768) {
    if (ntap == 1) return launch_wpp_v<T, 1>(pl, bigbytes, smallbytes, st);
    if (ntap == 2) return launch_wpp_v<T, 2>(pl, bigbytes, smallbytes, st);
    return launch_wpp_v<T, 4>(pl, bigbytes, smallbytes, st);
  }
  return launch_wpp_v<T, 0>(pl, bigbytes, smallbytes, st);
}


extern "C" {
int gan_conv_wgrad(const GanWgradDesc* d, gan_stream_t stream) {
  if (!d || d->struct_size != sizeof(GanWgradDesc)) return GAN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t es = d->dtype == GAN_F32 ? 4 : 2;
  size_t bb = (((size_t)d->big.n * d->big.h * d->big.w - 1) * d->big.pitch + d->big.c) * es;
  size_t sb = (((size_t)d->small.n * d->small.h * d->small.w - 1) * d->small.pitch + d->small.c) * es;
  const bool dma_ok = bb < 0x7fffffffull && sb < 0x7fffffffull && !(((uintptr_t)d->big.ptr | (uintptr_t)d->small.ptr) & 15);
  WgradPlan pl;
  int rc = plan_wgrad(d, &pl, dma_ok);         // the role-swapped plan exists only in the LDS-DMA kernel
  if (rc) return rc;
  if (pl.slab_bytes > d->workspace_bytes || (pl.slab_bytes && !d->workspace)) return GAN_E_WORKSPACE;
  // the caller skips its own optimiser pass for this kernel on the strength of gan_wgrad_adam_fused() == 1: a request this
  // launch's plan cannot honour (a planner option changed since the query) must fail, not leave the kernel without its update
  if (d->adam_fuse && !(pl.p.adam == 2 || (pl.p.adam == 1 && !pl.pp))) return GAN_E_SHAPE;
  if (d->dw_wire && !pl.p.wire) return GAN_E_SHAPE;        // (as for adam_fuse: the caller asks gan_wgrad_wire_direct() first)
  const int reduce_adam = pl.p.adam == 2;
  if (reduce_adam) pl.p.adam = 0;                      // (the GEMM kernels' own epilogue switch: they write plain slabs)
  if (pl.p.swap) { const size_t t = bb; bb = sb; sb = t; }
  if (pl.pp) rc = d->dtype == GAN_F16 ? launch_wpp<f16_t>(pl, (unsigned)bb, (unsigned)sb, st) : launch_wpp<bf16_t>(pl, (unsigned)bb, (unsigned)sb, st);
  else if (dma_ok) rc = d->dtype == GAN_F32 ? launch_wgrad_dma<float>(pl, (unsigned)bb, (unsigned)sb, st)
                   : d->dtype == GAN_F16 ? launch_wgrad_dma<f16_t>(pl, (unsigned)bb, (unsigned)sb, st)
                                         : launch_wgrad_dma<bf16_t>(pl, (unsigned)bb, (unsigned)sb, st);
  else if (d->dtype == GAN_F32) rc = launch_wgrad<float, false>(pl, st);
  else if (d->dtype == GAN_F16) rc = launch_wgrad<f16_t, true>(pl, st);
  else rc = launch_wgrad<bf16_t, true>(pl, st);
  if (rc) return rc;
  if (pl.p.splits > 1 && reduce_adam) {
    const WgradParams& q = pl.p;
    const long long count4 = (long long)4 * q.CaReal * q.CbReal;
    const int log2sg = wgrad_reduce_log2sg(count4, q.splits);
    const unsigned tiles = 16u * (unsigned)((q.CaReal + 63) / 64) * (unsigned)((q.CbReal + 63) / 64);
    const size_t per_split = (size_t)16 * q.CaReal * q.CbReal;
    if (d->dtype == GAN_F16)
      GAN_LAUNCH(wgrad_reduce_adam_kernel<f16_t>, dim3(tiles), dim3(256), 0, st, (const float*)q.slab, per_split, q.splits, log2sg, q.CaReal,
                         q.CbReal, q.aw, q.am, q.av, (f16_t*)q.anat, (f16_t*)q.atr, q.alr, q.omb1, q.omb2, q.aeps);
    else
      GAN_LAUNCH(wgrad_reduce_adam_kernel<bf16_t>, dim3(tiles), dim3(256), 0, st, (const float*)q.slab, per_split, q.splits, log2sg, q.CaReal,
                         q.CbReal, q.aw, q.am, q.av, (bf16_t*)q.anat, (bf16_t*)q.atr, q.alr, q.omb1, q.omb2, q.aeps);
    GAN_CHECK_LAUNCH();
  } else if (pl.p.splits > 1) {
    const long long count4 = (long long)4 * pl.p.CaReal * pl.p.CbReal;     // 16 taps * Ca * Cb floats, as float4
    const int log2sg = wgrad_reduce_log2sg(count4, pl.p.splits);
    const int EV = 256 >> log2sg;
    GAN_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((count4 + EV - 1) / EV)), dim3(256), 0, st,
                       (const float*)pl.p.slab, pl.p.dw, count4, pl.p.splits, pl.p.accumulate, log2sg, pl.p.wire);
    GAN_CHECK_LAUNCH();
  }
  return 0;
}
int gan_wgrad_plan_info(const GanWgradDesc* d, int32_t* info /*[4]: TA, TB, splits, fold*/) {
  WgradPlan pl;
  GanWgradDesc t = *d;
  if (!t.big.ptr) t.big.ptr = (void*)16;
  if (!t.small.ptr) t.small.ptr = (void*)16;
  if (!t.dw) t.dw = (float*)16;
  int rc = plan_wgrad(&t, &pl);
  if (rc) return rc;
  info[0] = pl.TA; info[1] = pl.TB; info[2] = pl.p.splits; info[3] = pl.p.fold;
  return 0;
}
int gan_wgrad_adam_fused(const GanWgradDesc* d) {
  if (!d || d->struct_size != sizeof(GanWgradDesc)) return GAN_E_ARG;
  const size_t es = d->dtype == GAN_F32 ? 4 : 2;
  const size_t bb = (((size_t)d->big.n * d->big.h * d->big.w - 1) * d->big.pitch + d->big.c) * es;
  const size_t sb = (((size_t)d->small.n * d->small.h * d->small.w - 1) * d->small.pitch + d->small.c) * es;
  const bool dma_ok = bb < 0x7fffffffull && sb < 0x7fffffffull && !(((uintptr_t)d->big.ptr | (uintptr_t)d->small.ptr) & 15);
  WgradPlan pl;
  const int rc = plan_wgrad(d, &pl, dma_ok);
  if (rc) return rc;
  return (pl.p.adam == 2 || (pl.p.adam == 1 && !pl.pp)) ? 1 : 0;
}
int gan_wgrad_wire_direct(const GanWgradDesc* d) {
  if (!d || d->struct_size != sizeof(GanWgradDesc)) return GAN_E_ARG;
  const size_t es = d->dtype == GAN_F32 ? 4 : 2;
  const size_t bb = (((size_t)d->big.n * d->big.h * d->big.w - 1) * d->big.pitch + d->big.c) * es;
  const size_t sb = (((size_t)d->small.n * d->small.h * d->small.w - 1) * d->small.pitch + d->small.c) * es;
  const bool dma_ok = bb < 0x7fffffffull && sb < 0x7fffffffull && !(((uintptr_t)d->big.ptr | (uintptr_t)d->small.ptr) & 15);
  WgradPlan pl;
  const int rc = plan_wgrad(d, &pl, dma_ok);
  if (rc) return rc;
  return pl.p.wire ? 1 : 0;
}
size_t gan_wgrad_workspace_bytes(const GanWgradDesc* d) {
  WgradPlan pl;
  GanWgradDesc t = *d;
  if (!t.big.ptr) t.big.ptr = (void*)16;
  if (!t.small.ptr) t.small.ptr = (void*)16;
  if (!t.dw) t.dw = (float*)16;
  if (plan_wgrad(&t, &pl)) return 0;
  return pl.slab_bytes;
}
}
