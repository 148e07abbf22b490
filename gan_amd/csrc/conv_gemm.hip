// Implicit-GEMM convolution for gfx950 (MI355X): Conv2D k4 (s1|s2) forward, Conv2DTranspose k4 s2
// forward and both input-gradients, all as ONE tap-gather GEMM kernel:
//
//   Y[m, n] = sum_{tap, c}  X[src(m, tap), c] * W[widx(tap)][n][c]
//
// rows m = (image, gy, gx) on a "GEMM grid"; src(m,tap) = (gy*S + dy(tap), gx*S + dx(tap)) with zero
// fill outside the source map; output pixel = (gy*OS + py, gx*OS + px).  A stride-2 transposed conv
// (and the dgrad of a stride-2 conv) is four output-parity sub-GEMMs with 4 taps each (K = 4*Cin),
// selected by blockIdx.z, so no zero-inserted input is ever materialised.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 128 bytes of K; LDS double-buffered, rows padded
// to 144 B so the 16 rows of a ds_read_b128 fragment read land on 16 distinct 16-B slots; global->VGPR
// prefetch of tile k+1 is issued before the MFMAs of tile k and written to the other LDS buffer after
// them (one barrier per K step).  bf16: v_mfma_f32_16x16x32_bf16; fp32 (parity path): exact
// v_mfma_f32_16x16x4_f32.  Small problems split K across blockIdx.z into fp32 slabs + a reduce kernel.
#include "common.h"

struct GemmParams {
  const void* x; const void* w; void* y; float* slab; const float* bias;
  int Nimg, Hs, Ws, xpitch, Cin, log2_cvecs;
  int Hg, Wg, M;
  int S, TWlog2, T;
  int dy0, dx0, dstep, wy0, wx0, wstep;
  int parity;
  int Wrows;
  int Ho, Wo, ypitch, Cout, OS;
  int splits, kchunks, NslabPitch;
  int act; float slope; int out_f32;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ void store_out(const GemmParams& p, size_t pix_off, int n, float v) {
  if (p.bias) v += p.bias[n];
  v = apply_act(v, p.act, p.slope);
  if (p.out_f32) ((float*)p.y)[pix_off + n] = v;
  else st_f((T*)p.y + pix_off + n, v);
}

__device__ __forceinline__ size_t out_pixel_offset(const GemmParams& p, int m, int py, int px) {
  int gx = m % p.Wg;
  int t = m / p.Wg;
  int gy = t % p.Hg;
  int img = t / p.Hg;
  return ((size_t)(img * p.Ho + gy * p.OS + py) * p.Wo + (gx * p.OS + px)) * (size_t)p.ypitch;
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmParams p) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int ROWB = 144;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, MT = WTM / 16, NT = WTN / 16;
  constexpr int AI = (BM + 31) / 32, BI = (BN + 31) / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * BM * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 15, q = lane >> 4;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  int par = 0, split = blockIdx.z;
  if (p.parity) { par = blockIdx.z / p.splits; split = blockIdx.z % p.splits; }
  const int py = par >> 1, px = par & 1;
  int dy0 = p.dy0, dx0 = p.dx0, wy0 = p.wy0, wx0 = p.wx0;
  if (p.parity) { dy0 = py; dx0 = px; wy0 = 1 - py; wx0 = 1 - px; }

  const int vec = tid & 7, row0 = tid >> 3;
  int ay[AI], ax[AI], ab[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int row = row0 + 32 * i;
    int m = bm0 + row;
    if (row < BM && m < p.M) {
      int gx = m % p.Wg;
      int t = m / p.Wg;
      int gy = t % p.Hg;
      ab[i] = (t / p.Hg) * p.Hs;
      ay[i] = gy * p.S;
      ax[i] = gx * p.S;
    } else {
      ab[i] = 0; ay[i] = -(1 << 20); ax[i] = -(1 << 20);
    }
  }
  const T* xg = (const T*)p.x;
  const T* wg = (const T*)p.w;
  const int cmask = (1 << p.log2_cvecs) - 1;
  const int twmask = (1 << p.TWlog2) - 1;

  uint4 ra[AI], rb[BI];
  auto gload = [&](int kc) {
    int kvec = kc * 8 + vec;
    int tap = kvec >> p.log2_cvecs;
    int cv = kvec & cmask;
    int ty = tap >> p.TWlog2, tx = tap & twmask;
    int dy = dy0 + ty * p.dstep, dx = dx0 + tx * p.dstep;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int sy = ay[i] + dy, sx = ax[i] + dx;
      bool ok = (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      size_t off = ((size_t)(ab[i] + sy) * p.Ws + sx) * (size_t)p.xpitch + (size_t)cv * VEC;
      ra[i] = ok ? *(const uint4*)(xg + off) : make_uint4(0, 0, 0, 0);
    }
    int widx = (wy0 + ty * p.wstep) * 4 + (wx0 + tx * p.wstep);
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int row = row0 + 32 * i;
      int n = bn0 + row;
      bool ok = row < BN && n < p.Wrows;
      size_t off = ((size_t)widx * p.Wrows + n) * (size_t)p.Cin + (size_t)cv * VEC;
      rb[i] = ok ? *(const uint4*)(wg + off) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int row = row0 + 32 * i;
      if (BM % 32 == 0 || row < BM) *(uint4*)(As + (buf * BM + row) * ROWB + vec * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int row = row0 + 32 * i;
      if (BN % 32 == 0 || row < BN) *(uint4*)(Bs + (buf * BN + row) * ROWB + vec * 16) = rb[i];
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
  if (kc_begin < kc_end) {
    gload(kc_begin);
    lstore(0);
  }
  __syncthreads();
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const bool more = kc + 1 < kc_end;
    if (more) gload(kc + 1);
    const unsigned char* Ab = As + (buf * BM + wm * WTM + r) * ROWB + q * 16;
    const unsigned char* Bb = Bs + (buf * BN + wn * WTN + r) * ROWB + q * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const uint4*)(Ab + i * 16 * ROWB + s * 64);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = *(const uint4*)(Bb + j * 16 * ROWB + s * 64);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], af[i], bfr[j]);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  if (p.splits > 1) {
    float* slab = p.slab + (size_t)blockIdx.z * p.M * p.NslabPitch;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int m = bm0 + wm * WTM + i * 16 + q * 4 + e;
        if (m < p.M) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            int n = bn0 + wn * WTN + j * 16 + r;
            slab[(size_t)m * p.NslabPitch + n] = acc[i][j][e];
          }
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int m = bm0 + wm * WTM + i * 16 + q * 4 + e;
        if (m < p.M) {
          size_t po = out_pixel_offset(p, m, py, px);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            int n = bn0 + wn * WTN + j * 16 + r;
            if (n < p.Cout) store_out<T>(p, po, n, acc[i][j][e]);
          }
        }
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, int P) {
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)P * p.M * p.Cout;
  if (idx >= total) return;
  int n = (int)(idx % p.Cout);
  long long t = idx / p.Cout;
  int m = (int)(t % p.M);
  int par = (int)(t / p.M);
  float s = 0.f;
  for (int k = 0; k < p.splits; ++k)
    s += p.slab[((size_t)(par * p.splits + k) * p.M + m) * p.NslabPitch + n];
  store_out<T>(p, out_pixel_offset(p, m, par >> 1, par & 1), n, s);
}

// ------------------------------------------------------------------------------------------------
struct GemmPlan {
  GemmParams p;
  int BM, BN, P;
  dim3 grid;
  size_t slab_bytes;
};

static int plan_gemm(const GanConvDesc* d, int op, GemmPlan* pl) {
  if (!d || !d->x.ptr || !d->y.ptr || !d->w) return GAN_E_ARG;
  if (d->dtype != GAN_F32 && d->dtype != GAN_BF16) return GAN_E_ARG;
  const int vec = d->dtype == GAN_F32 ? 4 : 8;
  const GanTensor &x = d->x, &y = d->y;
  if (x.c <= 0 || x.c % 8 || x.pitch % 8 || x.pitch < x.c || y.pitch < y.c || y.c <= 0) return GAN_E_SHAPE;
  if (x.n != y.n || d->w_rows < y.c) return GAN_E_SHAPE;
  int l2 = ilog2_exact(x.c / vec);
  if (l2 < 0) return GAN_E_SHAPE;
  GemmParams& p = pl->p;
  p.x = x.ptr; p.w = d->w; p.y = y.ptr; p.bias = d->bias; p.slab = (float*)d->workspace;
  p.Nimg = x.n; p.Hs = x.h; p.Ws = x.w; p.xpitch = x.pitch; p.Cin = x.c; p.log2_cvecs = l2;
  p.Wrows = d->w_rows; p.Ho = y.h; p.Wo = y.w; p.ypitch = y.pitch; p.Cout = y.c;
  p.act = d->act; p.slope = d->slope; p.out_f32 = d->y_f32 || d->dtype == GAN_F32;
  p.parity = 0; p.OS = 1; p.wy0 = p.wx0 = 0; p.wstep = 1; p.TWlog2 = 2; p.T = 16;
  const bool parity = (op == 1 && d->stride == 2) || op == 2;
  if (parity) {            // convT forward / stride-2 conv dgrad: 4 output-parity sub-GEMMs
    if (d->stride != 2 || y.h != 2 * x.h || y.w != 2 * x.w) return GAN_E_SHAPE;
    p.parity = 1; p.OS = 2; p.S = 1; p.TWlog2 = 1; p.T = 4; p.dstep = -1; p.wstep = 2;
    p.dy0 = p.dx0 = 0; p.Hg = x.h; p.Wg = x.w;
  } else if (op == 0 || op == 3) {   // conv forward (stride s) / convT dgrad (= conv s2 over dy)
    int s = (op == 3) ? 2 : d->stride;
    if (s != 1 && s != 2) return GAN_E_SHAPE;
    if (y.h != (x.h + 2 - 4) / s + 1 || y.w != (x.w + 2 - 4) / s + 1) return GAN_E_SHAPE;
    p.S = s; p.dy0 = p.dx0 = -1; p.dstep = 1; p.Hg = y.h; p.Wg = y.w;
  } else {                 // stride-1 conv dgrad: dx[i] = sum_k dy[i - k + 1] w[k]
    if (d->stride != 1 || y.h != x.h + 1 || y.w != x.w + 1) return GAN_E_SHAPE;
    p.S = 1; p.dy0 = p.dx0 = 1; p.dstep = -1; p.Hg = y.h; p.Wg = y.w;
  }
  long long M = (long long)x.n * p.Hg * p.Wg;
  if (M <= 0 || M > 0x7fffffffLL) return GAN_E_SHAPE;
  p.M = (int)M;
  const int bke = 128 / (d->dtype == GAN_F32 ? 4 : 2);
  long long K = (long long)p.T * x.c;
  if (K % bke) return GAN_E_SHAPE;
  p.kchunks = (int)(K / bke);
  const int P = parity ? 4 : 1;
  pl->P = P;
  int BN = y.c > 64 ? 128 : (y.c > 16 ? 64 : 16);
  int tilesN = (y.c + BN - 1) / BN;
  int BM = 128;
  if (((M + 127) / 128) * tilesN * P < 192) BM = 64;
  if (BM == 64 && M <= 32 && BN != 16) BM = 16;
  long long blocks = ((M + BM - 1) / BM) * tilesN * P;
  int splits = 1;
  if (blocks < 256) {
    splits = (int)((512 + blocks - 1) / blocks);
    int maxs = p.kchunks / 2; if (maxs < 1) maxs = 1;
    if (splits > maxs) splits = maxs;
    if (splits > 32) splits = 32;
  }
  p.splits = splits;
  p.NslabPitch = tilesN * BN;
  pl->BM = BM; pl->BN = BN;
  pl->grid = dim3((unsigned)((M + BM - 1) / BM), (unsigned)tilesN, (unsigned)(P * splits));
  pl->slab_bytes = splits > 1 ? (size_t)P * splits * (size_t)M * p.NslabPitch * sizeof(float) : 0;
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr size_t smem = 2 * (BM + BN) * 144;
  auto kern = conv_gemm_kernel<T, BM, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, pl.grid, dim3(256), smem, st, pl.p);
  GAN_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int launch_gemm(const GemmPlan& pl, hipStream_t st) {
  int rc;
  const int key = pl.BM * 1000 + pl.BN;
  switch (key) {
    case 128128: rc = launch_cfg<T, 128, 128, 2, 2>(pl, st); break;
    case 128064: rc = launch_cfg<T, 128, 64, 2, 2>(pl, st); break;
    case 128016: rc = launch_cfg<T, 128, 16, 4, 1>(pl, st); break;
    case 64128: rc = launch_cfg<T, 64, 128, 2, 2>(pl, st); break;
    case 64064: rc = launch_cfg<T, 64, 64, 2, 2>(pl, st); break;
    case 64016: rc = launch_cfg<T, 64, 16, 4, 1>(pl, st); break;
    case 16128: rc = launch_cfg<T, 16, 128, 1, 4>(pl, st); break;
    case 16064: rc = launch_cfg<T, 16, 64, 1, 4>(pl, st); break;
    default: return GAN_E_SHAPE;
  }
  if (rc) return rc;
  if (pl.p.splits > 1) {
    long long total = (long long)pl.P * pl.p.M * pl.p.Cout;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pl.p, pl.P);
    GAN_CHECK_LAUNCH();
  }
  return 0;
}

static int run_gemm(const GanConvDesc* d, int op, gan_stream_t stream) {
  GemmPlan pl;
  int rc = plan_gemm(d, op, &pl);
  if (rc) return rc;
  if (pl.slab_bytes > d->workspace_bytes || (pl.slab_bytes && !d->workspace)) return GAN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == GAN_F32 ? launch_gemm<float>(pl, st) : launch_gemm<bf16_t>(pl, st);
}

extern "C" {
int gan_conv2d_fwd(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 0, s); }
int gan_conv2d_dgrad(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 1, s); }
int gan_convT2d_fwd(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 2, s); }
int gan_convT2d_dgrad(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 3, s); }
int gan_conv_plan_info(const GanConvDesc* d, int op, int32_t* info /*[4]: BM, BN, splits, parities*/) {
  GemmPlan pl;
  GanConvDesc t = *d;
  if (!t.x.ptr) t.x.ptr = (void*)16;
  if (!t.y.ptr) t.y.ptr = (void*)16;
  if (!t.w) t.w = (void*)16;
  int rc = plan_gemm(&t, op, &pl);
  if (rc) return rc;
  info[0] = pl.BM; info[1] = pl.BN; info[2] = pl.p.splits; info[3] = pl.P;
  return 0;
}
size_t gan_conv_workspace_bytes(const GanConvDesc* d, int op) {
  GemmPlan pl;
  GanConvDesc t = *d;
  if (!t.x.ptr) t.x.ptr = (void*)16;   // planning only looks at shapes
  if (!t.y.ptr) t.y.ptr = (void*)16;
  if (!t.w) t.w = (void*)16;
  if (plan_gemm(&t, op, &pl)) return 0;
  return pl.slab_bytes;
}
}
