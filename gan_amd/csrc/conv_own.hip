// Column-owner kernel for the SMALLEST layers (<= 64 GEMM rows per parity: the 2x2 / 1x1 maps of the U-Net's bottleneck) - ONE launch
// instead of a split-K GEMM into fp32 slabs plus the slab reduce that finishes the layer (GanNormFuse).
//
// A workgroup owns 8 output channels of EVERY row: it needs no other workgroup's data to take the normalisation statistics, so the
// whole layer - gather GEMM, statistics (+ moving averages), normalise + dropout + activation forward, or the complete normalisation
// backward (dz, dgamma, dbeta, dy) - is one kernel with no slab round trip and no second launch.  Its weights are 8 rows of the NK copy
// per tap (Cin contiguous elements each): a contiguous stream; taps that never meet the map at this shape (12 of 16 for a 2x2 -> 1x1
// layer) are not read at all.  The reduction is split over the 16 waves of the workgroup (a wave serves one output parity; K steps of 32 dealt round-robin, all in flight at once), every wave
// multiplies v_mfma_f32_16x16x32 tiles with the weights as the "A" operand (rows 8..15 duplicate rows 0..7: the matrix pipe is idle
// anyway) straight from global memory - no LDS staging, the operands are read once - and the partial tiles are added in wave order
// by the finishing body (splitk_norm.h), i.e. in a fixed order.
//
// Bound: launch + dependent-load latency (weights 32-128 KB and activations <= 64 KB x live taps per workgroup).
#include "common.h"
#include "conv_params.h"
#include "splitk_norm.h"

static constexpr int OWN_NW = 16;          // waves per workgroup
#ifdef GAN_DIAG   // diagnostic build only (tools/diag_build.sh, tools/diag_own.py): per-block stamps, 100 MHz ticks
#define OWN_STAMP(i) do { if (p.diag && threadIdx.x == 0) p.diag[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OWN_STAMP(i) do {} while (0)
#endif

template <typename T, int MODE, int PP, int NRB>      // PP parities (1 | 4), NRB blocks of 16 rows per parity
__global__ __launch_bounds__(64 * OWN_NW) void conv_own_kernel(const GemmParams p, unsigned livemask) {
  constexpr int MPAD = 16 * NRB, NW = OWN_NW, ES = 2, UN = NRB == 1 ? 8 : 4, NWP = NW / PP;
  extern __shared__ __attribute__((aligned(16))) unsigned char own_smem[];
  float* part = (float*)own_smem;                               // [NW / PP][PP][MPAD][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int bx = blockIdx.x;
  const int twmask = (1 << p.TWlog2) - 1;
  OWN_STAMP(0);
  // a wave works for ONE parity (wave % PP) and takes every (NW / PP)-th K step of it: one flat list per wave, all of it in flight.
  // No gather table: a lane decodes its rows once (registers) and adds the tap; the live taps of the parity - those that meet the
  // map for at least one row, from the host (`livemask`, bit parity * taps + tap) - are packed 4 bits each into `taps`.
  const int par = PP == 4 ? (wave & 3) : 0, kw = PP == 4 ? (wave >> 2) : wave;
  const int dy0 = p.parity ? (par >> 1) : p.dy0, dx0 = p.parity ? (par & 1) : p.dx0;
  const int wy0 = p.parity ? 1 - (par >> 1) : p.wy0, wx0 = p.parity ? 1 - (par & 1) : p.wx0;
  unsigned long long taps = 0;
  int ntap = 0;
  {
    const unsigned bits = (livemask >> (par * p.T)) & ((1u << p.T) - 1u);
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (bits & (1u << t)) { taps |= (unsigned long long)t << (4 * ntap); ++ntap; }
  }
  int sy0[NRB], sx0[NRB], ibase[NRB];                           // source row / column of tap (0, 0) and the image's first pixel; rows past M: far outside
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const int m = rb * 16 + r;
    const unsigned t = fdiv((unsigned)m, p.divWg);
    const int gx = m - (int)t * p.Wg;
    const unsigned img = fdiv(t, p.divHg);
    const int gy = (int)t - (int)img * p.Hg;
    sy0[rb] = m < p.M ? gy * p.S + dy0 : -0x10000;
    sx0[rb] = gx * p.S + dx0;
    ibase[rb] = (int)img * p.Hs * p.Ws;
  }
  OWN_STAMP(1);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const int wtapbytes = p.Wrows * p.Cin * ES, pixbytes = p.xpitch * ES;
  const int log2kc = p.log2_cvecs - 2;                          // K steps of 32 elements (4 vectors of 8) per tap
  const int kcmask = (1 << log2kc) - 1;
  const int wrow = (bx * 8 + (r & 7)) * p.Cin * ES + q * 16;    // this lane's weight row (rows 8..15 of the MFMA tile repeat 0..7)
  f32x4 acc[NRB];
#pragma unroll
  for (int b = 0; b < NRB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto ldw = [&](int i, u32x4_t& a, u32x4_t (&b)[NRB]) {
    const int tap = (int)(taps >> (4 * (i >> log2kc))) & 15, kc = i & kcmask;
    const int ty = tap >> p.TWlog2, tx = tap & twmask;
    const int koff = kc * 64;
    a = __builtin_amdgcn_raw_buffer_load_b128(rw, ((wy0 + ty * p.wstep) * 4 + (wx0 + tx * p.wstep)) * wtapbytes + wrow + koff, 0, 0);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int sy = sy0[rb] + ty * p.dstep, sx = sx0[rb] + tx * p.dstep;
      const bool in = (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;     // else: zero padding (range check of the descriptor)
      const int off = in ? (ibase[rb] + sy * p.Ws + sx) * pixbytes + koff + q * 16 : (int)0x80000000;
      b[rb] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  };
  {
    const int n = ntap << log2kc;
    int i = kw;
    for (; i + (UN - 1) * NWP < n; i += UN * NWP) {             // UN K steps in flight per wave
      u32x4_t a[UN], b[UN][NRB];
#pragma unroll
      for (int u = 0; u < UN; ++u) ldw(i + u * NWP, a[u], b[u]);
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb] = mma16<T>(*(const uint4*)&a[u], *(const uint4*)&b[u][rb], acc[rb]);
    }
    for (; i < n; i += NWP) {
      u32x4_t a0, b0[NRB];
      ldw(i, a0, b0);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) acc[rb] = mma16<T>(*(const uint4*)&a0, *(const uint4*)&b0[rb], acc[rb]);
    }
  }
  OWN_STAMP(2);
  // accumulators: channel q * 4 + e of pixel r (q < 2: the 8 owned channels); wave = k * PP + parity -> part[k][parity][row][8]
  if (q < 2) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) *(f32x4*)(part + ((size_t)wave * MPAD + rb * 16 + r) * 8 + q * 4) = acc[rb];
  }
  __syncthreads();
  OWN_STAMP(3);
  if (tid >= 256) return;                                       // whole waves leave: the barriers below count the remaining ones
  splitk_norm_body<T, MODE, (PP * MPAD > 128 ? 2 : 1), false, 1>(p, PP, bx, 0, 1, part, NWP, MPAD);
  OWN_STAMP(4);
}

template <typename T, int MODE, int PP, int NRB>
static int own_launch_v(const GemmParams& p, unsigned live, hipStream_t st) {
  static bool attr_set = false;
  constexpr size_t smem = (size_t)OWN_NW * 16 * NRB * 8 * sizeof(float);
  auto kern = conv_own_kernel<T, MODE, PP, NRB>;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  GAN_LAUNCH(kern, dim3((unsigned)(p.Cout / 8)), dim3(64 * OWN_NW), smem, st, p, live);
  GAN_CHECK_LAUNCH();
  return 0;
}

template <typename T, int MODE>
static int own_launch_m(const GemmParams& p, int P, unsigned live, hipStream_t st) {
  const int nrb = p.M <= 16 ? 1 : 4;
  if (P == 1) return nrb == 1 ? own_launch_v<T, MODE, 1, 1>(p, live, st) : own_launch_v<T, MODE, 1, 4>(p, live, st);
  return nrb == 1 ? own_launch_v<T, MODE, 4, 1>(p, live, st) : own_launch_v<T, MODE, 4, 4>(p, live, st);
}

// (parity, tap) pairs that meet the map for at least one row: bit parity * taps + tap
static unsigned own_live_mask(const GemmParams& p, int P) {
  unsigned mask = 0;
  const int tw = 1 << p.TWlog2;
  for (int par = 0; par < P; ++par)
    for (int tap = 0; tap < p.T; ++tap) {
      const int dy0 = p.parity ? (par >> 1) : p.dy0, dx0 = p.parity ? (par & 1) : p.dx0;
      const int ty = tap / tw, tx = tap % tw;
      bool any_y = false, any_x = false;
      for (int gy = 0; gy < p.Hg; ++gy) any_y |= (unsigned)(gy * p.S + dy0 + ty * p.dstep) < (unsigned)p.Hs;
      for (int gx = 0; gx < p.Wg; ++gx) any_x |= (unsigned)(gx * p.S + dx0 + tx * p.dstep) < (unsigned)p.Ws;
      if (any_y && any_x) mask |= 1u << (par * p.T + tap);
    }
  return mask;
}

// conv_params.h: can this planned layer (p.skn set by plan_gemm) run on the column-owner kernel?  Only where a workgroup's operand
// stream is short: every workgroup reads ALL rows of the live taps (64 workgroups do, for 512 channels) and one CU takes in ~150 GB/s,
// so a layer whose 16 taps all meet the map (1.1 MB per workgroup at 64 rows x 8192) belongs to the split-K launch that spreads it
// over the chip (measured: -3 % on the Pix2Pix step with those layers here, CycleGAN batch 1 -16 %).
bool conv_own_eligible(const GemmParams& p, int P, int dtype) {
  const int max_rows = gan_opt("conv.own_max_rows");            // 0: never
  if (!p.skn || dtype == GAN_F32 || max_rows <= 0) return false;
  if (p.M > 64 || p.M > max_rows || p.Cin % 32 || p.Cout % 8 || P * p.T != 16) return false;
  if ((long long)P * (p.M / p.skn_groups) > 256) return false;  // rows per statistics group the finishing body holds (KR = 2)
  const int live = __builtin_popcount(own_live_mask(p, P));
  const long long bytes = (long long)live * p.Cin * 2 * (8 + p.M);      // per workgroup: 8 weight rows + the M gathered rows per live tap
  return bytes <= (long long)gan_opt("conv.own_max_kb") * 1024;
}

int conv_own_launch(const GemmParams& p, int P, int dtype, hipStream_t st) {
  if (!conv_own_eligible(p, P, dtype)) return GAN_E_SHAPE;
  const unsigned live = own_live_mask(p, P);
  if (dtype == GAN_F16) return p.skn == 1 ? own_launch_m<f16_t, 1>(p, P, live, st) : own_launch_m<f16_t, 2>(p, P, live, st);
  return p.skn == 1 ? own_launch_m<bf16_t, 1>(p, P, live, st) : own_launch_m<bf16_t, 2>(p, P, live, st);
}
