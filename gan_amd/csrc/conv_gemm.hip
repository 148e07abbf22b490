// Implicit-GEMM convolution for gfx950 (MI355X): Conv2D k4 (s1|s2) forward, Conv2DTranspose k4 s2
// forward and both input-gradients, all as ONE tap-gather GEMM kernel:
//
//   Y[m, n] = sum_{tap, c}  X[src(m, tap), c] * W[widx(tap)][n][c]
//
// rows m = (image, gy, gx) on a "GEMM grid"; src(m,tap) = (gy*S + dy(tap), gx*S + dx(tap)) with zero
// fill outside the source map; output pixel = (gy*OS + py, gx*OS + px).  A stride-2 transposed conv
// (and the dgrad of a stride-2 conv) is four output-parity sub-GEMMs with 4 taps each (K = 4*Cin),
// enumerated parity-fastest in blockIdx.x, so no zero-inserted input is ever materialised.
//
// Data path: the im2col gather goes global -> LDS directly (buffer_load_dwordx4 ... lds: per-lane SOURCE offset =
// the gather, read from a per-block table in LDS; lane-linear LDS destination), no VGPR staging and no ds_write.
// An LDS tile row is 128 (or 64) bytes of K, 16-byte slots XOR-swizzled through the source address so the
// ds_read_b128 fragment reads are bank-conflict free.  Padding taps and out-of-range rows carry the offset
// 0x80000000, which the buffer descriptor's range check turns into zeros.  2 or 3 LDS stages (counted vmcnt, raw
// s_barrier, inline-asm fragment reads); 4 or 8 waves; block tile BM x BN chosen per layer (256x256 ... 16x128, see
// plan_gemm); bf16: v_mfma_f32_16x16x32_bf16, fp32 (parity path): exact v_mfma_f32_16x16x4_f32, with the weights
// as the MFMA "A" operand so that a lane's accumulators are 4 consecutive channels of one pixel.  The epilogue
// stages the tile in LDS and writes whole 16-byte channel vectors (bias + activation fused, optional
// normalisation-statistics partials).  Small problems split K across blockIdx.z into fp32 slabs + a reduce kernel.
// Block ids are remapped so that each XCD (own L2) works on a contiguous range of M tiles and all their N tiles.
// The <= 8-channel layers do not come here: thin.hip.  DESIGN.md section 4 has the measured LDS-bandwidth model.
#include "common.h"
#include <type_traits>
#include "conv_params.h"
#include "splitk_norm.h"

#ifdef GAN_DIAG   // diagnostic build only (tools/diag_build.sh): in-kernel wall-clock stamps per block, 100 MHz ticks
static unsigned long long* g_diag = nullptr;
extern "C" void gan_diag_set(void* ptr) { g_diag = (unsigned long long*)ptr; }
#define DIAG_STAMP(i) do { if (p.diag && tid == 0) p.diag[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per-segment cycle sums of the ping-pong loop (wave 0 of the block): SEG_T0 at a segment's start, SEG_ADD(k) at its end
#define SEG_DECL unsigned long long seg_t = 0, seg_sum[6] = {0, 0, 0, 0, 0, 0}
#define SEG_T0 do { if (p.diag) { __builtin_amdgcn_sched_barrier(0); seg_t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define SEG_ADD(k) do { if (p.diag) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); seg_sum[k] += n_ - seg_t; seg_t = n_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define SEG_STORE do { if (p.diag && tid == 0) for (int k_ = 0; k_ < 6; ++k_) p.diag[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 16 + 8 + k_] = seg_sum[k_]; } while (0)
#else
#define DIAG_STAMP(i) do {} while (0)
#define SEG_DECL
#define SEG_T0 do {} while (0)
#define SEG_ADD(k) do {} while (0)
#define SEG_STORE do {} while (0)
#endif

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) { acc = mma16<f16_t>(a, b, acc); }
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

// Fused backward epilogue (GanBwdFuse): bf_mode = 1 norm+LeakyReLU, 2 norm+ReLU, 3 norm+ReLU+dropout mask, 4 LeakyReLU on the
// saved activation.  One 16-byte channel vector: g = da (+ add) in, dz out; s1 += dz, s2 += dz*xhat (norm modes).
// The arithmetic is reduce_partial_kernel's (norm.hip), so that fused and unfused paths agree to rounding of dz.
template <typename T, int MODE, int VEC>
__device__ __forceinline__ uint4 bwd_fuse_vec(const GemmParams& p, uint4 raw, const uint4& refv, const uint4& addv, const uint2& maskv,
                                              const float* mu, const float* rs, const float* ga, const float* be, float* s1, float* s2) {
  float g[VEC], rf[VEC];
  unpack16<T>(raw, g);
  if (p.bf_add) {
    float a2[VEC];
    unpack16<T>(addv, a2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] += a2[e];
  }
  unpack16<T>(refv, rf);
  if constexpr (MODE == 4) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = rf[e] > 0.f ? g[e] : g[e] * p.bf_slope;
  } else {
    float mk[VEC];
    if constexpr (MODE == 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mk[e] = 2.f * (float)((maskv.x >> (8 * e)) & 0xff);
        if constexpr (VEC == 8) mk[4 + e] = 2.f * (float)((maskv.y >> (8 * e)) & 0xff);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (rf[e] - mu[e]) * rs[e];
      const float z = fmaf(ga[e], xh, be[e]);
      const float zd = MODE == 3 ? z * mk[e] : z;
      float d = MODE == 3 ? g[e] * mk[e] : g[e];
      if (MODE == 1) d = zd > 0.f ? d : d * p.bf_slope;
      else d = zd > 0.f ? d : 0.f;
      s1[e] += d; s2[e] = fmaf(d, xh, s2[e]);
      g[e] = d;
    }
  }
  return pack16<T>(g);
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Epilogue shared by the GEMM kernels.  The MFMAs ran with the weights as the "A" operand, so the accumulators hold
// the transposed tile: acc[i][j][e] = Y[pixel row i*16 + (lane & 15)][channel j*16 + (lane >> 4)*4 + e] - four
// consecutive channels of one pixel per lane, which pack into one 8-byte (bf16) / 16-byte (fp32) write.  SMEMB bytes of
// LDS at `smem` are free for staging (the caller has passed a block barrier after its last LDS read).
// PARCOLS (conv_par_kernel): the tile's BN = 4 x PCOLS columns are (output parity, channel) pairs - column block v / PCOLS
// goes to the output pixel of parity v / PCOLS, channel bn0 + v % PCOLS; statistics partials come out as one chunk per parity.
// BF = false: the fused-backward-epilogue paths (GanBwdFuse) are not compiled in.  They cost registers - the 256x128 ping-pong kernel
// is 238 VGPRs with them (U = 8 rows of reference / skip / mask loads in flight), 175 with U = 2, 132 without; 256x256: 256 / 216 - and a
// 512-thread block puts TWO waves on every SIMD: at 238-256 registers nothing else fits beside them (512 per SIMD lane), at <= 216 an
// 80-register streaming wave does.  The kernels are therefore instantiated twice and a launch without bwd_fuse takes the lean one
// (round 4, tools/probe_overlap.py: overlap of a 256x128 GEMM stream with a normalisation stream 0.25 -> 0.45).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int SMEMB, bool PARCOLS = false, bool COH = false, bool BF = true>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[BM / WAVES_M / 16][BN / WAVES_N / 16],
                                              unsigned char* smem, int bm0, int bn0, int par, int P, int split) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int PCOLS = PARCOLS ? BN / 4 : BN;               // channels per parity block of columns
  constexpr int PVECS = PCOLS / VEC;
  constexpr int NW = WAVES_M * WAVES_N, NTHREADS = 64 * NW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, MT = WTM / 16, NT = WTN / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 15, q = lane >> 4;
  const int py = par >> 1, px = par & 1;
  if (p.splits > 1) {
    float* slab = p.slab + (size_t)(par * p.splits + split) * p.M * p.NslabPitch;
    const CohBuf cb = coh_buf(p.slab);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = bm0 + wm * WTM + i * 16 + r;
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = bn0 + wn * WTN + j * 16 + q * 4;
          // (splitk_norm_kernel reads 8-channel slices of all rows: its slabs are laid out [slice][row][8] so that a slice is contiguous)
          if (p.skn) st_f4<COH>(cb, slab + ((size_t)(col >> 3) * p.M + m) * 8 + (col & 7), acc[i][j]);
          else st_f4<COH>(cb, slab + (size_t)m * p.NslabPitch + col, acc[i][j]);
        }
      }
    }
  } else if (p.vec_store) {
    // stage the tile (bias + activation applied) as T in LDS, then coalesced 16-byte row stores.  The whole
    // stage/table area is free now, so as many 16-row groups per wave-row as fit are staged per pass (one
    // pass for bf16 tiles): IPP = largest divisor of MT whose rows fit.
    constexpr int CS = BN * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
    constexpr int VPR = BN / VEC;
    constexpr int MAXG = SMEMB / (CS * WAVES_M * 16);          // 16-row groups per wave-row that fit
    constexpr int IPP = MAXG >= MT ? MT : (MAXG >= MT / 2 && MT % 2 == 0 ? MT / 2 : (MAXG >= MT / 4 && MT % 4 == 0 ? MT / 4 : 1));
    static_assert(MAXG >= 1, "staging tile does not fit");
    unsigned char* Cs = smem;
    float bv[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = bn0 + (wn * WTN + j * 16 + q * 4 + e) % PCOLS;
        bv[j][e] = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
      }
    // fused normalisation statistics: per-column (sum, sum of squares) of the STORED values of this tile
    // a thread sums one 16-byte column group (VEC channels) over a slice of the staged rows
    constexpr int CG = BN / VEC, SL = NTHREADS / CG;          // column groups; row slices (NTHREADS >= BN)
    const int scg = tid % CG, sslice = tid / CG;
    static_assert(NTHREADS % VPR == 0 && CG == VPR, "a thread keeps one column group in the store loop");
    float ssum[VEC], ssq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ssum[e] = ssq[e] = 0.f;
    float cmu[VEC], crs[VEC], cga[VEC], cbe[VEC];              // fused backward epilogue: this thread's per-channel constants
    if (BF && p.bf_mode >= 1 && p.bf_mode <= 3 && bn0 + (scg % PVECS) * VEC < p.bf_cols) {
      const int grp = p.stats_tpg ? (bm0 / BM) / p.stats_tpg : 0, c0 = bn0 + (scg % PVECS) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        cmu[e] = p.bf_mean[grp * p.bf_cols + c0 + e]; crs[e] = p.bf_rstd[grp * p.bf_cols + c0 + e];
        cga[e] = p.bf_gamma[c0 + e]; cbe[e] = p.bf_beta[c0 + e];
      }
    }
#pragma unroll
    for (int ip = 0; ip < MT / IPP; ++ip) {
      if (ip) __syncthreads();
      // activation resolved once per pass, not per element (the run-time select chain over 64 accumulators per
      // lane was ~3 us of the epilogue)
      auto stage = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_c<ACT>(acc[ip * IPP + ii][j][e] + bv[j][e], p.slope);
            T* dst = (T*)(Cs + ((wm * IPP + ii) * 16 + r) * CS) + wn * WTN + j * 16 + q * 4;
            if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            else *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
          }
      };
      if (p.act == GAN_ACT_NONE) stage(std::integral_constant<int, GAN_ACT_NONE>{});
      else if (p.act == GAN_ACT_LRELU) stage(std::integral_constant<int, GAN_ACT_LRELU>{});
      else if (p.act == GAN_ACT_RELU) stage(std::integral_constant<int, GAN_ACT_RELU>{});
      else stage(std::integral_constant<int, GAN_ACT_TANH>{});
      __syncthreads();
      auto store_rows = [&](auto modec) {
        constexpr int MODE = decltype(modec)::value;
        constexpr int TOTAL = WAVES_M * IPP * 16 * VPR;
        if constexpr (MODE == 0 || TOTAL % NTHREADS != 0) {
          for (int idx = tid; idx < TOTAL; idx += NTHREADS) {
            const int sr = idx / VPR, v = idx % VPR;             // v == scg for every idx of a thread (NTHREADS % VPR == 0)
            const int g16 = sr >> 4;
            const int m = bm0 + (g16 / IPP) * WTM + (ip * IPP + g16 % IPP) * 16 + (sr & 15), n = bn0 + (v % PVECS) * VEC;
            if (m < p.M && n < p.Cout) {
              uint4 raw = *(const uint4*)(Cs + sr * CS + v * 16);
              const size_t pix = PARCOLS ? out_pixel_index(p, m, (v / PVECS) >> 1, (v / PVECS) & 1) : out_pixel_index(p, m, py, px);
              if constexpr (MODE != 0) {
                if (n < p.bf_cols) {
                  const uint4 rv = *(const uint4*)((const T*)p.bf_ref + pix * (size_t)p.bf_refpitch + n);
                  uint4 av = make_uint4(0, 0, 0, 0); uint2 mv = make_uint2(0, 0);
                  if (p.bf_add) av = *(const uint4*)((const T*)p.bf_add + pix * (size_t)p.bf_addpitch + n);
                  if constexpr (MODE == 3) {
                    if constexpr (VEC == 8) mv = *(const uint2*)(p.bf_mask + pix * (size_t)p.bf_maskpitch + n);
                    else mv.x = *(const uint32_t*)(p.bf_mask + pix * (size_t)p.bf_maskpitch + n);
                  }
                  raw = bwd_fuse_vec<T, MODE, VEC>(p, raw, rv, av, mv, cmu, crs, cga, cbe, ssum, ssq);
                }
              }
              *(uint4*)((T*)p.y + pix * (size_t)p.ypitch + n) = raw;
            }
          }
        } else {
          // fused backward epilogue: the reference / skip / mask vectors of U rows are requested together before any of them
          // is used (one dependent global load per iteration made this loop pure latency: +9..+40 us per launch)
          constexpr int ITERS = TOTAL / NTHREADS, U = (BN == 128 && !PARCOLS && ITERS % 4 == 0) ? 4 : (ITERS % 2 == 0 ? 2 : 1);      // (8 / 4 rows at a time on the 128-column tiles until round 4: +63 registers)
          const int n = bn0 + (scg % PVECS) * VEC;
          const int cpy = PARCOLS ? (scg / PVECS) >> 1 : py, cpx = PARCOLS ? (scg / PVECS) & 1 : px;
          const bool colok = n < p.Cout, fuse = n < p.bf_cols;
          for (int it0 = 0; it0 < ITERS; it0 += U) {
            uint4 rv[U], av[U]; uint2 mv[U]; size_t pix[U]; bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int sr = (tid + (it0 + u) * NTHREADS) / VPR, g16 = sr >> 4;
              const int m = bm0 + (g16 / IPP) * WTM + (ip * IPP + g16 % IPP) * 16 + (sr & 15);
              ok[u] = m < p.M && colok;
              pix[u] = ok[u] ? out_pixel_index(p, m, cpy, cpx) : 0;
              rv[u] = av[u] = make_uint4(0, 0, 0, 0); mv[u] = make_uint2(0, 0);
              if (ok[u] && fuse) {
                rv[u] = *(const uint4*)((const T*)p.bf_ref + pix[u] * (size_t)p.bf_refpitch + n);
                if (p.bf_add) av[u] = *(const uint4*)((const T*)p.bf_add + pix[u] * (size_t)p.bf_addpitch + n);
                if constexpr (MODE == 3) {
                  if constexpr (VEC == 8) mv[u] = *(const uint2*)(p.bf_mask + pix[u] * (size_t)p.bf_maskpitch + n);
                  else mv[u].x = *(const uint32_t*)(p.bf_mask + pix[u] * (size_t)p.bf_maskpitch + n);
                }
              }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (!ok[u]) continue;
              const int sr = (tid + (it0 + u) * NTHREADS) / VPR;
              uint4 raw = *(const uint4*)(Cs + sr * CS + scg * 16);
              if (fuse) raw = bwd_fuse_vec<T, MODE, VEC>(p, raw, rv[u], av[u], mv[u], cmu, crs, cga, cbe, ssum, ssq);
              *(uint4*)((T*)p.y + pix[u] * (size_t)p.ypitch + n) = raw;
            }
          }
        }
      };
      if constexpr (BF) {
        switch (p.bf_mode) {
          case 1: store_rows(std::integral_constant<int, 1>{}); break;
          case 2: store_rows(std::integral_constant<int, 2>{}); break;
          case 3: store_rows(std::integral_constant<int, 3>{}); break;
          case 4: store_rows(std::integral_constant<int, 4>{}); break;
          default: store_rows(std::integral_constant<int, 0>{}); break;
        }
      } else {
        store_rows(std::integral_constant<int, 0>{});
      }
      if (p.stats && !p.bf_mode) {
        for (int sr = sslice; sr < WAVES_M * IPP * 16; sr += SL) {
          const int g16 = sr >> 4;
          const int m = bm0 + (g16 / IPP) * WTM + (ip * IPP + g16 % IPP) * 16 + (sr & 15);
          if (m < p.M) {
            float v[VEC];
            unpack16<T>(*(const uint4*)(Cs + sr * CS + scg * 16), v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
          }
        }
      }
    }
    if (p.stats) {
      __syncthreads();
      float* red = (float*)smem;                               // [SL][BN][2]
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        red[(sslice * BN + scg * VEC + e) * 2] = ssum[e]; red[(sslice * BN + scg * VEC + e) * 2 + 1] = ssq[e];
      }
      __syncthreads();
      if (tid < BN && bn0 + tid % PCOLS < p.stats_C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < SL; ++k) { a += red[(k * BN + tid) * 2]; b += red[(k * BN + tid) * 2 + 1]; }
        // chunk index: (tile within its group) * P + parity; groups are whole numbers of M tiles
        const int tm = bm0 / BM, grp = tm / p.stats_tpg, chunk = (tm % p.stats_tpg) * P + (PARCOLS ? tid / PCOLS : par);
        float* dst = p.stats + (((size_t)grp * p.stats_tpg * P + chunk) * p.stats_C + bn0 + tid % PCOLS) * 2;
        dst[0] = a; dst[1] = b;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = bm0 + wm * WTM + i * 16 + r;
      if (m < p.M) {
        const size_t po = out_pixel_offset(p, m, py, px);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = bn0 + wn * WTN + j * 16 + q * 4 + e;
            if (n < p.Cout) store_out<T>(p, po, n, acc[i][j][e]);
          }
      }
    }
  }
}

// BKB: bytes of K per LDS row / pipeline step (128 or 64); NS: LDS stages (NS-1 tiles in flight)
// One block tile of the tap-gather GEMM: logical block index bx (what blockIdx.x is for the stand-alone kernel) and K split `split`.
// COH: the A operand was written earlier in the SAME launch by other workgroups (conv_stack_kernel): its LDS-DMA loads go around the
// L2 (sc0 sc1) and the slabs are written through.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NS, bool COH>
__device__ __forceinline__ void conv_gemm_body(const GemmParams& p, unsigned char* smem, int bx, int split) {
#if defined(__HIP_DEVICE_COMPILE__)   // body uses device-only buffer-descriptor builtins; the host pass only needs the stub
  constexpr int VEC = VecOf<T>::N;
  constexpr int NW = WAVES_M * WAVES_N, NTHREADS = 64 * NW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, MT = WTM / 16, NT = WTN / 16;
  constexpr int SLOTS = BKB / 16, RPI = 1024 / BKB;      // 16-B slots per row; rows per 1-KiB wave-instruction
  constexpr int AINS = BM / RPI, BINS = BN / RPI;        // 1-KiB pieces per tile
  constexpr int AI = (AINS + NW - 1) / NW, BI = (BINS + NW - 1) / NW;  // per wave
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int KSTEPS = BKB / 64;                       // 64-byte MFMA k-steps per row
  // counted vmcnt needs the same number of pieces from every wave: a tile with fewer A pieces than waves (16 rows: 2 pieces, 4 waves)
  // lets the other waves fetch the same pieces again (identical bytes to the same LDS rows)
  constexpr bool AWRAP = NS > 2 && AINS < NW && NW % AINS == 0;
  static_assert(NS == 2 || ((AINS % NW == 0 || AWRAP) && BINS % NW == 0), "counted vmcnt needs equal pieces per wave");
  // slot swizzle so that the 16 rows of a ds_read_b128 fragment read hit 16 distinct 16-B bank groups
  auto fsw = [](int row) { return BKB == 128 ? (row & 7) : ((0 - (row >> 2)) & 3); };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 15, q = lane >> 4;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
  // run of logical tiles, N tiles fastest, so an A tile is fetched into ONE L2 and re-used by its N tiles.
  // blockIdx.x enumerates (M tile, N tile, parity) with the parity fastest: the 4 parity sub-GEMMs of a tile
  // gather the same source rows, so they run next to each other on one XCD and share them in L2.
  const int P = p.parity ? 4 : 1;
  int bid = bx;
  const int nb = p.tilesM * p.tilesN * P;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int par = bid % P;
  bid /= P;
  const int bm0 = (bid / p.tilesN) * BM, bn0 = (bid % p.tilesN) * BN;
  const int py = par >> 1, px = par & 1;
  int dy0 = p.dy0, dx0 = p.dx0, wy0 = p.wy0, wx0 = p.wx0;
  if (p.parity) { dy0 = py; dx0 = px; wy0 = 1 - py; wx0 = 1 - px; }

  // this lane's slot in every 8-row x 128-B LDS piece, and the K chunk it must fetch for it (swizzle)
  const int lrow = lane / SLOTS, slot = lane % SLOTS;
  const int chunk = slot ^ fsw(lrow);
  const int cmask = (1 << p.log2_cvecs) - 1;
  const int twmask = (1 << p.TWlog2) - 1;

  // Gather table, built once per block: byte offset of the source pixel for every (tap, tile row), or
  // 0x80000000 for zero padding / rows past M.  Loads go through buffer descriptors, whose range check turns
  // such offsets into zeros, so the K loop spends one LDS read + one add per 1-KiB piece on addressing.
  int* tbl = (int*)(smem + NS * STAGE);       // [T][BM]
  for (int e = tid; e < p.T * BM; e += NTHREADS) {
    const int row = e % BM, tap = e / BM;
    const int m = bm0 + row;
    int off = (int)0x80000000;
    if (m < p.M) {
      const unsigned t = fdiv((unsigned)m, p.divWg);      // (multiply-high division: four runtime integer divisions per entry were ~1 us per block)
      const int gx = m - (int)t * p.Wg;
      const unsigned img = fdiv(t, p.divHg);
      const int gy = (int)t - (int)img * p.Hg;
      const int ty = tap >> p.TWlog2, tx = tap & twmask;
      const int sy = gy * p.S + dy0 + ty * p.dstep, sx = gx * p.S + dx0 + tx * p.dstep;
      if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
        off = (int)((((size_t)(img * p.Hs + sy) * p.Ws + sx) * (size_t)p.xpitch) * sizeof(T));
    }
    tbl[e] = off;
  }
  int nbo[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = bn0 + (wave + NW * i) * RPI + lrow;
    nbo[i] = n < p.Wrows ? (int)((size_t)n * p.Cin * sizeof(T)) : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const int wtapbytes = p.Wrows * p.Cin * (int)sizeof(T);
  __syncthreads();

  // prep(kc): per-piece byte offsets of K step kc (gather-table lookup + add); fire(idx, stage): issue piece idx
  // (A pieces first, then B) of the prepared step into `stage`.  Split so the 1-KiB LDS-DMA issues can be
  // spread between the MFMAs of the step being multiplied instead of stalling the wave in one burst.
  constexpr int PT = AI + BI;                           // pieces per wave per tile
  int va[AI], vb[BI];
  auto prep = [&](int kc) {
    // K order: channel-chunk major, tap minor (when a chunk lies inside one tap): the 16 (or 4) taps of one
    // channel chunk re-read the same input lines shifted by a pixel, so consecutive K steps hit in L2.
    int tap, coffB;
    if ((1 << p.log2_cvecs) >= SLOTS) {
      tap = kc & (p.T - 1);
      coffB = ((kc >> p.log2T) * SLOTS + chunk) * 16;
    } else {
      const int kvec = kc * SLOTS + chunk;
      tap = kvec >> p.log2_cvecs;
      coffB = (kvec & cmask) * 16;
    }
    const int* trow = tbl + tap * BM + lrow;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int ia = AWRAP ? (wave + NW * i) % AINS : wave + NW * i;
      va[i] = (ia < AINS) ? trow[ia * RPI] + coffB : (int)0x80000000;
    }
    const int ty = tap >> p.TWlog2, tx = tap & twmask;
    const int wofs = ((wy0 + ty * p.wstep) * 4 + (wx0 + tx * p.wstep)) * wtapbytes + coffB;
#pragma unroll
    for (int i = 0; i < BI; ++i) vb[i] = nbo[i] + wofs;
  };
  auto fire = [&](int idx, int stage) {
    unsigned char* As = smem + stage * STAGE;
    if (idx < AI) {
      const int ia = AWRAP ? (wave + NW * idx) % AINS : wave + NW * idx;
      if (ia < AINS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(As + ia * 1024), 16, va[idx], 0, 0, COH ? 0x11 : 0);
    } else {
      const int ib = wave + NW * (idx - AI);
      if (ib < BINS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(As + BM * BKB + ib * 1024), 16,
                                                 vb[idx - AI], 0, 0, 0);
    }
  };
  auto issue = [&](int kc, int stage) {
    prep(kc);
#pragma unroll
    for (int idx = 0; idx < PT; ++idx) fire(idx, stage);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row R = base + r, slot ((s*4+q) ^ (R&7)) with R&7 == r&7 (bases are multiples of 16)
  // Fragment reads are issued as inline-asm ds_read_b128: hipcc orders every compiler-visible LDS read behind
  // ALL in-flight LDS-DMA (it inserts s_waitcnt vmcnt(0)), which would drain the multi-stage pipeline each
  // step.  Ordering is ours instead: counted vmcnt + barrier before a stage is read (K loop below), and
  // counted lgkmcnt + sched_barrier before the MFMAs that consume the fragments.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned sw0 = ((q ^ fsw(r)) << 4), sw1 = (((4 + q) ^ fsw(r)) << 4);
  const unsigned a_off = (wm * WTM + r) * BKB, b_off = BM * BKB + (wn * WTN + r) * BKB;
  auto compute = [&](int stage, bool refill, int stage_i) {
    const unsigned sbase = lds_base + stage * STAGE;
    uint4 af[KSTEPS][MT], bfr[KSTEPS][NT];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const unsigned sw = s ? sw1 : sw0;
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[s][i]) : "v"(sbase + a_off + i * 16 * BKB + sw));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[s][j]) : "v"(sbase + b_off + j * 16 * BKB + sw));
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + 1 < KSTEPS) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], bfr[s][j], af[s][i]);   // D^T: lane = pixel, 4 channels
        // spread this wave's PT piece issues over the KSTEPS*MT row groups of MFMAs
        constexpr int G = KSTEPS * MT;
        const int g = s * MT + i;
        if (refill) {
#pragma unroll
          for (int idx = 0; idx < PT; ++idx)
            if (idx >= g * PT / G && idx < (g + 1) * PT / G) fire(idx, stage_i);
        }
      }
    }
  };

  const int kc_begin = (int)((long long)p.kchunks * split / p.splits);
  const int kc_end = (int)((long long)p.kchunks * (split + 1) / p.splits);
  const int nk = kc_end - kc_begin;
  // NS-1 tiles stay in flight.  At step i: wait until all but the newest (NS-2) tiles of THIS wave have landed
  // (counted vmcnt: LDS-DMA pieces retire in issue order), barrier (everyone's pieces of tile i are in LDS and
  // everyone has finished reading tile i-1), refill the stage tile i-1 occupied, multiply tile i.
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(kc_begin + s, s);
  int st_c = 0, st_i = NS - 1;                          // stage of tile i / of tile i+NS-1
  for (int i = 0; i < nk; ++i) {
    const int pending = nk - 1 - i < NS - 2 ? nk - 1 - i : NS - 2;
    if (NS >= 5 && pending == 3) wait_vmcnt<3 * PT>();
    else if (NS >= 4 && pending == 2) wait_vmcnt<2 * PT>();
    else if (NS >= 3 && pending == 1) wait_vmcnt<PT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // (spreading the piece issues between the MFMAs instead of this burst measured 5 % slower)
    if (i + NS - 1 < nk) issue(kc_begin + i + NS - 1, st_i);
    compute(st_c, false, st_i);
    st_c = st_c + 1 == NS ? 0 : st_c + 1;
    st_i = st_i + 1 == NS ? 0 : st_i + 1;
  }
  __syncthreads();

  gemm_epilogue<T, BM, BN, WAVES_M, WAVES_N, NS * STAGE + 16 * BM * 4, false, COH>(p, acc, smem, bm0, bn0, par, P, split);
#endif
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NS>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  conv_gemm_body<T, BM, BN, WAVES_M, WAVES_N, BKB, NS, false>(p, smem, (int)blockIdx.x, (int)blockIdx.z);
}

// ------------------------------------------------------------------------------------------------
// "Ping-pong" kernel for the 256-row tiles (256x256 and 256x128; 8 waves as 2 (M) x 4 (N), one block per CU).
//
// A K tile (128 bytes of K per row) is multiplied in PH phases of 16 MFMA steps per wave (a 64-row x 32-column piece of
// the wave's accumulator tile).  Each phase is a LOAD segment (fragment ds_reads for this phase + PP LDS-DMA pieces of a
// tile DP phases ahead + a counted vmcnt) and a MATH segment (the MFMAs), with a block barrier after each.  Waves 4..7
// (the second wave on every SIMD) run one barrier behind waves 0..3, so on every SIMD one wave is in its MATH segment
// while its partner is in its LOAD segment: the matrix pipe is fed without any wave having to overlap its own loads
// with its own math.  LDS-DMA pieces stay in flight across barriers (counted vmcnt, never 0 inside the loop).
//
// Ordering rules (g = phase number; group 0 = waves 0..3, group 1 = waves 4..7; group 0 runs LOAD(g) in barrier interval
// 2g and MATH(g) in 2g+1, group 1 in 2g+1 and 2g+2):
//   RAW: a piece first read in phase r must be covered by the vmcnt at the end of LOAD(r-1) of EVERY wave: after issuing
//        the pieces of phase r-1+DP at most VMW pieces may be outstanding.
//   WAR: a piece slot is re-filled NB K tiles later; its LDS-DMA is issued in LOAD(g') with g' >= (last phase whose
//        LOAD segment reads the old contents) + 2, because those reads are only known complete at the lgkmcnt(0) that
//        opens the reader's MATH segment.  The B fragments of the first column half are kept in registers for the
//        tile's last phase (256x256) so that no phase re-reads a half that is about to be re-staged.
// Piece slots of a K tile per wave, in staging order, and the phase that first reads them:
//   256x256 (PH 4, PP 2, DP 5, VMW 6, NB 2): A0 A0 | B0 B0 | B1 B1 | A1 A1   first read in phase 0 0 0 0 1 1 2 2
//   256x128 (PH 2, PP 3, DP 4, VMW 8, NB 3): A0 A0 B | B A1 A1               first read in phase 0 0 0 0 1 1
// (Issuing the pieces between the MFMAs of the MATH segment instead measured 25 % slower.)
// (A0/A1: the first/second 64 rows of each wave-row's 128 rows; B0/B1: the first/second 32 columns of each wave's 64.)
// K tiles past the end of the block's K range are staged as out-of-range pieces (zeros, no memory traffic), which keeps
// the vmcnt arithmetic uniform to the last phase.
//
// The im2col gather needs no table: a lane serves the same 4 tile rows for every K tile, so it keeps, per row, the byte
// offset of the row's window origin and a 16-bit mask of the taps that fall inside the source map; a piece's source
// offset is origin + (wave-uniform tap/channel offset) or 0x80000000 (range-checked to zeros).
template <int IMM> __device__ __forceinline__ void lds_read128(uint4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

template <typename T, int BN, bool BF>
__global__ __launch_bounds__(512) void conv_gemm_pp_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, WAVES_M = 2, WAVES_N = 4, BKB = 128;
  constexpr int WTN = BN / 4, MT = 8, NT = WTN / 16;        // wave tile 128 x (64 | 32)
  constexpr int PH = BN == 256 ? 4 : 2;
  constexpr int NB = BN == 256 ? 2 : 3;
  constexpr int PP = BN == 256 ? 2 : 3;
  constexpr int DP = BN == 256 ? 5 : 4;
  constexpr int VMW = BN == 256 ? 6 : 8;
  constexpr int QS = PP * PH;                               // piece slots per wave per K tile
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int ES = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  DIAG_STAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;

  // tile order: as conv_gemm_kernel (XCD-contiguous, parity fastest)
  const int P = p.parity ? 4 : 1;
  int bid = blockIdx.x;
  const int nb = p.tilesM * p.tilesN * P;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int par = bid % P;
  bid /= P;
  const int bm0 = (bid / p.tilesN) * BM, bn0 = (bid % p.tilesN) * BN;
  const int split = blockIdx.z;
  const int py = par >> 1, px = par & 1;
  int dy0 = p.dy0, dx0 = p.dx0, wy0 = p.wy0, wx0 = p.wx0;
  if (p.parity) { dy0 = py; dx0 = px; wy0 = 1 - py; wx0 = 1 - px; }
  const int twmask = (1 << p.TWlog2) - 1, TW = 1 << p.TWlog2;

  // ---- per-lane gather state ---------------------------------------------------------------------------
  const int lrow = lane >> 3, slot = lane & 7;
  const int chunkB = ((slot ^ lrow) << 4);                  // swizzled 16-byte K chunk this lane fetches for its LDS slot
  int a_org[2][2];                                          // [half][k]: byte offset of the row's window origin (+ chunk)
  unsigned a_msk[2][2];                                     // taps inside the source map
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = bm0 + (k * 16 + 8 * h + wave) * 8 + lrow;
      unsigned msk = 0;
      int org = 0;
      if (m < p.M) {
        const unsigned t = fdiv((unsigned)m, p.divWg);
        const int gx = m - (int)t * p.Wg;
        const unsigned img = fdiv(t, p.divHg);
        const int gy = (int)t - (int)img * p.Hg;
        const int sy0 = gy * p.S + dy0, sx0 = gx * p.S + dx0;
        unsigned vx = 0;
        for (int tx = 0; tx < TW; ++tx) vx |= ((unsigned)(sx0 + tx * p.dstep) < (unsigned)p.Ws ? 1u : 0u) << tx;
        for (int ty = 0; ty < TW; ++ty)
          if ((unsigned)(sy0 + ty * p.dstep) < (unsigned)p.Hs) msk |= vx << (ty << p.TWlog2);
        org = (int)((((long long)((int)img * p.Hs + sy0) * p.Ws + sx0) * (long long)p.xpitch) * ES) + chunkB;
      }
      a_org[h][k] = org; a_msk[h][k] = msk;
    }
  constexpr int BK_ = BN == 256 ? 4 : 2;                    // B pieces per wave per K tile
  int b_org[BK_];
#pragma unroll
  for (int i = 0; i < BK_; ++i) {
    // 256: i = h*2 + k -> piece (wave>>2)*8 + 16k + 4h + (wave&3);  128: i = k -> piece wave + 8k
    const int pb = BN == 256 ? ((wave >> 2) * 8 + 16 * (i & 1) + 4 * (i >> 1) + (wave & 3)) : (wave + 8 * i);
    const int n = bn0 + pb * 8 + lrow;
    b_org[i] = n < p.Wrows ? (int)((size_t)n * p.Cin * ES) + chunkB : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const int wtapbytes = p.Wrows * p.Cin * ES;
  const int pixbytes = p.xpitch * ES;

  const int kc_begin = (int)((long long)p.kchunks * split / p.splits);
  const int kc_end = (int)((long long)p.kchunks * (split + 1) / p.splits);
  const int nk = kc_end - kc_begin;

  // issue piece slots [I0, I0 + CNT) of K tile ts (ts >= nk: out-of-range pieces)
  auto issue_slots = [&](auto I0c, auto CNTc, int ts) {
    constexpr int I0 = decltype(I0c)::value, CNT = decltype(CNTc)::value;
    const int kc = kc_begin + ts;
    const bool live = ts < nk;
    const int tap = kc & (p.T - 1);
    const int coff = (kc >> p.log2T) * BKB;                // K order: channel chunk major, tap minor
    const int ty = tap >> p.TWlog2, tx = tap & twmask;
    const unsigned s_bit = live ? (1u << tap) : 0u;
    const int s_aoff = (ty * p.dstep * p.Ws + tx * p.dstep) * pixbytes + coff;
    const int s_boff = live ? ((wy0 + ty * p.wstep) * 4 + (wx0 + tx * p.wstep)) * wtapbytes + coff : (int)0x80000000;
    unsigned char* st = smem + (ts % NB) * STAGE;
    static_for<CNT>([&](auto Ic) {
      constexpr int i = I0 + decltype(Ic)::value;
      // slot -> (operand, half, k)
      constexpr bool isA = BN == 256 ? (i < 2 || i >= 6) : (i < 2 || i >= 4);
      if constexpr (isA) {
        constexpr int h = i < 2 ? 0 : 1, k = i & 1;
        const int pa = k * 16 + 8 * h + wave;
        const int off = (a_msk[h][k] & s_bit) ? a_org[h][k] + s_aoff : (int)0x80000000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(st + pa * 1024), 16, off, 0, 0, 0);
      } else {
        constexpr int bi = BN == 256 ? (i - 2) : (i - 2);  // 256: 0..3 = (h, k) as h*2+k;  128: 0..1 = k
        constexpr int h = BN == 256 ? (bi >> 1) : 0, k = bi & 1;
        const int pb = BN == 256 ? ((wave >> 2) * 8 + 16 * k + 4 * h + (wave & 3)) : (wave + 8 * k);
        const int boff = b_org[BN == 256 ? h * 2 + k : k] + s_boff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + BM * BKB + pb * 1024), 16, boff, 0, 0, 0);
      }
    });
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: tile row R, 16-byte K slot c at R*128 + ((c ^ (R & 7)) << 4); k-step s uses slots 4s + q
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned swz0 = (unsigned)((q ^ (r & 7)) << 4), swz1 = (unsigned)(((4 + q) ^ (r & 7)) << 4);
  const unsigned a_row = lds_base + (wr * 128 + r) * BKB, b_row = lds_base + BM * BKB + (wc * WTN + r) * BKB;

  DIAG_STAMP(1);
  // prologue: the first DP phases' worth of pieces
  static_for<(PP * DP + QS - 1) / QS>([&](auto Tc) {
    constexpr int t = decltype(Tc)::value;
    constexpr int cnt = (PP * DP - t * QS) < QS ? (PP * DP - t * QS) : QS;
    issue_slots(std::integral_constant<int, 0>{}, std::integral_constant<int, cnt>{}, t);
  });
  wait_vmcnt<VMW>();
  __builtin_amdgcn_s_barrier();
  DIAG_STAMP(2);
  if (wr == 1) __builtin_amdgcn_s_barrier();              // waves 4..7 run one barrier behind

  uint4 af[2][4], bf0[2][2], bf1[2][2];
  SEG_DECL;
  SEG_T0;
  for (int t = 0; t < nk; ++t) {
    const unsigned so = (unsigned)((t % NB) * STAGE);
    const unsigned aA0 = a_row + so + swz0, aA1 = a_row + so + swz1, bB0 = b_row + so + swz0, bB1 = b_row + so + swz1;
    static_for<PH>([&](auto Pc) {
      constexpr int ph = decltype(Pc)::value;
      // ---- LOAD segment ----
      constexpr int mh = BN == 256 ? ((ph == 0 || ph == 1) ? 0 : 1) : ph;     // row half multiplied in this phase
      constexpr int nh = BN == 256 ? ((ph == 1 || ph == 2) ? 1 : 0) : 0;      // column half (256x256 only)
      constexpr bool readA = BN == 256 ? (ph == 0 || ph == 2) : true;
      constexpr bool readB = BN == 256 ? (ph == 0 || ph == 1) : (ph == 0);
      constexpr int x0 = PP * (ph + DP);                   // slots of this phase, relative to K tile t
      issue_slots(std::integral_constant<int, x0 % QS>{}, std::integral_constant<int, PP>{}, t + x0 / QS);
      SEG_ADD(5);                                          // LDS-DMA issue
      if constexpr (readB) {
        static_for<2>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          constexpr int imm = (nh * 2 + j) * 16 * BKB;
          if constexpr (nh == 0) { lds_read128<imm>(bf0[0][j], bB0); lds_read128<imm>(bf0[1][j], bB1); }
          else { lds_read128<imm>(bf1[0][j], bB0); lds_read128<imm>(bf1[1][j], bB1); }
        });
      }
      if constexpr (readA) {
        static_for<4>([&](auto Ic) {
          constexpr int i = decltype(Ic)::value;
          constexpr int imm = (mh * 4 + i) * 16 * BKB;
          lds_read128<imm>(af[0][i], aA0); lds_read128<imm>(af[1][i], aA1);
        });
      }
      SEG_ADD(0);                                          // issue + fragment read issue
      wait_vmcnt<VMW>();
      SEG_ADD(1);                                          // waiting for older pieces
      __builtin_amdgcn_s_barrier();
      SEG_ADD(2);                                          // waiting for the partner group's MATH segment
      // ---- MATH segment ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            Mma<T>::run(acc[mh * 4 + i][nh * 2 + j], nh == 0 ? bf0[s2][j] : bf1[s2][j], af[s2][i]);
      __builtin_amdgcn_s_setprio(0);
      SEG_ADD(3);                                          // fragment wait + MFMAs
      __builtin_amdgcn_s_barrier();
      SEG_ADD(4);                                          // waiting for the partner group's LOAD segment
    });
  }
  SEG_STORE;
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  __syncthreads();
  DIAG_STAMP(3);
  gemm_epilogue<T, BM, BN, WAVES_M, WAVES_N, NB * STAGE, false, false, BF>(p, acc, smem, bm0, bn0, par, P, split);
  DIAG_STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------------
// Tap-shared ping-pong kernel (round 4): the ping-pong kernel above with the A operand of SH taps staged ONCE.
//
// The taps of one kernel row read the same source pixels, shifted: tap tx of output column gx reads source column
// gx*S + dx0 + tx*dstep, i.e. tap tx + NC (NC = S tap classes per row) reads what tap tx reads for the NEIGHBOURING output column
// gx + dstep.  A K "super tile" = (channel chunk, tap row ty, class c) therefore stages the 256 tile rows of the class's first tap
// once and multiplies SH = TW / NC taps (2: stride-2 convolutions and the parity sub-GEMMs, 4: stride-1 convolutions) against it:
// sub tile j reads the A fragment of tile row m from LDS row m + j*dir (dir = sign of dstep).  Where that leaves the row's SEGMENT
// (the part of an image row inside the row's 64-row block of the tile) the pixel is a HALO entry: per block 16 LDS rows behind the
// tile (256 + 16*block + segment*(SH-1) + k), staged by one extra piece per wave and super tile with the same validity logic as a tile
// row at the virtual column.  Staged bytes per K tile: 256x128 tile 48 KB -> 34 (SH 2) / 26 (SH 4); 256x256 64 -> 50 / 42 - the loop
// of the kernel above is as long as its LOAD segment, these pieces and the address work in front of them included.
//
// K order: chunk major, then tap row, class, and the SH taps of the class (the kernel above: chunk major, tap minor), so sums differ
// from it in fp32 rounding order only.  LDS: A ring of 2 super tiles x (256 + 64 halo rows) x 128 B | B ring of NBB K tiles.
// Schedule: phases as above (PH per K tile, NPH = PH*SH per super tile); the pieces of a super tile are issued by the table
// PsSched<BN, SH> (entry: what, and the phase relative to the super tile's first phase in whose LOAD segment it is issued); the
// counted vmcnt of every phase follows from the table (ps_vm), the RAW / WAR rules of the kernel above are static_asserts below:
//   RAW: a piece first read in phase r is covered by the vmcnt at the end of LOAD(r-1)   (need - issue phase >= 2 is asserted)
//   WAR: a slot is re-filled >= 2 phases after the last LOAD segment that reads its old contents
//        A0 blocks (rows 0-63 of each wave row): last read in phase (SH-1)*PH (sub tile SH-1's first-half phase; segments are cut at the
//        64-row blocks so that a shifted read never leaves its block + halo), A1 blocks and halo: last phase that reads A.
enum { PS_A0 = 0, PS_A1 = 1, PS_H = 2, PS_B = 3 };
struct PsEntry { int kind, k, j, h, p; };     // piece k of the kind (B: of sub tile j, column half h), issue phase p
template <int BN, int SH, int PH> struct PsSched;        // PH: phases per K tile
template <int BN, int SH, int PH> constexpr int ps_need(const PsEntry& e) {   // phase (of the super tile) whose LOAD segment first reads it
  return e.kind == PS_A0 ? 0 : e.kind == PS_A1 ? (PH == 4 ? 2 : 1) : e.kind == PS_H ? PH : e.j * PH + (PH == 4 ? e.h : 0);
}
// The table: every piece is issued PS_LA phases before the phase that first reads it, or as early as its slot's WAR rule allows
// if that is later; entries in issue order (stable: the order below inside one phase).  PS_LA = 3 from a sweep (2: latency shows,
// +10 %; 4-6: more pieces in flight lengthen the LOAD segments, up to +45 % on the 256x256 tiles)
#ifndef PS_LA
#define PS_LA 3
#endif
#ifndef PS_LA_A128
#define PS_LA_A128 4      // A pieces of the 256x128 tiles (activations: HBM / Infinity Cache): -1 % on the class; on the 256x256 tiles +6 %; B pieces at 2: +9 %
#endif
template <int BN, int SH> struct PsTable { static constexpr int Q = 4 + 1 + SH * (BN == 256 ? 4 : 2); PsEntry e[Q]; };
template <int BN, int SH, int PH> constexpr PsTable<BN, SH> ps_make() {
  constexpr int NPH = PH * SH, NBB = BN == 256 ? 2 : 4, Q = PsTable<BN, SH>::Q;
  constexpr int lastA0 = (SH - 1) * PH, lastA1 = (SH - 1) * PH + (PH == 4 ? 2 : 1);
  PsTable<BN, SH> t{};
  int n = 0;
  for (int k = 0; k < 2; ++k) t.e[n++] = PsEntry{PS_A0, k, 0, 0, 0};
  for (int k = 0; k < 2; ++k) t.e[n++] = PsEntry{PS_A1, k, 0, 0, 0};
  t.e[n++] = PsEntry{PS_H, 0, 0, 0, 0};
  for (int j = 0; j < SH; ++j)
    for (int h = 0; h < (BN == 256 ? 2 : 1); ++h)
      for (int k = 0; k < 2; ++k) t.e[n++] = PsEntry{PS_B, k, j, h, 0};
  for (int i = 0; i < Q; ++i) {
    const int need = ps_need<BN, SH, PH>(t.e[i]);
    const int last = t.e[i].kind == PS_A0 ? lastA0 : lastA1;
    const int most = t.e[i].kind == PS_B ? NBB * PH - 2 : 2 * NPH + need - last - 2;
    const int la = (t.e[i].kind != PS_B && BN == 128) ? PS_LA_A128 : PS_LA;
    t.e[i].p = need - (la < most ? la : most);
  }
  for (int i = 1; i < Q; ++i) {                              // insertion sort by issue phase
    const PsEntry x = t.e[i];
    int m = i - 1;
    while (m >= 0 && t.e[m].p > x.p) { t.e[m + 1] = t.e[m]; --m; }
    t.e[m + 1] = x;
  }
  return t;
}
template <int BN, int SH, int PH> struct PsSched {
  static constexpr int Q = PsTable<BN, SH>::Q;
  static constexpr PsTable<BN, SH> tab = ps_make<BN, SH, PH>();
};
template <int BN, int SH, int PH> constexpr int ps_phase_of(const PsEntry& e) { constexpr int NPH = PH * SH; return ((e.p % NPH) + NPH) % NPH; }
template <int BN, int SH, int PH> constexpr int ps_ahead_of(const PsEntry& e) { constexpr int NPH = PH * SH; return e.p < 0 ? (-e.p + NPH - 1) / NPH : 0; }
// pieces that may stay outstanding at the end of LOAD(gph) in the steady state: everything issued after the newest piece that phase
// gph + 1 reads (LDS-DMA pieces retire in issue order; program order inside a phase = table order)
template <int BN, int SH, int PH> constexpr int ps_vm(int gph) {
  using S = PsSched<BN, SH, PH>;
  constexpr int NPH = PH * SH;
  const int g = 3 * NPH + gph;
  int seq = 0, newest_needed = -1;
  for (int gg = -2 * NPH; gg <= g; ++gg)
    for (int i = 0; i < S::Q; ++i) {
      const int d = gg - S::tab.e[i].p;
      if (d % NPH != 0) continue;
      const int U = d / NPH;                                   // the super tile this issue belongs to
      if (U * NPH + ps_need<BN, SH, PH>(S::tab.e[i]) <= g + 1) newest_needed = seq;
      ++seq;
    }
  return seq - 1 - newest_needed;
}
template <int BN, int SH, int PH> constexpr bool ps_sched_ok() {
  using S = PsSched<BN, SH, PH>;
  constexpr int NPH = PH * SH, NBB = BN == 256 ? 2 : 4;
  constexpr int lastA0 = (SH - 1) * PH, lastA1 = (SH - 1) * PH + (PH == 4 ? 2 : 1);
  int count = 0;
  for (int i = 0; i < S::Q; ++i) {
    const PsEntry& e = S::tab.e[i];
    const int need = ps_need<BN, SH, PH>(e), ahead = need - e.p;
    if (ahead < 2) return false;
    if (i && e.p < S::tab.e[i - 1].p) return false;                // table in issue order
    const int last = e.kind == PS_A0 ? lastA0 : lastA1;
    if (e.kind == PS_B ? ahead > NBB * PH - 2 : ahead > 2 * NPH + need - last - 2) return false;
    ++count;
  }
  return count == 4 + 1 + SH * (BN == 256 ? 4 : 2);
}
static_assert(ps_sched_ok<128, 2, 2>() && ps_sched_ok<128, 4, 2>() && ps_sched_ok<256, 2, 4>() && ps_sched_ok<256, 4, 4>(), "tap-shared piece schedule");

template <typename T, int BN, int SH, int PH, bool BF>
__global__ __launch_bounds__(512) void conv_gemm_ps_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Sched = PsSched<BN, SH, PH>;
  constexpr int BM = 256, WAVES_M = 2, WAVES_N = 4, BKB = 128;
  constexpr int WTN = BN / 4, MT = 8, NT = WTN / 16;        // wave tile 128 x (64 | 32)
  constexpr int NPH = PH * SH;
  constexpr int NBB = BN == 256 ? 2 : 4;
  constexpr int LOG2SH = SH == 4 ? 2 : 1;
  constexpr int ASTAGE = (BM + 64) * BKB, BOFF = 2 * ASTAGE, BSTAGE = BN * BKB;
  constexpr int ES = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  DIAG_STAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;

  const int P = p.parity ? 4 : 1;
  int bid = blockIdx.x;
  const int nb = p.tilesM * p.tilesN * P;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int par = bid % P;
  bid /= P;
  const int bm0 = (bid / p.tilesN) * BM, bn0 = (bid % p.tilesN) * BN;
  const int split = blockIdx.z;
  const int py = par >> 1, px = par & 1;
  int dy0 = p.dy0, dx0 = p.dx0, wy0 = p.wy0, wx0 = p.wx0;
  if (p.parity) { dy0 = py; dx0 = px; wy0 = 1 - py; wx0 = 1 - px; }
  const int twmask = (1 << p.TWlog2) - 1, TW = 1 << p.TWlog2;
  const int dir = p.dstep > 0 ? 1 : -1;

  // ---- per-lane gather state: 4 tile rows + 1 halo row ---------------------------------------------------
  const int lrow = lane >> 3, slot = lane & 7;
  const int chunkB = ((slot ^ lrow) << 4);
  // window origin (+ chunk) and tap mask of the (possibly virtual) grid position (image row t, column gx)
  auto pix_state = [&](unsigned t, int gx, int& org, unsigned& msk) {
    const unsigned img = fdiv(t, p.divHg);
    const int gy = (int)t - (int)img * p.Hg;
    const int sy0 = gy * p.S + dy0, sx0 = gx * p.S + dx0;
    unsigned vx = 0, m_ = 0;
    for (int tx = 0; tx < TW; ++tx) vx |= ((unsigned)(sx0 + tx * p.dstep) < (unsigned)p.Ws ? 1u : 0u) << tx;
    for (int ty = 0; ty < TW; ++ty)
      if ((unsigned)(sy0 + ty * p.dstep) < (unsigned)p.Hs) m_ |= vx << (ty << p.TWlog2);
    msk = m_;
    org = (int)((((long long)((int)img * p.Hs + sy0) * p.Ws + sx0) * (long long)p.xpitch) * ES) + chunkB;
  };
  int a_org[2][2];
  unsigned a_msk[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = bm0 + (k * 16 + 8 * h + wave) * 8 + lrow;
      a_org[h][k] = 0; a_msk[h][k] = 0;
      if (m < p.M) {
        const unsigned t = fdiv((unsigned)m, p.divWg);
        pix_state(t, m - (int)t * p.Wg, a_org[h][k], a_msk[h][k]);
      }
    }
  int h_org = 0;
  unsigned h_msk = 0;
  {
    const int hb = wave >> 1, sub = (wave & 1) * 8 + lrow;           // halo row 16*hb + sub of block hb
    const int seg = sub / (SH - 1), k = sub - seg * (SH - 1);
    const int g0 = bm0 + hb * 64, g1 = g0 + 63;
    const unsigned t = fdiv((unsigned)g0, p.divWg) + (unsigned)seg;
    const long long rs = (long long)t * p.Wg;                        // first GEMM row of image row t
    if (rs <= g1 && rs < p.M) {
      const int gs = g0 > (int)rs ? g0 - (int)rs : 0, ge = g1 - (int)rs < p.Wg - 1 ? g1 - (int)rs : p.Wg - 1;
      pix_state(t, dir > 0 ? ge + 1 + k : gs - 1 - k, h_org, h_msk);
    }
  }
  constexpr int BK_ = BN == 256 ? 4 : 2;                    // B pieces per wave per K tile
  int b_org[BK_];
#pragma unroll
  for (int i = 0; i < BK_; ++i) {
    const int pb = BN == 256 ? ((wave >> 2) * 8 + 16 * (i & 1) + 4 * (i >> 1) + (wave & 3)) : (wave + 8 * i);
    const int n = bn0 + pb * 8 + lrow;
    b_org[i] = n < p.Wrows ? (int)((size_t)n * p.Cin * ES) + chunkB : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const int wtapbytes = p.Wrows * p.Cin * ES;
  const int pixbytes = p.xpitch * ES;

  const int su_total = p.kchunks >> LOG2SH;
  const int su_begin = (int)((long long)su_total * split / p.splits);
  const int nsu = (int)((long long)su_total * (split + 1) / p.splits) - su_begin;
  const int log2NC = p.TWlog2 - LOG2SH;                     // tap classes per kernel row = TW / SH

  // fragment addresses.  Unshifted (sub tile 0): tile row R, 16-byte K slot c at R*128 + ((c ^ (R & 7)) << 4), as above.
  // Shifted (sub tile j): per fragment and lane the LDS row of (tile row + j*dir) or of its halo entry; slot q, the second k step = ^ 64.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned swz0 = (unsigned)((q ^ (r & 7)) << 4), swz1 = (unsigned)(((4 + q) ^ (r & 7)) << 4);
  const unsigned a_row = lds_base + (wr * 128 + r) * BKB, b_row = lds_base + BOFF + (wc * WTN + r) * BKB;
  unsigned a_tbl[8][SH - 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = wr * 128 + i * 16 + r, hb = wr * 2 + (i >> 2);
    const int g0 = bm0 + hb * 64;
    const unsigned t = fdiv((unsigned)(bm0 + m), p.divWg);
    const int rs = (int)t * p.Wg, gx = bm0 + m - rs;
    const int gs = g0 > rs ? g0 - rs : 0, ge = g0 + 63 - rs < p.Wg - 1 ? g0 + 63 - rs : p.Wg - 1;
    const int seg = (int)t - (int)fdiv((unsigned)g0, p.divWg);
#pragma unroll
    for (int j = 1; j < SH; ++j) {
      const int gv = gx + j * dir;
      const int L = (gv >= gs && gv <= ge) ? m + j * dir : BM + hb * 16 + seg * (SH - 1) + (dir > 0 ? gv - ge - 1 : gs - 1 - gv);
      a_tbl[i][j - 1] = lds_base + (unsigned)L * BKB + (unsigned)((q ^ (L & 7)) << 4);
    }
  }

  // issue table entry E for the block's super tile us (us >= nsu: out-of-range pieces).  (Computing these offsets behind the wave's own
  // MFMAs, one phase ahead, measured slower: the work is a serial chain wherever it sits; conv_gemm_pt_kernel removes it instead.)
  auto issue_entry = [&](auto Ec, int us) {
    constexpr PsEntry e = Sched::tab.e[decltype(Ec)::value];
    const bool live = us < nsu;
    const int su = su_begin + us;
    const int c = su & ((1 << log2NC) - 1);
    const int ty = (su >> log2NC) & twmask;
    const int coff = (su >> (log2NC + p.TWlog2)) * BKB;
    if constexpr (e.kind == PS_B) {
      const int tx = c + (e.j << log2NC);
      const int s_boff = live ? ((wy0 + ty * p.wstep) * 4 + (wx0 + tx * p.wstep)) * wtapbytes + coff : (int)0x80000000;
      constexpr int bi = BN == 256 ? e.h * 2 + e.k : e.k;
      const int pb = BN == 256 ? ((wave >> 2) * 8 + 16 * e.k + 4 * e.h + (wave & 3)) : (wave + 8 * e.k);
      unsigned char* dst = smem + BOFF + ((us * SH + e.j) & (NBB - 1)) * BSTAGE + pb * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, b_org[bi] + s_boff, 0, 0, 0);
    } else {
      const unsigned s_bit = live ? (1u << ((ty << p.TWlog2) + c)) : 0u;
      const int s_aoff = (ty * p.dstep * p.Ws + c * p.dstep) * pixbytes + coff;
      unsigned char* stA = smem + (us & 1) * ASTAGE;
      if constexpr (e.kind == PS_H) {
        const int off = (h_msk & s_bit) ? h_org + s_aoff : (int)0x80000000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(stA + BM * BKB + wave * 1024), 16, off, 0, 0, 0);
      } else {
        constexpr int h = e.kind == PS_A1 ? 1 : 0;
        const int pa = e.k * 16 + 8 * h + wave;
        const int off = (a_msk[h][e.k] & s_bit) ? a_org[h][e.k] + s_aoff : (int)0x80000000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(stA + pa * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  DIAG_STAMP(1);
  // prologue: what the steady state would have issued before the first phase (issue phase < 0), in its order
  static_for<2 * NPH>([&](auto Gc) {
    constexpr int gg = decltype(Gc)::value - 2 * NPH;
    static_for<Sched::Q>([&](auto Ec) {
      constexpr PsEntry e = Sched::tab.e[decltype(Ec)::value];
      if constexpr ((gg - e.p) % NPH == 0 && gg - e.p >= 0) issue_entry(Ec, (gg - e.p) / NPH);
    });
  });
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  DIAG_STAMP(2);
  if (wr == 1) __builtin_amdgcn_s_barrier();              // waves 4..7 run one barrier behind

  uint4 af[2][4], bf0[2][2], bf1[2][2];
  for (int u = 0; u < nsu; ++u) {
    const unsigned soA = (unsigned)((u & 1) * ASTAGE);
    const unsigned aA0 = a_row + soA + swz0, aA1 = a_row + soA + swz1;
    static_for<NPH>([&](auto Gc) {
      constexpr int gph = decltype(Gc)::value, jsub = gph / PH, ph = gph % PH;
      // ---- LOAD segment ----
      constexpr int mh = PH == 4 ? ((ph == 0 || ph == 1) ? 0 : 1) : ph;       // row half multiplied in this phase
      constexpr int nh = PH == 4 ? ((ph == 1 || ph == 2) ? 1 : 0) : 0;        // column half (256x256 only)
      constexpr bool readA = PH == 4 ? (ph == 0 || ph == 2) : true;
      constexpr bool readB = PH == 4 ? (ph == 0 || ph == 1) : (ph == 0);
      static_for<Sched::Q>([&](auto Ec) {
        constexpr PsEntry e = Sched::tab.e[decltype(Ec)::value];
        if constexpr (ps_phase_of<BN, SH, PH>(e) == gph) issue_entry(Ec, u + ps_ahead_of<BN, SH, PH>(e));
      });
      if constexpr (readB) {
        const unsigned soB = (unsigned)(((u * SH + jsub) & (NBB - 1)) * BSTAGE);
        const unsigned bB0 = b_row + soB + swz0, bB1 = b_row + soB + swz1;
        static_for<2>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          constexpr int imm = (nh * 2 + j) * 16 * BKB;
          if constexpr (nh == 0) { lds_read128<imm>(bf0[0][j], bB0); lds_read128<imm>(bf0[1][j], bB1); }
          else { lds_read128<imm>(bf1[0][j], bB0); lds_read128<imm>(bf1[1][j], bB1); }
        });
      }
      if constexpr (readA) {
        static_for<4>([&](auto Ic) {
          constexpr int i = decltype(Ic)::value;
          if constexpr (jsub == 0) {
            constexpr int imm = (mh * 4 + i) * 16 * BKB;
            lds_read128<imm>(af[0][i], aA0); lds_read128<imm>(af[1][i], aA1);
          } else {
            const unsigned a0 = a_tbl[mh * 4 + i][jsub - 1] + soA;
            lds_read128<0>(af[0][i], a0); lds_read128<0>(af[1][i], a0 ^ 64u);
          }
        });
      }
      wait_vmcnt<ps_vm<BN, SH, PH>(gph)>();
      __builtin_amdgcn_s_barrier();
      // ---- MATH segment ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            Mma<T>::run(acc[mh * 4 + i][nh * 2 + j], nh == 0 ? bf0[s2][j] : bf1[s2][j], af[s2][i]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    });
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  __syncthreads();
  DIAG_STAMP(3);
  gemm_epilogue<T, BM, BN, WAVES_M, WAVES_N, BOFF + NBB * BSTAGE, false, false, BF>(p, acc, smem, bm0, bn0, par, P, split);
  DIAG_STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------------
// Table-driven form of the tap-shared kernel (round 4): a LOAD segment of the kernels above is a serial chain - ~40 scalar instructions
// that decode the K tile into tap / channel offsets, 4 vector instructions per piece that apply them and the validity mask, then the
// pieces, then the fragment reads - and it, not the MFMAs, sets the length of a phase (knock-outs on the kernel above, 256x128 class:
// no scalar decode -14 %, no vector work -19 %, neither -27 %; pieces out of range or not makes no difference; moving the same work behind
// the wave's own MFMAs gained nothing).  Here the work is done ONCE per block:
//   * every A piece's offset is a loop-invariant register: atab[row][combo] = window origin + tap offset of the (tap row, class) combination,
//     or 0x80000000 where that tap leaves the map (5 rows x 8 | 4 | 2 combinations);
//   * the K loop is `for (chunk) unroll (combination) unroll (phase)`: what differs from chunk to chunk is ONE scalar (the channel
//     offset, handed to the instruction as its scalar offset), LDS stage / ring-slot offsets are immediates, B pieces take the lane's
//     row offset register + a scalar tap offset;
//   * pieces issued for the chunk after the block's last one use a descriptor with zero records (zeros, no vector instruction).
// A LOAD segment is then <= 3 x (m0, scalar offset, buffer_load ... lds) + 12 ds_read_b128 with immediates.  Arithmetic, schedule and
// results are those of conv_gemm_ps_kernel.
template <typename T, int BN, int SH, int TWV, bool BF>
__global__ __launch_bounds__(512) void conv_gemm_pt_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PH = BN == 256 ? 4 : 2;
  using Sched = PsSched<BN, SH, PH>;
  constexpr int BM = 256, WAVES_M = 2, WAVES_N = 4, BKB = 128;
  constexpr int WTN = BN / 4, MT = 8, NT = WTN / 16;
  constexpr int NPH = PH * SH;
  constexpr int NBB = BN == 256 ? 2 : 4;
  constexpr int NC = TWV / SH, NCOMBO = TWV * NC;           // tap classes per kernel row; (tap row, class) combinations = super tiles per chunk
  constexpr int LOG2T = TWV == 4 ? 4 : 2;
  constexpr int ASTAGE = (BM + 64) * BKB, BOFF = 2 * ASTAGE, BSTAGE = BN * BKB;
  constexpr int ES = sizeof(T);
  static_assert((NCOMBO * SH) % NBB == 0 && NCOMBO % 2 == 0, "ring slots are compile-time inside a chunk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  DIAG_STAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;

  const int P = p.parity ? 4 : 1;
  int bid = blockIdx.x;
  const int nb = p.tilesM * p.tilesN * P;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int par = bid % P;
  bid /= P;
  const int bm0 = (bid / p.tilesN) * BM, bn0 = (bid % p.tilesN) * BN;
  const int split = blockIdx.z;
  const int py = par >> 1, px = par & 1;
  int dy0 = p.dy0, dx0 = p.dx0, wy0 = p.wy0, wx0 = p.wx0;
  if (p.parity) { dy0 = py; dx0 = px; wy0 = 1 - py; wx0 = 1 - px; }
  const int dir = p.dstep > 0 ? 1 : -1;
  const int pixbytes = p.xpitch * ES;

  DIAG_STAMP(5);
  // ---- per-lane piece offsets: 4 tile rows + 1 halo row, one per (tap row, class) -----------------------
  const int lrow = lane >> 3, slot = lane & 7;
  const int chunkB = ((slot ^ lrow) << 4);
  int atab[5][NCOMBO];
  // (Straight-line, branch-free and short: written with && / ?: per entry this block compiled to ~1,100 instructions with 55 exec-masked
  // branches and a scalar multiply chain per entry - 3.2 us per block in front of the first piece, tools/diag_gemm.py.)
  int s_aoff[NCOMBO];                                        // tap offsets of the combinations: scalars, shared by the five rows
#pragma unroll
  for (int cc = 0; cc < NCOMBO; ++cc) s_aoff[cc] = ((cc / NC) * p.dstep * p.Ws + (cc % NC) * p.dstep) * pixbytes;
  auto fill_row = [&](auto Rc, bool exists, unsigned t, int gx) {
    constexpr int row = decltype(Rc)::value;
    const unsigned img = fdiv(t, p.divHg);
    const int gy = (int)t - (int)img * p.Hg;
    const int sy0 = gy * p.S + dy0, sx0 = gx * p.S + dx0;
    const int org = (int)((((long long)((int)img * p.Hs + sy0) * p.Ws + sx0) * (long long)p.xpitch) * ES) + chunkB;
    unsigned vx = 0, vyx = 0;                                // validity bits: classes; (tap row, class) combinations
#pragma unroll
    for (int c = 0; c < NC; ++c) vx |= (unsigned)((unsigned)(sx0 + c * p.dstep) < (unsigned)p.Ws) << c;
#pragma unroll
    for (int ty = 0; ty < TWV; ++ty) vyx |= (vx & (0u - (unsigned)((unsigned)(sy0 + ty * p.dstep) < (unsigned)p.Hs))) << (ty * NC);
    vyx &= 0u - (unsigned)exists;
    static_for<NCOMBO>([&](auto Cc) {
      constexpr int cc = decltype(Cc)::value;
      const int m = __builtin_amdgcn_sbfe((int)vyx, cc, 1);           // 0 | -1
      atab[row][cc] = ((org + s_aoff[cc]) & m) | ((int)0x80000000 & ~m);
    });
  };
  static_for<4>([&](auto Rc) {
    constexpr int row = decltype(Rc)::value, h = row >> 1, k = row & 1;          // row = 2 * half + piece, as the table's A0 / A1 entries
    const int m = bm0 + (k * 16 + 8 * h + wave) * 8 + lrow;
    const unsigned t = fdiv((unsigned)m, p.divWg);
    fill_row(Rc, m < p.M, t, m - (int)t * p.Wg);
  });
  {
    const int hb = wave >> 1, sub = (wave & 1) * 8 + lrow;           // halo row 16*hb + sub of block hb
    const int seg = sub / (SH - 1), k = sub - seg * (SH - 1);
    const int g0 = bm0 + hb * 64, g1 = g0 + 63;
    const unsigned t = fdiv((unsigned)g0, p.divWg) + (unsigned)seg;
    const long long rs = (long long)t * p.Wg;
    const bool exists = rs <= g1 && rs < p.M;
    const int gs = g0 > (int)rs ? g0 - (int)rs : 0, ge = g1 - (int)rs < p.Wg - 1 ? g1 - (int)rs : p.Wg - 1;
    fill_row(std::integral_constant<int, 4>{}, exists, t, dir > 0 ? ge + 1 + k : gs - 1 - k);
  }
  DIAG_STAMP(6);
  constexpr int BK_ = BN == 256 ? 4 : 2;
  int b_org[BK_];
#pragma unroll
  for (int i = 0; i < BK_; ++i) {
    const int pb = BN == 256 ? ((wave >> 2) * 8 + 16 * (i & 1) + 4 * (i >> 1) + (wave & 3)) : (wave + 8 * i);
    const int n = bn0 + pb * 8 + lrow;
    b_org[i] = n < p.Wrows ? (int)((size_t)n * p.Cin * ES) + chunkB : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const int wtapbytes = p.Wrows * p.Cin * ES;
  const int wstepb = p.wstep * wtapbytes, wbase = (wy0 * 4 + wx0) * wtapbytes;      // weight tap (ty, tx) at wbase + (4 ty + tx) * wstepb

  const int ch_total = p.kchunks >> LOG2T;                  // 128-byte channel chunks
  const int ch_begin = (int)((long long)ch_total * split / p.splits);
  const int nch = (int)((long long)ch_total * (split + 1) / p.splits) - ch_begin;

  DIAG_STAMP(7);
  // fragment addresses: stage and ring-slot offsets are immediates; the shifted reads take per-lane tables (both k steps)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned swz0 = (unsigned)((q ^ (r & 7)) << 4), swz1 = (unsigned)(((4 + q) ^ (r & 7)) << 4);
  const unsigned a_base0 = lds_base + (wr * 128 + r) * BKB + swz0, a_base1 = lds_base + (wr * 128 + r) * BKB + swz1;
  const unsigned b_base0 = lds_base + BOFF + (wc * WTN + r) * BKB + swz0, b_base1 = lds_base + BOFF + (wc * WTN + r) * BKB + swz1;
  static_assert(BN == 128, "256 columns: no room for the tables (measured with the second table replaced by a v_xor per read: 0.164 -> 0.187 ms per step for the class, spills)");
  unsigned a_tbl[8][SH - 1], a_tbx[8][SH - 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = wr * 128 + i * 16 + r, hb = wr * 2 + (i >> 2);
    const int g0 = bm0 + hb * 64;
    const unsigned t = fdiv((unsigned)(bm0 + m), p.divWg);
    const int rs = (int)t * p.Wg, gx = bm0 + m - rs;
    const int gs = g0 > rs ? g0 - rs : 0, ge = g0 + 63 - rs < p.Wg - 1 ? g0 + 63 - rs : p.Wg - 1;
    const int seg = (int)t - (int)fdiv((unsigned)g0, p.divWg);
#pragma unroll
    for (int j = 1; j < SH; ++j) {
      const int gv = gx + j * dir;
      const int L = (gv >= gs && gv <= ge) ? m + j * dir : BM + hb * 16 + seg * (SH - 1) + (dir > 0 ? gv - ge - 1 : gs - 1 - gv);
      a_tbl[i][j - 1] = lds_base + (unsigned)L * BKB + (unsigned)((q ^ (L & 7)) << 4);
      a_tbx[i][j - 1] = a_tbl[i][j - 1] ^ 64u;
    }
  }
  unsigned char* const a_dst = smem + wave * 1024;           // this wave's piece inside an A panel / B panel
  unsigned char* const b_dst = smem + BOFF + wave * 1024;

  // table entry E for combination CT counted from the current chunk's first (CT >= NCOMBO: the next chunk's; `rxn` / `rwn` have zero records
  // when there is none), coff = the current chunk's byte offset
  auto issue = [&](auto Ec, auto CTc, int coff, const __amdgpu_buffer_rsrc_t& rxn, const __amdgpu_buffer_rsrc_t& rwn) {
    constexpr PsEntry e = Sched::tab.e[decltype(Ec)::value];
    constexpr int ct = decltype(CTc)::value, cc = ct % NCOMBO, ty = cc / NC, c = cc % NC;
    constexpr bool wrap = ct >= NCOMBO;
    const int so = coff + (wrap ? BKB : 0);
    if constexpr (e.kind == PS_B) {
      constexpr int tapmul = 4 * ty + c + e.j * NC;
      constexpr int bslot = (ct * SH + e.j) & (NBB - 1);
      constexpr int pbo = BN == 256 ? (16 * e.k + 4 * e.h) * 1024 : 8 * e.k * 1024;
      unsigned char* dst = (BN == 256 ? smem + BOFF + ((wave >> 2) * 8 + (wave & 3)) * 1024 : b_dst) + bslot * BSTAGE + pbo;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrap ? rwn : rw, (__attribute__((address_space(3))) void*)dst, 16, b_org[BN == 256 ? e.h * 2 + e.k : e.k],
                                               wbase + tapmul * wstepb + so, 0, 0);
    } else {
      constexpr int row = e.kind == PS_H ? 4 : (e.kind == PS_A1 ? 2 : 0) + e.k;
      constexpr int pa = e.kind == PS_H ? 32 : e.k * 16 + 8 * (e.kind == PS_A1 ? 1 : 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrap ? rxn : rx, (__attribute__((address_space(3))) void*)(a_dst + (ct & 1) * ASTAGE + pa * 1024), 16,
                                               atab[row][cc], so, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  DIAG_STAMP(1);
  // prologue: what the steady state would have issued before the first phase
  static_for<2 * NPH>([&](auto Gc) {
    constexpr int gg = decltype(Gc)::value - 2 * NPH;
    static_for<Sched::Q>([&](auto Ec) {
      constexpr PsEntry e = Sched::tab.e[decltype(Ec)::value];
      if constexpr ((gg - e.p) % NPH == 0 && gg - e.p >= 0) {
        static_assert((gg - e.p) / NPH < NCOMBO, "look-ahead inside one chunk");
        issue(Ec, std::integral_constant<int, (gg - e.p) / NPH>{}, ch_begin * BKB, rx, rw);
      }
    });
  });
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  DIAG_STAMP(2);
  if (wr == 1) __builtin_amdgcn_s_barrier();              // waves 4..7 run one barrier behind

  uint4 af[2][4], bf0[2][2], bf1[2][2];
  for (int cu = 0; cu < nch; ++cu) {
    const int coff = (ch_begin + cu) * BKB;
    const bool more = cu + 1 < nch;
    const __amdgpu_buffer_rsrc_t rxn = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, more ? p.xbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwn = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, more ? p.wbytes : 0u, 0x00020000);
    static_for<NCOMBO>([&](auto Cc) {
      constexpr int combo = decltype(Cc)::value;
      static_for<NPH>([&](auto Gc) {
        constexpr int gph = decltype(Gc)::value, jsub = gph / PH, ph = gph % PH;
        constexpr int mh = PH == 4 ? ((ph == 0 || ph == 1) ? 0 : 1) : ph;
        constexpr int nh = PH == 4 ? ((ph == 1 || ph == 2) ? 1 : 0) : 0;
        constexpr bool readA = PH == 4 ? (ph == 0 || ph == 2) : true;
        constexpr bool readB = PH == 4 ? (ph == 0 || ph == 1) : (ph == 0);
        // ---- LOAD segment: scalar and memory instructions only ----
        static_for<Sched::Q>([&](auto Ec) {
          constexpr PsEntry e = Sched::tab.e[decltype(Ec)::value];
          if constexpr (ps_phase_of<BN, SH, PH>(e) == gph) {
            static_assert(ps_ahead_of<BN, SH, PH>(e) <= 1, "look-ahead of at most one super tile");
            issue(Ec, std::integral_constant<int, combo + ps_ahead_of<BN, SH, PH>(e)>{}, coff, rxn, rwn);
          }
        });
        if constexpr (readB) {
          constexpr int bslot = (combo * SH + jsub) & (NBB - 1);
          static_for<2>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            constexpr int imm = bslot * BSTAGE + (nh * 2 + j) * 16 * BKB;
            if constexpr (nh == 0) { lds_read128<imm>(bf0[0][j], b_base0); lds_read128<imm>(bf0[1][j], b_base1); }
            else { lds_read128<imm>(bf1[0][j], b_base0); lds_read128<imm>(bf1[1][j], b_base1); }
          });
        }
        if constexpr (readA) {
          constexpr int sA = (combo & 1) * ASTAGE;
          static_for<4>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if constexpr (jsub == 0) {
              constexpr int imm = sA + (mh * 4 + i) * 16 * BKB;
              lds_read128<imm>(af[0][i], a_base0); lds_read128<imm>(af[1][i], a_base1);
            } else {
              lds_read128<sA>(af[0][i], a_tbl[mh * 4 + i][jsub - 1]); lds_read128<sA>(af[1][i], a_tbx[mh * 4 + i][jsub - 1]);
            }
          });
        }
        wait_vmcnt<ps_vm<BN, SH, PH>(gph)>();
        __builtin_amdgcn_s_barrier();
        // ---- MATH segment ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              Mma<T>::run(acc[mh * 4 + i][nh * 2 + j], nh == 0 ? bf0[s2][j] : bf1[s2][j], af[s2][i]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
      });
    });
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  __syncthreads();
  DIAG_STAMP(3);
  gemm_epilogue<T, BM, BN, WAVES_M, WAVES_N, BOFF + NBB * BSTAGE, false, false, BF>(p, acc, smem, bm0, bn0, par, P, split);
  DIAG_STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------------
// "Parity-patch" kernel for the stride-2 transposed convolutions / stride-2 conv input gradients with few output channels
// (generator up6 forward, down1 dgrad, PatchGAN down1 dgrad: 64 output channels, K = 4 taps x Cin per output parity).
//
// As four separate parity sub-GEMMs these layers stage every input pixel 16 times (4 parities x 4 taps) for 64 columns of
// output: 51 FLOP per staged byte, a third of what the LDS-DMA path needs to keep the matrix pipe busy.
// Here ONE block computes all four parities of 256 input-grid positions (R whole image rows) x 64 channels, i.e. 1024 output
// pixels, from an input PATCH that is staged once per channel chunk: (R+2) rows of Wg+2 pixels (a one-pixel halo, zeros outside the
// map; row pitch Wg+8) x 32 channels (64-byte LDS rows).  The 16 (parity, tap) products of a chunk read it at nine different shifts
//     output (2gy+py, 2gx+px) += x[gy + py - ty, gx + px - tx] . W[1-py+2ty][1-px+2tx],   ty, tx in {0, 1}
// so an A fragment is an ordinary ds_read_b128 at (pixel + shift) * 64 bytes.  Per chunk 32-40 KB of patch + 64 KB of weights
// are staged for 16.8 MFLOP: ~170 FLOP per staged byte.
//
// Waves: 8 = 2 (row halves of 128) x 4 (parities); a wave's accumulators are 128 rows x 64 channels of ONE parity, the same
// register tile as the 256x256 ping-pong kernel (8 x 4 MFMA tiles), whose phase structure is kept: a K step = one tap of one
// chunk = 32 MFMAs per wave; LOAD segment (LDS-DMA issue + 12 fragment reads + counted vmcnt) | barrier | MATH segment |
// barrier, waves 4..7 one barrier behind waves 0..3.
//
// LDS: [B ring: 4 slots (= the 4 taps of a chunk) x 4 parities x 64 columns x 64 B = 64 KB][patch buffer 0][patch buffer 1],
// a patch buffer = NPW * 8 pieces of 1 KB (16 pixels each).  16-byte slots of a 64-byte row are XOR-swizzled with
// ((row >> 2) & 1) << 1, which makes the 16-row fragment reads conflict-free at EVERY pixel offset (the shifts move the base).
// Ordering (s = K step; group 0 runs LOAD(s) in barrier interval 2s, group 1 in 2s+1):
//   B tiles of step s+2 are issued in LOAD(s) (2 pieces per wave) into ring slot (s+2) % 4, last read in LOAD(s-2): WAR needs
//     the issue >= 2 phases after the last reading LOAD segment (its reads are only known complete at the reader's lgkmcnt(0));
//   patch pieces of chunk c+1 are issued in LOAD(4c+1) and LOAD(4c+2) into buffer (c+1) & 1, last read in LOAD(4c-1);
//   RAW: everything first read in LOAD(s+1) is covered by the vmcnt at the end of LOAD(s) of every wave: at that point only
//     the pieces issued in LOAD(s) itself may be outstanding (2 B pieces + the patch pieces of that step).
// (Round 4: templated on log2 of the grid width so that every fragment address is `per-lane register + immediate`, the chunk loop unrolled
// by the patch buffer, piece offsets = a loop-invariant register + a scalar chunk offset, chunks past the end through a zero-record
// descriptor: no vector instruction and no address arithmetic left in a LOAD segment, as in conv_gemm_pt_kernel.)
template <typename T, int LW>
__global__ __launch_bounds__(512) void conv_par_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NPW = LW == 7 ? 5 : 4;                       // patch pieces per wave: ceil((256 / Wg + 2) * (Wg + 8) / 16 / 8)
  constexpr int BM = 256, PC = 64, BN = 4 * PC, CKB = 64;    // rows, channels per parity, virtual columns, bytes of K per LDS row
  constexpr int BSTEP = 4 * PC * CKB;                        // B tiles of one K step (4 parities): 16 KB
  constexpr int BRING = 4 * BSTEP;
  constexpr int PATCH = NPW * 8 * 1024;
  constexpr int P1 = (NPW + 1) / 2, P2 = NPW / 2;            // patch pieces per wave issued in steps 4c+1 / 4c+2
  constexpr int ES = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  DIAG_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, par = wave & 3, py = par >> 1, px = par & 1;
  const int r = lane & 15, q = lane >> 4;

  int bid = blockIdx.x;
  const int nb = p.tilesM * p.tilesN;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int bm0 = (bid / p.tilesN) * BM, bn0 = (bid % p.tilesN) * PC;
  constexpr int Wg = 1 << LW, PW = Wg + 8;                   // patch row: halo | Wg pixels | halo | 6 unused (PW % 8 == 0, below)
  constexpr int lw = LW;                                     // Wg is a power of two (16..128)
  const int Rrows = BM >> lw;
  // tile origin (R whole rows of one image)
  const int img = bm0 / (p.Hg * Wg), gy0 = (bm0 - img * p.Hg * Wg) >> lw;

  // ---- staging state ---------------------------------------------------------------------------------------------------
  const int spix = lane >> 2, sslot = lane & 3;              // this lane's pixel / 16-byte slot inside a 1-KB piece
  int a_src[NPW];                                            // patch pieces wave + 8j: source byte offset of chunk 0, or out of range
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int pp = (wave + 8 * j) * 16 + spix;
    const int prow = pp / PW, pcol = pp - prow * PW;
    const int y = gy0 - 1 + prow, x = pcol - 1;
    int off = (int)0x80000000;
    if (prow < Rrows + 2 && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws)
      off = (int)((((long long)(img * p.Hs + y) * p.Ws + x) * (long long)p.xpitch) * ES) + ((sslot ^ (((pp >> 2) & 1) << 1)) << 4);
    a_src[j] = off;
  }
  int b_src;                                                 // B pieces wave and wave + 8: columns (wave & 3) * 16 + spix of parities (wave >> 2), +2
  {
    const int col = (wave & 3) * 16 + spix, n = bn0 + col;
    b_src = n < p.Wrows ? (int)((size_t)n * p.Cin * ES) + ((sslot ^ (((col >> 2) & 1) << 1)) << 4) : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);     // zero records: chunks past the end
  const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0u, 0x00020000);
  const int wtapbytes = p.Wrows * p.Cin * ES;
  const int nchunks = p.kchunks;                             // channel chunks of CKB bytes

  auto issue_b = [&](auto Tc, int cc) {                      // weights of (chunk cc, tap T) for the 4 parities -> ring slot T
    constexpr int t = decltype(Tc)::value, ty = t >> 1, tx = t & 1;
    const bool live = cc < nchunks;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pb = (wave >> 2) + 2 * h, pyb = pb >> 1, pxb = pb & 1;
      const int wtap = (1 - pyb + 2 * ty) * 4 + (1 - pxb + 2 * tx);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? rw : rw0, (__attribute__((address_space(3))) void*)(smem + t * BSTEP + (wave + 8 * h) * 1024), 16, b_src,
                                               wtap * wtapbytes + cc * CKB, 0, 0);
    }
  };
  auto issue_patch = [&](auto J0c, auto CNTc, int cc) {      // pieces wave + 8j, j in [J0, J0 + CNT), of chunk cc
    constexpr int J0 = decltype(J0c)::value, CNT = decltype(CNTc)::value;
    const bool live = cc < nchunks;
    unsigned char* dst = smem + BRING + (cc & 1) * PATCH;
    static_for<CNT>([&](auto Jc) {
      constexpr int j = J0 + decltype(Jc)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? rx : rx0, (__attribute__((address_space(3))) void*)(dst + (wave + 8 * j) * 1024), 16, a_src[j], cc * CKB, 0, 0);
    });
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing -----------------------------------------------------------------------------------------------
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // B: column j*16 + r of this wave's parity, slot q (swizzle term depends on r only)
  const unsigned b_row = lds_base + par * (PC * CKB) + r * CKB + (((unsigned)q ^ ((((unsigned)r >> 2) & 1) << 1)) << 4);
  // A: fragment i = tile rows wr*128 + i*16 + (0..15): 16 consecutive pixels of one image row (Wg >= 16), patch pixel
  // (row_i + 1 + dy) * PW + col_i + r + 1 + dx at shift (dy, dx).  PW % 8 == 0 and col_i % 16 == 0, so the swizzle term depends
  // on r + 1 + dx only: byte address = [wave-uniform (row_i + 1 + dy) * PW + col_i] * 64 + lane_off[dx], one add per fragment.
  unsigned a_base[2];                                        // tx = 0, 1 -> dx = px - tx; + this wave's row half and its parity's row shift
#pragma unroll
  for (int tx = 0; tx < 2; ++tx) {
    const unsigned c = (unsigned)(r + 1 + px - tx);
    a_base[tx] = lds_base + BRING + (c << 6) + ((((unsigned)q) ^ (((c >> 2) & 1) << 1)) << 4) + (unsigned)((wr * (128 >> lw) + py) * PW * CKB);
  }
  // fragment i of (tap row ty, patch buffer b): a_base[tx] + [(16 i >> lw) * PW + (16 i & (Wg - 1)) + (1 - ty) * PW] * 64 + b * PATCH: an immediate

  DIAG_STAMP(1);
  // prologue: patch of chunk 0, weights of steps 0 and 1
  issue_patch(std::integral_constant<int, 0>{}, std::integral_constant<int, NPW>{}, 0);
  issue_b(std::integral_constant<int, 0>{}, 0);
  issue_b(std::integral_constant<int, 1>{}, 0);
  wait_vmcnt<2>();
  __builtin_amdgcn_s_barrier();
  DIAG_STAMP(2);
  if (wr == 1) __builtin_amdgcn_s_barrier();                 // waves 4..7 run one barrier behind

  uint4 af[8], bfr[4];
  SEG_DECL;
  SEG_T0;
  auto chunk = [&](auto PBc, int c) {                        // one channel chunk: 4 K steps (taps) on patch buffer PB
    constexpr int PB = decltype(PBc)::value;
    static_for<4>([&](auto Tc) {
      constexpr int t = decltype(Tc)::value, ty = t >> 1, tx = t & 1;
      // ---- LOAD segment ----
      static_for<4>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        lds_read128<t * BSTEP + j * 16 * CKB>(bfr[j], b_row);
      });
      static_for<8>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        constexpr int imm = (((16 * i) >> lw) * PW + ((16 * i) & (Wg - 1)) + (1 - ty) * PW) * CKB + PB * PATCH;
        static_assert(imm >= 0 && imm < 65536, "ds_read immediate");
        lds_read128<imm>(af[i], a_base[tx]);
      });
      SEG_ADD(0);                                            // fragment read issue
      if constexpr (t < 2) issue_b(std::integral_constant<int, t + 2>{}, c);
      else issue_b(std::integral_constant<int, t - 2>{}, c + 1);
      if constexpr (t == 1) issue_patch(std::integral_constant<int, 0>{}, std::integral_constant<int, P1>{}, c + 1);
      if constexpr (t == 2) issue_patch(std::integral_constant<int, P1>{}, std::integral_constant<int, P2>{}, c + 1);
      SEG_ADD(5);                                            // LDS-DMA issue
      wait_vmcnt<2 + (t == 1 ? P1 : (t == 2 ? P2 : 0))>();
      SEG_ADD(1);                                            // waiting for older pieces
      __builtin_amdgcn_s_barrier();
      SEG_ADD(2);                                            // waiting for the partner group's MATH segment
      // ---- MATH segment ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::run(acc[i][j], bfr[j], af[i]);
      __builtin_amdgcn_s_setprio(0);
      SEG_ADD(3);                                            // fragment wait + MFMAs
      __builtin_amdgcn_s_barrier();
      SEG_ADD(4);                                            // waiting for the partner group's LOAD segment
    });
  };
  for (int c = 0; c < nchunks; c += 2) {
    chunk(std::integral_constant<int, 0>{}, c);
    if (c + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, c + 1);
  }
  SEG_STORE;
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  __syncthreads();
  DIAG_STAMP(3);
  gemm_epilogue<T, BM, BN, 2, 4, BRING + 2 * PATCH, true>(p, acc, smem, bm0, bn0, 0, 4, 0);
  DIAG_STAMP(4);
#endif
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

// 4 channels per thread: float4 slab reads, one 8/16-byte store.  Needs Cout % 4 == 0 and a 4-element aligned destination
// (vec_store layers).  rg_arg: row groups per workgroup of a launch that emits statistics partials (an argument, not a GemmParams
// field: one more int in the struct cost conv_gemm_kernel<64,128> 16 more spilled SGPRs and 20 % of its speed).
template <typename T, int U>       // U: row groups walked at a time (<= the row groups per workgroup)
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const GemmParams p, int P, int rg_arg) {
  const int c4 = p.Cout >> 2;
  const long long total = (long long)P * p.M * c4;
  // With fused statistics a workgroup walks RG consecutive groups of RB = 256 / c4 whole rows (all of one parity and one statistics group:
  // the planner guarantees it) and emits ONE chunk of partial sums for them: RG = 1 gave one chunk per 2 rows of a 512-channel layer,
  // thousands per launch, and the planner then left the statistics to a separate pass over the tensor (round 5).  The row groups are
  // walked U at a time with every load of the U groups requested before the first is used: one group after the other was a chain of
  // up to 16 dependent round trips (25 us for a launch that moves 50 MB).
  const int RG = p.stats ? rg_arg : 1;
  struct Raw4 { f32x4 f; uint2 h; };
  auto ld4raw = [](const void* base, size_t off) {
    Raw4 q;
    if constexpr (sizeof(T) == 4) q.f = *(const f32x4*)((const float*)base + off); else q.h = *(const uint2*)((const T*)base + off);
    return q;
  };
  auto unraw = [](const Raw4& q, float* out) {
    if constexpr (sizeof(T) == 4) { out[0] = q.f[0]; out[1] = q.f[1]; out[2] = q.f[2]; out[3] = q.f[3]; }
    else { float t8[8]; unpack16<T>(make_uint4(q.h.x, q.h.y, 0u, 0u), t8); out[0] = t8[0]; out[1] = t8[1]; out[2] = t8[2]; out[3] = t8[3]; }
  };
  float acc8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc8[e] = 0.f;
  const long long idx0 = (long long)blockIdx.x * RG * 256 + threadIdx.x;
  if (idx0 >= total && !p.stats) return;
  const int n = (int)(idx0 % c4) * 4;                          // the same for every row group of the thread (256 % c4 == 0 where RG > 1)
  const int m_first = (int)((idx0 / c4) % p.M), par_first = (int)((idx0 / c4) / p.M);
  const size_t sstride = (size_t)p.M * p.NslabPitch;
  const bool bf = p.bf_mode && n < p.bf_cols;
  const bool bfn = bf && p.bf_mode != 4;
  f32x4 mu = f32x4{0.f, 0.f, 0.f, 0.f}, rs = mu, ga = mu, be = mu;
  f32x4 bias4 = mu;
  if (bfn && idx0 < total) {                                    // one statistics group per workgroup: its constants once
    const int RBq = (256 / c4) * RG, grp = (m_first / RBq) / p.stats_tpg;
    mu = *(const f32x4*)(p.bf_mean + grp * p.bf_cols + n); rs = *(const f32x4*)(p.bf_rstd + grp * p.bf_cols + n);
    ga = *(const f32x4*)(p.bf_gamma + n); be = *(const f32x4*)(p.bf_beta + n);
  }
  if (p.bias && idx0 < total) bias4 = f32x4{p.bias[n], p.bias[n + 1], p.bias[n + 2], p.bias[n + 3]};
  for (int it0 = 0; it0 < RG; it0 += U) {
    bool ok[U];
    size_t pix[U];
    const float* src[U];
    Raw4 rraw[U], araw[U];
    uint32_t mraw[U];
    f32x4 sacc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long idx = idx0 + (long long)(it0 + u) * 256;
      ok[u] = it0 + u < RG && idx < total;
      const long long t = (ok[u] ? idx : (idx0 < total ? idx0 : 0)) / c4;     // a row that is not this thread's: a valid row read again, nothing stored
                                                                               // (unconditional loads: a branch per row serialised them)
      const int m = (int)(t % p.M), par = (int)(t / p.M);
      src[u] = p.slab + ((size_t)par * p.splits * p.M + m) * p.NslabPitch + n;
      pix[u] = out_pixel_index(p, m, par >> 1, par & 1);
      sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      mraw[u] = 0u;
      if (bf) {
        if (p.bf_add) araw[u] = ld4raw(p.bf_add, pix[u] * (size_t)p.bf_addpitch + n);
        rraw[u] = ld4raw(p.bf_ref, pix[u] * (size_t)p.bf_refpitch + n);
        if (p.bf_mode == 3) mraw[u] = *(const uint32_t*)(p.bf_mask + pix[u] * (size_t)p.bf_maskpitch + n);
      }
    }
    int k = 0;
    for (; k + 4 <= p.splits; k += 4) {                         // 16 slab loads in flight
      f32x4 a[U], b[U], c[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = *(const f32x4*)(src[u] + (size_t)k * sstride); b[u] = *(const f32x4*)(src[u] + (size_t)(k + 1) * sstride);
        c[u] = *(const f32x4*)(src[u] + (size_t)(k + 2) * sstride); d[u] = *(const f32x4*)(src[u] + (size_t)(k + 3) * sstride);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { sacc[u] += a[u]; sacc[u] += b[u]; sacc[u] += c[u]; sacc[u] += d[u]; }
    }
    for (; k + 2 <= p.splits; k += 2) {
      f32x4 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = *(const f32x4*)(src[u] + (size_t)k * sstride); b[u] = *(const f32x4*)(src[u] + (size_t)(k + 1) * sstride); }
#pragma unroll
      for (int u = 0; u < U; ++u) { sacc[u] += a[u]; sacc[u] += b[u]; }
    }
    for (; k < p.splits; ++k) {
#pragma unroll
      for (int u = 0; u < U; ++u) sacc[u] += *(const f32x4*)(src[u] + (size_t)k * sstride);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act(sacc[u][e] + bias4[e], p.act, p.slope);
      const size_t o = pix[u] * (size_t)p.ypitch + n;
      float sx[4] = {0.f, 0.f, 0.f, 0.f};       // fused backward epilogue: dz * xhat
      if (bf) {                                  // GanBwdFuse on 4 channels of one pixel (same arithmetic as bwd_fuse_vec)
        float rf[4], a2[4];
        if (!p.out_f32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (float)(T)v[e];     // da as the unfused path would have stored and re-read it
        }
        if (p.bf_add) {
          unraw(araw[u], a2);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += a2[e];
        }
        unraw(rraw[u], rf);
        if (p.bf_mode == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rf[e] > 0.f ? v[e] : v[e] * p.bf_slope;
        } else {
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (p.bf_mode == 3) {
            const uint32_t w = mraw[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) mk[e] = 2.f * (float)((w >> (8 * e)) & 0xff);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (rf[e] - mu[e]) * rs[e];
            const float z = fmaf(ga[e], xh, be[e]);
            const float zd = z * mk[e];
            float d = v[e] * mk[e];
            d = zd > 0.f ? d : (p.bf_mode == 1 ? d * p.bf_slope : 0.f);
            v[e] = d; sx[e] = d * xh;
          }
        }
      }
      if (p.out_f32) *(f32x4*)((float*)p.y + o) = f32x4{v[0], v[1], v[2], v[3]};
      else *(uint2*)((T*)p.y + o) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float w = p.out_f32 ? v[e] : (float)(T)v[e];                  // as stored (bf16-rounded on the fast path)
          acc8[2 * e] += p.bf_mode ? v[e] : w; acc8[2 * e + 1] += p.bf_mode ? sx[e] : w * w;
        }
      }
    }
  }   // row groups
  if (p.stats) {
    // fused normalisation statistics of a split-K layer: per-channel (sum, sum^2) of the STORED values over the workgroup's RG * RB rows
    // = one chunk, laid out like the tile partials of the unsplit epilogue: [group][chunk][C][2]
    // (fused backward epilogue: (sum dz, sum dz*xhat) of the channels < bf_cols instead)
    __shared__ float red[256][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc8[e];
    __syncthreads();
    const int RB = 256 / c4;
    if ((int)threadIdx.x < c4) {
      float tot8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tot8[e] = 0.f;
      for (int rr = 0; rr < RB; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) tot8[e] += red[rr * c4 + threadIdx.x][e];
      const int n0 = (int)threadIdx.x * 4;
      const int mb = m_first / (RB * RG);                      // this thread's first row is the workgroup's first row
      const int rpb = p.stats_tpg;                             // chunks per group (per parity)
      const int grp = mb / rpb, chunk = (mb % rpb) * P + par_first;
      float* dst = p.stats + (((size_t)grp * rpb * P + chunk) * p.stats_C + n0) * 2;
      if (n0 < p.stats_C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = tot8[e];
      }
    }
  }
}

template <typename T, int MODE, int KR>
__global__ __launch_bounds__(256) void splitk_norm_kernel(const GemmParams p, int P) {
  splitk_norm_body<T, MODE, KR, false>(p, P, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
// the same on 512 threads (256 row slots): groups of more than conv.skn512_min_rows rows - half the rows per thread
template <typename T, int MODE, int KR>
__global__ __launch_bounds__(512) void splitk_norm512_kernel(const GemmParams p, int P) {
  splitk_norm_body<T, MODE, KR, false, 0, 512>(p, P, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// ------------------------------------------------------------------------------------------------
// pipeline shape per tile (measured): 128-byte K rows and 2 stages for the 8-wave 256-row tiles (one block per CU),
// 64-byte rows for 256x64 and 2 stages for 128x128 so that two blocks share a CU, 3 stages with counted vmcnt for the
// 4-wave 128x64 / 64x128 tiles.
#ifndef GAN_NS64
#define GAN_NS64 3        // stages of the 64x128 tile (experiment hook: -DGAN_NS64=2)
#endif
static constexpr int cfg_bkb(int BM, int BN) { return (BM == 256 && BN == 64) ? 64 : 128; }   // 256x64: two blocks per CU
static constexpr int cfg_ns(int BM, int BN) {
  // (round 5: 5 stages on the <= 64-row tiles - every K tile of a 32-way split block in flight at once - measured +-0 / -6 % on those
  // kernels alone and -2 % on the step: at 127 KB of LDS a block no longer shares its CU with the other lanes' kernels)
  return BM == 256 ? 2 : (BM == 128 && BN == 128) ? 2 : (BM == 128 && BN >= 64) ? 3 : (BM == 64 && BN == 128) ? GAN_NS64 : 2;
}

struct GemmPlan {
  GemmParams p;
  int BM, BN, P, stats_chunks;
  int stats_rg;            // split-K slab reduce emitting the partials: row groups (of 256 / (Cout / 4) rows) per workgroup = per chunk
  bool pp;                 // 256-row tile on the ping-pong kernel
  int ps_sh;               // > 0: on the tap-shared ping-pong kernel, taps per staged A tile (2 | 4)
  bool ps_table;           // ... in its table-driven form (conv_gemm_pt_kernel)
  int par_npw;             // > 0: parity-patch kernel (all four parities of 256 grid positions per block), patch pieces per wave
  bool own;                // column-owner kernel (conv_own.hip): the whole GanNormFuse layer in one launch, no slabs
  bool bf_requested;       // the caller asked for a fused backward epilogue (carried iff p.bf_mode != 0)
  dim3 grid;
  size_t slab_bytes;
};

static int plan_gemm(const GanConvDesc* d, int op, GemmPlan* pl) {
  if (!d || d->struct_size != sizeof(GanConvDesc) || !d->x.ptr || !d->y.ptr || !d->w) return GAN_E_ARG;
  if (!gan_dtype_ok(d->dtype)) return GAN_E_ARG;
  const int vec = d->dtype == GAN_F32 ? 4 : 8;
  const GanTensor &x = d->x, &y = d->y;
  if (x.c <= 0 || x.c % 8 || x.pitch % 8 || x.pitch < x.c || y.pitch < y.c || y.c <= 0) return GAN_E_SHAPE;
  if (x.n != y.n || d->w_rows < y.c) return GAN_E_SHAPE;
  if (((uintptr_t)x.ptr | (uintptr_t)d->w) & 15) return GAN_E_ARG;   // 16-byte LDS-DMA pieces
  int l2 = ilog2_exact(x.c / vec);
  if (l2 < 0) return GAN_E_SHAPE;
  GemmParams& p = pl->p;
  p.x = x.ptr; p.w = d->w; p.y = y.ptr; p.bias = d->bias; p.slab = (float*)d->workspace;
  {
    const size_t es = d->dtype == GAN_F32 ? 4 : 2;
    const size_t xb = (((size_t)x.n * x.h * x.w - 1) * x.pitch + x.c) * es, wb = (size_t)16 * d->w_rows * x.c * es;
    if (xb >= 0x7fffffffull || wb >= 0x7fffffffull) return GAN_E_SHAPE;
    p.xbytes = (unsigned)xb; p.wbytes = (unsigned)wb;
  }
  p.Nimg = x.n; p.Hs = x.h; p.Ws = x.w; p.xpitch = x.pitch; p.Cin = x.c; p.log2_cvecs = l2;
  p.Wrows = d->w_rows; p.Ho = y.h; p.Wo = y.w; p.ypitch = y.pitch; p.Cout = y.c;
  p.act = d->act; p.slope = d->slope; p.out_f32 = d->y_f32 || d->dtype == GAN_F32;
  p.vec_store = (!d->y_f32 || d->dtype == GAN_F32) && y.c % vec == 0 && y.pitch % vec == 0 && ((uintptr_t)y.ptr % 16) == 0;
  p.parity = 0; p.OS = 1; p.wy0 = p.wx0 = 0; p.wstep = 1; p.TWlog2 = 2; p.T = 16; p.log2T = 4;
  const bool parity = (op == 1 && d->stride == 2) || op == 2;
  if (parity) {            // convT forward / stride-2 conv dgrad: 4 output-parity sub-GEMMs
    if (d->stride != 2 || y.h != 2 * x.h || y.w != 2 * x.w) return GAN_E_SHAPE;
    p.parity = 1; p.OS = 2; p.S = 1; p.TWlog2 = 1; p.T = 4; p.log2T = 2; p.dstep = -1; p.wstep = 2;
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
  p.divWg = make_fastdiv((uint32_t)p.Wg); p.divHg = make_fastdiv((uint32_t)p.Hg);
  const long long Kbytes = (long long)p.T * x.c * (d->dtype == GAN_F32 ? 4 : 2);
  if (Kbytes % 128) return GAN_E_SHAPE;
  const int P = parity ? 4 : 1;
  pl->P = P;
  // Tile choice.  The 128x128 tile needs ~38 TB/s of L2->LDS traffic at the MFMA peak (65 FLOP per byte
  // staged), more than the L2s deliver, so big layers use 256-row tiles (512 threads, one block per CU);
  // efficiency of a candidate = relative tile quality x how full its last round of blocks is.
  int BN = y.c > 64 ? 128 : (y.c > 16 ? 64 : 16);
  int BM = 128;
  if (((M + 127) / 128) * ((y.c + BN - 1) / BN) * P < 192) BM = 64;
  if (BM == 64 && M <= 32 && BN != 16) BM = 16;
  if (y.c >= 128 && M >= 256 && gan_opt("conv.big_tiles")) {
    auto fill = [](long long blocks, long long slots) { return (double)blocks / (double)(((blocks + slots - 1) / slots) * slots); };
    const long long b128 = ((M + 127) / 128) * ((y.c + 127) / 128) * P;
    const long long b256n = ((M + 255) / 256) * ((y.c + 127) / 128) * P;
    const long long b256 = ((M + 255) / 256) * ((y.c + 255) / 256) * P;
    const int q128 = gan_opt("conv.q128"), q256n = gan_opt("conv.q256n"), minb = gan_opt("conv.big_min_blocks");
    double best = 0.01 * q128 * fill(b128, 512);
    if (b256n >= minb && 0.01 * q256n * fill(b256n, 256) > best) { best = 0.01 * q256n * fill(b256n, 256); BM = 256; BN = 128; }
    if (y.c >= 256 && b256 >= minb && 1.0 * fill(b256, 256) > best) { BM = 256; BN = 256; }
  }
  // 64-channel outputs on big maps: 256x64 tiles stage 17 % fewer bytes per FLOP than 128x64 and halve the tile count
  const int tall64 = gan_opt("conv.tall64");
  if (tall64 && BN == 64 && BM == 128 && ((M + 255) / 256) * P >= 1024) BM = 256;
  // Parity-patch kernel (conv_par_kernel): stride-2 transposed conv / stride-2 conv dgrad with 64 output channels per block,
  // all four parities from one staged input patch.  Needs tiles of whole image rows and enough of them to fill the chip.
  pl->par_npw = 0;
  {
    const int use_par = gan_opt("conv.parity_patch"), par_maxn = gan_opt("conv.parity_patch_max_n"), par_minb = gan_opt("conv.parity_patch_min_blocks");
    const int Wg = p.Wg;
    if (use_par && parity && d->dtype != GAN_F32 && y.c % 64 == 0 && y.c <= par_maxn && x.c % 32 == 0 && p.vec_store &&
        Wg >= 16 && Wg <= 128 && (Wg & (Wg - 1)) == 0 && p.Hg % (256 / Wg) == 0 && M % 256 == 0 &&
        (M / 256) * (y.c / 64) >= par_minb) {
      pl->par_npw = (((256 / Wg + 2) * (Wg + 8) + 15) / 16 + 7) / 8;      // ceil(patch pixels / 16) pieces over 8 waves (rows of Wg + 8)
      BM = 256; BN = 64;
    }
  }
  p.kchunks = pl->par_npw ? (int)(((long long)x.c * 2) / 64) : (int)(Kbytes / cfg_bkb(BM, BN));
  int tilesN = (y.c + BN - 1) / BN;
  long long tilesM = (M + BM - 1) / BM;
  long long blocks = tilesM * tilesN * P;
  int splits = 1;
  const int t_small = gan_opt("conv.split_target"), t_skinny = gan_opt("conv.split_target_skinny"), t_big = gan_opt("conv.split_target_big");
  // 256-row tiles: a half-full chip (128..160 blocks) is worth a 2-way K split only when K is long enough
  // to amortise the fp32 slab round trip (measured: K>=4096 +25..40 %, K=2048 neutral)
  const long long target = BM == 256 ? (p.kchunks >= 48 && blocks <= 160 ? t_big : gan_opt("conv.split_target_256")) : (BN == 16 ? t_skinny : t_small);   // BN=16: streaming layers want more, shorter blocks
  if (blocks < target && !pl->par_npw) {
    splits = (int)((target + blocks - 1) / blocks);
    const int mink = gan_opt("conv.split_min_ktiles"), maxsp = gan_opt("conv.split_max");
    int maxs = p.kchunks / mink; if (maxs < 1) maxs = 1;
    if (splits > maxs) splits = maxs;
    if (splits > maxsp) splits = maxsp;
  }
  p.splits = splits;
  p.stats = nullptr; p.stats_tpg = 0; p.stats_C = y.c; pl->stats_rg = 1;
  pl->stats_chunks = 0;
  // fused backward epilogue request (GanBwdFuse): validated here, honoured below if this launch shape can carry it
  const GanBwdFuse* bf = d->bwd_fuse;
  int bf_mode = 0;
  p.bf_ref = p.bf_add = nullptr; p.bf_mean = p.bf_rstd = p.bf_gamma = p.bf_beta = nullptr; p.bf_mask = nullptr;
  p.bf_refpitch = p.bf_addpitch = p.bf_maskpitch = p.bf_mode = p.bf_cols = 0; p.bf_slope = 0.f;
  pl->bf_requested = bf != nullptr;
  if (bf) {
    if ((op != 1 && op != 3) || !bf->ref.ptr || bf->cols <= 0 || bf->cols % 8 || bf->cols > y.c || d->bias || d->act != GAN_ACT_NONE || d->y_f32)
      return GAN_E_ARG;
    if (bf->ref.n != y.n || bf->ref.h != y.h || bf->ref.w != y.w || bf->ref.c < bf->cols || bf->ref.pitch % vec ||
        ((uintptr_t)bf->ref.ptr & 15))
      return GAN_E_SHAPE;
    if (bf->add.ptr && (bf->add.n != y.n || bf->add.h != y.h || bf->add.w != y.w || bf->add.c < bf->cols || bf->add.pitch % vec ||
                        ((uintptr_t)bf->add.ptr & 15)))
      return GAN_E_SHAPE;
    if (bf->mean) {
      if (!bf->rstd || !bf->gamma || !bf->beta || d->stats_groups <= 0 || !d->stats_partial) return GAN_E_ARG;
      if (bf->dropmask && (bf->mask_pitch % 8 || bf->mask_pitch < bf->cols || ((uintptr_t)bf->dropmask & 7))) return GAN_E_SHAPE;
      bf_mode = bf->act == GAN_ACT_LRELU && !bf->dropmask ? 1 : bf->act == GAN_ACT_RELU ? (bf->dropmask ? 3 : 2) : 0;
    } else {
      bf_mode = bf->act == GAN_ACT_LRELU && !bf->dropmask ? 4 : 0;
    }
    if (!bf_mode) return GAN_E_ARG;
    p.stats_C = bf_mode == 4 ? y.c : bf->cols;
  }
  const bool reduce4_ok = p.vec_store && y.c % 4 == 0 && (p.out_f32 || d->dtype != GAN_F32);
  const bool want_stats = d->stats_groups > 0 && bf_mode != 4;
  // GanNormFuse: a small split-K layer is finished by its slab-reduce kernel (splitk_norm_kernel): <= 1024 rows per group
  p.skn = 0; p.skn_groups = 0;
  if (const GanNormFuse* nf = d->norm_fuse; nf && gan_opt("conv.norm_fuse") && splits > 1 && reduce4_ok && want_stats && y.c % 8 == 0 &&
      !d->bias && d->act == GAN_ACT_NONE && x.n % d->stats_groups == 0 && M % d->stats_groups == 0 &&
      (long long)P * (M / d->stats_groups) <= 1024) {
    const int G = d->stats_groups;
    const int oc = bf_mode ? bf->cols : y.c;
    const bool out_ok = nf->out.ptr && nf->out.n == y.n && nf->out.h == y.h && nf->out.w == y.w && nf->out.c >= oc && nf->out.pitch % 4 == 0 &&
                        ((uintptr_t)nf->out.ptr & 7) == 0;
    if (!bf_mode && out_ok && nf->gamma && nf->beta && nf->mean && nf->rstd && (!nf->moving_mean || nf->moving_var)) {
      p.skn = 1;
    } else if (bf_mode >= 1 && bf_mode <= 3 && out_ok && (G == 1 || (long long)P * M <= 4096) && bf->cols % 8 == 0) {
      p.skn = 2;
      p.bf_mode = bf_mode; p.bf_cols = bf->cols; p.bf_slope = bf->slope;
      p.bf_ref = bf->ref.ptr; p.bf_refpitch = bf->ref.pitch;
      p.bf_add = bf->add.ptr; p.bf_addpitch = bf->add.pitch;
      p.bf_mean = bf->mean; p.bf_rstd = bf->rstd; p.bf_gamma = bf->gamma; p.bf_beta = bf->beta;
      p.bf_mask = bf->dropmask; p.bf_maskpitch = bf->mask_pitch;
    }
    if (p.skn) {
      p.skn_groups = G; p.skn_out = nf->out.ptr; p.skn_outpitch = nf->out.pitch;
      p.skn_gamma = nf->gamma; p.skn_beta = nf->beta; p.skn_mean = nf->mean; p.skn_rstd = nf->rstd;
      p.skn_mmean = nf->moving_mean; p.skn_mvar = nf->moving_var; p.skn_eps = nf->eps; p.skn_momentum = nf->momentum;
      p.skn_mask = nf->dropmask; p.skn_act = nf->act; p.skn_slope = nf->slope;
      p.skn_dgamma = nf->dgamma; p.skn_dbeta = nf->dbeta; p.skn_accumulate = nf->accumulate;
      pl->stats_chunks = -1;                                   // plan_info()[4]: the launch finishes the layer
    }
  }
  if (p.skn) {
  } else
  if (want_stats && splits == 1 && p.vec_store && BN <= 64 * 8 && x.n % d->stats_groups == 0) {
    const long long rpg = M / d->stats_groups;            // GEMM rows per statistics group (per parity)
    // groups are whole numbers of M tiles - or there is ONE group, whose last tile may be ragged (the epilogue sums rows < M only):
    // PatchGAN's 31 x 31 layer (base_gan.py:146-151) with D(real) and D(fake) as separate launches
    if ((rpg % BM == 0 || d->stats_groups == 1) && (BM == 256 ? 512 : 256) >= BN) {
      p.stats_tpg = (int)((rpg + BM - 1) / BM);
      pl->stats_chunks = p.stats_tpg * P;
      p.stats = d->stats_partial;
    }
  } else if (want_stats && splits > 1 && reduce4_ok && x.n % d->stats_groups == 0) {
    // split-K layer: the slab-reduce kernel emits the partials (one chunk per 256-thread block = RB whole rows)
    const int c4 = y.c / 4;
    const long long rpg = M / d->stats_groups;
    if (c4 <= 256 && 256 % c4 == 0 && rpg % (256 / c4) == 0) {
      // one chunk per workgroup = RG groups of RB = 256 / c4 whole rows: the smallest RG <= 16 that brings the launch under ~512 chunks
      // (beyond ~1k chunks the finalize's walk costs more than a separate pass over the tensor) and still divides the group's rows
      const long long rb = 256 / c4;
      int rg = 1;
      const int rgmax = gan_opt("conv.reduce_stats_rg");
      while (rg < rgmax && rpg / (rb * rg) * P > 512 && rpg % (rb * rg * 2) == 0) rg *= 2;
      if (rpg / (rb * rg) * P <= 1024) {
        p.stats_tpg = (int)(rpg / (rb * rg)); pl->stats_rg = rg;
        pl->stats_chunks = p.stats_tpg * P;
        p.stats = d->stats_partial;
      }
    }
  }
  if (bf_mode && !p.skn) {
    // the slab-reduce kernel of a split-K launch carries it at no cost (streaming kernel, the slabs are read anyway); a tile
    // epilogue re-reads the reference tensor at the tile's strided pixel order - break-even per launch on the 128/256-column
    // tiles, a gain on the 64-column ones (their act_bwd pass streamed 4 tensors): option conv.bwd_fuse_tile = 0 never, 1 always,
    // 2 not on the 64-column tiles, 3 only on them
    const int bf_tile = gan_opt("conv.bwd_fuse_tile");   // (the op tests exercise every carrier)
    const bool tile_ok = bf_tile == 1 || (bf_tile == 2 && BN != 64) || (bf_tile == 3 && BN == 64);
    const bool carrier = p.vec_store && (splits == 1 ? tile_ok : reduce4_ok);
    if (carrier && (bf_mode == 4 || p.stats)) {
      p.bf_mode = bf_mode; p.bf_cols = bf->cols; p.bf_slope = bf->slope;
      p.bf_ref = bf->ref.ptr; p.bf_refpitch = bf->ref.pitch;
      p.bf_add = bf->add.ptr; p.bf_addpitch = bf->add.pitch;
      p.bf_mean = bf->mean; p.bf_rstd = bf->rstd; p.bf_gamma = bf->gamma; p.bf_beta = bf->beta;
      p.bf_mask = bf->dropmask; p.bf_maskpitch = bf->mask_pitch;
      if (bf_mode == 4) pl->stats_chunks = 1;          // plan_info()[4] > 0 = "the epilogue is fused"
    } else {
      p.stats = nullptr; p.stats_tpg = 0; pl->stats_chunks = 0; p.stats_C = y.c;
    }
  }
  p.NslabPitch = tilesN * BN;
  p.tilesM = (int)tilesM; p.tilesN = tilesN;
#ifdef GAN_DIAG
  p.diag = g_diag;
#endif
  const int use_pp = gan_opt("conv.pingpong");
  pl->pp = use_pp && BM == 256 && (BN == 256 || BN == 128) && ((long long)x.c * (d->dtype == GAN_F32 ? 4 : 2)) % 128 == 0;
  // tap-shared variant (conv_gemm_ps_kernel): the taps of a kernel row that read the same pixels one grid column apart share one staged
  // A tile: 2 for the stride-2 shapes and the parity sub-GEMMs, 4 for the stride-1 convolutions.  Needs the halo of a 64-row block
  // (one entry per image-row segment and shift) to fit its 16 LDS rows.  Option bit 0: 128-column tiles, bit 1: 256-column tiles.
  pl->ps_sh = 0;
  if (pl->pp && d->dtype != GAN_F32 && (gan_opt("conv.tap_share") & (BN == 256 ? 2 : 1))) {
    const int sh = (p.S == 2 || p.parity) ? 2 : 4;
    const int segs = (63 + p.Wg - 1) / p.Wg + 1;
    if ((1 << p.TWlog2) % sh == 0 && p.kchunks % sh == 0 && segs * (sh - 1) <= 16) pl->ps_sh = sh;
  }
  pl->ps_table = pl->ps_sh && BN == 128 && (gan_opt("conv.tap_share") & 4) && (p.kchunks >> (2 * p.TWlog2)) >= splits;
  pl->BM = BM; pl->BN = BN;
  pl->grid = dim3((unsigned)(tilesM * tilesN * (pl->par_npw ? 1 : P)), 1, (unsigned)splits);
  pl->slab_bytes = splits > 1 ? (size_t)P * splits * (size_t)M * p.NslabPitch * sizeof(float) : 0;
  pl->own = conv_own_eligible(p, P, d->dtype);
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr int BKB = cfg_bkb(BM, BN), NS = cfg_ns(BM, BN);
  constexpr size_t stage2 = (size_t)NS * (BM + BN) * BKB + 16 * BM * sizeof(int);   // stages + gather table
  constexpr size_t smem = stage2;   // the epilogue staging tile is carved out of the same area
  auto kern = conv_gemm_kernel<T, BM, BN, WM, WN, BKB, NS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  GAN_LAUNCH(kern, pl.grid, dim3(64 * WM * WN), smem, st, pl.p);
  GAN_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BN, bool BF>
static int launch_pp_v(const GemmPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr size_t smem = (size_t)(BN == 256 ? 2 : 3) * (256 + BN) * 128;
  auto kern = conv_gemm_pp_kernel<T, BN, BF>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  GAN_LAUNCH(kern, pl.grid, dim3(512), smem, st, pl.p);
  GAN_CHECK_LAUNCH();
  return 0;
}
template <typename T, int BN>
static int launch_pp(const GemmPlan& pl, hipStream_t st) {       // the lean instantiation unless the launch carries a fused backward epilogue
  const bool full = pl.p.bf_mode != 0 || !gan_opt("conv.lean_epilogue");
  return full ? launch_pp_v<T, BN, true>(pl, st) : launch_pp_v<T, BN, false>(pl, st);
}

template <typename T, int BN, int SH, int TWV, bool BF>
static int launch_pt_v(const GemmPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr size_t smem = (size_t)2 * (256 + 64) * 128 + (size_t)(BN == 256 ? 2 : 4) * BN * 128;
  if constexpr (sizeof(T) == 2) {
    auto kern = conv_gemm_pt_kernel<T, BN, SH, TWV, BF>;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    GAN_LAUNCH(kern, pl.grid, dim3(512), smem, st, pl.p);
    GAN_CHECK_LAUNCH();
    return 0;
  } else {
    return GAN_E_SHAPE;
  }
}
template <typename T, int BN, int SH, int PH, bool BF>
static int launch_ps_v(const GemmPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr size_t smem = (size_t)2 * (256 + 64) * 128 + (size_t)(BN == 256 ? 2 : 4) * BN * 128;
  if constexpr (sizeof(T) == 2) {
    auto kern = conv_gemm_ps_kernel<T, BN, SH, PH, BF>;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    GAN_LAUNCH(kern, pl.grid, dim3(512), smem, st, pl.p);
    GAN_CHECK_LAUNCH();
    return 0;
  } else {
    return GAN_E_SHAPE;       // (the planner keeps fp32 on the kernel above)
  }
}
template <typename T, int BN>
static int launch_ps(const GemmPlan& pl, hipStream_t st) {
  const bool full = pl.p.bf_mode != 0 || !gan_opt("conv.lean_epilogue");
  if constexpr (BN == 128) {
    if (pl.ps_table) {                        // table-driven form
      const int twv = 1 << pl.p.TWlog2;
      if (pl.ps_sh == 4) return full ? launch_pt_v<T, BN, 4, 4, true>(pl, st) : launch_pt_v<T, BN, 4, 4, false>(pl, st);
      if (twv == 4) return full ? launch_pt_v<T, BN, 2, 4, true>(pl, st) : launch_pt_v<T, BN, 2, 4, false>(pl, st);
      return full ? launch_pt_v<T, BN, 2, 2, true>(pl, st) : launch_pt_v<T, BN, 2, 2, false>(pl, st);
    }
  }
  constexpr int PH = BN == 256 ? 4 : 2;
  if (pl.ps_sh == 2) return full ? launch_ps_v<T, BN, 2, PH, true>(pl, st) : launch_ps_v<T, BN, 2, PH, false>(pl, st);
  return full ? launch_ps_v<T, BN, 4, PH, true>(pl, st) : launch_ps_v<T, BN, 4, PH, false>(pl, st);
}

template <typename T, int LW>
static int launch_par(const GemmPlan& pl, hipStream_t st) {
  static bool attr_set = false;
  constexpr int NPW = LW == 7 ? 5 : 4;
  constexpr size_t smem = (size_t)4 * 4 * 64 * 64 + 2 * (size_t)NPW * 8 * 1024;
  if constexpr (sizeof(T) == 2) {
    auto kern = conv_par_kernel<T, LW>;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    GAN_LAUNCH(kern, pl.grid, dim3(512), smem, st, pl.p);
    GAN_CHECK_LAUNCH();
    return 0;
  } else {
    return GAN_E_SHAPE;       // (the planner never sends the fp32 parity path here)
  }
}

template <typename T>
static int launch_gemm(const GemmPlan& pl, hipStream_t st) {
  int rc;
  if (pl.par_npw) {
    switch (pl.p.Wg) {                          // (the planner admits power-of-two grid widths 16..128; par_npw = 5 at 128, else 4)
      case 16: return launch_par<T, 4>(pl, st);
      case 32: return launch_par<T, 5>(pl, st);
      case 64: return launch_par<T, 6>(pl, st);
      case 128: return launch_par<T, 7>(pl, st);
      default: return GAN_E_SHAPE;
    }
  }
  const int key = pl.pp ? pl.BN : pl.BM * 1000 + pl.BN;
  switch (key) {
    case 256: rc = pl.ps_sh ? launch_ps<T, 256>(pl, st) : launch_pp<T, 256>(pl, st); break;
    case 128: rc = pl.ps_sh ? launch_ps<T, 128>(pl, st) : launch_pp<T, 128>(pl, st); break;
    case 256256: rc = launch_cfg<T, 256, 256, 2, 4>(pl, st); break;
    case 256128: rc = launch_cfg<T, 256, 128, 4, 2>(pl, st); break;
    case 128128: rc = launch_cfg<T, 128, 128, 2, 2>(pl, st); break;
    case 256064: rc = launch_cfg<T, 256, 64, 4, 2>(pl, st); break;
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
    if (pl.p.skn) {
      const dim3 g((unsigned)(pl.p.Cout / 8), (unsigned)((pl.p.skn == 1 && !(pl.p.skn_mmean && pl.p.skn_groups > 1)) ? pl.p.skn_groups : 1));
      const long long rg = (long long)pl.P * (pl.p.M / pl.p.skn_groups);
      const int min512 = gan_opt("conv.skn512_min_rows");
      if (min512 > 0 && rg >= min512 && rg > 256) {            // 512 threads: 256 row slots
        const int kr = rg <= 512 ? 2 : 4;
#define SKN_LAUNCH(MODE, KRV) GAN_LAUNCH((splitk_norm512_kernel<T, MODE, KRV>), g, dim3(512), 0, st, pl.p, pl.P)
        if (pl.p.skn == 1) { if (kr == 2) SKN_LAUNCH(1, 2); else SKN_LAUNCH(1, 4); }
        else { if (kr == 2) SKN_LAUNCH(2, 2); else SKN_LAUNCH(2, 4); }
#undef SKN_LAUNCH
      } else {
      const int kr = rg <= 128 ? 1 : rg <= 256 ? 2 : rg <= 512 ? 4 : 8;
#define SKN_LAUNCH(MODE, KRV) GAN_LAUNCH((splitk_norm_kernel<T, MODE, KRV>), g, dim3(256), 0, st, pl.p, pl.P)
      if (pl.p.skn == 1) { if (kr == 1) SKN_LAUNCH(1, 1); else if (kr == 2) SKN_LAUNCH(1, 2); else if (kr == 4) SKN_LAUNCH(1, 4); else SKN_LAUNCH(1, 8); }
      else { if (kr == 1) SKN_LAUNCH(2, 1); else if (kr == 2) SKN_LAUNCH(2, 2); else if (kr == 4) SKN_LAUNCH(2, 4); else SKN_LAUNCH(2, 8); }
#undef SKN_LAUNCH
      }
    } else if (pl.p.vec_store && pl.p.Cout % 4 == 0 && (pl.p.out_f32 || sizeof(T) == 2)) {
      long long total = (long long)pl.P * pl.p.M * (pl.p.Cout / 4);
      const long long per_wg = 256LL * (pl.p.stats ? pl.stats_rg : 1);
      const dim3 rgrid((unsigned)((total + per_wg - 1) / per_wg));
      const int rgs = pl.p.stats ? pl.stats_rg : 1;
      if (rgs >= 4) GAN_LAUNCH((splitk_reduce4_kernel<T, 4>), rgrid, dim3(256), 0, st, pl.p, pl.P, pl.stats_rg);
      else if (rgs >= 2) GAN_LAUNCH((splitk_reduce4_kernel<T, 2>), rgrid, dim3(256), 0, st, pl.p, pl.P, pl.stats_rg);
      else GAN_LAUNCH((splitk_reduce4_kernel<T, 1>), rgrid, dim3(256), 0, st, pl.p, pl.P, pl.stats_rg);
    } else {
      long long total = (long long)pl.P * pl.p.M * pl.p.Cout;
      GAN_LAUNCH(splitk_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pl.p, pl.P);
    }
    GAN_CHECK_LAUNCH();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Layer stack (round 4): a run of consecutive SMALL split-K layers - each a GEMM into fp32 slabs plus the slab reduce that finishes
// the layer (GanNormFuse: statistics + normalise + activation forward, dz / dy backward) - in ONE launch: a resident grid walks
// the layers, GEMM tiles and finishing work units are dealt round-robin to the workgroups, and a grid barrier separates the phases
// (2 per layer).  What crosses workgroups inside the launch (slabs, a layer's output = the next layer's GEMM operand, the skip
// gradient) is written through / read around the per-XCD L2s (COH = true bodies above), so the barrier itself is atomics only:
// hierarchical, 8 group counters (workgroup & 7 = its XCD under round-robin dispatch) -> a master counter -> 8 release words,
// all monotonic (no reset, valid from one launch to the next as long as the grid size of a plan never changes).  Measured 2.2 us per
// barrier at 256 workgroups (tools/probes/gridbar_probe.hip).  Every wait is bounded (~2 s): a grid that cannot become resident
// sets *err and runs to completion instead of hanging the GPU.  LDS <= 78 KB and 256 threads per workgroup: two of these kernels
// fit side by side on every CU (the two chains of the CycleGAN step).
struct StackLayer {
  GemmParams p;
  int P, tile;               // parities; tile 0 = 64 x 128 (4 waves 2 x 2), 1 = 16 x 128 (4 waves 1 x 4)
  int gx, items;             // GEMM work items = gx (block tiles) x splits
  int fin_mode, fin_kr, fin_gx, fin_gy;      // finishing work units = fin_gx (8-channel slices) x fin_gy (groups, or 1)
};
struct StackBar { unsigned long long grp[8][16]; unsigned long long master[16]; unsigned long long rel[8][16]; };

__device__ __forceinline__ void stack_barrier(StackBar* bar, unsigned nblocks, int* err) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 7, per = nblocks >> 3;
    const unsigned long long seen = __hip_atomic_load(&bar->rel[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long old = __hip_atomic_fetch_add(&bar->grp[g][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old % per == per - 1) {                                         // last of its group
      const unsigned long long m = __hip_atomic_fetch_add(&bar->master[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (m % 8 == 7)                                                   // last group: release everyone
        for (int k = 0; k < 8; ++k) __hip_atomic_fetch_add(&bar->rel[k][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(&bar->rel[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seen) {
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { *err = 1; break; }      // 2 s at 100 MHz: give up, never hang
    }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_stack_kernel(const StackLayer* __restrict__ layers, int n, StackBar* bar, int* err) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = (int)gridDim.x;
  for (int l = 0; l < n; ++l) {
    const StackLayer& L = layers[l];
    // ---- GEMM phase: block tiles x K splits -> fp32 slabs ----
    for (int item = (int)blockIdx.x; item < L.items; item += nb) {
      const int bx = item % L.gx, split = item / L.gx;
      if (L.tile == 0) conv_gemm_body<T, 64, 128, 2, 2, 128, 3, true>(L.p, smem, bx, split);
      else conv_gemm_body<T, 16, 128, 1, 4, 128, 2, true>(L.p, smem, bx, split);
      __syncthreads();                                                  // (the next item rebuilds the gather table in the same LDS)
    }
    stack_barrier(bar, (unsigned)nb, err);
    // ---- finishing phase: slab sums + the layer's normalisation (forward) / dz, dy (backward) ----
    for (int item = (int)blockIdx.x; item < L.fin_gx * L.fin_gy; item += nb) {
      const int bx = item % L.fin_gx, by = item / L.fin_gx;
#define SKN_BODY(MODE, KRV) splitk_norm_body<T, MODE, KRV, true>(L.p, L.P, bx, by, L.fin_gy)
      // (groups of up to 512 rows only: the 1024-row variant needs ~350 registers, and this kernel is held to 256 so that two
      // of its workgroups fit on a CU)
      if (L.fin_mode == 1) { if (L.fin_kr == 1) SKN_BODY(1, 1); else if (L.fin_kr == 2) SKN_BODY(1, 2); else SKN_BODY(1, 4); }
      else { if (L.fin_kr == 1) SKN_BODY(2, 1); else if (L.fin_kr == 2) SKN_BODY(2, 2); else SKN_BODY(2, 4); }
#undef SKN_BODY
      __syncthreads();
    }
    if (l + 1 < n) stack_barrier(bar, (unsigned)nb, err);
  }
#endif
}

struct StackHeader { uint32_t magic, n, grid, dtype; size_t smem; };
static constexpr uint32_t STACK_MAGIC = 0x474e5334u;

template <typename T>
static int launch_stack(const StackHeader* h, const void* dev_plan, void* bar, int32_t* err, hipStream_t st) {
  static size_t attr_smem = 0;
  auto kern = conv_stack_kernel<T>;
  if (h->smem > attr_smem) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) return (int)e;
    attr_smem = h->smem;
  }
  GAN_LAUNCH(kern, dim3(h->grid), dim3(256), h->smem, st, (const StackLayer*)((const char*)dev_plan + 256), (int)h->n, (StackBar*)bar, (int*)err);
  GAN_CHECK_LAUNCH();
  return 0;
}

static int run_gemm(const GanConvDesc* d, int op, gan_stream_t stream) {
  GemmPlan pl;
  int rc = plan_gemm(d, op, &pl);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int fam = thin_family(d, op, pl.p);
  if (pl.bf_requested && (fam || !pl.p.bf_mode)) return GAN_E_SHAPE;   // the caller must consult gan_conv_plan_info()[4] first
  // ... and likewise for GanNormFuse: the caller drops its follow-up normalisation launches on the strength of plan_info()[4] == -1,
  // so a request this launch's plan cannot honour (a planner option changed since, another shape) must fail, not silently skip
  if (d->norm_fuse && (fam || !pl.p.skn)) return GAN_E_SHAPE;
  if (!fam && pl.p.stats && pl.stats_chunks > 0 && (size_t)d->stats_groups * pl.stats_chunks * pl.p.stats_C * 2 * sizeof(float) > d->stats_partial_bytes)
    return GAN_E_WORKSPACE;                                            // the caller's partial-sums region is too small for this plan
  if (fam) return thin_launch(fam, d, pl.p, st);   // <= 8-channel streaming layers
  if (pl.own) return conv_own_launch(pl.p, pl.P, d->dtype, st);
  if (pl.slab_bytes > d->workspace_bytes || (pl.slab_bytes && !d->workspace)) return GAN_E_WORKSPACE;
  return d->dtype == GAN_F32 ? launch_gemm<float>(pl, st) : d->dtype == GAN_F16 ? launch_gemm<f16_t>(pl, st) : launch_gemm<bf16_t>(pl, st);
}

static void plan_only_desc(GanConvDesc* t) {   // planning only looks at shapes and alignment
  if (!t->x.ptr) t->x.ptr = (void*)16;
  if (!t->y.ptr) t->y.ptr = (void*)16;
  if (!t->w) t->w = (void*)16;
}

extern "C" {
int gan_conv2d_fwd(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 0, s); }
int gan_conv2d_dgrad(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 1, s); }
int gan_convT2d_fwd(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 2, s); }
int gan_convT2d_dgrad(const GanConvDesc* d, gan_stream_t s) { return run_gemm(d, 3, s); }
int gan_conv_plan_info(const GanConvDesc* d, int op, int32_t* info /*[5]: BM, BN, splits, parities, stats chunks*/) {
  if (!d || !info || d->struct_size != sizeof(GanConvDesc)) return GAN_E_ARG;    // before the copy: a caller built against a
  GemmPlan pl;                                                                    // smaller struct must not be read past its end
  GanConvDesc t = *d;
  plan_only_desc(&t);
  int rc = plan_gemm(&t, op, &pl);
  if (rc) return rc;
  info[0] = pl.par_npw ? 4 * pl.BM : pl.BM;      // parity-patch kernel: 1024 output pixels (4 parities x 256 positions) per block
  info[1] = pl.BN; info[2] = pl.own ? 1 : pl.p.splits; info[3] = pl.P;
  if (pl.own) { info[0] = 0; info[1] = 8; }          // column-owner kernel: 8 channels of every row per workgroup
  info[4] = pl.stats_chunks;          // > 0: this launch can emit normalisation-statistics partials (chunks per group)
  if (const int fam = thin_family(&t, op, pl.p)) { info[0] = 0; info[1] = fam; info[2] = 1; info[4] = 0; }   // thin.hip kernels
  return 0;
}
int gan_conv_tap_shared(const GanConvDesc* d, int op) {
  if (!d || d->struct_size != sizeof(GanConvDesc)) return GAN_E_ARG;
  GemmPlan pl;
  GanConvDesc t = *d;
  plan_only_desc(&t);
  const int rc = plan_gemm(&t, op, &pl);
  if (rc) return rc;
  return (thin_family(&t, op, pl.p) || pl.par_npw || !pl.pp) ? 0 : pl.ps_sh;
}
/* ---- layer stack: plan on the host, launch from a device copy of the plan ---- */
size_t gan_conv_stack_plan_bytes(int32_t n) { return n > 0 ? 256 + (size_t)n * sizeof(StackLayer) : 0; }

// one layer's plan as a stack layer; GAN_E_SHAPE when the launch cannot join a stack
static int stack_layer_plan(const GanConvDesc* d, int op, GemmPlan& pl, int* tile, long long* rg) {
  if (!d || d->struct_size != sizeof(GanConvDesc) || op < 0 || op > 3) return GAN_E_ARG;
  int rc = plan_gemm(d, op, &pl);
  if (rc) return rc;
  if (thin_family(d, op, pl.p) || !pl.p.skn || pl.p.splits <= 1 || pl.pp || pl.par_npw) return GAN_E_SHAPE;   // small split-K layers finished by their reduce only
  if (pl.bf_requested && !pl.p.bf_mode) return GAN_E_SHAPE;
  if (pl.BM == 64 && pl.BN == 128) *tile = 0; else if (pl.BM == 16 && pl.BN == 128) *tile = 1; else return GAN_E_SHAPE;
  *rg = (long long)pl.P * (pl.p.M / pl.p.skn_groups);
  if (*rg > 512) return GAN_E_SHAPE;                                  // (conv_stack_kernel: register budget of two workgroups per CU)
  return 0;
}

int gan_conv_stack_eligible(const GanConvDesc* d, int op) {
  if (!d || d->struct_size != sizeof(GanConvDesc)) return GAN_E_ARG;
  GemmPlan pl;
  GanConvDesc t = *d;
  plan_only_desc(&t);
  int tile; long long rg;
  const int rc = stack_layer_plan(&t, op, pl, &tile, &rg);
  return rc == 0 ? 1 : (rc == GAN_E_SHAPE ? 0 : rc);
}

int gan_conv_stack_plan(const GanConvDesc* const* descs, const int32_t* ops, int32_t n, void* host_plan, size_t plan_bytes) {
  if (!descs || !ops || n <= 0 || n > 32 || !host_plan || plan_bytes < gan_conv_stack_plan_bytes(n)) return GAN_E_ARG;
  StackHeader* h = (StackHeader*)host_plan;
  StackLayer* L = (StackLayer*)((char*)host_plan + 256);
  size_t smem = 0;
  for (int i = 0; i < n; ++i) {
    const GanConvDesc* d = descs[i];
    if (!d || d->dtype != descs[0]->dtype) return GAN_E_ARG;
    GemmPlan pl;
    int tile; long long rg;
    int rc = stack_layer_plan(d, ops[i], pl, &tile, &rg);
    if (rc) return rc;
    if (pl.slab_bytes > d->workspace_bytes || !d->workspace) return GAN_E_WORKSPACE;
    StackLayer& s = L[i];
    s.p = pl.p; s.P = pl.P; s.tile = tile;
    s.gx = (int)pl.grid.x; s.items = (int)(pl.grid.x * pl.grid.z);
    s.fin_mode = pl.p.skn;
    s.fin_gx = pl.p.Cout / 8;
    s.fin_gy = (pl.p.skn == 1 && !(pl.p.skn_mmean && pl.p.skn_groups > 1)) ? pl.p.skn_groups : 1;
    s.fin_kr = rg <= 128 ? 1 : rg <= 256 ? 2 : 4;
    const size_t sm = tile == 0 ? (size_t)3 * (64 + 128) * 128 + 16 * 64 * sizeof(int) : (size_t)2 * (16 + 128) * 128 + 16 * 16 * sizeof(int);
    if (sm > smem) smem = sm;
  }
  h->magic = STACK_MAGIC; h->n = (uint32_t)n; h->dtype = (uint32_t)descs[0]->dtype; h->smem = smem;
  int grid = gan_opt("conv.stack_blocks");
  if (grid < 8) grid = 8;
  // every workgroup of the grid must be RESIDENT (the grid barriers spin): at most two 256-thread workgroups of this kernel fit on a CU
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (grid > 2 * cus) return GAN_E_ARG;
  h->grid = (uint32_t)(grid & ~7);
  return 0;
}

int gan_conv_stack_launch(const void* host_plan, const void* dev_plan, void* barrier_state, int32_t* err_flag, gan_stream_t stream) {
  const StackHeader* h = (const StackHeader*)host_plan;
  if (!h || h->magic != STACK_MAGIC || !dev_plan || !barrier_state || !err_flag || ((uintptr_t)barrier_state & 127)) return GAN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  return h->dtype == GAN_F32 ? launch_stack<float>(h, dev_plan, barrier_state, err_flag, st)
       : h->dtype == GAN_F16 ? launch_stack<f16_t>(h, dev_plan, barrier_state, err_flag, st)
                             : launch_stack<bf16_t>(h, dev_plan, barrier_state, err_flag, st);
}
size_t gan_conv_stack_barrier_bytes(void) { return sizeof(StackBar); }

size_t gan_conv_workspace_bytes(const GanConvDesc* d, int op) {
  if (!d || d->struct_size != sizeof(GanConvDesc)) return 0;
  GemmPlan pl;
  GanConvDesc t = *d;
  plan_only_desc(&t);
  if (plan_gemm(&t, op, &pl)) return 0;
  if (const int fam = thin_family(&t, op, pl.p)) return thin_workspace_bytes(fam, &t, pl.p);
  return pl.slab_bytes;
}
}
