// Device helpers shared by conv_gemm.hip and conv_own.hip: output-pixel decode, the L2-coherent access wrappers and the body that
// finishes a small layer (GanNormFuse): statistics + normalise + dropout + activation forward, the whole normalisation backward.
#pragma once
#include "common.h"
#include "conv_params.h"

__device__ __forceinline__ size_t out_pixel_index(const GemmParams& p, int m, int py, int px) {
  const unsigned t = fdiv((unsigned)m, p.divWg);
  const int gx = m - t * p.Wg;
  const unsigned img = fdiv(t, p.divHg);
  const int gy = t - img * p.Hg;
  return (size_t)(img * p.Ho + gy * p.OS + py) * p.Wo + (gx * p.OS + px);
}
__device__ __forceinline__ size_t out_pixel_offset(const GemmParams& p, int m, int py, int px) {
  return out_pixel_index(p, m, py, px) * (size_t)p.ypitch;
}

// Cross-workgroup exchange INSIDE one launch (conv_stack_kernel): the L2s of the 8 XCDs are not coherent with each other, so data
// that another workgroup reads later in the same kernel is written through (sc0 sc1 stores) and read around the L2 (sc0 sc1 loads)
// - measured coherent without any cache maintenance (tools/probes/gridbar_probe.hip).  Buffer instructions so that the compiler
// tracks the stores' data registers (an inline-asm store re-used them too early).  COH = false: plain accesses.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
struct CohBuf { __amdgpu_buffer_rsrc_t r; const unsigned char* base; };
__device__ __forceinline__ CohBuf coh_buf(const void* base) {
  CohBuf b;
  b.r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xfffffff0u, 0x00020000);
  b.base = (const unsigned char*)base;
  return b;
}
template <bool COH> __device__ __forceinline__ f32x4 ld_f4(const CohBuf& b, const float* ptr) {
  if constexpr (COH) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (unsigned)((const unsigned char*)ptr - b.base), 0, 0x11);
    return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  } else {
    return *(const f32x4*)ptr;
  }
}
template <bool COH> __device__ __forceinline__ void st_f4(const CohBuf& b, float* ptr, const f32x4& v) {
  if constexpr (COH) {
    const u32x4_t u = u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(u, b.r, (unsigned)((const unsigned char*)ptr - b.base), 0, 0x11);
  } else {
    *(f32x4*)ptr = v;
  }
}
template <bool COH> __device__ __forceinline__ uint2 ld_u2(const CohBuf& b, const void* ptr) {
  if constexpr (COH) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(b.r, (unsigned)((const unsigned char*)ptr - b.base), 0, 0x11);
    return make_uint2(v.x, v.y);
  } else {
    return *(const uint2*)ptr;
  }
}
template <bool COH> __device__ __forceinline__ void st_u2(const CohBuf& b, void* ptr, const uint2& v) {
  if constexpr (COH) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{v.x, v.y}, b.r, (unsigned)((const unsigned char*)ptr - b.base), 0, 0x11);
  else *(uint2*)ptr = v;
}

// ------------------------------------------------------------------------------------------------
// Slab reduce of a SMALL split-K layer that also finishes the layer (GanNormFuse): a workgroup owns 8 channels of every row of
// a statistics group (<= 1024 rows: 128 row slots x <= 8 rows per thread, kept in registers), so summing the K slabs, the
// normalisation statistics and the apply pass need no second launch and no second read.
//   MODE 1 (forward):  y = sum of slabs (stored); mean, rstd (+ moving averages); out = act(dropout(gamma*(y-mean)*rstd+beta))
//   MODE 2 (backward): da = sum of slabs (+ add); dz = da * act'(z) * mask with z from the saved y (bf_ref);
//                      out = gamma*rstd*(dz - sum(dz)/R - xhat*sum(dz*xhat)/R); dgamma, dbeta; channels >= bf_cols: y = da
// Arithmetic per element = splitk_reduce4_kernel + stats_finalize / bwd_finalize + norm_act_fwd / norm_act_bwd (norm.hip); the
// per-channel sums are taken in a different (fixed) order.  gridDim.y == groups: one group per workgroup; gridDim.y == 1: the
// workgroup walks the groups in order (moving averages of successive BatchNormalization calls; dgamma / dbeta over the groups).
// bx / by / gy: the block coordinates and the y extent of the grid of the stand-alone kernel.  COH (conv_stack_kernel): the slabs and the skip
// gradient were written earlier in the same launch by other workgroups and the outputs are read later in it: loads around / stores through
// the L2 (sc0 sc1).
// SRC = 1 (conv_own_kernel): the layer's values are not in slabs but in the workgroup's LDS - `part` holds `nparts` partial sums
// [part][parity * mpad + row][8] (fp32) of its 8 channels, added here in order.
// NTH: threads of the workgroup that run the body (256, or 512 for the big groups: half the rows - and half the serial loads - per thread)
template <typename T, int MODE, int KR, bool COH, int SRC = 0, int NTH = 256>       // KR: rows per thread (1, 2, 4, 8): (NTH / 2) * KR >= rows per group
__device__ __forceinline__ void splitk_norm_body(const GemmParams& p, int P, int bx, int by, int gy, const float* part = nullptr, int nparts = 0,
                                                 int mpad = 0) {
  constexpr int RS = NTH / 2, NWV = NTH / 64, UNR = 16 / KR;         // 16 slab loads in flight per thread (a row at a time the kernel is latency-bound; 32 in flight: no faster, measured)
  __shared__ double red[NWV][8][2];
  __shared__ float bc[8][4];
  const int tid = threadIdx.x, cv = tid & 1, rs = tid >> 1, lane = tid & 63, wave = tid >> 6;
  const int n = bx * 8 + cv * 4;
  const int groups = p.skn_groups, Mg = p.M / groups, Rg = P * Mg;
  const size_t sstride = (size_t)p.M * p.NslabPitch;
  const int g0 = gy > 1 ? by : 0, g1 = gy > 1 ? g0 + 1 : groups;
  const CohBuf cslab = coh_buf(p.slab), cout = coh_buf(p.skn_out), cy = coh_buf(p.y), cadd = coh_buf(p.bf_add);
  auto ld4 = [](const void* base, size_t off, float* out) {
    if constexpr (sizeof(T) == 4) { const f32x4 q = *(const f32x4*)((const float*)base + off); out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3]; }
    else { float t8[8]; const uint2 q = *(const uint2*)((const T*)base + off); unpack16<T>(make_uint4(q.x, q.y, 0u, 0u), t8); out[0] = t8[0]; out[1] = t8[1]; out[2] = t8[2]; out[3] = t8[3]; }
  };
  auto ld4c = [](const CohBuf& cbuf, const void* base, size_t off, float* out) {     // the same through the coherent path (COH)
    if constexpr (sizeof(T) == 4) { const f32x4 q = ld_f4<COH>(cbuf, (const float*)base + off); out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3]; }
    else { float t8[8]; const uint2 q = ld_u2<COH>(cbuf, (const T*)base + off); unpack16<T>(make_uint4(q.x, q.y, 0u, 0u), t8); out[0] = t8[0]; out[1] = t8[1]; out[2] = t8[2]; out[3] = t8[3]; }
  };
  // the same in two halves - issue (raw) early, unpack where the value is needed: the loads of one row slot are independent of the slab
  // sums, and a dependent round trip costs 2-4.5 us here (cold lines, 64-128 workgroups: latency-bound, not bandwidth-bound)
  struct Raw4 { f32x4 f; uint2 h; };
  auto ld4raw = [](const void* base, size_t off) {
    Raw4 q;
    if constexpr (sizeof(T) == 4) q.f = *(const f32x4*)((const float*)base + off); else q.h = *(const uint2*)((const T*)base + off);
    return q;
  };
  auto ld4craw = [](const CohBuf& cbuf, const void* base, size_t off) {
    Raw4 q;
    if constexpr (sizeof(T) == 4) q.f = ld_f4<COH>(cbuf, (const float*)base + off); else q.h = ld_u2<COH>(cbuf, (const T*)base + off);
    return q;
  };
  auto unraw = [](const Raw4& q, float* out) {
    if constexpr (sizeof(T) == 4) { out[0] = q.f[0]; out[1] = q.f[1]; out[2] = q.f[2]; out[3] = q.f[3]; }
    else { float t8[8]; unpack16<T>(make_uint4(q.h.x, q.h.y, 0u, 0u), t8); out[0] = t8[0]; out[1] = t8[1]; out[2] = t8[2]; out[3] = t8[3]; }
  };
  auto st4 = [](const CohBuf& cbuf, void* base, size_t off, const float* v, bool f32) {
    if (f32 || sizeof(T) == 4) st_f4<COH>(cbuf, (float*)base + off, f32x4{v[0], v[1], v[2], v[3]});
    else st_u2<COH>(cbuf, (T*)base + off, make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])));
  };
  // slabs of this kernel's launches: [parity * splits + split][8-channel slice][row][8] (gemm_epilogue, p.skn)
  auto lds_sum = [&](int par, int m) {
    const float* src = part + (size_t)(par * mpad + m) * 8 + cv * 4;
    const size_t pstride = (size_t)P * mpad * 8;
    f32x4 sacc = *(const f32x4*)src;
    for (int k = 1; k < nparts; ++k) sacc += *(const f32x4*)(src + (size_t)k * pstride);
    return sacc;
  };
  auto slab_sum = [&](int par, int m, float* v) {
    if constexpr (SRC == 1) {
      const f32x4 sacc = lds_sum(par, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (p.out_f32 || sizeof(T) == 4) ? sacc[e] : (float)(T)sacc[e];
      return;
    }
    const float* src = p.slab + (size_t)par * p.splits * sstride + ((size_t)bx * p.M + m) * 8 + cv * 4;
    f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= p.splits; k += 4) {
      const f32x4 a = ld_f4<COH>(cslab, src + (size_t)k * sstride), b = ld_f4<COH>(cslab, src + (size_t)(k + 1) * sstride);
      const f32x4 c = ld_f4<COH>(cslab, src + (size_t)(k + 2) * sstride), d = ld_f4<COH>(cslab, src + (size_t)(k + 3) * sstride);
      sacc += a; sacc += b; sacc += c; sacc += d;
    }
    for (; k < p.splits; ++k) sacc += ld_f4<COH>(cslab, src + (size_t)k * sstride);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (p.out_f32 || sizeof(T) == 4) ? sacc[e] : (float)(T)sacc[e];      // as stored
  };
  // sum of two values per channel over the workgroup's row slots: lanes of equal cv inside a wave, then the 4 waves
  // (a thread's <= 8 rows are summed in fp32, everything across threads in double, like the finalize kernels of norm.hip)
  auto block_sums = [&](const float (&f1)[4], const float (&f2)[4], double* t1, double* t2) {     // t1/t2: totals of channel (tid & 7), all threads
    double s1[4], s2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] = (double)f1[e]; s2[e] = (double)f2[e];
#pragma unroll
      for (int o = 2; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    }
    __syncthreads();                                          // (red is re-used group after group)
    if (lane < 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[wave][lane * 4 + e][0] = s1[e]; red[wave][lane * 4 + e][1] = s2[e]; }
    }
    __syncthreads();
    const int c = tid & 7;
    double a1 = red[0][c][0], a2 = red[0][c][1];
#pragma unroll
    for (int w = 1; w < NWV; ++w) { a1 += red[w][c][0]; a2 += red[w][c][1]; }
    *t1 = a1; *t2 = a2;
  };
  if (MODE == 2 && bx * 8 >= p.bf_cols) {                     // skip half of a decoder concat: plain gradient
    for (int row = rs; row < P * p.M; row += RS) {
      const int par = row / p.M, m = row - par * p.M;
      float v[4];
      slab_sum(par, m, v);
      st4(cy, p.y, out_pixel_index(p, m, par >> 1, par & 1) * (size_t)p.ypitch + n, v, p.out_f32);
    }
    return;
  }
  double tg = 0, tb = 0;                                       // MODE 2: dgamma / dbeta of channel (tid & 7) over the groups
  for (int g = g0; g < g1; ++g) {
    float va[KR][4], vb[KR][4];                                // MODE 1: y | MODE 2: dz, xhat
    size_t pix[KR];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4], rsd[4], ga[4], be[4];
    if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mu[e] = p.bf_mean[g * p.bf_cols + n + e]; rsd[e] = p.bf_rstd[g * p.bf_cols + n + e];
        ga[e] = p.bf_gamma[n + e]; be[e] = p.bf_beta[n + e];
      }
    }
    // the thread's rows, and everything of them that does not depend on the slab sums, requested first
    Raw4 rraw[KR], araw[KR];
    uint32_t mraw[KR];
    float cgam = 0.f, cbet = 0.f, cmm = 0.f, cmv = 0.f;         // MODE 1, tid < 8: this channel's gamma, beta, moving statistics
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      int row = rs + RS * k;
      if (row >= Rg) row = rs < Rg ? rs : 0;                   // (clamped: loaded, never used)
      const int par = row / Mg, m = g * Mg + (row - par * Mg);
      pix[k] = out_pixel_index(p, m, par >> 1, par & 1);
      mraw[k] = 0u;
      if (MODE == 2) {
        if (p.bf_add) araw[k] = ld4craw(cadd, p.bf_add, pix[k] * (size_t)p.bf_addpitch + n);
        rraw[k] = ld4raw(p.bf_ref, pix[k] * (size_t)p.bf_refpitch + n);
        if (p.bf_mode == 3) mraw[k] = *(const uint32_t*)(p.bf_mask + pix[k] * (size_t)p.bf_maskpitch + n);
      } else if (p.skn_mask) {
        mraw[k] = *(const uint32_t*)(p.skn_mask + pix[k] * (size_t)p.Cout + n);
      }
    }
    if (MODE == 1 && tid < 8) {
      const int cn0 = bx * 8 + tid;
      cgam = p.skn_gamma[cn0]; cbet = p.skn_beta[cn0];
      if (p.skn_mmean) { cmm = p.skn_mmean[cn0]; cmv = p.skn_mvar[cn0]; }
    }
    // slab sums of the thread's KR rows: 16 loads in flight per thread (a row at a time the kernel was latency-bound: 24 us for
    // 256 KB per workgroup); the splits of a row are added in order, as splitk_reduce4_kernel adds them
    if constexpr (SRC == 1) {
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        int row = rs + RS * k;
        if (row >= Rg) row = rs < Rg ? rs : 0;                 // (clamped: read, never used)
        const int par = row / Mg, m = g * Mg + (row - par * Mg);
        const f32x4 sacc = lds_sum(par, m);
#pragma unroll
        for (int e = 0; e < 4; ++e) va[k][e] = (p.out_f32 || sizeof(T) == 4) ? sacc[e] : (float)(T)sacc[e];      // as stored
      }
    } else {
      const float* src[KR];
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        int row = rs + RS * k;
        if (row >= Rg) row = rs < Rg ? rs : 0;                 // (clamped: loaded, never used)
        const int par = row / Mg, m = g * Mg + (row - par * Mg);
        src[k] = p.slab + (size_t)par * p.splits * sstride + ((size_t)bx * p.M + m) * 8 + cv * 4;
      }
      f32x4 sacc[KR];
#pragma unroll
      for (int k = 0; k < KR; ++k) sacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      int sp = 0;
      for (; sp + UNR <= p.splits; sp += UNR) {
        f32x4 t[UNR][KR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int k = 0; k < KR; ++k) t[u][k] = ld_f4<COH>(cslab, src[k] + (size_t)(sp + u) * sstride);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int k = 0; k < KR; ++k) sacc[k] += t[u][k];
      }
      for (; sp < p.splits; ++sp) {
#pragma unroll
        for (int k = 0; k < KR; ++k) sacc[k] += ld_f4<COH>(cslab, src[k] + (size_t)sp * sstride);
      }
#pragma unroll
      for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) va[k][e] = (p.out_f32 || sizeof(T) == 4) ? sacc[k][e] : (float)(T)sacc[k][e];      // as stored
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int row = rs + RS * k;
      if (row < Rg) {
        if (MODE == 1) {
          st4(cy, p.y, pix[k] * (size_t)p.ypitch + n, va[k], p.out_f32);
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += va[k][e]; s2[e] = fmaf(va[k][e], va[k][e], s2[e]); }
        } else {
          float rf[4], a2[4];
          if (p.bf_add) {
            unraw(araw[k], a2);
#pragma unroll
            for (int e = 0; e < 4; ++e) va[k][e] += a2[e];
          }
          unraw(rraw[k], rf);
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (p.bf_mode == 3) {
            const uint32_t w = mraw[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) mk[e] = 2.f * (float)((w >> (8 * e)) & 0xff);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (rf[e] - mu[e]) * rsd[e];
            const float z = fmaf(ga[e], xh, be[e]);
            const float zd = z * mk[e];
            float d = va[k][e] * mk[e];
            d = zd > 0.f ? d : (p.bf_mode == 1 ? d * p.bf_slope : 0.f);
            va[k][e] = d; vb[k][e] = xh;
            s1[e] += d; s2[e] = fmaf(d, xh, s2[e]);
          }
        }
      }
    }
    double t1, t2;
    block_sums(s1, s2, &t1, &t2);
    const int c = tid & 7, cn = bx * 8 + c;
    if (MODE == 1) {
      if (tid < 8) {
        const double rows = (double)Rg;
        const double m = t1 / rows;
        double var = t2 / rows - m * m;
        if (var < 0) var = 0;
        const float r = 1.0f / sqrtf((float)var + p.skn_eps);
        p.skn_mean[g * p.Cout + cn] = (float)m; p.skn_rstd[g * p.Cout + cn] = r;
        if (p.skn_mmean) {
          const double adj = rows / (double)(Rg > 1 ? Rg - 1 : 1);
          p.skn_mmean[cn] = cmm + ((float)m - cmm) * (1.f - p.skn_momentum);
          p.skn_mvar[cn] = cmv + ((float)(var * adj) - cmv) * (1.f - p.skn_momentum);
        }
        bc[c][0] = (float)m; bc[c][1] = cgam * r; bc[c][2] = cbet;
      }
      __syncthreads();
      float mu1[4], A[4], b1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { mu1[e] = bc[cv * 4 + e][0]; A[e] = bc[cv * 4 + e][1]; b1[e] = bc[cv * 4 + e][2]; }
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        if (rs + RS * k < Rg) {
          float o[4];
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (p.skn_mask) {
            const uint32_t w = mraw[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) mk[e] = 2.f * (float)((w >> (8 * e)) & 0xff);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float z = fmaf(va[k][e] - mu1[e], A[e], b1[e]);
            if (p.skn_mask) z *= mk[e];
            o[e] = apply_act(z, p.skn_act, p.skn_slope);
          }
          st4(cout, p.skn_out, pix[k] * (size_t)p.skn_outpitch + n, o, false);
        }
      }
    } else {
      if (tid < 8) {
        const float A = p.bf_gamma[cn] * p.bf_rstd[g * p.bf_cols + cn], invR = 1.0f / (float)Rg;
        bc[c][0] = A; bc[c][1] = -A * ((float)t1 * invR); bc[c][2] = -A * ((float)t2 * invR);
        tb += t1; tg += t2;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        if (rs + RS * k < Rg) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaf(va[k][e], bc[cv * 4 + e][0], fmaf(vb[k][e], bc[cv * 4 + e][2], bc[cv * 4 + e][1]));
          st4(cout, p.skn_out, pix[k] * (size_t)p.skn_outpitch + n, o, false);
        }
      }
    }
  }
  if (MODE == 2 && tid < 8) {
    const int cn = bx * 8 + tid;
    if (p.skn_dgamma) p.skn_dgamma[cn] = (p.skn_accumulate ? p.skn_dgamma[cn] : 0.f) + (float)tg;
    if (p.skn_dbeta) p.skn_dbeta[cn] = (p.skn_accumulate ? p.skn_dbeta[cn] : 0.f) + (float)tb;
  }
}

