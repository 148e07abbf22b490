// Normalisation (BatchNorm / InstanceNorm as "statistics groups") + dropout + activation, forward and
// backward, plus activation-only backward and bias gradients.  All HBM-bound: 16-byte vector accesses on
// NHWC rows, 4 independent rows in flight per thread (memory-level parallelism instead of a dependent
// load-compute chain), per-channel parameters fetched as float4.  Reductions are two-stage and
// deterministic: block partials [group][chunk][C][2] -> a chunk-parallel finalize in double precision.
#include "common.h"
#include "splitk_norm.h"      // CohBuf / ld_f4 / st_f4: accesses that are coherent across the XCDs' L2s inside one launch

struct RedGeom {
  int C, cvecs;            // channels, 16-byte vectors per row
  long long rows_per_group;
  int chunks;              // row chunks per group (gridDim.x)
  int hw, gsize;           // pixels per image, images per group
  int rslots;              // row slots per 256-thread block = 256 / min(cvecs, 256)
  int log2cv;              // cvecs is a power of two
  FastDiv divRpg;          // row -> statistics group (rows < 2^31; 64-bit divisions cost more than the arithmetic)
};

struct NormP {
  const void* y; int ypitch;
  const void* da; int dapitch;
  const void* da2; int da2pitch;
  void* out; int outpitch;          // a (fwd) or dy (bwd)
  const float* gamma; const float* beta; const float* mean; const float* rstd;
  const uint8_t* mask;
  const float* sums;                // bwd apply: [G][C][2] = (sum dz, sum dz*xhat)
  int act; float slope;
  int has_norm;                     // 0: plain activation backward on saved a
};

template <int VEC>
__device__ __forceinline__ void ldp(const float* base, float* out) {   // VEC consecutive floats, 16-B aligned
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    float4 v = *(const float4*)(base + 4 * i);
    out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
  }
}

template <int VEC>
__device__ __forceinline__ void ld_mask(const uint8_t* m, float* out) {  // VEC consecutive 0/1 bytes -> 2*mask
  if (VEC == 8) {
    uint2 w = *(const uint2*)m;
#pragma unroll
    for (int e = 0; e < 4; ++e) { out[e] = 2.f * (float)((w.x >> (8 * e)) & 0xff); out[4 + e] = 2.f * (float)((w.y >> (8 * e)) & 0xff); }
  } else {
    uint32_t w = *(const uint32_t*)m;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = 2.f * (float)((w >> (8 * e)) & 0xff);
  }
}

// Activation and mask are compile-time in the streaming kernels: with run-time selects these passes were VALU-bound
// (56 instructions per element), not HBM-bound.
// dz = d(loss)/d(z) for a = act(mk * z): da * act'(mk*z) * mk   (mk = 2*mask, or 1 without dropout)
template <int ACT, bool MASK> __device__ __forceinline__ float dz_c(float da, float z, float mk, float slope) {
  const float zd = MASK ? z * mk : z;
  float d = MASK ? da * mk : da;
  if (ACT == GAN_ACT_LRELU) d = zd > 0.f ? d : d * slope;
  else if (ACT == GAN_ACT_RELU) d = zd > 0.f ? d : 0.f;
  return d;
}

// MODE 0: stats (y, y^2).  MODE 1: norm backward sums (dz, dz*xhat).  MODE 2: column sum of `da` (bias grad).
template <typename T, int MODE, int ACT = GAN_ACT_NONE, bool MASK = false>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const NormP p, const RedGeom g, float* partial) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int U = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];     // [rslots][C][2]
  const int grp = blockIdx.y, chunk = blockIdx.x;
  const int cvl = g.cvecs < 256 ? g.cvecs : 256;
  const int rslots = g.rslots;
  const int cv = threadIdx.x % cvl, rslot = threadIdx.x / cvl;
  const long long r0 = g.rows_per_group * chunk / g.chunks, r1 = g.rows_per_group * (chunk + 1) / g.chunks;
  const long long rowbase = (long long)grp * g.rows_per_group;
  float* out2 = partial + ((size_t)grp * g.chunks + chunk) * g.C * 2;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
  // xhat = (y - mu)*rs with the subtraction first: y*rs - mu*rs cancels catastrophically when |mu| >> sigma
  float ga[VEC], be[VEC], mu[VEC], rs[VEC];
  if (MODE == 1) {
    ldp<VEC>(p.mean + grp * g.C + cv * VEC, mu); ldp<VEC>(p.rstd + grp * g.C + cv * VEC, rs);
    ldp<VEC>(p.gamma + cv * VEC, ga); ldp<VEC>(p.beta + cv * VEC, be);
  }
  for (long long rr = r0 + rslot; rr < r1; rr += (long long)rslots * U) {
    uint4 vy[U], vd[U], v2[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long row = rowbase + rr + (long long)u * rslots;
      ok[u] = rr + (long long)u * rslots < r1;
      vy[u] = vd[u] = v2[u] = make_uint4(0, 0, 0, 0);
      if (ok[u]) {
        if (MODE != 2) vy[u] = *(const uint4*)((const T*)p.y + row * p.ypitch + cv * VEC);
        if (MODE != 0) vd[u] = *(const uint4*)((const T*)p.da + row * p.dapitch + cv * VEC);
        if (MODE == 1 && p.da2) v2[u] = *(const uint4*)((const T*)p.da2 + row * p.da2pitch + cv * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      if (MODE == 0) {
        float v[VEC];
        unpack16<T>(vy[u], v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      } else if (MODE == 1) {
        float yv[VEC], dv[VEC], d2[VEC], mk[VEC];
        unpack16<T>(vy[u], yv); unpack16<T>(vd[u], dv);
        if (p.da2) {
          unpack16<T>(v2[u], d2);
#pragma unroll
          for (int e = 0; e < VEC; ++e) dv[e] += d2[e];
        }
        if (MASK) ld_mask<VEC>(p.mask + (rowbase + rr + (long long)u * rslots) * g.C + cv * VEC, mk);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xh = (yv[e] - mu[e]) * rs[e];
          const float z = fmaf(ga[e], xh, be[e]);
          const float dz = dz_c<ACT, MASK>(dv[e], z, MASK ? mk[e] : 1.f, p.slope);
          s1[e] += dz; s2[e] = fmaf(dz, xh, s2[e]);
        }
      } else {
        float dv[VEC];
        unpack16<T>(vd[u], dv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s1[e] += dv[e];
      }
    }
  }
  // (16-byte stores of {s1, s2, s1, s2}: scalar stores 16 dwords apart from lane to lane were 16-way bank conflicts, 89 % of
  // the kernel's LDS cycles in the SQ counters)
#pragma unroll
  for (int e = 0; e < VEC; e += 2)
    *(f32x4*)(lds + ((size_t)rslot * g.C + cv * VEC + e) * 2) = f32x4{s1[e], s2[e], s1[e + 1], s2[e + 1]};
  __syncthreads();
  for (int c = threadIdx.x; c < g.C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < rslots; ++k) { const float2 t = *(const float2*)(lds + ((size_t)k * g.C + c) * 2); a += t.x; b += t.y; }
    out2[c * 2] = a; out2[c * 2 + 1] = b;
  }
}

// Sum one (group, channel) pair over chunks: 32 chunk lanes x 8 channels per block (FC channels per block).
constexpr int FC = 8, FL = 32;
__device__ __forceinline__ void chunk_sum(const float* partial, int g, int chunks, int C, int c, int kl, double* red,
                                          double& s1, double& s2) {
  double a = 0, b = 0;
  if (c < C) {
    int k = kl;
    for (; k + 3 * FL < chunks; k += 4 * FL) {          // 4 independent loads in flight per lane
      const float* q = partial + (((size_t)g * chunks + k) * C + c) * 2;
      const float2 v0 = *(const float2*)q, v1 = *(const float2*)(q + (size_t)FL * C * 2);
      const float2 v2 = *(const float2*)(q + (size_t)2 * FL * C * 2), v3 = *(const float2*)(q + (size_t)3 * FL * C * 2);
      a += v0.x; b += v0.y; a += v1.x; b += v1.y; a += v2.x; b += v2.y; a += v3.x; b += v3.y;
    }
    for (; k < chunks; k += FL) {
      float2 v = *(const float2*)(partial + (((size_t)g * chunks + k) * C + c) * 2);
      a += v.x; b += v.y;
    }
  }
  const int cl = threadIdx.x % FC;
  red[(kl * FC + cl) * 2] = a; red[(kl * FC + cl) * 2 + 1] = b;
  __syncthreads();
  s1 = s2 = 0;
  if (kl == 0)
    for (int k = 0; k < FL; ++k) { s1 += red[(k * FC + cl) * 2]; s2 += red[(k * FC + cl) * 2 + 1]; }
  __syncthreads();
}

// ---- finalize carried by the apply launch (GanNormDesc.sync / GanNormBwdDesc.sync) -------------------------------------------
// A finalize launch between the pass that wrote the partials and the apply pass is 4.8 us of kernel for a few KB of work and - on the
// step - 8 us each (marginal cost measured by leaving them out, tools/marginal.py): the launch boundary is the cost.  With a
// sync area the apply launch does it itself: its first `nfin` workgroups (always dispatched first, they wait for nobody) run the
// finalize body, publish mean / rstd (or the backward sums) with write-through stores and count themselves in; every workgroup has
// its rows' loads in flight by then, waits for the count, reads the constants around the L2 and goes on.  Same arithmetic, same
// summation order as the stand-alone finalize kernels: bit-identical.  The sync area cleans itself: every workgroup takes a
// departure ticket after the wait (32 sub-counters, a cache line each, so that 4,096 workgroups do not queue on one address); the
// last one out zeroes the counters for the next launch that uses the area (launches sharing an area must be stream-ordered).
// Every wait is bounded (2 s): a launch whose first workgroups never publish sets the error word and runs on with stale constants.
constexpr int SYNC_ARRIVE = 0, SYNC_ERR = 8, SYNC_TOP = 16, SYNC_SUB = 32, SYNC_WORDS = 32 + 16 * 32;
struct FinP {
  const float* partial; unsigned* sync;
  int G, chunks, cb, nfin;                  // cb: channel blocks of FC; nfin = finalize work units = first workgroups that do them
  long long rows; float eps, momentum; float* mean; float* rstd; float* mmean; float* mvar;      // forward
  float* sums; float* dgamma; float* dbeta; int accumulate;                                        // backward
};
template <bool COH> __device__ __forceinline__ void st_f1(const CohBuf& b, float* ptr, float v) {
  if constexpr (COH) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b.r, (unsigned)((const unsigned char*)ptr - b.base), 0, 0x11);
  else *ptr = v;
}
template <int VEC, bool COH> __device__ __forceinline__ void ldp_c(const CohBuf& b, const float* base, float* out) {
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const f32x4 v = ld_f4<COH>(b, base + 4 * i);
    out[4 * i] = v[0]; out[4 * i + 1] = v[1]; out[4 * i + 2] = v[2]; out[4 * i + 3] = v[3];
  }
}
__device__ __forceinline__ void fin_arrive(unsigned* sync) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + SYNC_ARRIVE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every thread of the workgroup; returns thread 0's departure ticket
__device__ __forceinline__ unsigned fin_wait(unsigned* sync, int nfin) {
  unsigned t = 0;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(sync + SYNC_ARRIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nfin) {
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {      // 2 s at 100 MHz: give up, never hang
        __hip_atomic_store(sync + SYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    t = __hip_atomic_fetch_add(sync + SYNC_SUB + 16 * (blockIdx.x & 31u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return t;
}
__device__ __forceinline__ void fin_depart(unsigned* sync, unsigned ticket) {
  if (threadIdx.x == 0) {
    const unsigned grid = gridDim.x, i = blockIdx.x & 31u, cnt = (grid - i + 31u) >> 5;      // workgroups with this residue
    if (ticket == cnt - 1) {
      __hip_atomic_store(sync + SYNC_SUB + 16 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned ng = grid < 32u ? grid : 32u;
      if (__hip_atomic_fetch_add(sync + SYNC_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1) {      // everybody has passed the wait
        __hip_atomic_store(sync + SYNC_TOP, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + SYNC_ARRIVE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// groups are processed in order inside a block when moving averages are updated (two successive
// BatchNormalization calls update them one after the other); otherwise by = the group (gridDim.y = groups).
template <bool COH>
__device__ __forceinline__ void stats_finalize_body(const float* partial, int G, int chunks, int C, long long rows, float eps, float* mean,
                                                    float* rstd, float* mmean, float* mvar, float momentum, int bx, int by, double* red) {
  const int c = bx * FC + threadIdx.x % FC, kl = threadIdx.x / FC;
  const int g0 = mmean ? 0 : by, g1 = mmean ? G : by + 1;
  const CohBuf cm = coh_buf(mean), cr = coh_buf(rstd);
  for (int g = g0; g < g1; ++g) {
    double s, s2;
    chunk_sum(partial, g, chunks, C, c, kl, red, s, s2);
    if (kl == 0 && c < C) {
      double m = s / (double)rows;
      double var = s2 / (double)rows - m * m;
      if (var < 0) var = 0;
      st_f1<COH>(cm, mean + g * C + c, (float)m);
      st_f1<COH>(cr, rstd + g * C + c, 1.0f / sqrtf((float)var + eps));
      if (mmean) {
        double adj = (double)rows / (double)(rows > 1 ? rows - 1 : 1);
        mmean[c] += ((float)m - mmean[c]) * (1.f - momentum);
        mvar[c] += ((float)(var * adj) - mvar[c]) * (1.f - momentum);
      }
    }
  }
}
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* partial, int G, int chunks, int C, long long rows,
                                                             float eps, float* mean, float* rstd, float* mmean, float* mvar,
                                                             float momentum) {
  __shared__ double red[FL * FC * 2];
  stats_finalize_body<false>(partial, G, chunks, C, rows, eps, mean, rstd, mmean, mvar, momentum, blockIdx.x, blockIdx.y, red);
}

template <bool COH>
__device__ __forceinline__ void bwd_finalize_body(const float* partial, int G, int chunks, int C, float* sums, float* dgamma, float* dbeta,
                                                  int accumulate, int bx, double* red) {
  const int c = bx * FC + threadIdx.x % FC, kl = threadIdx.x / FC;
  const CohBuf cs = coh_buf(sums);
  double tg = 0, tb = 0;
  for (int g = 0; g < G; ++g) {
    double s1, s2;
    chunk_sum(partial, g, chunks, C, c, kl, red, s1, s2);
    if (kl == 0 && c < C) {
      if (sums) { st_f1<COH>(cs, sums + (g * C + c) * 2, (float)s1); st_f1<COH>(cs, sums + (g * C + c) * 2 + 1, (float)s2); }
      tb += s1; tg += s2;
    }
  }
  if (kl == 0 && c < C) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)tg;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)tb;
  }
}
__global__ __launch_bounds__(256) void bwd_finalize_kernel(const float* partial, int G, int chunks, int C, float* sums,
                                                           float* dgamma, float* dbeta, int accumulate) {
  __shared__ double red[FL * FC * 2];
  bwd_finalize_body<false>(partial, G, chunks, C, sums, dgamma, dbeta, accumulate, blockIdx.x, red);
}

// forward: a = act(dropout(gamma*(y-mean)*rstd + beta)) = act(mk * ((y-mean)*A + beta)); each thread: one channel vector x 4 rows.
// The per-(group, channel) constants are loaded once when the thread's rows lie in one statistics group (always,
// except at group boundaries / 1x1 InstanceNorm maps).  FIN: the launch also finalizes the statistics (FinP above).
template <typename T, int ACT, bool MASK, bool FIN>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const NormP p, const RedGeom g, long long rows, const FinP f) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int U = 4;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int cv = (int)(idx & (unsigned)(g.cvecs - 1));
  const long long rb = (long long)(idx >> g.log2cv) * U;
  const bool live = rb < rows;
  if (!FIN && !live) return;
  uint4 vy[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    vy[u] = rb + u < rows ? *(const uint4*)((const T*)p.y + (rb + u) * p.ypitch + cv * VEC) : make_uint4(0, 0, 0, 0);
  unsigned ticket = 0;
  if constexpr (FIN) {
    __shared__ double red[FL * FC * 2];
    if (blockIdx.x < (unsigned)f.nfin) {
      stats_finalize_body<true>(f.partial, f.G, f.chunks, g.C, f.rows, f.eps, f.mean, f.rstd, f.mmean, f.mvar, f.momentum,
                                (int)(blockIdx.x % (unsigned)f.cb), (int)(blockIdx.x / (unsigned)f.cb), red);
      fin_arrive(f.sync);
    }
    ticket = fin_wait(f.sync, f.nfin);
  }
  if (live) {
    float A[VEC], be[VEC], mu[VEC];        // z = (y - mu)*A + beta, subtraction first (no cancellation when |mu| >> sigma)
    const CohBuf cm = coh_buf(p.mean), cr = coh_buf(p.rstd);
    auto consts = [&](int grp) {
      float ga[VEC], rs[VEC];
      ldp<VEC>(p.gamma + cv * VEC, ga); ldp<VEC>(p.beta + cv * VEC, be);
      ldp_c<VEC, FIN>(cm, p.mean + grp * g.C + cv * VEC, mu); ldp_c<VEC, FIN>(cr, p.rstd + grp * g.C + cv * VEC, rs);
#pragma unroll
      for (int e = 0; e < VEC; ++e) A[e] = ga[e] * rs[e];
    };
    const long long rl = rb + U - 1 < rows ? rb + U - 1 : rows - 1;
    const int g0 = (int)fdiv((unsigned)rb, g.divRpg), gl = (int)fdiv((unsigned)rl, g.divRpg);
    const bool uni = g0 == gl;
    if (uni) consts(g0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = rb + u;
      if (row >= rows) break;
      if (!uni) consts((int)fdiv((unsigned)row, g.divRpg));
      float v[VEC], o[VEC], mk[VEC];
      unpack16<T>(vy[u], v);
      if (MASK) ld_mask<VEC>(p.mask + row * g.C + cv * VEC, mk);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float z = fmaf(v[e] - mu[e], A[e], be[e]);
        if (MASK) z *= mk[e];
        o[e] = act_c<ACT>(z, p.slope);
      }
      *(uint4*)((T*)p.out + row * p.outpitch + cv * VEC) = pack16<T>(o);
    }
  }
  if constexpr (FIN) fin_depart(f.sync, ticket);
}

// backward apply: dy = gamma*rstd*(dz - S1/R - xhat*S2/R) = dz*A + xhat*N2 + N1.  FIN: the launch also finalizes the sums.
template <typename T, int ACT, bool MASK, bool FIN>
__global__ __launch_bounds__(256) void norm_act_bwd_kernel(const NormP p, const RedGeom g, long long rows, const FinP f) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int U = 4;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int cv = (int)(idx & (unsigned)(g.cvecs - 1));
  const long long rb = (long long)(idx >> g.log2cv) * U;
  const bool live = rb < rows;
  if (!FIN && !live) return;
  uint4 vy[U], vd[U], v2[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool ok = rb + u < rows;
    vy[u] = ok ? *(const uint4*)((const T*)p.y + (rb + u) * p.ypitch + cv * VEC) : make_uint4(0, 0, 0, 0);
    vd[u] = ok ? *(const uint4*)((const T*)p.da + (rb + u) * p.dapitch + cv * VEC) : make_uint4(0, 0, 0, 0);
    v2[u] = (ok && p.da2) ? *(const uint4*)((const T*)p.da2 + (rb + u) * p.da2pitch + cv * VEC) : make_uint4(0, 0, 0, 0);
  }
  unsigned ticket = 0;
  if constexpr (FIN) {
    __shared__ double red[FL * FC * 2];
    if (blockIdx.x < (unsigned)f.nfin) {
      bwd_finalize_body<true>(f.partial, f.G, f.chunks, g.C, f.sums, f.dgamma, f.dbeta, f.accumulate, (int)blockIdx.x, red);
      fin_arrive(f.sync);
    }
    ticket = fin_wait(f.sync, f.nfin);
  }
  if (live) {
    float ga[VEC], be[VEC], rs[VEC], mu[VEC], A[VEC], N1[VEC], N2[VEC];
    const float invR = 1.0f / (float)g.rows_per_group;
    const CohBuf cs = coh_buf(p.sums);
    auto consts = [&](int grp) {
      float t[2 * VEC];
      ldp<VEC>(p.gamma + cv * VEC, ga); ldp<VEC>(p.beta + cv * VEC, be);
      ldp<VEC>(p.mean + grp * g.C + cv * VEC, mu); ldp<VEC>(p.rstd + grp * g.C + cv * VEC, rs);
      ldp_c<2 * VEC, FIN>(cs, p.sums + ((size_t)grp * g.C + cv * VEC) * 2, t);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        A[e] = ga[e] * rs[e];
        N1[e] = -A[e] * (t[2 * e] * invR);
        N2[e] = -A[e] * (t[2 * e + 1] * invR);
      }
    };
    const long long rl = rb + U - 1 < rows ? rb + U - 1 : rows - 1;
    const int g0 = (int)fdiv((unsigned)rb, g.divRpg), gl = (int)fdiv((unsigned)rl, g.divRpg);
    const bool uni = g0 == gl;
    if (uni) consts(g0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = rb + u;
      if (row >= rows) break;
      if (!uni) consts((int)fdiv((unsigned)row, g.divRpg));
      float yv[VEC], dv[VEC], d2[VEC], o[VEC], mk[VEC];
      unpack16<T>(vy[u], yv); unpack16<T>(vd[u], dv);
      if (p.da2) {
        unpack16<T>(v2[u], d2);
#pragma unroll
        for (int e = 0; e < VEC; ++e) dv[e] += d2[e];
      }
      if (MASK) ld_mask<VEC>(p.mask + row * g.C + cv * VEC, mk);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xh = (yv[e] - mu[e]) * rs[e];
        const float z = fmaf(ga[e], xh, be[e]);
        const float dz = dz_c<ACT, MASK>(dv[e], z, MASK ? mk[e] : 1.f, p.slope);
        o[e] = fmaf(dz, A[e], fmaf(xh, N2[e], N1[e]));
      }
      *(uint4*)((T*)p.out + row * p.outpitch + cv * VEC) = pack16<T>(o);
    }
  }
  if constexpr (FIN) fin_depart(f.sync, ticket);
}

// activation-only backward on the saved activation a: dy = (da + da2) * act'(a)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const NormP p, const RedGeom g, long long rows) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int U = 4;
  // a thread's U rows are a quarter of the tensor apart, so that the lanes of a wave sit on CONSECUTIVE rows: for the 8-channel
  // tensors this kernel mostly sees (the tanh head: one 16-byte vector per row) four consecutive rows per thread made every load
  // instruction touch 64 lines 64 bytes apart (20.6 us for 67 MB)
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const int cv = (int)(idx & (unsigned)(g.cvecs - 1));
  const long long rq = (rows + U - 1) / U, rb = (long long)(idx >> g.log2cv);
  if (rb >= rq) return;
  uint4 vy[U], vd[U], v2[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = rb + u * rq;
    const bool ok = row < rows;
    vy[u] = ok ? *(const uint4*)((const T*)p.y + row * p.ypitch + cv * VEC) : make_uint4(0, 0, 0, 0);
    vd[u] = ok ? *(const uint4*)((const T*)p.da + row * p.dapitch + cv * VEC) : make_uint4(0, 0, 0, 0);
    v2[u] = (ok && p.da2) ? *(const uint4*)((const T*)p.da2 + row * p.da2pitch + cv * VEC) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = rb + u * rq;
    if (row >= rows) continue;
    float av[VEC], dv[VEC], d2[VEC], o[VEC];
    unpack16<T>(vy[u], av); unpack16<T>(vd[u], dv); unpack16<T>(v2[u], d2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float a = av[e], d = dv[e] + d2[e];
      if (ACT == GAN_ACT_LRELU) o[e] = a > 0.f ? d : d * p.slope;
      else if (ACT == GAN_ACT_RELU) o[e] = a > 0.f ? d : 0.f;
      else if (ACT == GAN_ACT_TANH) o[e] = d * (1.f - a * a);
      else o[e] = d;
    }
    *(uint4*)((T*)p.out + row * p.outpitch + cv * VEC) = pack16<T>(o);
  }
}

// ------------------------------------------------------------------------------------------------
static int pick_chunks(long long rows_per_group, int rslots, int groups) {
  // one 4-row round per thread while the tensor is small (a block's serial rounds are pure latency), more rows
  // per block only once the chunk count hits the cap
  long long ch = (rows_per_group + (long long)rslots * 4 - 1) / ((long long)rslots * 4);
  // finalize walks the chunks: keep it short, but big tensors need >2 blocks per CU in flight to stream at HBM rate
  long long cap = 512 / groups; if (cap < 1) cap = 1;
  if (ch > cap) ch = cap;
  if (ch < 1) ch = 1;
  return (int)ch;
}
static int red_geom(const GanTensor& t, int groups, int dtype, RedGeom* g) {
  const int vec = dtype == GAN_F32 ? 4 : 8;
  if (t.c <= 0 || t.c % 8 || t.pitch % 8 || groups <= 0 || t.n % groups) return GAN_E_SHAPE;
  g->C = t.c; g->cvecs = t.c / vec; g->hw = t.h * t.w; g->gsize = t.n / groups;
  if (g->cvecs > 256 || (g->cvecs & (g->cvecs - 1))) return GAN_E_SHAPE;
  g->rslots = 256 / g->cvecs;
  g->rows_per_group = (long long)g->gsize * g->hw;
  if ((long long)t.n * g->hw >= 0x7fffffffLL / 4) return GAN_E_SHAPE;      // 32-bit row / thread indices
  g->log2cv = ilog2_exact(g->cvecs);
  g->divRpg = make_fastdiv((uint32_t)g->rows_per_group);
  g->chunks = pick_chunks(g->rows_per_group, g->rslots, groups);
  return 0;
}
static size_t red_ws_bytes(int groups, int chunks, int c) {
  return ((size_t)groups * chunks * c * 2 + (size_t)groups * c * 2) * sizeof(float);
}

template <typename T, int MODE, int ACT = GAN_ACT_NONE, bool MASK = false>
static int launch_partial(const NormP& p, const RedGeom& g, int groups, float* partial, hipStream_t st) {
  size_t lds = (size_t)g.rslots * g.C * 2 * sizeof(float);
  GAN_LAUNCH((reduce_partial_kernel<T, MODE, ACT, MASK>), dim3(g.chunks, groups), dim3(256), lds, st, p, g, partial);
  GAN_CHECK_LAUNCH();
  return 0;
}

static long long apply_blocks(const RedGeom& g, long long rows) {
  return ((rows + 3) / 4 * g.cvecs + 255) / 256;
}
template <typename K>
static int launch_rows(K kern, const NormP& p, const RedGeom& g, long long rows, hipStream_t st) {
  GAN_LAUNCH(kern, dim3((unsigned)apply_blocks(g, rows)), dim3(256), 0, st, p, g, rows);
  GAN_CHECK_LAUNCH();
  return 0;
}
template <typename K>
static int launch_rows_fin(K kern, const NormP& p, const RedGeom& g, long long rows, const FinP& f, hipStream_t st) {
  GAN_LAUNCH(kern, dim3((unsigned)apply_blocks(g, rows)), dim3(256), 0, st, p, g, rows, f);
  GAN_CHECK_LAUNCH();
  return 0;
}
// the apply launch can carry the finalize: a sync area, the option, and at least as many workgroups as finalize work units
static bool fin_fits(const void* sync, const RedGeom& g, long long rows, int nfin) {
  return sync && gan_opt("norm.fin_in_apply") && (long long)nfin <= apply_blocks(g, rows);
}

// run F<ACT, MASK>() for the run-time (act, mask) pair
#define GAN_ACT_MASK_SWITCH(act, mask, F)                                                                   \
  switch ((act) * 2 + ((mask) ? 1 : 0)) {                                                                   \
    case GAN_ACT_LRELU * 2: return F(GAN_ACT_LRELU, false);                                                 \
    case GAN_ACT_LRELU * 2 + 1: return F(GAN_ACT_LRELU, true);                                              \
    case GAN_ACT_RELU * 2: return F(GAN_ACT_RELU, false);                                                   \
    case GAN_ACT_RELU * 2 + 1: return F(GAN_ACT_RELU, true);                                                \
    case GAN_ACT_TANH * 2: return F(GAN_ACT_TANH, false);                                                   \
    case GAN_ACT_TANH * 2 + 1: return F(GAN_ACT_TANH, true);                                                \
    case GAN_ACT_NONE * 2 + 1: return F(GAN_ACT_NONE, true);                                                \
    default: return F(GAN_ACT_NONE, false);                                                                 \
  }

// fin != nullptr: the launch also finalizes (FinP)
template <typename T>
static int launch_fwd(const NormP& p, const RedGeom& g, long long rows, const FinP* fin, hipStream_t st) {
  if (fin) {
#define F(A, M) launch_rows_fin(norm_act_fwd_kernel<T, A, M, true>, p, g, rows, *fin, st)
    GAN_ACT_MASK_SWITCH(p.act, p.mask != nullptr, F)
#undef F
  }
  const FinP none = {};
#define F(A, M) launch_rows_fin(norm_act_fwd_kernel<T, A, M, false>, p, g, rows, none, st)
  GAN_ACT_MASK_SWITCH(p.act, p.mask != nullptr, F)
#undef F
}
template <typename T>
static int launch_bwd_apply(const NormP& p, const RedGeom& g, long long rows, const FinP* fin, hipStream_t st) {
  if (fin) {
#define F(A, M) launch_rows_fin(norm_act_bwd_kernel<T, A, M, true>, p, g, rows, *fin, st)
    GAN_ACT_MASK_SWITCH(p.act, p.mask != nullptr, F)
#undef F
  }
  const FinP none = {};
#define F(A, M) launch_rows_fin(norm_act_bwd_kernel<T, A, M, false>, p, g, rows, none, st)
  GAN_ACT_MASK_SWITCH(p.act, p.mask != nullptr, F)
#undef F
}
template <typename T>
static int launch_bwd_partial(const NormP& p, const RedGeom& g, int groups, float* partial, hipStream_t st) {
#define F(A, M) launch_partial<T, 1, A, M>(p, g, groups, partial, st)
  GAN_ACT_MASK_SWITCH(p.act, p.mask != nullptr, F)
#undef F
}
template <typename T>
static int launch_act_bwd(const NormP& p, const RedGeom& g, long long rows, hipStream_t st) {
  switch (p.act) {
    case GAN_ACT_LRELU: return launch_rows(act_bwd_kernel<T, GAN_ACT_LRELU>, p, g, rows, st);
    case GAN_ACT_RELU: return launch_rows(act_bwd_kernel<T, GAN_ACT_RELU>, p, g, rows, st);
    case GAN_ACT_TANH: return launch_rows(act_bwd_kernel<T, GAN_ACT_TANH>, p, g, rows, st);
    default: return launch_rows(act_bwd_kernel<T, GAN_ACT_NONE>, p, g, rows, st);
  }
}

// finalize (sums, dgamma, dbeta) + apply of a normalisation backward: inside the apply launch when the descriptor has a sync area
// and the grid can carry it, a finalize launch in front of it otherwise
static int bwd_finalize_apply(const GanNormBwdDesc* d, const NormP& p, const RedGeom& g, long long rows, const float* partial, int chunks,
                              float* sums, hipStream_t st) {
  FinP f = {};
  f.partial = partial; f.sync = d->sync; f.G = d->groups; f.chunks = chunks; f.cb = (g.C + FC - 1) / FC; f.nfin = f.cb;
  f.sums = sums; f.dgamma = d->dgamma; f.dbeta = d->dbeta; f.accumulate = d->accumulate;
  const FinP* fin = nullptr;
  if (fin_fits(d->sync, g, rows, f.nfin)) {
    fin = &f;
  } else {
    GAN_LAUNCH(bwd_finalize_kernel, dim3(f.cb), dim3(256), 0, st, partial, d->groups, chunks, g.C, sums, d->dgamma, d->dbeta, d->accumulate);
    GAN_CHECK_LAUNCH();
  }
  return d->dtype == GAN_F32 ? launch_bwd_apply<float>(p, g, rows, fin, st) : d->dtype == GAN_F16 ? launch_bwd_apply<f16_t>(p, g, rows, fin, st)
                                                                                                 : launch_bwd_apply<bf16_t>(p, g, rows, fin, st);
}

extern "C" {

size_t gan_norm_workspace_bytes(int32_t groups, int32_t c, int64_t rows_per_group) {
  int cvecs = c / 4; if (cvecs < 1) cvecs = 1; if (cvecs > 256) cvecs = 256;   // fp32 layout = upper bound
  return red_ws_bytes(groups, pick_chunks(rows_per_group, 256 / cvecs, groups), c);
}

static int norm_stats_partial(const GanNormDesc* d, RedGeom* g, hipStream_t st) {
  if (!d || d->struct_size != sizeof(GanNormDesc) || !d->y.ptr || !d->mean || !d->rstd || !d->workspace) return GAN_E_ARG;
  int rc = red_geom(d->y, d->groups, d->dtype, g);
  if (rc) return rc;
  if (red_ws_bytes(d->groups, g->chunks, g->C) > d->workspace_bytes) return GAN_E_WORKSPACE;
  NormP p = {};
  p.y = d->y.ptr; p.ypitch = d->y.pitch;
  float* partial = (float*)d->workspace;
  return d->dtype == GAN_F32 ? launch_partial<float, 0>(p, *g, d->groups, partial, st)
       : d->dtype == GAN_F16 ? launch_partial<f16_t, 0>(p, *g, d->groups, partial, st)
                             : launch_partial<bf16_t, 0>(p, *g, d->groups, partial, st);
}

int gan_norm_stats(const GanNormDesc* d, gan_stream_t stream) {
  RedGeom g;
  hipStream_t st = (hipStream_t)stream;
  int rc = norm_stats_partial(d, &g, st);
  if (rc) return rc;
  GAN_LAUNCH(stats_finalize_kernel, dim3((g.C + FC - 1) / FC, d->moving_mean ? 1 : d->groups), dim3(256), 0, st,
                     (const float*)d->workspace, d->groups, g.chunks, g.C, g.rows_per_group, d->eps, d->mean, d->rstd,
                     d->moving_mean, d->moving_var, d->momentum);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_norm_stats_partial(const GanNormDesc* d, gan_stream_t stream) {
  RedGeom g;
  return norm_stats_partial(d, &g, (hipStream_t)stream);
}

size_t gan_norm_sync_bytes(void) { return (size_t)SYNC_WORDS * sizeof(unsigned); }
size_t gan_norm_sync_error_offset(void) { return (size_t)SYNC_ERR * sizeof(unsigned); }

int gan_norm_stats_finalize(const GanNormDesc* d, int32_t chunks, gan_stream_t stream) {
  if (!d || d->struct_size != sizeof(GanNormDesc) || !d->y.ptr || !d->mean || !d->rstd || !d->workspace || chunks <= 0) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->y, d->groups, d->dtype, &g);
  if (rc) return rc;
  if ((size_t)d->groups * chunks * g.C * 2 * sizeof(float) > d->workspace_bytes) return GAN_E_WORKSPACE;   // the producer's partials
  GAN_LAUNCH(stats_finalize_kernel, dim3((g.C + FC - 1) / FC, d->moving_mean ? 1 : d->groups), dim3(256), 0, (hipStream_t)stream,
                     (const float*)d->workspace, d->groups, chunks, g.C, g.rows_per_group, d->eps, d->mean, d->rstd,
                     d->moving_mean, d->moving_var, d->momentum);
  GAN_CHECK_LAUNCH();
  return 0;
}

static int norm_act_fwd_impl(const GanNormDesc* d, const FinP* fin_in, int32_t chunks, hipStream_t st) {
  if (!d || d->struct_size != sizeof(GanNormDesc) || !d->y.ptr || !d->a.ptr || !d->mean || !d->rstd || !d->gamma || !d->beta) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->y, d->groups, d->dtype, &g);
  if (rc) return rc;
  if (d->a.pitch % 8 || d->a.c != d->y.c) return GAN_E_SHAPE;
  NormP p = {};
  p.y = d->y.ptr; p.ypitch = d->y.pitch; p.out = d->a.ptr; p.outpitch = d->a.pitch;
  p.gamma = d->gamma; p.beta = d->beta; p.mean = d->mean; p.rstd = d->rstd; p.mask = d->dropmask;
  p.act = d->act; p.slope = d->slope;
  long long rows = (long long)d->y.n * g.hw;
  FinP f = {};
  const FinP* fin = nullptr;
  if (fin_in) {              // finalize + apply: one launch when the apply grid can carry the finalize, two otherwise
    if (!d->workspace) return GAN_E_ARG;
    if (chunks <= 0) chunks = g.chunks;        // the partials gan_norm_stats_partial() wrote
    if ((size_t)d->groups * chunks * g.C * 2 * sizeof(float) > d->workspace_bytes) return GAN_E_WORKSPACE;
    f.partial = (const float*)d->workspace; f.sync = d->sync; f.G = d->groups; f.chunks = chunks;
    f.cb = (g.C + FC - 1) / FC; f.nfin = f.cb * (d->moving_mean ? 1 : d->groups);
    f.rows = g.rows_per_group; f.eps = d->eps; f.momentum = d->momentum; f.mean = d->mean; f.rstd = d->rstd;
    f.mmean = d->moving_mean; f.mvar = d->moving_var;
    if (fin_fits(d->sync, g, rows, f.nfin)) {
      fin = &f;
    } else {
      GAN_LAUNCH(stats_finalize_kernel, dim3(f.cb, d->moving_mean ? 1 : d->groups), dim3(256), 0, st, f.partial, d->groups, chunks, g.C,
                 g.rows_per_group, d->eps, d->mean, d->rstd, d->moving_mean, d->moving_var, d->momentum);
      GAN_CHECK_LAUNCH();
    }
  }
  return d->dtype == GAN_F32 ? launch_fwd<float>(p, g, rows, fin, st) : d->dtype == GAN_F16 ? launch_fwd<f16_t>(p, g, rows, fin, st) : launch_fwd<bf16_t>(p, g, rows, fin, st);
}

int gan_norm_act_fwd(const GanNormDesc* d, gan_stream_t stream) {
  return norm_act_fwd_impl(d, nullptr, 0, (hipStream_t)stream);
}

int gan_norm_finalize_act_fwd(const GanNormDesc* d, int32_t chunks, gan_stream_t stream) {
  const FinP want = {};
  return norm_act_fwd_impl(d, &want, chunks, (hipStream_t)stream);
}

int gan_norm_act_bwd(const GanNormBwdDesc* d, gan_stream_t stream) {
  if (!d || d->struct_size != sizeof(GanNormBwdDesc) || !d->y.ptr || !d->da.ptr || !d->dy.ptr || !d->mean || !d->rstd || !d->gamma ||
      !d->beta || !d->workspace)
    return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->y, d->groups, d->dtype, &g);
  if (rc) return rc;
  if (d->da.pitch % 8 || d->dy.pitch % 8 || (d->da2.ptr && d->da2.pitch % 8)) return GAN_E_SHAPE;
  if (red_ws_bytes(d->groups, g.chunks, g.C) > d->workspace_bytes) return GAN_E_WORKSPACE;
  NormP p = {};
  p.y = d->y.ptr; p.ypitch = d->y.pitch; p.da = d->da.ptr; p.dapitch = d->da.pitch;
  p.da2 = d->da2.ptr; p.da2pitch = d->da2.pitch; p.out = d->dy.ptr; p.outpitch = d->dy.pitch;
  p.gamma = d->gamma; p.beta = d->beta; p.mean = d->mean; p.rstd = d->rstd; p.mask = d->dropmask;
  p.act = d->act; p.slope = d->slope; p.has_norm = 1;
  float* partial = (float*)d->workspace;
  float* sums = partial + (size_t)d->groups * g.chunks * g.C * 2;
  p.sums = sums;
  hipStream_t st = (hipStream_t)stream;
  rc = d->dtype == GAN_F32 ? launch_bwd_partial<float>(p, g, d->groups, partial, st)
       : d->dtype == GAN_F16 ? launch_bwd_partial<f16_t>(p, g, d->groups, partial, st)
                             : launch_bwd_partial<bf16_t>(p, g, d->groups, partial, st);
  if (rc) return rc;
  long long rows = (long long)d->y.n * g.hw;
  return bwd_finalize_apply(d, p, g, rows, partial, g.chunks, sums, st);
}

/* Second half of a normalisation backward whose first half ran in the producing dgrad's epilogue (GanBwdFuse): d->da holds
 * dz, d->workspace the producer's partial sums [groups][chunks][C][2] followed by room for the finalized [groups][C][2]. */
int gan_norm_act_bwd_fused(const GanNormBwdDesc* d, int32_t chunks, gan_stream_t stream) {
  if (!d || d->struct_size != sizeof(GanNormBwdDesc) || !d->y.ptr || !d->da.ptr || !d->dy.ptr || !d->mean || !d->rstd || !d->gamma ||
      !d->beta || !d->workspace || chunks <= 0 || d->da2.ptr)
    return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->y, d->groups, d->dtype, &g);
  if (rc) return rc;
  if (d->da.pitch % 8 || d->dy.pitch % 8) return GAN_E_SHAPE;
  if (red_ws_bytes(d->groups, chunks, g.C) > d->workspace_bytes) return GAN_E_WORKSPACE;
  NormP p = {};
  p.y = d->y.ptr; p.ypitch = d->y.pitch; p.da = d->da.ptr; p.dapitch = d->da.pitch;
  p.out = d->dy.ptr; p.outpitch = d->dy.pitch;
  p.gamma = d->gamma; p.beta = d->beta; p.mean = d->mean; p.rstd = d->rstd;
  p.act = GAN_ACT_NONE; p.slope = d->slope; p.has_norm = 1;        // dz already carries activation derivative and mask
  float* partial = (float*)d->workspace;
  float* sums = partial + (size_t)d->groups * chunks * g.C * 2;
  p.sums = sums;
  long long rows = (long long)d->y.n * g.hw;
  return bwd_finalize_apply(d, p, g, rows, partial, chunks, sums, (hipStream_t)stream);
}

static int bias_grad_impl(int32_t dtype, const GanTensor& dy, float* dbias, int32_t accumulate, void* workspace,
                          size_t workspace_bytes, hipStream_t st) {
  RedGeom g;
  int rc = red_geom(dy, 1, dtype, &g);
  if (rc) return rc;
  if (!workspace || red_ws_bytes(1, g.chunks, g.C) > workspace_bytes) return GAN_E_WORKSPACE;
  NormP q = {};
  q.da = dy.ptr; q.dapitch = dy.pitch;
  float* partial = (float*)workspace;
  rc = dtype == GAN_F32 ? launch_partial<float, 2>(q, g, 1, partial, st) : dtype == GAN_F16 ? launch_partial<f16_t, 2>(q, g, 1, partial, st) : launch_partial<bf16_t, 2>(q, g, 1, partial, st);
  if (rc) return rc;
  GAN_LAUNCH(bwd_finalize_kernel, dim3((g.C + FC - 1) / FC), dim3(256), 0, st, (const float*)partial, 1, g.chunks, g.C,
                     (float*)nullptr, (float*)nullptr, dbias, accumulate);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_act_bwd(const GanActBwdDesc* d, gan_stream_t stream) {
  if (!d || d->struct_size != sizeof(GanActBwdDesc) || !d->a.ptr || !d->da.ptr || !d->dy.ptr) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->a, 1, d->dtype, &g);
  if (rc) return rc;
  if (d->da.pitch % 8 || d->dy.pitch % 8 || (d->da2.ptr && d->da2.pitch % 8)) return GAN_E_SHAPE;
  NormP p = {};
  p.y = d->a.ptr; p.ypitch = d->a.pitch; p.da = d->da.ptr; p.dapitch = d->da.pitch;
  p.da2 = d->da2.ptr; p.da2pitch = d->da2.pitch; p.out = d->dy.ptr; p.outpitch = d->dy.pitch;
  p.act = d->act; p.slope = d->slope; p.has_norm = 0;
  hipStream_t st = (hipStream_t)stream;
  long long rows = (long long)d->a.n * g.hw;
  rc = d->dtype == GAN_F32 ? launch_act_bwd<float>(p, g, rows, st) : d->dtype == GAN_F16 ? launch_act_bwd<f16_t>(p, g, rows, st) : launch_act_bwd<bf16_t>(p, g, rows, st);
  if (rc) return rc;
  if (d->dbias) return bias_grad_impl(d->dtype, d->dy, d->dbias, d->accumulate, d->workspace, d->workspace_bytes, st);
  return 0;
}

int gan_bias_grad(int32_t dtype, const GanTensor* dy, float* dbias, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, gan_stream_t stream) {
  if (!dy || !dy->ptr || !dbias) return GAN_E_ARG;
  return bias_grad_impl(dtype, *dy, dbias, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}
}  // extern "C"
