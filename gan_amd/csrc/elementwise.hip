// HBM-bound kernels of the training step: normalisation statistics, norm+dropout+activation forward and
// backward, activation backward, BCE-from-logits / L1 losses with fused gradients, TF-form Adam, weight
// layout preparation, dropout-mask generation and fp32<->typed packing.  All global accesses are 16-byte
// vectors (8 bf16 / 4 fp32 channels per lane) on NHWC rows; per-channel reductions are two-stage
// (deterministic block partials -> double-precision finalize), never atomics.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// generic per-(group, channel) two-value reduction over NHWC rows
// ------------------------------------------------------------------------------------------------
struct RedGeom {
  int C, cvecs;            // channels, 16-byte vectors per row
  long long rows_per_group;
  int chunks;              // row chunks per group (gridDim.x)
  int hw, gsize;           // pixels per image, images per group
};

// Block reduction: thread (cvi, rslot) holds VEC pairs; result for channel c lands in out2[c*2 + {0,1}].
template <int VEC>
__device__ __forceinline__ void block_reduce_cols(float (&s1)[VEC], float (&s2)[VEC], int cv, int rslot, int rslots,
                                                  int C, float* lds, float* out2) {
  // lds: [rslots][C][2]
  if (cv >= 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      lds[((size_t)rslot * C + cv * VEC + e) * 2 + 0] = s1[e];
      lds[((size_t)rslot * C + cv * VEC + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < rslots; ++k) { a += lds[((size_t)k * C + c) * 2]; b += lds[((size_t)k * C + c) * 2 + 1]; }
    out2[c * 2] = a; out2[c * 2 + 1] = b;
  }
}

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

// dz for one element (shared by the bwd reduce and apply passes)
__device__ __forceinline__ float bwd_dz(float da, float z, float mk, int act, float slope) {
  float zd = z * mk;                // mk = 2*mask or 1
  float g;
  if (act == GAN_ACT_LRELU) g = zd > 0.f ? 1.f : slope;
  else if (act == GAN_ACT_RELU) g = zd > 0.f ? 1.f : 0.f;
  else g = 1.f;
  return da * g * mk;
}

// MODE 0: stats (y, y^2).  MODE 1: norm backward sums (dz, dz*xhat).  MODE 2: column sum of `da` (bias grad).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const NormP p, const RedGeom g, float* partial) {
  constexpr int VEC = VecOf<T>::N;
  extern __shared__ float lds[];
  const int grp = blockIdx.y, chunk = blockIdx.x;
  const int cvl = g.cvecs < 256 ? g.cvecs : 256;
  const int rslots = 256 / cvl;
  const int cvi = threadIdx.x % cvl, rslot = threadIdx.x / cvl;
  const long long r0 = g.rows_per_group * chunk / g.chunks, r1 = g.rows_per_group * (chunk + 1) / g.chunks;
  const long long rowbase = (long long)grp * g.rows_per_group;
  float* out2 = partial + ((size_t)grp * g.chunks + chunk) * g.C * 2;
  for (int cv = cvi; cv < g.cvecs; cv += cvl) {   // loop only runs >1x when C/VEC > 256
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
    float sc[VEC], sh[VEC], mu[VEC], rs[VEC];
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        int c = cv * VEC + e;
        mu[e] = p.mean[grp * g.C + c]; rs[e] = p.rstd[grp * g.C + c];
        sc[e] = p.gamma[c]; sh[e] = p.beta[c];
      }
    }
    if (rslot < rslots) {
      for (long long rr = r0 + rslot; rr < r1; rr += rslots) {
        long long row = rowbase + rr;
        if (MODE == 0) {
          float v[VEC];
          unpack16<T>(*(const uint4*)((const T*)p.y + row * p.ypitch + cv * VEC), v);
#pragma unroll
          for (int e = 0; e < VEC; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
        } else if (MODE == 1) {
          float yv[VEC], dv[VEC];
          unpack16<T>(*(const uint4*)((const T*)p.y + row * p.ypitch + cv * VEC), yv);
          unpack16<T>(*(const uint4*)((const T*)p.da + row * p.dapitch + cv * VEC), dv);
          if (p.da2) {
            float d2[VEC];
            unpack16<T>(*(const uint4*)((const T*)p.da2 + row * p.da2pitch + cv * VEC), d2);
#pragma unroll
            for (int e = 0; e < VEC; ++e) dv[e] += d2[e];
          }
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float xh = (yv[e] - mu[e]) * rs[e];
            float z = sc[e] * xh + sh[e];
            float mk = p.mask ? 2.f * (float)p.mask[row * g.C + cv * VEC + e] : 1.f;
            float dz = bwd_dz(dv[e], z, mk, p.act, p.slope);
            s1[e] += dz; s2[e] += dz * xh;
          }
        } else {
          float dv[VEC];
          unpack16<T>(*(const uint4*)((const T*)p.da + row * p.dapitch + cv * VEC), dv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) s1[e] += dv[e];
        }
      }
    }
    block_reduce_cols<VEC>(s1, s2, rslot < rslots ? cv : -1, rslot, rslots, g.C, lds, out2);
    __syncthreads();
  }
}

// finalize stats: one thread per channel, groups in order (moving averages are updated once per group,
// like two successive BatchNormalization calls).
__global__ void stats_finalize_kernel(const float* partial, int G, int chunks, int C, long long rows, float eps,
                                      float* mean, float* rstd, float* mmean, float* mvar, float momentum) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  for (int g = 0; g < G; ++g) {
    double s = 0, s2 = 0;
    for (int k = 0; k < chunks; ++k) {
      s += partial[(((size_t)g * chunks + k) * C + c) * 2];
      s2 += partial[(((size_t)g * chunks + k) * C + c) * 2 + 1];
    }
    double m = s / (double)rows;
    double var = s2 / (double)rows - m * m;
    if (var < 0) var = 0;
    float vf = (float)var;
    mean[g * C + c] = (float)m;
    rstd[g * C + c] = 1.0f / sqrtf(vf + eps);
    if (mmean) {
      double adj = (double)rows / (double)(rows > 1 ? rows - 1 : 1);
      mmean[c] += ((float)m - mmean[c]) * (1.f - momentum);
      mvar[c] += ((float)(var * adj) - mvar[c]) * (1.f - momentum);
    }
  }
}

__global__ void bwd_finalize_kernel(const float* partial, int G, int chunks, int C, float* sums, float* dgamma,
                                    float* dbeta, int accumulate) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double tg = 0, tb = 0;
  for (int g = 0; g < G; ++g) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k < chunks; ++k) {
      s1 += partial[(((size_t)g * chunks + k) * C + c) * 2];
      s2 += partial[(((size_t)g * chunks + k) * C + c) * 2 + 1];
    }
    if (sums) { sums[(g * C + c) * 2] = (float)s1; sums[(g * C + c) * 2 + 1] = (float)s2; }
    tb += s1; tg += s2;
  }
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)tg;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)tb;
}

// forward: a = act(dropout(gamma*(y-mean)*rstd + beta))
template <typename T>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const NormP p, const RedGeom g, long long nvec) {
  constexpr int VEC = VecOf<T>::N;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nvec) return;
  long long row = idx / g.cvecs;
  int cv = (int)(idx % g.cvecs);
  int grp = (int)((row / g.hw) / g.gsize);
  float v[VEC], o[VEC];
  unpack16<T>(*(const uint4*)((const T*)p.y + row * p.ypitch + cv * VEC), v);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    int c = cv * VEC + e;
    float xh = (v[e] - p.mean[grp * g.C + c]) * p.rstd[grp * g.C + c];
    float z = p.gamma[c] * xh + p.beta[c];
    if (p.mask) z *= 2.f * (float)p.mask[row * g.C + c];
    o[e] = apply_act(z, p.act, p.slope);
  }
  *(uint4*)((T*)p.out + row * p.outpitch + cv * VEC) = pack16<T>(o);
}

// backward apply: dy = gamma*rstd*(dz - S1/R - xhat*S2/R)      (has_norm)
//                 dy = (da+da2) * act'(a)                       (!has_norm; y holds the saved activation a)
template <typename T>
__global__ __launch_bounds__(256) void norm_act_bwd_kernel(const NormP p, const RedGeom g, long long nvec) {
  constexpr int VEC = VecOf<T>::N;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nvec) return;
  long long row = idx / g.cvecs;
  int cv = (int)(idx % g.cvecs);
  float yv[VEC], dv[VEC], o[VEC];
  unpack16<T>(*(const uint4*)((const T*)p.y + row * p.ypitch + cv * VEC), yv);
  unpack16<T>(*(const uint4*)((const T*)p.da + row * p.dapitch + cv * VEC), dv);
  if (p.da2) {
    float d2[VEC];
    unpack16<T>(*(const uint4*)((const T*)p.da2 + row * p.da2pitch + cv * VEC), d2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) dv[e] += d2[e];
  }
  if (p.has_norm) {
    int grp = (int)((row / g.hw) / g.gsize);
    float invR = 1.0f / (float)g.rows_per_group;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      int c = cv * VEC + e;
      float rs = p.rstd[grp * g.C + c];
      float xh = (yv[e] - p.mean[grp * g.C + c]) * rs;
      float z = p.gamma[c] * xh + p.beta[c];
      float mk = p.mask ? 2.f * (float)p.mask[row * g.C + c] : 1.f;
      float dz = bwd_dz(dv[e], z, mk, p.act, p.slope);
      float s1 = p.sums[(grp * g.C + c) * 2], s2 = p.sums[(grp * g.C + c) * 2 + 1];
      o[e] = p.gamma[c] * rs * (dz - s1 * invR - xh * s2 * invR);
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float a = yv[e], gq;
      if (p.act == GAN_ACT_LRELU) gq = a > 0.f ? 1.f : p.slope;
      else if (p.act == GAN_ACT_RELU) gq = a > 0.f ? 1.f : 0.f;
      else if (p.act == GAN_ACT_TANH) gq = 1.f - a * a;
      else gq = 1.f;
      o[e] = dv[e] * gq;
    }
  }
  *(uint4*)((T*)p.out + row * p.outpitch + cv * VEC) = pack16<T>(o);
}

// ------------------------------------------------------------------------------------------------
static int red_geom(const GanTensor& t, int groups, int dtype, RedGeom* g) {
  const int vec = dtype == GAN_F32 ? 4 : 8;
  if (t.c <= 0 || t.c % 8 || t.pitch % 8 || groups <= 0 || t.n % groups) return GAN_E_SHAPE;
  g->C = t.c; g->cvecs = t.c / vec; g->hw = t.h * t.w; g->gsize = t.n / groups;
  g->rows_per_group = (long long)g->gsize * g->hw;
  long long ch = g->rows_per_group / 512;
  long long cap = 1024 / groups; if (cap < 1) cap = 1;
  if (ch > cap) ch = cap;
  if (ch < 1) ch = 1;
  g->chunks = (int)ch;
  return 0;
}
static size_t red_ws_bytes(int groups, int chunks, int c) {
  return ((size_t)groups * chunks * c * 2 + (size_t)groups * c * 2) * sizeof(float);
}

template <typename T, int MODE>
static int launch_partial(const NormP& p, const RedGeom& g, int groups, float* partial, hipStream_t st) {
  constexpr int VEC = VecOf<T>::N;
  size_t lds = (size_t)256 * VEC * 2 * sizeof(float);
  if ((size_t)g.C * 2 * sizeof(float) * (256 / (g.cvecs < 256 ? g.cvecs : 256)) > lds)
    lds = (size_t)g.C * 2 * sizeof(float) * (256 / (g.cvecs < 256 ? g.cvecs : 256));
  hipLaunchKernelGGL((reduce_partial_kernel<T, MODE>), dim3(g.chunks, groups), dim3(256), lds, st, p, g, partial);
  GAN_CHECK_LAUNCH();
  return 0;
}

extern "C" {

size_t gan_norm_workspace_bytes(int32_t groups, int32_t c, int64_t rows_per_group) {
  long long ch = rows_per_group / 512;
  long long cap = 1024 / groups; if (cap < 1) cap = 1;
  if (ch > cap) ch = cap;
  if (ch < 1) ch = 1;
  return red_ws_bytes(groups, (int)ch, c);
}

int gan_norm_stats(const GanNormDesc* d, gan_stream_t stream) {
  if (!d || !d->y.ptr || !d->mean || !d->rstd || !d->workspace) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->y, d->groups, d->dtype, &g);
  if (rc) return rc;
  if (red_ws_bytes(d->groups, g.chunks, g.C) > d->workspace_bytes) return GAN_E_WORKSPACE;
  NormP p = {};
  p.y = d->y.ptr; p.ypitch = d->y.pitch;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)d->workspace;
  rc = d->dtype == GAN_F32 ? launch_partial<float, 0>(p, g, d->groups, partial, st)
                           : launch_partial<bf16_t, 0>(p, g, d->groups, partial, st);
  if (rc) return rc;
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((g.C + 127) / 128), dim3(128), 0, st, (const float*)partial,
                     d->groups, g.chunks, g.C, g.rows_per_group, d->eps, d->mean, d->rstd, d->moving_mean,
                     d->moving_var, d->momentum);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_norm_act_fwd(const GanNormDesc* d, gan_stream_t stream) {
  if (!d || !d->y.ptr || !d->a.ptr || !d->mean || !d->rstd || !d->gamma || !d->beta) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->y, d->groups, d->dtype, &g);
  if (rc) return rc;
  if (d->a.pitch % 8 || d->a.c != d->y.c) return GAN_E_SHAPE;
  NormP p = {};
  p.y = d->y.ptr; p.ypitch = d->y.pitch; p.out = d->a.ptr; p.outpitch = d->a.pitch;
  p.gamma = d->gamma; p.beta = d->beta; p.mean = d->mean; p.rstd = d->rstd; p.mask = d->dropmask;
  p.act = d->act; p.slope = d->slope;
  long long nvec = (long long)d->y.n * g.hw * g.cvecs;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((nvec + 255) / 256));
  if (d->dtype == GAN_F32) hipLaunchKernelGGL(norm_act_fwd_kernel<float>, grid, dim3(256), 0, st, p, g, nvec);
  else hipLaunchKernelGGL(norm_act_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, p, g, nvec);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_norm_act_bwd(const GanNormBwdDesc* d, gan_stream_t stream) {
  if (!d || !d->y.ptr || !d->da.ptr || !d->dy.ptr || !d->mean || !d->rstd || !d->gamma || !d->beta || !d->workspace)
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
  rc = d->dtype == GAN_F32 ? launch_partial<float, 1>(p, g, d->groups, partial, st)
                           : launch_partial<bf16_t, 1>(p, g, d->groups, partial, st);
  if (rc) return rc;
  hipLaunchKernelGGL(bwd_finalize_kernel, dim3((g.C + 127) / 128), dim3(128), 0, st, (const float*)partial, d->groups,
                     g.chunks, g.C, sums, d->dgamma, d->dbeta, d->accumulate);
  GAN_CHECK_LAUNCH();
  long long nvec = (long long)d->y.n * g.hw * g.cvecs;
  dim3 grid((unsigned)((nvec + 255) / 256));
  if (d->dtype == GAN_F32) hipLaunchKernelGGL(norm_act_bwd_kernel<float>, grid, dim3(256), 0, st, p, g, nvec);
  else hipLaunchKernelGGL(norm_act_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, p, g, nvec);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_act_bwd(const GanActBwdDesc* d, gan_stream_t stream) {
  if (!d || !d->a.ptr || !d->da.ptr || !d->dy.ptr) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(d->a, 1, d->dtype, &g);
  if (rc) return rc;
  if (d->da.pitch % 8 || d->dy.pitch % 8 || (d->da2.ptr && d->da2.pitch % 8)) return GAN_E_SHAPE;
  NormP p = {};
  p.y = d->a.ptr; p.ypitch = d->a.pitch; p.da = d->da.ptr; p.dapitch = d->da.pitch;
  p.da2 = d->da2.ptr; p.da2pitch = d->da2.pitch; p.out = d->dy.ptr; p.outpitch = d->dy.pitch;
  p.act = d->act; p.slope = d->slope; p.has_norm = 0;
  hipStream_t st = (hipStream_t)stream;
  long long nvec = (long long)d->a.n * g.hw * g.cvecs;
  dim3 grid((unsigned)((nvec + 255) / 256));
  if (d->dtype == GAN_F32) hipLaunchKernelGGL(norm_act_bwd_kernel<float>, grid, dim3(256), 0, st, p, g, nvec);
  else hipLaunchKernelGGL(norm_act_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, p, g, nvec);
  GAN_CHECK_LAUNCH();
  if (d->dbias) {
    if (!d->workspace || red_ws_bytes(1, g.chunks, g.C) > d->workspace_bytes) return GAN_E_WORKSPACE;
    NormP q = {};
    q.da = d->dy.ptr; q.dapitch = d->dy.pitch;
    float* partial = (float*)d->workspace;
    rc = d->dtype == GAN_F32 ? launch_partial<float, 2>(q, g, 1, partial, st) : launch_partial<bf16_t, 2>(q, g, 1, partial, st);
    if (rc) return rc;
    // dbias has the REAL channel count <= 8-padded C: finalize writes C entries, caller passes an 8-padded buffer
    hipLaunchKernelGGL(bwd_finalize_kernel, dim3((g.C + 127) / 128), dim3(128), 0, st, (const float*)partial, 1, g.chunks,
                       g.C, (float*)nullptr, (float*)nullptr, d->dbias, d->accumulate);
    GAN_CHECK_LAUNCH();
  }
  return 0;
}
}  // extern "C"

// ------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void bce_kernel(const float* x, long long count, float target, float loss_scale,
                                                   int loss_acc, float* loss_out, float grad_scale, T* dx, int dx_pitch) {
  __shared__ float red[16];
  float s = 0.f;
  const float inv = 1.0f / (float)count;
  for (long long i = threadIdx.x; i < count; i += 1024) {
    float v = x[i];
    float e = expf(-fabsf(v));
    s += fmaxf(v, 0.f) - v * target + log1pf(e);
    if (dx) {
      float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      st_f(dx + i * dx_pitch, grad_scale * (sig - target) * inv);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    t = t * inv * loss_scale;
    loss_out[0] = loss_acc ? loss_out[0] + t : t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void l1_kernel(const T* a, int apitch, const T* b, int bpitch, int C, long long pixels,
                                                 float gscale, T* da, int dapitch, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
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
__global__ void l1_finalize_kernel(const float* partial, int n, double inv_count, float loss_scale, int acc, float* loss_out) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += partial[i];
  float t = (float)(s * inv_count) * loss_scale;
  loss_out[0] = acc ? loss_out[0] + t : t;
}

// ------------------------------------------------------------------------------------------------
// Adam (TF form), weight prep, dropout, pack
// ------------------------------------------------------------------------------------------------
__global__ void adam_begin_kernel(int32_t* step, float* lr_t, float lr, float b1, float b2) {
  int t = *step + 1;
  *step = t;
  double v = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
  *lr_t = (float)v;
}
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v,
                                                   const float4* __restrict__ g, long long nvec, const float* lr_t,
                                                   float omb1, float omb2, float eps, float gscale) {
  const float lr = *lr_t;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    float4 pp = p[i], mm = m[i], vv = v[i], gg = g[i];
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

template <typename T>
__global__ __launch_bounds__(256) void wprep_kernel(const float* master, int A, int B, T* nat, T* tr) {
  const int B8 = (B + 7) & ~7, A8 = (A + 7) & ~7;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (nat) {
    long long tot = (long long)16 * A * B8;
    if (i < tot) {
      int b = (int)(i % B8);
      long long t = i / B8;
      int a = (int)(t % A), tap = (int)(t / A);
      st_f(nat + i, b < B ? master[((size_t)tap * A + a) * B + b] : 0.f);
    }
  }
  if (tr) {
    long long tot = (long long)16 * B * A8;
    if (i < tot) {
      int a = (int)(i % A8);
      long long t = i / A8;
      int b = (int)(t % B), tap = (int)(t / B);
      st_f(tr + i, a < A ? master[((size_t)tap * A + a) * B + b] : 0.f);
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

extern "C" {

int gan_bce_logits(const float* x, int64_t count, float target, float loss_scale, int32_t loss_accumulate,
                   float* loss_out, float grad_scale, int32_t dtype, void* dx, int32_t dx_pitch, gan_stream_t stream) {
  if (!x || count <= 0 || !loss_out) return GAN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GAN_F32)
    hipLaunchKernelGGL(bce_kernel<float>, dim3(1), dim3(1024), 0, st, x, (long long)count, target, loss_scale,
                       loss_accumulate, loss_out, grad_scale, (float*)dx, dx_pitch);
  else
    hipLaunchKernelGGL(bce_kernel<bf16_t>, dim3(1), dim3(1024), 0, st, x, (long long)count, target, loss_scale,
                       loss_accumulate, loss_out, grad_scale, (bf16_t*)dx, dx_pitch);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_l1(int32_t dtype, const GanTensor* a, const GanTensor* b, float loss_scale, int32_t loss_accumulate,
           float* loss_out, float grad_scale, const GanTensor* da, float* workspace, gan_stream_t stream) {
  if (!a || !b || !a->ptr || !b->ptr || !loss_out || !workspace) return GAN_E_ARG;
  if (a->n != b->n || a->h != b->h || a->w != b->w || a->c != b->c) return GAN_E_SHAPE;
  long long pixels = (long long)a->n * a->h * a->w;
  long long total = pixels * a->c;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  float gs = grad_scale / (float)total;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GAN_F32)
    hipLaunchKernelGGL(l1_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)a->ptr, a->pitch, (const float*)b->ptr,
                       b->pitch, a->c, pixels, gs, da ? (float*)da->ptr : nullptr, da ? da->pitch : 0, workspace);
  else
    hipLaunchKernelGGL(l1_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)a->ptr, a->pitch,
                       (const bf16_t*)b->ptr, b->pitch, a->c, pixels, gs, da ? (bf16_t*)da->ptr : nullptr,
                       da ? da->pitch : 0, workspace);
  GAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(1), 0, st, (const float*)workspace, blocks, 1.0 / (double)total,
                     loss_scale, loss_accumulate, loss_out);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_adam_begin(int32_t* step, float* lr_t, float lr, float beta1, float beta2, gan_stream_t stream) {
  if (!step || !lr_t) return GAN_E_ARG;
  hipLaunchKernelGGL(adam_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, lr_t, lr, beta1, beta2);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_adam_tf(float* param, float* m, float* v, const float* grad, int64_t count, const float* lr_t, float beta1,
                float beta2, float eps, float grad_scale, gan_stream_t stream) {
  if (!param || !m || !v || !grad || !lr_t || count <= 0 || count % 4) return GAN_E_ARG;
  long long nvec = count / 4;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param, (float4*)m,
                     (float4*)v, (const float4*)grad, nvec, lr_t, 1.f - beta1, 1.f - beta2, eps, grad_scale);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_weights_prepare(const float* master, int32_t A, int32_t B, int32_t dtype, void* nk_native, void* nk_transposed,
                        gan_stream_t stream) {
  if (!master || A <= 0 || B <= 0 || (!nk_native && !nk_transposed)) return GAN_E_ARG;
  long long B8 = (B + 7) & ~7, A8 = (A + 7) & ~7;
  long long t1 = nk_native ? 16LL * A * B8 : 0, t2 = nk_transposed ? 16LL * B * A8 : 0;
  long long tot = t1 > t2 ? t1 : t2;
  dim3 grid((unsigned)((tot + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GAN_F32)
    hipLaunchKernelGGL(wprep_kernel<float>, grid, dim3(256), 0, st, master, A, B, (float*)nk_native, (float*)nk_transposed);
  else
    hipLaunchKernelGGL(wprep_kernel<bf16_t>, grid, dim3(256), 0, st, master, A, B, (bf16_t*)nk_native, (bf16_t*)nk_transposed);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_dropout_mask(uint8_t* mask, int64_t count, uint64_t seed, const int32_t* step, uint32_t stream_id, gan_stream_t stream) {
  if (!mask || count <= 0 || !step) return GAN_E_ARG;
  long long words = (count + 7) / 8;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask,
                     (long long)count, seed, step, stream_id);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_pack(int32_t dtype, const float* src, const GanTensor* dst, gan_stream_t stream) {
  if (!src || !dst || !dst->ptr) return GAN_E_ARG;
  long long total = (long long)dst->n * dst->h * dst->w * dst->c;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == GAN_F32)
    hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst->ptr, dst->c, dst->pitch, total);
  else
    hipLaunchKernelGGL(pack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst->ptr, dst->c, dst->pitch, total);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_unpack(int32_t dtype, const GanTensor* src, float* dst, gan_stream_t stream) {
  if (!src || !dst || !src->ptr) return GAN_E_ARG;
  long long total = (long long)src->n * src->h * src->w * src->c;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == GAN_F32)
    hipLaunchKernelGGL(unpack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src->ptr, dst, src->c, src->pitch, total);
  else
    hipLaunchKernelGGL(unpack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src->ptr, dst, src->c, src->pitch, total);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_copy_view(int32_t dtype, const GanTensor* src, const GanTensor* dst, gan_stream_t stream) {
  if (!src || !dst || !src->ptr || !dst->ptr) return GAN_E_ARG;
  if (src->n != dst->n || src->h != dst->h || src->w != dst->w || src->c != dst->c) return GAN_E_SHAPE;
  long long total = (long long)src->n * src->h * src->w * src->c;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == GAN_F32)
    hipLaunchKernelGGL(copy_view_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src->ptr, src->pitch,
                       (float*)dst->ptr, dst->pitch, src->c, total);
  else
    hipLaunchKernelGGL(copy_view_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src->ptr, src->pitch,
                       (bf16_t*)dst->ptr, dst->pitch, src->c, total);
  GAN_CHECK_LAUNCH();
  return 0;
}

int gan_bias_grad(int32_t dtype, const GanTensor* dy, float* dbias, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, gan_stream_t stream) {
  if (!dy || !dy->ptr || !dbias || !workspace) return GAN_E_ARG;
  RedGeom g;
  int rc = red_geom(*dy, 1, dtype, &g);
  if (rc) return rc;
  if (red_ws_bytes(1, g.chunks, g.C) > workspace_bytes) return GAN_E_WORKSPACE;
  NormP q = {};
  q.da = dy->ptr; q.dapitch = dy->pitch;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  rc = dtype == GAN_F32 ? launch_partial<float, 2>(q, g, 1, partial, st) : launch_partial<bf16_t, 2>(q, g, 1, partial, st);
  if (rc) return rc;
  hipLaunchKernelGGL(bwd_finalize_kernel, dim3((g.C + 127) / 128), dim3(128), 0, st, (const float*)partial, 1, g.chunks,
                     g.C, (float*)nullptr, (float*)nullptr, dbias, accumulate);
  GAN_CHECK_LAUNCH();
  return 0;
}

const char* gan_version(void) { return "gan_amd 0.1 (gfx950)"; }
}
