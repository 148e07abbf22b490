// Launch parameters of the convolution kernels (conv_gemm.hip plans them; thin.hip reuses the geometry).
#pragma once
#include "common.h"

struct GemmParams {
  const void* x; const void* w; void* y; float* slab; const float* bias;
  int Nimg, Hs, Ws, xpitch, Cin, log2_cvecs;
  int Hg, Wg, M;
  int S, TWlog2, T, log2T;
  int dy0, dx0, dstep, wy0, wx0, wstep;
  int parity;
  int Wrows;
  int Ho, Wo, ypitch, Cout, OS;
  int splits, kchunks, NslabPitch;
  int tilesM, tilesN;
  int act; float slope; int out_f32; int vec_store;
  unsigned xbytes, wbytes;   // extents for the buffer descriptors (out-of-range offsets read zeros)
  float* stats; int stats_tpg, stats_C;   // fused per-channel (sum, sum^2) partials: tiles per group, channel count
  FastDiv divWg, divHg;      // GEMM-grid width / height (row -> (image, gy, gx) decode)
  // fused backward epilogue of a dgrad launch (GanBwdFuse, include/gan_amd.h): the stored value becomes
  // dz = (da + add) * act'(.) [* 2*dropmask] for channels < bf_cols; with bf_mean the (sum dz, sum dz*xhat) partials go to `stats`
  const void* bf_ref; const void* bf_add;
  const float* bf_mean; const float* bf_rstd; const float* bf_gamma; const float* bf_beta;
  const unsigned char* bf_mask;
  int bf_refpitch, bf_addpitch, bf_maskpitch, bf_mode, bf_cols; float bf_slope;
  // small split-K layer finished inside its slab-reduce kernel (GanNormFuse): skn = 1 forward, 2 backward
  int skn, skn_groups; void* skn_out; int skn_outpitch;
  const float* skn_gamma; const float* skn_beta; float* skn_mean; float* skn_rstd; float* skn_mmean; float* skn_mvar;
  float skn_eps, skn_momentum; const unsigned char* skn_mask; int skn_act; float skn_slope;
  float* skn_dgamma; float* skn_dbeta; int skn_accumulate;
#ifdef GAN_DIAG
  unsigned long long* diag;  // diagnostic build: per-block stamps
#endif

};

// conv_own.hip: column-owner kernel - a layer of <= 64 rows per parity that plan_gemm lets its slab reduce finish (p.skn) in ONE launch
bool conv_own_eligible(const GemmParams& p, int P, int dtype);
int conv_own_launch(const GemmParams& p, int P, int dtype, hipStream_t st);

// thin.hip: streaming kernels for the layers with <= 8 channels on one side (HBM-bound, no LDS tiling).
// thin_family(): 0 = use the tiled implicit GEMM, 1 = "thin-N" (few output channels), 2 = "thin-K" (8-channel input)
int thin_family(const GanConvDesc* d, int op, const GemmParams& p);
size_t thin_workspace_bytes(int family, const GanConvDesc* d, const GemmParams& p);
int thin_launch(int family, const GanConvDesc* d, const GemmParams& p, hipStream_t st);
